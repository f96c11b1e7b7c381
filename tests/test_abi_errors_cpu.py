"""CPU: argument checking of the round-2 entry points of libvitadapter_hip.so (include/vitadapter_hip.h).  Every call
here is rejected by the host-side validation before anything touches a device: wrong shapes, strides, alignments
and null pointers come back as VAH_E_* codes with a message in vah_last_error (the reference's extension printf()s and
carries on, ms_deform_attn_cuda.cu:50-52)."""
import ctypes

import pytest

import _vah

lib = _vah.lib
E_NULL, E_SHAPE, E_ALIGN = -1, -2, -4
P = 4096           # a non-null, 16-byte aligned fake pointer: never dereferenced by a rejected call


def _err():
    return lib.vah_last_error().decode()


def test_conv_taps_rejects_bad_shapes():
    ty = (ctypes.c_int * 9)(*([0] * 9))
    tx = (ctypes.c_int * 9)(*([0] * 9))
    # Cin must be 16 or a multiple of 64
    assert lib.vah_conv_taps_nhwc_bf16(P, 1, 8, 8, 24, P, 64, 9, ty, tx, 1, P, 8, 8, 8, 8, 1, 0, 0, None) == E_SHAPE
    assert 'Cin' in _err()
    # more than 9 taps, stride 3
    assert lib.vah_conv_taps_nhwc_bf16(P, 1, 8, 8, 64, P, 64, 10, ty, tx, 1, P, 8, 8, 8, 8, 1, 0, 0, None) == E_SHAPE
    assert lib.vah_conv_taps_nhwc_bf16(P, 1, 8, 8, 64, P, 64, 9, ty, tx, 3, P, 8, 8, 8, 8, 1, 0, 0, None) == E_SHAPE
    # output positions outside the output tensor
    assert lib.vah_conv_taps_nhwc_bf16(P, 1, 8, 8, 64, P, 64, 9, ty, tx, 1, P, 8, 8, 8, 8, 2, 1, 1, None) == E_SHAPE
    # tap offsets beyond +-4
    ty[0] = 7
    assert lib.vah_conv_taps_nhwc_bf16(P, 1, 8, 8, 64, P, 64, 9, ty, tx, 1, P, 8, 8, 8, 8, 1, 0, 0, None) == E_SHAPE
    ty[0] = 0
    assert lib.vah_conv_taps_nhwc_bf16(None, 1, 8, 8, 64, P, 64, 9, ty, tx, 1, P, 8, 8, 8, 8, 1, 0, 0, None) == E_NULL
    assert lib.vah_conv_taps_nhwc_bf16(P + 2, 1, 8, 8, 64, P, 64, 9, ty, tx, 1, P, 8, 8, 8, 8, 1, 0, 0, None) == E_ALIGN
    # nothing to do is not an error
    assert lib.vah_conv_taps_nhwc_bf16(None, 0, 8, 8, 64, None, 64, 9, ty, tx, 1, None, 8, 8, 8, 8, 1, 0, 0, None) == 0


def test_conv_gradients_reject_inconsistent_geometry():
    # OH must be (H - 1) // S + 1
    assert lib.vah_conv3x3_dgrad_nhwc_bf16(P, 1, 5, 4, 64, P, 64, 2, P, 8, 8, None) == E_SHAPE
    assert lib.vah_conv3x3_dgrad_nhwc_bf16(P, 1, 4, 4, 64, P, 48, 2, P, 8, 8, None) == E_SHAPE          # Cin % 64
    assert lib.vah_conv3x3_dgrad_nhwc_bf16(None, 1, 4, 4, 64, P, 64, 2, P, 8, 8, None) == E_NULL
    assert lib.vah_conv3x3_wgrad_nhwc_bf16(P, 1, 8, 8, 64, P, 5, 4, 64, 2, P, 1 << 20, P, None) == E_SHAPE
    assert lib.vah_conv3x3_wgrad_nhwc_bf16(P, 1, 8, 8, 64, P, 4, 4, 64, 2, None, 1 << 20, P, None) == E_NULL
    assert lib.vah_conv3x3_wgrad_ws_floats(64, 64) == 256 * 64 * 9 * 64
    assert lib.vah_conv3x3_wgrad_ws_floats(16, 64) == 256 * 64 * 9 * 16


def test_nhwc_batchnorm_and_pool_reject_bad_channels():
    assert lib.vah_bn_nhwc_stats(P, 100, 96, P, P, None) == E_SHAPE              # not a power of two
    assert lib.vah_bn_nhwc_stats(P, 100, 512, P, P, None) == E_SHAPE             # > 256
    assert lib.vah_bn_nhwc_apply(P, 100, 64, None, P, None, None, 1, P, None) == E_NULL
    assert lib.vah_bn_nhwc_apply(None, 0, 64, None, None, None, None, 1, None, None) == 0
    assert lib.vah_bn_nhwc_bwd_stats(P, None, 100, 64, P, P, None, None, 1, P, P, None) == E_NULL
    assert lib.vah_bn_nhwc_bwd_apply(P, P, 100, 48, P, P, None, None, 1, P, P, P, None) == E_SHAPE
    assert lib.vah_maxpool3s2_nhwc_fwd_bf16(P, 1, 8, 8, 12, P, P, None) == E_SHAPE
    assert lib.vah_maxpool3s2_nhwc_bwd_bf16(None, P, 1, 8, 8, 16, P, None) == E_NULL
    assert lib.vah_image_to_nhwc16_bf16(P, 1, 0, 8, P, None) == E_SHAPE
    assert lib.vah_image_to_nhwc16_bf16(P, 1, 8, 8, P + 8, None) == E_ALIGN


def test_layout_kernels_reject_bad_widths():
    assert lib.vah_pixel_shuffle2_bf16(P, 1, 8, 4, 12, P, 0, None, None) == E_SHAPE           # w % 8
    assert lib.vah_pixel_shuffle2_bf16(P, 1, 8, 4, 16, None, 0, None, None) == E_NULL
    assert lib.vah_pixel_shuffle2_bf16(P + 8, 1, 8, 4, 16, P, 1, None, None) == E_ALIGN
    assert lib.vah_patchify_bf16(P, 1, 3, 64, 64, 12, P, None) == E_SHAPE                # patch size % 8
    assert lib.vah_patchify_bf16(P, 1, 3, 60, 64, 16, P, None) == E_SHAPE                # H % patch size
    assert lib.vah_patchify_bf16(None, 1, 3, 64, 64, 16, P, None) == E_NULL


def test_bias_attention_rejects_bad_bias_layout():
    args = (P, P, P, 192, 5 * 192, 1, 1, 5, 0.125)
    assert lib.vah_attn_bias_fwd_bf16(*args, P, 60, P, 64, P, None) == E_SHAPE           # ldb % 64
    assert lib.vah_attn_bias_fwd_bf16(*args, P, 0, P, 64, P, None) == E_SHAPE            # ldb < N
    assert 'ldb' in _err()
    assert lib.vah_attn_bias_fwd_bf16(*args, None, 64, P, 64, P, None) == E_NULL
    assert lib.vah_attn_bias_fwd_bf16(P, P, P, 192, 7, 1, 1, 5, 0.125, P, 64, P, 64, P, None) == E_SHAPE      # batch stride
    assert lib.vah_relpos_bias_build(P, P, 10, 2, 5, 4, P, P, None) == E_SHAPE           # ldb < N
    assert lib.vah_relpos_bias_build(None, P, 10, 2, 5, 64, P, P, None) == E_NULL
    assert lib.vah_relpos_bias_grad(P, P, 1, 2, 5, 64, 1 << 20, P, P, None) == E_SHAPE   # table too large for the LDS bins
    assert lib.vah_relpos_bias_grad_ws_floats(10, 2) == 32 * 2 * 10


@pytest.mark.parametrize('name', ['vah_conv_taps_nhwc_bf16', 'vah_conv3x3_dgrad_nhwc_bf16', 'vah_conv3x3_wgrad_nhwc_bf16',
                                  'vah_bn_nhwc_stats', 'vah_pixel_shuffle2_bf16', 'vah_patchify_bf16', 'vah_attn_bias_fwd_bf16',
                                  'vah_attn_bias_bwd_bf16', 'vah_relpos_bias_build', 'vah_relpos_bias_grad'])
def test_round2_symbols_are_exported(name):
    assert name in _vah.EXPORTS and hasattr(lib, name)
