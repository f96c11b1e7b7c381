"""ctypes loader for libvitadapter_hip.so (the C ABI in include/vitadapter_hip.h).

The library is the product: there is NO fallback.  If it has not been built
(`python vit-adapter_amd/build.py`), importing this module raises ImportError.
"""
import ctypes
import os

# torch MUST be imported before the library is loaded: the PyTorch-ROCm wheel bundles its own
# libamdhip64.so, and the kernels have to be registered with (and launched through) the same
# HIP runtime instance that owns torch's streams and allocations.  Loading our library first
# would pull in /opt/rocm's runtime as a second instance ("no ROCm-capable device" at launch).
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'lib', 'libvitadapter_hip.so')
ABI_VERSION = 36

if not os.path.exists(LIB_PATH):
    raise ImportError(
        'libvitadapter_hip.so is missing (%s): build it with `python vit-adapter_amd/build.py` '
        '(hipcc --offload-arch=gfx950).  There is no CPU or PyTorch fallback.' % LIB_PATH)

lib = ctypes.CDLL(LIB_PATH)

_i64 = ctypes.c_int64
_p = ctypes.c_void_p

lib.vah_abi_version.restype = ctypes.c_int
lib.vah_last_error.restype = ctypes.c_char_p
lib.vah_prof_enable.argtypes = [ctypes.c_int]
lib.vah_prof_enable.restype = ctypes.c_int
lib.vah_prof_filter.argtypes = [ctypes.c_char_p]
lib.vah_prof_filter.restype = ctypes.c_int
lib.vah_prof_report.argtypes = [ctypes.c_char_p, _i64]
lib.vah_prof_report.restype = _i64
for _sfx in ('f32', 'f64'):
    _f = getattr(lib, 'vah_msda_forward_' + _sfx)
    _f.argtypes = [_p] * 5 + [_i64] * 7 + [_p, _p]
    _f.restype = ctypes.c_int
    _b = getattr(lib, 'vah_msda_backward_' + _sfx)
    _b.argtypes = [_p] * 6 + [_i64] * 7 + [_p] * 4
    _b.restype = ctypes.c_int

lib.vah_attn_padded_len.argtypes = [_i64]
lib.vah_attn_padded_len.restype = _i64
lib.vah_attn_fwd_bf16.argtypes = [_p, _p, _p, _i64, _i64, _i64, _i64, _i64, ctypes.c_float, _p, _p, _i64, _p, _p]
lib.vah_attn_fwd_bf16.restype = ctypes.c_int

lib.vah_attn_bwd_workspace_bytes.argtypes = [_i64, _i64, _i64]
lib.vah_attn_bwd_workspace_bytes.restype = _i64
lib.vah_attn_bwd_bf16.argtypes = ([_p, _p, _p, _i64, _i64, _p, _p, _i64, _p, _i64, _i64, _i64, ctypes.c_float]
                                  + [_p] * 4 + [_i64, _i64, _p])
lib.vah_attn_bwd_bf16.restype = ctypes.c_int

lib.vah_attn_bias_fwd_bf16.argtypes = [_p, _p, _p, _i64, _i64, _i64, _i64, _i64, ctypes.c_float, _p, _i64, _p, _i64, _p, _p]
lib.vah_attn_bias_bwd_bf16.argtypes = ([_p, _p, _p, _i64, _i64, _p, _p, _i64, _p, _i64, _i64, _i64, ctypes.c_float, _p, _p, _i64, _p, _p,
                                        _p, _p, _p, _i64, _i64, _p])
lib.vah_relpos_bias_build.argtypes = [_p, _p, _i64, _i64, _i64, _i64, _p, _p, _p]
lib.vah_relpos_bias_grad_ws_floats.argtypes = [_i64, _i64]
lib.vah_relpos_bias_grad_ws_floats.restype = _i64
lib.vah_relpos_bias_grad.argtypes = [_p, _p, _i64, _i64, _i64, _i64, _i64, _p, _p, _p]
lib.vah_attn_win_fwd_bf16.argtypes = [_p, _p, _p, _i64, _i64, _i64, _i64, _i64, _i64, ctypes.c_float, _p, _p, _i64, _p, _p]
lib.vah_attn_win_fwd_bf16.restype = ctypes.c_int
lib.vah_attn_win_bwd_bf16.argtypes = [_p, _p, _p, _i64, _p, _p, _i64, _p, _i64, _i64, _i64, _i64, _i64, ctypes.c_float,
                                      _p, _p, _p, _p, _i64, _p]
lib.vah_attn_win_bwd_bf16.restype = ctypes.c_int
_ci = ctypes.c_int
lib.vah_msda_fused_supported.argtypes = [_i64, _i64, _i64]
lib.vah_msda_fused_supported.restype = ctypes.c_int
lib.vah_msda_fused_forward.argtypes = [_p, _ci, _p, _p, _p, _p, _ci, _i64, _i64, _p, _i64] + [_i64] * 7 + [_p, _p]
lib.vah_msda_fused_forward.restype = ctypes.c_int
lib.vah_msda_fused_backward.argtypes = [_p, _ci, _p, _p, _p, _p, _ci, _p, _i64, _p] + [_i64] * 7 + [_p] * 4
lib.vah_msda_fused_backward.restype = ctypes.c_int
lib.vah_msda_fused_forward_win.argtypes = [_p, _ci, _p, _p, _p, _p, _ci, _i64, _i64, _p] + [_i64] * 7 + [_p, _i64, _ci, _p, _p]
lib.vah_msda_win_ws_bytes.argtypes = [_i64, _i64]
lib.vah_msda_win_ws_bytes.restype = _i64
lib.vah_msda_fused_forward_win.restype = ctypes.c_int
lib.vah_msda_tile_ws_bytes.argtypes = [_i64] * 6
lib.vah_msda_tile_ws_bytes.restype = _i64
lib.vah_msda_backward_tiled_f32.argtypes = [_p] * 6 + [_i64] * 7 + [_p] * 3 + [_p, _i64, _p]
lib.vah_msda_backward_tiled_f32.restype = ctypes.c_int
lib.vah_msda_fused_backward_tiled.argtypes = ([_p, _ci, _p, _p, _p, _p, _ci, _i64, _i64, _p, _i64, _p] + [_i64] * 7
                                              + [_p, _ci, _p, _p, _ci, _i64, _i64, _p, _i64, _p])
lib.vah_msda_fused_backward_tiled.restype = ctypes.c_int
_f = ctypes.c_float
lib.vah_layernorm_fwd_f32_bf16.argtypes = [_p, _p, _p, _i64, _i64, _f, _p, _p, _p, _p]
lib.vah_layernorm_bwd_f32_bf16.argtypes = [_p, _p, _p, _p, _p, _p, _i64, _i64, _p, _p, _p, _p, _p]
lib.vah_colsum_bf16.argtypes = [_p, _i64, _i64, _p, _p, _p]
lib.vah_layernorm_dual_fwd.argtypes = [_p, _p, _p, _p, _p, _i64, _i64, _f, _p, _p, _p, _p, _p]
lib.vah_layernorm_dual_bwd.argtypes = [_p] * 8 + [_i64, _i64, _p, _p, _p, _p]
lib.vah_colsum_f32.argtypes = [_p, _i64, _i64, _i64, _i64, _p, _p, _p]
lib.vah_residual_layernorm_fwd.argtypes = [_p, _p, _p, _p, _i64, _i64, _i64, _p, _p, ctypes.c_float, _p, _p, _p, _p, _p]
lib.vah_residual_layernorm_bwd.argtypes = [_p] * 9 + [_i64] * 3 + [_p] * 7
_int = ctypes.c_int
lib.vah_gemm_set_tuning.argtypes = [_int, _int]
lib.vah_gemm_bf16.argtypes = [_int, _int, _i64, _i64, _i64, _p, _i64, _p, _i64, _p, _i64, _int, _int, _p, _int,
                              _p, _i64, _p]
lib.vah_gemm_bf16_fin.argtypes = [_int, _int, _i64, _i64, _i64, _p, _i64, _p, _i64, _p, _i64, _int, _p, _i64, _p, _i64, _i64,
                                  _p, _p]
lib.vah_colsum_bf16_partials.argtypes = [_p, _i64, _i64, _p, ctypes.POINTER(_i64), _p]
lib.vah_gemm_table_dump.argtypes = [ctypes.c_char_p, _i64]
lib.vah_gemm_table_dump.restype = _i64
lib.vah_gemm_table_load.argtypes = [ctypes.c_char_p]
lib.vah_gemm_library_version.argtypes = []
lib.vah_gemm_library_version.restype = _i64
lib.vah_gemm_rejected_candidates.argtypes = []
lib.vah_gemm_rejected_candidates.restype = _i64
_tail_in = [_p, _int, _p, _int, _p, _int, _i64, _i64, _i64, _i64]
lib.vah_bn_tail_ws_floats.argtypes = [_i64]
lib.vah_bn_tail_ws_floats.restype = _i64
lib.vah_bn_tail_stats.argtypes = _tail_in + [_p, _p, _p, _p]
lib.vah_bn_tail_apply.argtypes = _tail_in + [_p, _p, _p, _p, _int, _p, _p, _int, _p]
lib.vah_bn_tail_bwd_stats.argtypes = _tail_in + [_p, _p, _p, _p, _int, _p, _p, _int, _p, _p, _p]
lib.vah_bn_tail_bwd_apply.argtypes = _tail_in + [_p, _p, _p, _p, _int, _p, _p, _int, _p, _p, _p, _p, _p, _p]
lib.vah_bn_finalize_stats.argtypes = [_p, _i64, _f, _f, _p, _p, _p, _p, _p]
lib.vah_maxpool3s2_fwd_bf16.argtypes = [_p, _i64, _i64, _i64, _p, _p, _p]
lib.vah_maxpool3s2_bwd_bf16.argtypes = [_p, _p, _i64, _i64, _i64, _p, _p]
lib.vah_conv_taps_nhwc_bf16.argtypes = [_p, _i64, _i64, _i64, _i64, _p, _i64, _int, _p, _p, _int, _p, _i64, _i64, _i64, _i64, _int, _int,
                                        _int, _p]
lib.vah_conv3x3_dgrad_nhwc_bf16.argtypes = [_p, _i64, _i64, _i64, _i64, _p, _i64, _int, _p, _i64, _i64, _p]
lib.vah_conv3x3_wgrad_ws_floats.argtypes = [_i64, _i64]
lib.vah_conv3x3_wgrad_ws_floats.restype = _i64
lib.vah_conv3x3_wgrad_nhwc_bf16.argtypes = [_p, _i64, _i64, _i64, _i64, _p, _i64, _i64, _i64, _int, _p, _i64, _p, _p]
lib.vah_image_to_nhwc16_bf16.argtypes = [_p, _i64, _i64, _i64, _p, _p]
lib.vah_patchify_bf16.argtypes = [_p, _i64, _i64, _i64, _i64, _i64, _p, _p]
lib.vah_bn_nhwc_ws_floats.argtypes = [_i64]
lib.vah_bn_nhwc_ws_floats.restype = _i64
lib.vah_bn_nhwc_stats.argtypes = [_p, _i64, _i64, _p, _p, _p]
lib.vah_bn_nhwc_apply.argtypes = [_p, _i64, _i64, _p, _p, _p, _p, _int, _p, _p]
lib.vah_bn_nhwc_bwd_stats.argtypes = [_p, _p, _i64, _i64, _p, _p, _p, _p, _int, _p, _p, _p]
lib.vah_bn_nhwc_bwd_apply.argtypes = [_p, _p, _i64, _i64, _p, _p, _p, _p, _int, _p, _p, _p, _p]
lib.vah_maxpool3s2_nhwc_fwd_bf16.argtypes = [_p, _i64, _i64, _i64, _i64, _p, _p, _p]
lib.vah_maxpool3s2_nhwc_bwd_bf16.argtypes = [_p, _p, _i64, _i64, _i64, _i64, _p, _p]
lib.vah_pixel_shuffle2_bf16.argtypes = [_p, _i64, _i64, _i64, _i64, _p, _int, _p, _p]
lib.vah_transpose_tokens.argtypes = [_p, _i64, _i64, _i64, _i64, _i64, _p, _int, _int, _p, _p]
lib.vah_reduce_ws_floats.argtypes = [_i64]
lib.vah_reduce_ws_floats.restype = _i64
lib.vah_scale_residual_fwd.argtypes = [_p, _p, _p, _p, _i64, _i64, _i64, _p, _p]
lib.vah_scale_residual_bwd.argtypes = [_p, _p, _p, _p, _i64, _i64, _i64, _p, _p, _p, _p]
lib.vah_dwconv3x3_tokens_bf16.argtypes = [_p, _p, _p, _i64, _i64, _i64, _i64, ctypes.c_int, _p, _p]
lib.vah_dwconv3x3_tokens_wgrad_bf16.argtypes = [_p, _p, _i64, _i64, _i64, _i64, _p, _p, _p, _p]
for _n in ('vah_layernorm_fwd_f32_bf16', 'vah_layernorm_bwd_f32_bf16', 'vah_scale_residual_fwd',
           'vah_scale_residual_bwd', 'vah_dwconv3x3_tokens_bf16', 'vah_dwconv3x3_tokens_wgrad_bf16',
           'vah_colsum_bf16', 'vah_colsum_f32', 'vah_layernorm_dual_fwd', 'vah_layernorm_dual_bwd', 'vah_residual_layernorm_fwd', 'vah_residual_layernorm_bwd', 'vah_gemm_set_tuning', 'vah_gemm_bf16', 'vah_gemm_bf16_fin', 'vah_colsum_bf16_partials', 'vah_gemm_table_load',
           'vah_bn_tail_stats', 'vah_bn_tail_apply', 'vah_bn_tail_bwd_stats', 'vah_bn_tail_bwd_apply',
           'vah_bn_finalize_stats', 'vah_transpose_tokens', 'vah_maxpool3s2_fwd_bf16', 'vah_maxpool3s2_bwd_bf16',
           'vah_conv_taps_nhwc_bf16', 'vah_conv3x3_dgrad_nhwc_bf16', 'vah_conv3x3_wgrad_nhwc_bf16',
           'vah_pixel_shuffle2_bf16', 'vah_patchify_bf16', 'vah_attn_bias_fwd_bf16', 'vah_attn_bias_bwd_bf16', 'vah_relpos_bias_build', 'vah_relpos_bias_grad', 'vah_image_to_nhwc16_bf16', 'vah_bn_nhwc_stats', 'vah_bn_nhwc_apply', 'vah_bn_nhwc_bwd_stats', 'vah_bn_nhwc_bwd_apply',
           'vah_maxpool3s2_nhwc_fwd_bf16', 'vah_maxpool3s2_nhwc_bwd_bf16'):
    getattr(lib, _n).restype = ctypes.c_int

if lib.vah_abi_version() != ABI_VERSION:
    raise ImportError('libvitadapter_hip.so ABI %d != binding ABI %d: rebuild the library'
                      % (lib.vah_abi_version(), ABI_VERSION))

# every symbol include/vitadapter_hip.h declares (checked by tests/test_capi_symbols.py)
EXPORTS = (
    'vah_abi_version', 'vah_last_error', 'vah_prof_enable', 'vah_prof_filter', 'vah_prof_report',
    'vah_msda_forward_f32', 'vah_msda_forward_f64',
    'vah_msda_backward_f32', 'vah_msda_backward_f64',
    'vah_msda_fused_supported', 'vah_msda_fused_forward', 'vah_msda_fused_backward',
    'vah_msda_fused_forward_win', 'vah_msda_win_ws_bytes', 'vah_msda_tile_ws_bytes', 'vah_msda_backward_tiled_f32', 'vah_msda_fused_backward_tiled',
    'vah_pixel_shuffle2_bf16', 'vah_patchify_bf16', 'vah_attn_bias_fwd_bf16', 'vah_attn_bias_bwd_bf16', 'vah_relpos_bias_build', 'vah_relpos_bias_grad_ws_floats',
    'vah_relpos_bias_grad', 'vah_attn_padded_len', 'vah_attn_fwd_bf16', 'vah_attn_bwd_workspace_bytes', 'vah_attn_bwd_bf16',
    'vah_attn_win_fwd_bf16', 'vah_attn_win_bwd_bf16',
    'vah_reduce_ws_floats', 'vah_layernorm_fwd_f32_bf16', 'vah_layernorm_bwd_f32_bf16', 'vah_scale_residual_fwd',
    'vah_scale_residual_bwd', 'vah_dwconv3x3_tokens_bf16', 'vah_dwconv3x3_tokens_wgrad_bf16', 'vah_colsum_bf16', 'vah_colsum_f32',
    'vah_layernorm_dual_fwd', 'vah_layernorm_dual_bwd',
    'vah_residual_layernorm_fwd', 'vah_residual_layernorm_bwd',
    'vah_gemm_set_tuning', 'vah_gemm_bf16', 'vah_gemm_bf16_fin', 'vah_colsum_bf16_partials', 'vah_gemm_table_dump', 'vah_gemm_table_load', 'vah_gemm_library_version', 'vah_gemm_rejected_candidates',
    'vah_bn_tail_ws_floats', 'vah_bn_tail_stats', 'vah_bn_tail_apply', 'vah_bn_tail_bwd_stats', 'vah_bn_tail_bwd_apply',
    'vah_bn_finalize_stats', 'vah_transpose_tokens', 'vah_maxpool3s2_fwd_bf16', 'vah_maxpool3s2_bwd_bf16',
    'vah_image_to_nhwc16_bf16', 'vah_bn_nhwc_ws_floats', 'vah_bn_nhwc_stats', 'vah_bn_nhwc_apply', 'vah_bn_nhwc_bwd_stats',
    'vah_bn_nhwc_bwd_apply', 'vah_maxpool3s2_nhwc_fwd_bf16', 'vah_maxpool3s2_nhwc_bwd_bf16',
    'vah_conv_taps_nhwc_bf16', 'vah_conv3x3_dgrad_nhwc_bf16', 'vah_conv3x3_wgrad_ws_floats', 'vah_conv3x3_wgrad_nhwc_bf16',
)


class _NoSwitch:
    """Context manager that does nothing (the tensor's device is already current)."""

    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_SWITCH = _NoSwitch()


def on(device):
    """``with on(t.device):`` - make ``device`` current for the HIP calls inside.  torch.cuda.device costs ~5 us of
    host time per use (index normalisation + two exchange calls); a training step enters it ~400 times, always for
    the device that already is current: in that case nothing is switched."""
    import torch
    if device.index is None or torch.cuda.current_device() == device.index:
        return _NO_SWITCH
    return torch.cuda.device(device)


def raw_stream(device):
    """hipStream_t (as an int) of torch's current stream on ``device``.  torch._C._cuda_getCurrentRawStream skips the
    Stream object torch.cuda.current_stream builds (~5 us of host time per kernel launch, ~200 launches per step)."""
    import torch
    idx = device.index if device.index is not None else torch.cuda.current_device()
    try:
        return torch._C._cuda_getCurrentRawStream(idx)
    except AttributeError:                 # other torch builds
        return torch.cuda.current_stream(device).cuda_stream


def check(rc, what):
    """Turn a non-zero ABI return code into RuntimeError (the reference only printf()s)."""
    if rc != 0:
        msg = lib.vah_last_error().decode('utf-8', 'replace')
        raise RuntimeError('%s failed (code %d): %s' % (what, rc, msg))


def prof_enable(on, prefix=''):
    """Time the launches of the entry points whose profile name starts with ``prefix``."""
    lib.vah_prof_filter(prefix.encode())
    lib.vah_prof_enable(1 if on else 0)


def prof_report():
    """-> {kernel_name: dict(calls, total_ms, bytes, def_bytes, flops)}; synchronises the recorded events.
    bytes = algorithmic bytes for the IO dtypes the launches ran with, def_bytes = the operator's
    fp32-definition bytes, flops = matrix-core work (0 for the HBM-bound entry points)."""
    need = lib.vah_prof_report(None, 0)
    buf = ctypes.create_string_buffer(int(need) + 64)
    lib.vah_prof_report(buf, len(buf))
    out = {}
    for line in buf.value.decode().splitlines():
        name, calls, ms, nbytes, dbytes, flops = line.split()
        out[name] = dict(calls=int(calls), total_ms=float(ms), bytes=int(nbytes), def_bytes=int(dbytes),
                         flops=int(flops))
    return out


GEMM_EPI_NONE, GEMM_EPI_BIAS = 0, 1


def gemm_table_dump():
    """The GEMM algorithm cache as text: a '#hipblaslt <version>' line, then one problem per line
    (see include/vitadapter_hip.h)."""
    n = lib.vah_gemm_table_dump(None, 0)
    buf = ctypes.create_string_buffer(int(n))
    lib.vah_gemm_table_dump(buf, n)
    return '#hipblaslt %d\n' % lib.vah_gemm_library_version() + buf.value.decode()


def gemm_table_load(text):
    """Load a dumped table; a table of another hipBLASLt build is ignored (returns 0): its
    algorithm indices mean nothing here and every problem is simply timed again."""
    first = text.split('\n', 1)[0].split()
    if len(first) == 2 and first[0] == '#hipblaslt' and int(first[1]) != lib.vah_gemm_library_version():
        return 0
    n = lib.vah_gemm_table_load(text.encode())
    if n < 0:
        check(n, 'gemm_table_load')
    return n
