"""3x3 convolutions of the SpatialPriorModule on libvitadapter_hip.so (csrc/conv.hip): NHWC bf16 tensors, implicit
GEMMs on the matrix cores.  Reference: nn.Conv2d(k=3, padding=1, stride 1 | 2, bias=False) of
/root/reference/detection/mmdet_custom/models/backbones/adapter_modules.py:217-260 and its autograd.
There is no CPU path behind these functions: importing the module loads the HIP library."""
import ctypes

import torch

import _vah

_FWD_TAPS = [(dy - 1, dx - 1) for dy in range(3) for dx in range(3)]


def _stream(t):
    return _vah.raw_stream(t.device)


def _taps(call, x, w, taps, S, out, ny, nx, OS, oy0, ox0):
    N, IH, IW, Cin = x.shape
    Cout, T = w.shape[0], len(taps)
    ty = (ctypes.c_int * T)(*[t[0] for t in taps])
    tx = (ctypes.c_int * T)(*[t[1] for t in taps])
    with _vah.on(x.device):
        rc = _vah.lib.vah_conv_taps_nhwc_bf16(x.data_ptr(), N, IH, IW, Cin, w.data_ptr(), Cout, T, ty, tx, S, out.data_ptr(), ny,
                                              nx, out.shape[1], out.shape[2], OS, oy0, ox0, _stream(x))
    _vah.check(rc, call)


def forward_weight(weight):
    """(Cout, Cin, 3, 3) -> (Cout, 9, Cin) bf16, the tap-major layout the kernels read."""
    return weight.detach().permute(0, 2, 3, 1).reshape(weight.shape[0], 9, weight.shape[1]).to(torch.bfloat16).contiguous()


def conv3x3_forward(x, w9, stride):
    """x (N, H, W, Cin) bf16 NHWC, w9 = forward_weight(weight) -> (N, OH, OW, Cout) bf16."""
    N, H, W, _ = x.shape
    OH, OW = (H - 1) // stride + 1, (W - 1) // stride + 1
    out = torch.empty((N, OH, OW, w9.shape[0]), dtype=torch.bfloat16, device=x.device)
    _taps('vah_conv_taps_nhwc_bf16', x, w9, _FWD_TAPS, stride, out, OH, OW, 1, 0, 0)
    return out


def dgrad_weight(weight):
    """(Cout, Cin, 3, 3) -> (Cin, 9, Cout) bf16: the layout the input gradient reads."""
    return weight.detach().permute(1, 2, 3, 0).reshape(weight.shape[1], 9, weight.shape[0]).to(torch.bfloat16).contiguous()


def conv3x3_input_grad(gy, wt9, stride, in_hw):
    """gy (N, OH, OW, Cout) bf16 NHWC, wt9 = dgrad_weight(weight) -> d(loss)/d(input) (N, H, W, Cin) bf16."""
    H, W = in_hw
    N, OH, OW, cout = gy.shape
    cin = wt9.shape[0]
    gx = torch.empty((N, H, W, cin), dtype=torch.bfloat16, device=gy.device)
    with _vah.on(gy.device):
        rc = _vah.lib.vah_conv3x3_dgrad_nhwc_bf16(gy.data_ptr(), N, OH, OW, cout, wt9.data_ptr(), cin, stride, gx.data_ptr(), H, W,
                                                  _stream(gy))
    _vah.check(rc, 'vah_conv3x3_dgrad_nhwc_bf16')
    return gx


def conv3x3_weight_grad(x, gy, stride):
    """x (N, H, W, Cin), gy (N, OH, OW, Cout) bf16 NHWC -> d(loss)/d(weight) as (Cout, 3, 3, Cin) fp32."""
    N, H, W, cin = x.shape
    _, OH, OW, cout = gy.shape
    nws = _vah.lib.vah_conv3x3_wgrad_ws_floats(cin, cout)
    ws = torch.empty((nws,), dtype=torch.float32, device=x.device)
    dw = torch.empty((cout, 3, 3, cin), dtype=torch.float32, device=x.device)
    with _vah.on(x.device):
        rc = _vah.lib.vah_conv3x3_wgrad_nhwc_bf16(x.data_ptr(), N, H, W, cin, gy.data_ptr(), OH, OW, cout, stride, ws.data_ptr(),
                                                  nws, dw.data_ptr(), _stream(x))
    _vah.check(rc, 'vah_conv3x3_wgrad_nhwc_bf16')
    return dw
