"""Fused core of MSDeformAttn.forward: softmax(logits) + sampling locations + gather in one HIP
kernel (csrc/msda_fused.hip), gradients in one more.  Not part of the reference's surface (its
module does these steps with ~8 PyTorch ops around MSDeformAttnFunction,
/root/reference/detection/ops/modules/ms_deform_attn.py:108-128); ops.modules.MSDeformAttn uses it
when the shapes allow and otherwise keeps the reference sequence.
"""
import os

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

import _vah

_DT = {torch.float32: 0, torch.bfloat16: 1}


def fused_supported(value, offsets, logits, reference_points, n_levels, n_points):
    """True when the fused kernels cover this call (else use the unfused Function)."""
    if os.environ.get('VAH_MSDA_FUSED', '1') == '0':
        return False
    return (value.is_cuda and value.dim() == 4 and value.dtype in _DT and offsets.dtype in _DT
            and logits.dtype == offsets.dtype and reference_points.shape[0] == 1
            and reference_points.shape[-1] == 2 and reference_points.shape[2] in (1, n_levels)
            and bool(_vah.lib.vah_msda_fused_supported(value.shape[-1], n_levels, n_points))
            and value.numel() > 0 and offsets.numel() > 0)


class PullSchedule:
    """Tiles of the value maps + candidate query lists for the atomic-free grad_value pass
    (include/vitadapter_hip.h, "PULL SCHEDULE").  Static per (reference grid, value shapes)."""

    def __init__(self, tile_meta, cand, ntiles, radius, cap):
        self.tile_meta, self.cand = tile_meta, cand
        self.ntiles, self.radius, self.cap = int(ntiles), float(radius), int(cap)


_PULL_CACHE = {}


def dense_pull(value_dtype):
    """The matrix-core form of the pull pass (bf16 grad_out rows; VAH_MSDA_PULL_MODE=sort|dense)."""
    return value_dtype == torch.bfloat16 and os.environ.get('VAH_MSDA_PULL_MODE', 'dense') == 'dense'


def build_pull_schedule(reference_points, value_shapes, radius=None, tile=None, n_points=4, dense=False):
    """reference_points (1, Lq, 1|L, 2) on the GPU, value_shapes [(H, W)] host ints.
    Tile edge per level: 16 px where the level is sparsely sampled (few bucket entries per pixel),
    else 8 px, so that a tile's bucket store (~ pixels x entries per pixel) fits one workgroup."""
    radius = float(os.environ.get('VAH_MSDA_PULL_RADIUS', 5.0)) if radius is None else float(radius)
    tile_env = int(os.environ.get('VAH_MSDA_PULL_TILE', 0)) if tile is None else int(tile)
    if dense and tile_env <= 0:
        tile_env = 8                       # the dense kernel multiplies 64-pixel slabs
    split = max(64, int(os.environ.get('VAH_MSDA_PULL_SPLIT', 2048)))
    dev = reference_points.device
    ref = reference_points.detach().float()[0]                  # (Lq, RL, 2)
    metas, cands, start = [], [], 0
    # a near sample lies within `radius` px of its reference point and touches the pixels
    # floor(.) and floor(.) + 1: a query can reach a tile from at most radius + 1 px away
    margin = radius + 1.0 + 0.05
    for l, (H, W) in enumerate(value_shapes):
        per_px = ref.shape[0] * n_points * 4.0 / float(H * W)       # expected bucket entries per pixel
        tile = tile_env if tile_env > 0 else (16 if per_px <= 12.0 else 8)
        r = ref[:, l if ref.shape[1] > 1 else 0]
        px, py = r[:, 0] * W - 0.5, r[:, 1] * H - 0.5
        for y0 in range(0, H, tile):
            ny = min(tile, H - y0)
            in_y = (py >= y0 - margin) & (py <= y0 + ny - 1 + margin)
            idx_y = in_y.nonzero().squeeze(1)
            if idx_y.numel() == 0:
                continue
            pxs = px[idx_y]
            for x0 in range(0, W, tile):
                nx = min(tile, W - x0)
                sel = idx_y[(pxs >= x0 - margin) & (pxs <= x0 + nx - 1 + margin)]
                if sel.numel() == 0:
                    continue
                # the dense kernel walks a candidate list 64 at a time: long lists are cut into several
                # entries of the same tile (their partial sums meet in the atomics) so that the
                # workgroups are many and even
                step = split if dense else int(sel.numel())
                for s0 in range(0, int(sel.numel()), step):
                    part = sel[s0:s0 + step]
                    metas.append([l, y0, x0, ny, nx, 0, int(part.numel()), 0])
                    cands.append(part.to(torch.int32))
    if not metas:
        return None
    order = sorted(range(len(metas)), key=lambda i: -metas[i][6])       # longest lists first
    metas, cands = [metas[i] for i in order], [cands[i] for i in order]
    for mt, cd in zip(metas, cands):
        mt[5] = start
        start += int(cd.numel())
    meta = torch.tensor(metas, dtype=torch.int32, device=dev)
    cand = torch.cat(cands).contiguous()
    cap = int(os.environ.get('VAH_MSDA_PULL_CAP', 7168))      # 70 KB of LDS: two workgroups per CU
    return PullSchedule(meta, cand, len(metas), radius, 0 if dense else cap)      # cap 0 selects the dense kernel


def pull_schedule_for(reference_points, spatial_shapes, dense=False):
    """Cached per reference-point tensor (one host read of the (L, 2) shapes on the first call)."""
    if os.environ.get('VAH_MSDA_PULL', '1') == '0':
        return None
    key = (reference_points.data_ptr(), tuple(reference_points.shape), reference_points._version,
           spatial_shapes.data_ptr(), str(reference_points.device), bool(dense))
    hit = _PULL_CACHE.get(key)
    if hit is None:
        shapes = [tuple(int(v) for v in hw) for hw in spatial_shapes.tolist()]
        sched = build_pull_schedule(reference_points, shapes, dense=dense)
        if len(_PULL_CACHE) > 64:
            _PULL_CACHE.clear()
        _PULL_CACHE[key] = (sched, reference_points, spatial_shapes)     # keep the keys' tensors alive
        return sched
    return hit[0]


def tiled_backward(n_levels, n_points):
    """The atomic-free tile pass serves P == 4, L <= 4 (VAH_MSDA_TILED=0: the round-1 pull / atomic kernels)."""
    return n_points == 4 and 1 <= n_levels <= 4 and os.environ.get('VAH_MSDA_TILED', '1') != '0'


class MSDeformAttnFusedFunction(Function):
    """apply(value (N,S,M,32), spatial_shapes, level_start_index, offsets (N,Lq,M,L,P,2),
    logits (N,Lq,M,L*P), reference_points (1,Lq,1|L,2)) -> (N, Lq, M*32) in value's dtype."""

    @staticmethod
    def forward(ctx, value, spatial_shapes, level_start_index, offsets, logits, reference_points):
        N, S, M, D = value.shape
        _, Lq, _, L, P, _ = offsets.shape
        value, offsets, logits = value.contiguous(), offsets.contiguous(), logits.contiguous()
        ref = reference_points.detach().float().contiguous().view(Lq, -1, 2)
        out = torch.empty((N, Lq, M * D), dtype=value.dtype, device=value.device)
        with torch.cuda.device(value.device):
            rc = _vah.lib.vah_msda_fused_forward(
                value.data_ptr(), _DT[value.dtype], spatial_shapes.data_ptr(),
                level_start_index.data_ptr(), offsets.data_ptr(), logits.data_ptr(),
                _DT[offsets.dtype], ref.data_ptr(), ref.shape[1], N, S, M, D, L, Lq, P,
                out.data_ptr(), torch.cuda.current_stream(value.device).cuda_stream)
        _vah.check(rc, 'vah_msda_fused_forward')
        ctx.save_for_backward(value, spatial_shapes, level_start_index, offsets, logits, ref)
        ctx.tiled = tiled_backward(L, P)
        ctx.pull = None if ctx.tiled else pull_schedule_for(reference_points, spatial_shapes, dense_pull(value.dtype))
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        value, shapes, lsi, offsets, logits, ref = ctx.saved_tensors
        N, S, M, D = value.shape
        _, Lq, _, L, P, _ = offsets.shape
        grad_output = grad_output.contiguous().to(value.dtype)
        d_off = torch.empty_like(offsets)
        d_logit = torch.empty_like(logits)
        if ctx.tiled:
            # atomic-free tile pass (csrc/msda_tile.hip): grad_value is STORED, in the value's dtype
            sh_host, lsi_host, _ = _vah.host_geometry(shapes, lsi)
            ws_bytes = _vah.lib.vah_msda_tile_ws_bytes(N, S, M, L, Lq, P, sh_host, lsi_host)
            if ws_bytes >= 0:
                grad_value = torch.empty_like(value)
                ws = torch.empty(ws_bytes, dtype=torch.uint8, device=value.device)
                with torch.cuda.device(value.device):
                    rc = _vah.lib.vah_msda_fused_backward_tiled(
                        value.data_ptr(), _DT[value.dtype], shapes.data_ptr(), lsi.data_ptr(),
                        offsets.data_ptr(), logits.data_ptr(), _DT[offsets.dtype], ref.data_ptr(),
                        ref.shape[1], grad_output.data_ptr(), N, S, M, D, L, Lq, P, grad_value.data_ptr(),
                        _DT[value.dtype], d_off.data_ptr(), d_logit.data_ptr(), sh_host, lsi_host,
                        ws.data_ptr(), ws_bytes, torch.cuda.current_stream(value.device).cuda_stream)
                _vah.check(rc, 'vah_msda_fused_backward_tiled')
                return grad_value, None, None, d_off, d_logit, None
        grad_value = torch.zeros(value.shape, dtype=torch.float32, device=value.device)
        pull = ctx.pull if not ctx.tiled else None
        with torch.cuda.device(value.device):
            rc = _vah.lib.vah_msda_fused_backward(
                value.data_ptr(), _DT[value.dtype], shapes.data_ptr(), lsi.data_ptr(),
                offsets.data_ptr(), logits.data_ptr(), _DT[offsets.dtype], ref.data_ptr(),
                ref.shape[1], grad_output.data_ptr(), N, S, M, D, L, Lq, P, grad_value.data_ptr(),
                d_off.data_ptr(), d_logit.data_ptr(),
                pull.tile_meta.data_ptr() if pull else None, pull.cand.data_ptr() if pull else None,
                pull.ntiles if pull else 0, pull.radius if pull else -1.0, pull.cap if pull else 0,
                torch.cuda.current_stream(value.device).cuda_stream)
        _vah.check(rc, 'vah_msda_fused_backward')
        return grad_value.to(value.dtype), None, None, d_off, d_logit, None
