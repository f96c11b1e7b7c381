"""CPU: the host schedule of the LDS-window MSDA forward (ops/functions/ms_deform_attn_fused.py::build_window_schedule):
every query is in exactly one group, the groups are the 8 x 8-pixel tiles of the value map in raster order, and a
group's window holds every pixel within halo + 1 of its tile (so a sample whose offset stays within `halo` pixels has
all four bilinear corners inside the window; the kernel reads the others from global memory)."""
import pytest
import torch

from ops.functions.ms_deform_attn_fused import build_window_schedule
from oracle import cases


@pytest.mark.parametrize('H,W,qshapes,halo', [(64, 64, [(128, 128), (64, 64), (32, 32)], 5), (40, 40, [(80, 80), (40, 40), (20, 20)], 5),
                                               (50, 84, [(100, 168), (50, 84), (25, 42)], 3), (7, 9, [(14, 18)], 2)])
def test_window_schedule_invariants(H, W, qshapes, halo):
    ref = cases.reference_grid(qshapes)                     # (1, Lq, 1, 2) as the adapter builds it
    Lq = ref.shape[1]
    s = build_window_schedule(ref, H, W, start=0, tile=8, halo=halo)
    perm = s.perm.long()
    assert sorted(perm.tolist()) == list(range(Lq))                         # a permutation: each query once
    off = s.group_off.long()
    nty, ntx = (H + 7) // 8, (W + 7) // 8
    assert s.ngroups == nty * ntx and off.numel() == s.ngroups + 1
    assert off[0] == 0 and off[-1] == Lq and bool((off[1:] >= off[:-1]).all())
    xy = ref.reshape(-1, ref.shape[-2], 2)[:, 0]
    px, py = (xy[:, 0] * W), (xy[:, 1] * H)
    for g in range(s.ngroups):
        gy, gx = divmod(g, ntx)
        y0, x0, wh, ww = s.group_win[g].tolist()
        # the window: tile + halo + 1 on every side, clipped to the map
        assert y0 == max(gy * 8 - halo - 1, 0) and x0 == max(gx * 8 - halo - 1, 0)
        assert y0 + wh == min(gy * 8 + 8 + halo + 1, H) and x0 + ww == min(gx * 8 + 8 + halo + 1, W)
        assert wh * ww <= s.max_win_px
        q = perm[off[g]:off[g + 1]]
        if q.numel():
            # the group's queries have their reference point in the tile (last tile row / column takes the clamped rest)
            ty = (py[q] / 8).floor().clamp(0, nty - 1)
            tx = (px[q] / 8).floor().clamp(0, ntx - 1)
            assert bool((ty == gy).all()) and bool((tx == gx).all())
            # all four corners of a sample displaced by <= halo pixels lie inside the window (or outside the map)
            for d in (-halo, halo):
                yy = (py[q] - 0.5 + d).floor()
                xx = (px[q] - 0.5 + d).floor()
                for c in (0, 1):
                    ok_y = ((yy + c < 0) | (yy + c >= H) | ((yy + c >= y0) & (yy + c < y0 + wh)))
                    ok_x = ((xx + c < 0) | (xx + c >= W) | ((xx + c >= x0) & (xx + c < x0 + ww)))
                    assert bool(ok_y.all()) and bool(ok_x.all())
