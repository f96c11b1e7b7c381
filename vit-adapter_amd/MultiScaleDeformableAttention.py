"""Drop-in for the reference's compiled extension module ``MultiScaleDeformableAttention``.

The reference's Python does ``import MultiScaleDeformableAttention as MSDA`` and calls
``MSDA.ms_deform_attn_forward`` / ``MSDA.ms_deform_attn_backward``
(/root/reference/detection/ops/functions/ms_deform_attn_func.py:11,25,42; pybind exports
/root/reference/detection/ops/src/vision.cpp:13-16).  This module provides the same two
callables with the same argument order, checks and error behaviour, implemented as a ctypes
binding over the C ABI of libvitadapter_hip.so (include/vitadapter_hip.h).  The GPU work is
hand-written HIP for gfx950; there is no CPU or PyTorch fallback in here.

Checks mirrored from /root/reference/detection/ops/src/cuda/ms_deform_attn_cuda.cu:
  contiguity of every tensor (:28-32, :92-98), device residency (:34-38),
  ``batch % min(batch, im2col_step) == 0`` (:50-52), float/double only (:64,134).
CPU tensors raise "Not implemented on the CPU" as ms_deform_attn.h:38 does.
"""
import os
import threading

import torch

import _vah

__all__ = ['ms_deform_attn_forward', 'ms_deform_attn_backward', 'QuerySchedule',
           'build_query_schedule', 'query_schedule', 'current_schedule']


# ---------------------------------------------------------------------------------------------
# Query schedules (extension, not part of the reference's surface): a spatial grouping of the
# queries that lets the windowed kernels keep each group's value / grad_value rows in LDS.  A
# schedule never changes results; with none, the plain kernels run.
# ---------------------------------------------------------------------------------------------
class QuerySchedule:
    """perm (Lq,) int32 and group_off (n_groups+1,) int32 on the device, see
    include/vitadapter_hip.h ("Windowed variants")."""

    def __init__(self, perm, group_off, n_groups, fwd_px, bwd_px, bwd_stage, tiles=None,
                 max_group=None):
        self.perm = perm
        self.group_off = group_off
        self.n_groups = int(n_groups)
        if max_group is None:       # one host read at construction; schedules are built once and cached
            max_group = int((group_off[1:] - group_off[:-1]).max()) if group_off.numel() > 1 else 0
        self.max_group = max(1, int(max_group))
        self.Lq = int(perm.numel())
        self.fwd_px = int(fwd_px)          # 0 = use the plain forward kernel
        self.bwd_px = int(bwd_px)          # 0 = use the plain backward kernel
        self.bwd_stage = bool(bwd_stage)
        self.tiles = tiles

    def matches(self, value, Lq, D, L):
        return (self.Lq == Lq and D == 32 and L <= 4 and value.dtype == torch.float32
                and self.perm.device == value.device)


_LDS_CAP_PX = (160 * 1024 - 512) // 128          # 128-byte pixel rows that fit one workgroup's LDS


def _env_int(name, default):
    try:
        return int(os.environ.get(name, default))
    except ValueError:
        return default


def build_query_schedule(reference_points, value_shapes, halo=4.5, n_points=4):
    """Group queries by the tile of the value map their reference point falls into.

    reference_points: (1|N, Lq, L|1, 2) or (Lq, 2), (x, y) in [0,1] (only [0, :, 0] is used: the
    adapter's reference grids do not depend on the batch or the level).
    value_shapes: list of (H, W) of the value levels (host ints).
    The tile size is the largest power of two (in pixels of the finest level) whose estimated
    windows - tile + ``halo`` pixels on every side, summed over levels - fit the LDS budget."""
    ref = reference_points
    if ref.dim() == 4:
        ref = ref[0, :, 0, :]
    ref = ref.detach().float()
    Lq = ref.shape[0]
    H0, W0 = max(value_shapes, key=lambda s: s[0] * s[1])
    fwd_budget = _env_int('VAH_MSDA_FWD_PX', 600)
    bwd_budget = _env_int('VAH_MSDA_BWD_PX', 1200)

    def need(core):
        tot = 0
        for (h, w) in value_shapes:
            ch, cw = core * h / H0, core * w / W0
            tot += (min(ch + 2 * halo, h)) * (min(cw + 2 * halo, w))
        return tot

    L = len(value_shapes)
    lds_bytes = 160 * 1024 - 512

    def fits(core):
        """grad windows + the group's sample descriptors (32 B each) must fit one workgroup's LDS"""
        group = Lq * (core * core) / float(H0 * W0) * 1.1
        return need(core) * 1.15 * 128 + group * L * n_points * 32 <= lds_bytes

    core = _env_int('VAH_MSDA_TILE', 0)
    if core <= 0:
        core = 4
        while core * 2 <= max(H0, W0) and need(core * 2) <= bwd_budget and fits(core * 2):
            core *= 2
    th, tw = max(1, -(-H0 // core)), max(1, -(-W0 // core))
    ty = (ref[:, 1] * H0 / core).floor().clamp_(0, th - 1).long()
    tx = (ref[:, 0] * W0 / core).floor().clamp_(0, tw - 1).long()
    tile = ty * tw + tx
    perm = torch.argsort(tile, stable=True).to(torch.int32)
    counts = torch.bincount(tile, minlength=th * tw)
    group_off = torch.cat([counts.new_zeros(1), counts.cumsum(0)]).to(torch.int32)
    est = need(core)
    max_group = int(counts.max())
    # LDS left for windows once the group's sample descriptors (32 B / sample) are in
    cap_px = (160 * 1024 - 512 - max_group * len(value_shapes) * n_points * 32) // 128
    # forward: value windows only; backward: grad windows (+ value windows when both fit)
    fwd_px = int(min(cap_px, max(est * 1.15, 64))) if (est <= fwd_budget * 2 and cap_px >= 64) else 0
    bwd_stage = est * 1.15 * 2 <= cap_px
    bwd_px = int(min(cap_px // (2 if bwd_stage else 1), max(est * 1.15, 64))) if cap_px >= 64 else 0
    # Measured on MI355X (profiles/r01_msda_windowed_ablation.txt): ds_add_f32 retires ~0.4 lanes
    # per clock per CU, so the LDS-accumulating backward is 1.6x SLOWER than the plain kernel's
    # memory-side atomics, and the windowed forward loses to the plain gather on latency.  The
    # windowed kernels therefore stay opt-in (VAH_MSDA_WIN=1) until the atomic-free
    # (inverted-index) accumulation described in DESIGN.md replaces ds_add_f32.
    if _env_int('VAH_MSDA_WIN', 0) == 0:
        fwd_px = bwd_px = 0
    return QuerySchedule(perm.contiguous(), group_off.contiguous(), th * tw, fwd_px, bwd_px,
                         bwd_stage, tiles=(core, th, tw), max_group=max_group)


_tls = threading.local()


def current_schedule():
    return getattr(_tls, 'schedule', None)


class query_schedule:
    """``with query_schedule(s): MSDeformAttnFunction.apply(...)`` -- the Function picks the
    schedule up in forward and keeps it for its backward."""

    def __init__(self, schedule):
        self.schedule = schedule

    def __enter__(self):
        self.prev = current_schedule()
        _tls.schedule = self.schedule
        return self.schedule

    def __exit__(self, *exc):
        _tls.schedule = self.prev
        return False


def _require(cond, msg):
    if not cond:
        raise RuntimeError(msg)


def _common_checks(named, value, spatial_shapes, level_start_index, sampling_loc, attn_weight,
                   im2col_step):
    if not value.is_cuda:
        raise RuntimeError('Not implemented on the CPU')
    for name, t in named:
        _require(t.is_contiguous(), '%s tensor has to be contiguous' % name)
    for name, t in named:
        _require(t.is_cuda, '%s must be a CUDA tensor' % name)
        _require(t.device == value.device, '%s must be on the same device as value' % name)
    _require(value.dtype in (torch.float32, torch.float64),
             'ms_deform_attn: expected float32 or float64 value, got %s' % value.dtype)
    for name, t in named:
        if name in ('spatial_shapes', 'level_start_index'):
            _require(t.dtype == torch.int64, '%s must be int64' % name)
        else:
            _require(t.dtype == value.dtype, '%s must have the dtype of value' % name)
    _require(value.dim() == 4, 'value must be (N, S, M, D)')
    _require(spatial_shapes.dim() == 2 and spatial_shapes.size(1) == 2,
             'spatial_shapes must be (L, 2)')
    _require(sampling_loc.dim() == 6 and sampling_loc.size(5) == 2,
             'sampling_loc must be (N, Lq, M, L, P, 2)')
    N, S, M, D = value.shape
    L = spatial_shapes.size(0)
    Lq, P = sampling_loc.size(1), sampling_loc.size(4)
    _require(level_start_index.numel() == L, 'level_start_index must be (L,)')
    _require(tuple(sampling_loc.shape) == (N, Lq, M, L, P, 2),
             'sampling_loc shape %s does not match value/spatial_shapes' % (tuple(sampling_loc.shape),))
    _require(tuple(attn_weight.shape) == (N, Lq, M, L, P),
             'attn_weight shape %s does not match sampling_loc' % (tuple(attn_weight.shape),))
    step = min(N, int(im2col_step))
    _require(N == 0 or (step > 0 and N % step == 0),
             'batch(%d) must divide im2col_step(%d)' % (N, step))
    return N, S, M, D, L, Lq, P


def _stream_and_guard(t):
    dev = t.device
    stream = torch.cuda.current_stream(dev).cuda_stream
    return stream, (torch.cuda.device(dev) if torch.cuda.current_device() != dev.index else None)


def ms_deform_attn_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight,
                           im2col_step, schedule=None):
    """-> output (N, Lq, M*D).  Same contract as the reference's ms_deform_attn_forward;
    ``schedule`` (optional QuerySchedule) selects the LDS-windowed kernel."""
    named = (('value', value), ('spatial_shapes', spatial_shapes),
             ('level_start_index', level_start_index), ('sampling_loc', sampling_loc),
             ('attn_weight', attn_weight))
    N, S, M, D, L, Lq, P = _common_checks(named, value, spatial_shapes, level_start_index,
                                          sampling_loc, attn_weight, im2col_step)
    out = torch.empty((N, Lq, M * D), dtype=value.dtype, device=value.device)
    fn = _vah.lib.vah_msda_forward_f32 if value.dtype == torch.float32 else _vah.lib.vah_msda_forward_f64
    windowed = schedule is not None and schedule.fwd_px > 0 and schedule.matches(value, Lq, D, L)
    stream, guard = _stream_and_guard(value)
    if guard is not None:
        guard.__enter__()
    try:
        if windowed:
            rc = _vah.lib.vah_msda_forward_win_f32(
                value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(),
                sampling_loc.data_ptr(), attn_weight.data_ptr(), schedule.group_off.data_ptr(),
                schedule.perm.data_ptr(), schedule.n_groups, schedule.max_group, schedule.fwd_px,
                N, S, M, D, L, Lq, P,
                out.data_ptr(), stream)
        else:
            rc = fn(value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(),
                    sampling_loc.data_ptr(), attn_weight.data_ptr(), N, S, M, D, L, Lq, P,
                    out.data_ptr(), stream)
    finally:
        if guard is not None:
            guard.__exit__(None, None, None)
    _vah.check(rc, 'ms_deform_attn_forward')
    return out


def ms_deform_attn_backward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight,
                            grad_output, im2col_step, schedule=None):
    """-> [grad_value, grad_sampling_loc, grad_attn_weight] (shapes of the inputs)."""
    named = (('value', value), ('spatial_shapes', spatial_shapes),
             ('level_start_index', level_start_index), ('sampling_loc', sampling_loc),
             ('attn_weight', attn_weight), ('grad_output', grad_output))
    N, S, M, D, L, Lq, P = _common_checks(named, value, spatial_shapes, level_start_index,
                                          sampling_loc, attn_weight, im2col_step)
    _require(grad_output.numel() == N * Lq * M * D, 'grad_output must be (N, Lq, M*D)')
    grad_loc = torch.empty_like(sampling_loc)
    grad_attn = torch.empty_like(attn_weight)
    fn = _vah.lib.vah_msda_backward_f32 if value.dtype == torch.float32 else _vah.lib.vah_msda_backward_f64
    windowed = schedule is not None and schedule.bwd_px > 0 and schedule.matches(value, Lq, D, L)
    # atomic-free tiled grad_value (csrc/msda_tile.hip): fp32, D == 32, P == 4, L <= 4, levels inside [0, S)
    tiled = None
    if (not windowed and value.dtype == torch.float32 and D == 32 and P == 4 and 1 <= L <= 4 and N * Lq * M > 0
            and os.environ.get('VAH_MSDA_TILED', '1') != '0'):
        sh_host, lsi_host, _ = _vah.host_geometry(spatial_shapes, level_start_index)
        ws_bytes = _vah.lib.vah_msda_tile_ws_bytes(N, S, M, L, Lq, P, sh_host, lsi_host)
        if ws_bytes >= 0:
            tiled = (sh_host, lsi_host, torch.empty(ws_bytes, dtype=torch.uint8, device=value.device), ws_bytes)
    # the tile pass stores every element of grad_value; the scatter kernels accumulate into zeros
    grad_value = torch.empty_like(value) if tiled else torch.zeros_like(value)
    stream, guard = _stream_and_guard(value)
    if guard is not None:
        guard.__enter__()
    try:
        if tiled:
            rc = _vah.lib.vah_msda_backward_tiled_f32(
                value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(),
                sampling_loc.data_ptr(), attn_weight.data_ptr(), grad_output.data_ptr(),
                N, S, M, D, L, Lq, P, grad_value.data_ptr(), grad_loc.data_ptr(), grad_attn.data_ptr(),
                tiled[0], tiled[1], tiled[2].data_ptr(), tiled[3], stream)
        elif windowed:
            rc = _vah.lib.vah_msda_backward_win_f32(
                value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(),
                sampling_loc.data_ptr(), attn_weight.data_ptr(), grad_output.data_ptr(),
                schedule.group_off.data_ptr(), schedule.perm.data_ptr(), schedule.n_groups,
                schedule.max_group, schedule.bwd_px, 1 if schedule.bwd_stage else 0,
                N, S, M, D, L, Lq, P,
                grad_value.data_ptr(), grad_loc.data_ptr(), grad_attn.data_ptr(), stream)
        else:
            rc = fn(value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(),
                    sampling_loc.data_ptr(), attn_weight.data_ptr(), grad_output.data_ptr(),
                    N, S, M, D, L, Lq, P, grad_value.data_ptr(), grad_loc.data_ptr(),
                    grad_attn.data_ptr(), stream)
    finally:
        if guard is not None:
            guard.__exit__(None, None, None)
    _vah.check(rc, 'ms_deform_attn_backward')
    return [grad_value, grad_loc, grad_attn]
