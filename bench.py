#!/usr/bin/env python
"""bench.py -- ViT-Adapter-B 1024x1024 training step throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W          (N > 1: starts its own N ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One process per GPU; data parallel over RCCL (torch DDP, gradient all-reduce overlapped with
backward, SyncBatchNorm statistics exchanged per layer).  A "step" is one full pass of the hot
path over one synthetic batch: zero_grad -> ViTAdapter forward under bf16 autocast (the fused
MSDeformAttn core runs on bf16 values / offsets / logits with fp32 accumulation, the ViT attention on
bf16 MFMA tiles) -> loss = sum_k mean(f_k) -> backward (+ gradient all-reduce) -> fused AdamW
update.  Inputs are resident in HBM before the timed region.

One GPU (default `--hip-graph auto`): after the warm-up ONE step is captured into a HIP graph, three eager steps and
three replays are timed (still warm-up), and the timed region runs the faster form: K replays (0.3 ms of host time per
step) where the host is the longer pole - slow-CPU boxes, the 24-block presets - or the eager loop where the GPU is (it
overlaps the weight-gradient GEMMs on a side stream, a captured graph does not).  HIP events cannot be read back from
inside a graph, so after a replayed region the per-kernel rows are taken from K more EAGER steps of the same process
(`config.hip_graph`, `config.hip_graph_probe`, `config.kernel_rows`).  N > 1 (round 3): the same - the captured step then
holds the gradient all-reduce (`--dp-mode flat`, default: ONE flat all-reduce after the backward; `ddp`: torch DDP's buckets,
constructed and warmed up for 11 steps on the side stream as torch asks) and the SyncBatchNorm statistics all-reduces (RCCL
collectives are capturable); the ranks agree before the first replay, so a
rank whose capture failed sends everyone to the eager loop.  `--hip-graph off` / `VAH_DP_GRAPH=0` time the eager loop with
the events inside the timed region.

Rank 0 prints ONE JSON line with the contract fields plus
  "roofline":     the MSDA entry point with the largest total time in the timed region: bytes the
                  launch must move FOR THE DTYPES IT RUNS WITH / mean launch duration (HIP events
                  recorded by the library on the launch stream during the timed steps) against the
                  8 TB/s HBM3E peak; the op's fp32-definition bytes are reported
                  beside it as `frac_fp32_definition`, never as `frac`
  "kernels":      every timed entry point: the fused MSDA core and the ViT attention kernels
                  (bound "mfma": 4*B*heads*N^2*64 flops per forward launch against the 2.5 PFLOP/s
                  dense bf16 peak) inside the timed region, and - measured after it, same process -
                  the reference-precision boundary kernels (plain fp32 MSDeformAttnFunction
                  forward / backward at the BASELINE call shapes: cfg3 injector, cfg3 extractor, cfg1)
  "cpu_baseline": the oracle restatement of the same backbone (oracle/vit_adapter_ref.py, kind
                  "port") run fwd+bwd on the host cores on a bounded sample (one image), plus the
                  reference's pure-PyTorch MSDA core on BASELINE config 1.
"""
import argparse
import json
import os

os.environ.setdefault('MIOPEN_FIND_MODE', '2')     # UPerHead row: MIOpen convolutions of new shapes without an exhaustive search
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'vit-adapter_amd'))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK = 8.0e12     # B/s, MI355X_MICROARCH.md "HBM3E peak BW" (spec)
MFMA_PEAK = 2.5e15    # flop/s, dense bf16 MFMA (same guide; the 5 PF headline includes 2:1 sparsity)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--preset', default='base_det', help='vitadapter preset (base_det = BASELINE configs[2])')
    ap.add_argument('--size', type=int, nargs=2, default=[1024, 1024], metavar=('H', 'W'))
    ap.add_argument('--hip-graph', default='auto', choices=['auto', 'on', 'off'],
                    help='capture one whole step (zero_grad, forward, backward, AdamW - and, with a process group, DDP\'s bucketed '
                         'all-reduces and the SyncBatchNorm statistics all-reduces: RCCL collectives are capturable) into a HIP '
                         'graph after the warm-up and time K replays of it (0.3 ms of host time per step instead of 20-70 ms); the '
                         'per-kernel rows then come from K more EAGER steps right after the timed region, because HIP events '
                         'cannot be read back from inside a graph.  auto: replay if the capture succeeds ON EVERY RANK and three '
                         'replays beat three eager steps (maximum over the ranks) during the warm-up, else the eager loop; '
                         'VAH_DP_GRAPH=0 keeps steps with collectives eager; off: always the eager loop with the events inside '
                         'the timed region')
    ap.add_argument('--no-checkpoint', action='store_true',
                    help='presets with with_cp=True (the large models: the reference recomputes activations to fit 32 GB '
                         'cards): keep the activations instead - 288 GB of HBM holds them, same arithmetic')
    ap.add_argument('--batch', type=int, default=2, help='per-GPU batch')
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'fp32'])
    ap.add_argument('--cpu-baseline', default='auto', choices=['auto', 'full', 'small', 'none'],
                    help="auto: 'full' (one 1024^2 image) at N=1 on rank 0, none otherwise")
    ap.add_argument('--no-optimizer', action='store_true', help='diagnostic only: skip AdamW')
    ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'],
                    help="'gloo' lets several ranks share one GPU for a functional rehearsal of the "
                         "N>1 path on a 1-GPU box (not a performance mode)")
    ap.add_argument('--dp-mode', default='flat', choices=['flat', 'ddp'],
                    help='N > 1: gradient averaging as ONE flat all-reduce after the backward (default; few, large collectives '
                         'for point-to-point xGMI, no per-gradient bucket copies, a captured step with one collective on a side '
                         'stream instead of one per bucket) or torch DDP with 64 MB buckets overlapped with the backward')
    ap.add_argument('--family-steps', type=int, default=5,
                    help='untimed eager steps after the timed region that put every library entry point on the `kernels` list '
                         '(GEMMs by role with MFMA fractions, LayerNorm / residual / column-sum / conv / BatchNorm-tail rows); 0: off')
    ap.add_argument('--profile-kernels', default='msda_,attn_',
                    help='HIP-event timing inside the timed region for the library entry points whose name '
                         'starts with one of these comma-separated prefixes ("" = all of them, costs ~2 %%; '
                         '"none" = no events)')
    ap.add_argument('--boundary-iters', type=int, default=20,
                    help='iterations of the fp32 boundary-kernel measurement after the timed region (0 = skip)')
    ap.add_argument('--consumer-iters', type=int, default=5,
                    help='iterations of the pixel-decoder encoder / UPerHead rows after the timed region (0 = skip)')
    ap.add_argument('--mock-step', action='store_true',
                    help='CPU rehearsal of the launch path only (rank spawn, process group over gloo, barrier, '
                         'max-over-ranks, the one JSON line) around a toy step; used by tests/test_bench_launch_cpu.py, '
                         'never a measurement')
    ap.add_argument('--gemm-tuning', default='file', choices=['file', 'off', 'tune'],
                    help="hipBLASLt/rocBLAS solution selection through torch TunableOp: 'file' loads the "
                         "committed selections (no tuning at run time), 'tune' re-tunes and rewrites them")
    return ap.parse_args()


def host_cores():
    """CPU share of this process: cgroup quota if set, else the affinity mask (never the raw
    os.cpu_count() of a shared host, which oversubscribes the box and runs ~100x slower)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, int(os.environ.get("VAH_CPU_THREADS", "16"))))


TUNING_FILE = os.path.join(ROOT, 'vit-adapter_amd', 'tuning', 'tunableop_gemm_mi355x_base_det_1024.csv')


def setup_gemm_tuning(args):
    """The Linear layers run on hipBLASLt / rocBLAS through torch.  Their default heuristic picks
    slow tiles for several of this workload's shapes; torch's TunableOp selections, tuned once on an
    MI355X and committed (vit-adapter_amd/tuning/), are replayed here (57.7 vs 60.9 ms per step).
    Shapes that are not in the file keep the library default."""
    if args.gemm_tuning == 'off':
        return
    import torch.cuda.tunable as tunable
    tunable.enable(True)
    if args.gemm_tuning == 'tune':
        tunable.tuning_enable(True)
        tunable.set_max_tuning_duration(30)
        tunable.set_filename(TUNING_FILE)
    else:
        tunable.tuning_enable(False)
        if hasattr(tunable, 'write_file_on_exit'):
            tunable.write_file_on_exit(False)
        if os.path.exists(TUNING_FILE):
            tunable.read_file(TUNING_FILE)


FAMILY_PREFIXES = 'gemm_,colsum_,layernorm,residual_,scale_residual,dwconv_,bn_,conv_,gelu_,maxpool,transpose,pixel_shuffle,patchify,relpos,spm_'

MSDA_SOURCES = ('msda.hip', 'msda_fused.hip', 'msda_tile.hip', 'msda_fwd_win.hip', 'msda_common.h')


def msda_source_digest():
    """sha1 over the MSDA kernel sources: stamps a PMC pass with the code it measured."""
    import hashlib
    h = hashlib.sha1()
    for f in MSDA_SOURCES:
        path = os.path.join(ROOT, 'vit-adapter_amd', 'csrc', f)
        if os.path.exists(path):
            h.update(open(path, 'rb').read())
    return h.hexdigest()[:16]


def pmc_traffic(kernel, args, preset_kw):
    """HBM bytes per launch of the dominant MSDA entry point from the committed PMC pass
    (profiles/r03_msda_pmc.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs of
    tools/prof_msda_single.py, gfx950 FETCH_SIZE x2 correction, written by tools/pmc_msda_summary.py
    together with the digest of the kernel sources it measured).  Only for the exact shapes that
    pass measured (BASELINE configs[2]: 4 injector + 6 extractor calls per direction and step) and
    only while the kernel sources are the ones it measured; None (-> "traffic": null) otherwise."""
    if args.preset != 'base_det' or list(args.size) != [1024, 1024] or args.batch != 2:
        return None
    path = os.path.join(ROOT, 'profiles', 'r03_msda_pmc.json')
    try:
        pmc = json.load(open(path))
        if pmc.get('msda_source_digest') != msda_source_digest():
            return None
        d = 'fwd' if 'fwd' in kernel else 'bwd'
        inj = pmc['cfg3_inj_' + d]['hbm_bytes_per_launch']
        ext = pmc['cfg3_ext_' + d]['hbm_bytes_per_launch']
        return int((4 * inj + 6 * ext) / 10)
    except (OSError, KeyError, ValueError):
        return None


# MSDeformAttnFunction call shapes of the BASELINE configs:
# name -> (N, M, D, P, Lq, value level shapes, query grid shapes)
BOUNDARY_SHAPES = {
    'cfg3_inj': (2, 12, 32, 4, 4096, [(128, 128), (64, 64), (32, 32)], [(64, 64)]),
    'cfg3_ext': (2, 12, 32, 4, 21504, [(64, 64)], [(128, 128), (64, 64), (32, 32)]),
    'cfg1': (1, 8, 32, 4, 5440, [(64, 64), (32, 32), (16, 16), (8, 8)], [(64, 64), (32, 32), (16, 16), (8, 8)]),
}


def boundary_inputs(cfg, dev, seed=0):
    """fp32 inputs of one plain MSDeformAttnFunction call, "adapter" locations: the pixel-centre
    reference grid of the query maps + (ring bias (p+1)*dir_m + N(0,1) px) / (W_l, H_l)
    (ref ops/modules/ms_deform_attn.py:64-75,117-119; adapter_modules.py:13-25)."""
    import math
    N, M, D, P, Lq, shapes, qshapes = BOUNDARY_SHAPES[cfg]
    L, S = len(shapes), sum(h * w for h, w in shapes)
    g = torch.Generator(device=dev).manual_seed(seed)
    hw = torch.as_tensor(shapes, dtype=torch.long, device=dev)
    lsi = torch.cat((hw.new_zeros((1,)), hw.prod(1).cumsum(0)[:-1]))
    value = torch.randn(N, S, M, D, device=dev, generator=g)
    attn = torch.softmax(torch.randn(N, Lq, M, L * P, device=dev, generator=g), -1).view(N, Lq, M, L, P)
    pts = []
    for h, w in qshapes:
        ys = (torch.arange(h, dtype=torch.float32, device=dev) + 0.5) / h
        xs = (torch.arange(w, dtype=torch.float32, device=dev) + 0.5) / w
        gy, gx = torch.meshgrid(ys, xs, indexing='ij')
        pts.append(torch.stack((gx.reshape(-1), gy.reshape(-1)), -1))
    ref = torch.cat(pts, 0)
    assert ref.shape[0] == Lq
    th = torch.arange(M, dtype=torch.float32, device=dev) * (2.0 * math.pi / M)
    d = torch.stack([th.cos(), th.sin()], -1)
    d = (d / d.abs().max(-1, keepdim=True)[0]).view(M, 1, 1, 2).repeat(1, L, P, 1)
    d = d * torch.arange(1, P + 1, dtype=torch.float32, device=dev).view(1, 1, P, 1)
    off = d[None, None] + torch.randn(N, Lq, M, L, P, 2, device=dev, generator=g)
    loc = ref[None, :, None, None, None, :] + off / hw.flip(-1).float()[None, None, None, :, None, :]
    gout = torch.randn(N, Lq, M * D, device=dev, generator=g)
    return value, hw, lsi, loc.contiguous(), attn.contiguous(), gout


def boundary_kernels(dev, iters):
    """The reference-precision boundary: plain fp32 ms_deform_attn_forward / _backward
    of the drop-in extension module at the BASELINE call shapes, timed with the library's HIP events
    on the launch stream.  This is BASELINE's "MSDeformAttn GB/s vs HBM peak" at the precision of
    ref detection/ops/functions/ms_deform_attn_func.py:21 (custom_fwd casts to fp32)."""
    import MultiScaleDeformableAttention as MSDA
    import _vah
    rows = {}
    for cfg in BOUNDARY_SHAPES:
        v, s, i, l, a, g = boundary_inputs(cfg, dev)
        for _ in range(3):
            MSDA.ms_deform_attn_forward(v, s, i, l, a, 64)
            MSDA.ms_deform_attn_backward(v, s, i, l, a, g, 64)
        torch.cuda.synchronize()
        _vah.prof_enable(True, 'msda_fwd_f32,msda_bwd_f32')
        for _ in range(iters):
            MSDA.ms_deform_attn_forward(v, s, i, l, a, 64)
            MSDA.ms_deform_attn_backward(v, s, i, l, a, g, 64)
        torch.cuda.synchronize()
        _vah.prof_enable(False)
        for name, r in _vah.prof_report().items():
            if r['calls']:
                rows['%s[%s]' % (name, cfg)] = kernel_row(r, 1)
        del v, s, i, l, a, g
    return rows


def consumer_rows(dev, iters):
    """The two consumers of the backbone's maps that BASELINE configs[3] / configs[4] name, as rows beside the backbone's
    (they are not part of the timed step: the metric is the backbone's): the 6-layer deformable encoder of Mask2Former's
    pixel decoder at configs[4]'s per-GPU shapes (800 x 1344, batch 1: levels 25x42, 50x84, 100x168, 256 channels, 8
    heads; vitadapter/pixel_decoder.py) and UPerHead at configs[3]'s (640 x 640, batch 2: four 1024-channel maps at
    strides 4..32, 512 channels inside, 150 classes; vitadapter/heads.py), forward + backward under bf16 autocast, timed
    with HIP events on the launch stream; the MSDA launches inside the encoder with the library's own events."""
    import _vah
    from vitadapter.heads import UPerHead
    from vitadapter.pixel_decoder import MSDeformAttnEncoder, encoder_inputs
    rows = {}

    def timed(fn, n):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n

    torch.manual_seed(0)
    enc = MSDeformAttnEncoder().to(dev).train()
    with torch.no_grad():                           # the zero-initialised offsets / weights Linear as after some training
        for layer in enc.layers:
            layer.attentions[0].sampling_offsets.weight.normal_(0, 0.02)
            layer.attentions[0].attention_weights.weight.normal_(0, 0.05)
    shapes = [(25, 42), (50, 84), (100, 168)]
    query, pos, ref, ss, lsi = encoder_inputs(shapes, 1, 256, dev)
    query.requires_grad_(True)

    def enc_step():
        enc.zero_grad(set_to_none=True)
        with torch.autocast('cuda', dtype=torch.bfloat16):
            out = enc(query=query, query_pos=pos, spatial_shapes=ss, reference_points=ref, level_start_index=lsi)
        out.float().mean().backward()

    ms = timed(enc_step, iters)
    _vah.prof_enable(True, 'msda_')
    for _ in range(iters):
        enc_step()
    torch.cuda.synchronize()
    _vah.prof_enable(False)
    rows['pixel_decoder_encoder[cfg4]'] = {'what': '6 layers, Lq = S = %d, batch 1, 256 ch, 8 heads, 3 levels, fwd+bwd, bf16 autocast'
                                           % sum(h * w for h, w in shapes), 'ms': round(ms, 3)}
    for name, r in _vah.prof_report().items():
        if r['calls']:
            rows['%s[cfg4_pixdec]' % name] = kernel_row(r, iters)
    del enc, query, pos, ref
    head = UPerHead(in_channels=(1024,) * 4, channels=512, num_classes=150).to(dev).train()
    feats = [torch.randn(2, 1024, 160 >> i, 160 >> i, device=dev, requires_grad=True) for i in range(4)]

    def head_step():
        head.zero_grad(set_to_none=True)
        with torch.autocast('cuda', dtype=torch.bfloat16):
            out = head(feats)
        out.float().mean().backward()

    rows['uper_head[cfg3]'] = {'what': 'UPerHead 4 x 1024 ch in (160..20 px), 512 ch, 150 classes, batch 2, fwd+bwd, bf16 autocast '
                                       '(MIOpen / hipBLASLt convolutions)', 'ms': round(timed(head_step, iters), 3)}
    return rows


def kernel_row(r, steps):
    """One `kernels` entry from a library profile record (see _vah.prof_report)."""
    avg_s = r['total_ms'] * 1e-3 / r['calls']
    row = {'calls_per_step': r['calls'] / steps, 'avg_us': round(avg_s * 1e6, 2),
           'ms_per_step': round(r['total_ms'] / steps, 3)}
    if r['flops']:
        fl = r['flops'] / r['calls']
        row.update({'bound': 'mfma', 'flops_per_launch': int(fl),
                    'achieved_TFLOPs': round(fl / avg_s / 1e12, 1),
                    'frac_of_mfma_peak': round(fl / avg_s / MFMA_PEAK, 4)})
    else:
        per_launch, per_def = r['bytes'] / r['calls'], r['def_bytes'] / r['calls']
        row.update({'bound': 'hbm', 'bytes_per_launch': int(per_launch),
                    'achieved_GBps': round(per_launch / avg_s / 1e9, 1),
                    'frac_of_hbm_peak': round(per_launch / avg_s / HBM_PEAK, 4)})
        if per_def != per_launch:
            row.update({'fp32_definition_bytes_per_launch': int(per_def),
                        'frac_fp32_definition': round(per_def / avg_s / HBM_PEAK, 4)})
    return row


def cpu_baseline(args, preset_kw):
    """Oracle (CPU port of the reference) on the host cores; bounded sample, same workload."""
    from oracle import cases, msda as oracle_msda, seeded, vit_adapter_ref as ref
    cores = host_cores()
    torch.set_num_threads(cores)
    out = {'cores': cores, 'kind': 'port', 'unit': 'images/sec'}

    # (a) the reference's pure-PyTorch MSDA core on BASELINE config 1 (north-star CPU leg)
    N, M, D, P, Lq, shapes, _ = cases.bench_inputs('cfg1')
    L, S = len(shapes), sum(h * w for h, w in shapes)
    g = torch.Generator().manual_seed(0)
    value = (torch.rand(N, S, M, D, generator=g) * 0.01).requires_grad_(True)
    loc = torch.rand(N, Lq, M, L, P, 2, generator=g).requires_grad_(True)
    attn = torch.rand(N, Lq, M, L, P, generator=g) + 1e-5
    attn = (attn / attn.sum((-1, -2), keepdim=True)).requires_grad_(True)
    tf, tfb = [], []
    for it in range(2 + 5):
        t0 = time.perf_counter()
        o = oracle_msda.core_torch(value, shapes, loc, attn)
        t1 = time.perf_counter()
        o.sum().backward()
        t2 = time.perf_counter()
        if it >= 2:
            tf.append(t1 - t0)
            tfb.append(t2 - t0)
    fb = 4 * (N * S * M * D + 3 * N * Lq * M * L * P + N * Lq * M * D)
    out['msda_core_cfg1'] = {'fwd_ms': round(1e3 * sorted(tf)[len(tf) // 2], 2),
                             'fwd_bwd_ms': round(1e3 * sorted(tfb)[len(tfb) // 2], 2),
                             'fwd_GBps': round(fb / sorted(tf)[len(tf) // 2] / 1e9, 3)}

    # (b) the whole backbone step on one image
    mode = args.cpu_baseline
    H, W = args.size if mode == 'full' else (256, 256)
    from vitadapter.backbones import ViTAdapter
    kw = dict(preset_kw)
    shapes_sd = {k: tuple(v.shape) for k, v in ViTAdapter(**kw).state_dict().items()}
    sd = seeded.seeded_state_dict(shapes_sd, 0)
    for k, v in sd.items():
        if v.is_floating_point() and 'running_' not in k:
            v.requires_grad_(True)
    cfg = ref.Cfg(**kw)
    x = torch.randn(1, 3, H, W, generator=g)
    t0 = time.perf_counter()
    outs = ref.vit_adapter_forward(sd, x, cfg, training=True)
    sum(o.mean() for o in outs).backward()
    dt = time.perf_counter() - t0
    out['value'] = round(1.0 / dt, 4)
    out['sample'] = ('oracle ViT-Adapter (%s) fwd+bwd, fp32, 1 image %dx%d, 1 pass, %.1f s'
                     % (args.preset, H, W, dt))
    if (H, W) != tuple(args.size):
        out['sample'] += ' (reduced size: not the bench workload)'
    return out


def spawn_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no launcher environment: start N ranks as a child
    torch.distributed.run job (ref segmentation/dist_train.sh:8-9 does the same with
    torch.distributed.launch) and relay its output.  Called before anything touches the GPU - a
    process that has initialised HIP must not exec, and this one never does: it only waits."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    return subprocess.run(cmd, env=env).returncode


def main_mock(args):
    """--mock-step: everything bench.py does around the step, on CPU over gloo, with a toy step."""
    from vitadapter import data_parallel as dp
    rank, _, world = dp.init_from_env('gloo')
    assert world == args.gpus, '--gpus %d but the launcher started %d rank(s)' % (args.gpus, world)
    dev = torch.device('cpu')
    torch.manual_seed(0)
    net = dp.wrap(torch.nn.Linear(8, 8), dev)
    x = torch.randn(args.batch, 8, generator=torch.Generator().manual_seed(1234 + rank))
    for _ in range(args.warmup):
        net(x).sum().backward()
    dp.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        net(x).sum().backward()
    dp.barrier()
    dt = dp.max_over_ranks(time.perf_counter() - t0, dev)
    if rank == 0:
        print(json.dumps({'metric': 'mock', 'value': round(world * args.batch * args.steps / dt, 3), 'unit': 'images/sec',
                          'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'data': 'mock',
                          'config': {'workload': 'mock step (launch-path rehearsal on CPU)', 'parallelism': 'dp%d' % world,
                                     'hip_graph': False,
                       'rccl_ranks': dist.get_world_size() if dist.is_initialized() else 1,
                                     'backend': 'gloo'}}), flush=True)
    if world > 1:
        dp.barrier()
        dist.destroy_process_group()


def main():
    args = parse()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(spawn_ranks(args))
    if args.mock_step:
        return main_mock(args)
    assert torch.cuda.is_available(), 'bench.py needs a GPU'
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device('cuda', dev_index)

    import _vah
    from vitadapter import data_parallel as dp
    from vitadapter.backbones.vit_adapter import PRESETS, build_preset

    rank, local_rank, world = dp.init_from_env(args.backend)
    assert world == args.gpus, '--gpus %d but the launcher started %d rank(s)' % (args.gpus, world)

    setup_gemm_tuning(args)
    preset_kw = dict(PRESETS[args.preset])
    torch.manual_seed(0)
    model = build_preset(args.preset, **({'with_cp': False} if args.no_checkpoint else {})).to(dev).train()
    n_params = sum(p.numel() for p in model.parameters())
    # A process group (N > 1, or one rank with VAH_ONE_RANK_GROUP=1): the step contains DDP's bucketed all-reduces and the
    # SyncBatchNorm statistics all-reduces.  RCCL collectives are capturable; torch asks for the DDP constructor AND >= 11
    # warm-up steps on the stream the capture will use.  VAH_DP_GRAPH=0: never capture a step that has collectives.
    grouped = dist.is_initialized()
    dp_graph = grouped and args.backend == 'nccl' and os.environ.get('VAH_DP_GRAPH', '1') != '0'
    use_graph = args.hip_graph != 'off' and not args.no_optimizer and (not grouped or dp_graph)
    side = torch.cuda.Stream() if use_graph else None
    sync = None
    if grouped and args.dp_mode == 'flat':
        # one flat all-reduce per step after the backward (data_parallel.FlatGradSync); every rank seeds its model alike
        net, sync = model, dp.FlatGradSync(model.parameters())
    elif grouped and use_graph:
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            net = dp.wrap(model, dev, bucket_cap_mb=int(os.environ.get('VAH_DDP_BUCKET_MB', '64')))
        torch.cuda.current_stream().wait_stream(side)
    else:
        net = dp.wrap(model, dev, bucket_cap_mb=64)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-5, weight_decay=0.05, fused=True, capturable=use_graph)
    amp = torch.bfloat16 if args.dtype == 'bf16' else None

    H, W = args.size
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    x = torch.randn(args.batch, 3, H, W, device=dev, generator=gen)

    def step():
        opt.zero_grad(set_to_none=True)
        with torch.autocast('cuda', dtype=amp, enabled=amp is not None):
            feats = net(x)
        loss = sum(f.float().mean() for f in feats)
        loss.backward()
        if sync is not None:
            sync()
        if not args.no_optimizer:
            opt.step()
        return loss

    def fence():
        dp.barrier()
        torch.cuda.synchronize()

    graph = None
    probe = None
    if use_graph:
        # warm-up on a side stream (allocator pools, GEMM algorithm choices), then one captured step
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(args.warmup, 11 if grouped else 3)):
                step()
        torch.cuda.current_stream().wait_stream(side)
        fence()

        def agree(flag):
            """Every rank replays or none does: a rank whose capture failed must not leave the others inside a collective."""
            if not grouped:
                return flag
            t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            return bool(t.item())

        def wall(fn, n=3):
            torch.cuda.synchronize()
            p0 = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - p0) / n

        try:
            # auto: which loop is faster HERE is measured, three steps each, still inside the warm-up.  Replay wins where
            # the host is the longer pole (slow-CPU boxes, the 24-block presets); the eager loop wins where the GPU is
            # (it overlaps the weight-gradient GEMMs on a side stream, which a captured graph does not) and keeps the
            # HIP events inside the timed region.  The eager probe runs BEFORE the capture: replaying a graph after
            # eager steps have released tensors it captured (gradients, bf16 weight copies) is not safe.
            eager_ms = 1e3 * wall(step) if args.hip_graph == 'auto' else None
            # the parameters' AccumulateGrad nodes were made on the warm-up stream; the capture stream differs by design
            if hasattr(torch.autograd.graph, 'set_warn_on_accumulate_grad_stream_mismatch'):
                torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
            graph = torch.cuda.CUDAGraph()
            opt.zero_grad(set_to_none=True)
            captured, why = True, None
            mode = 'global'
            if grouped:
                # ProcessGroupNCCL's watchdog thread polls the events of the eager collectives enqueued so far; an event
                # query from ANY thread while a capture in the default (global) mode is open is an error that aborts the
                # process (seen once in ~6 one-rank runs: WorkNCCL::finishedGPUExecutionInternal from Watchdog::runLoop).
                # So: everything enqueued is finished and the watchdog has had time to retire it before the capture opens,
                # and the capture only polices its own thread.
                fence()
                time.sleep(1.0)
                mode = 'thread_local'
            try:
                with torch.cuda.graph(graph, capture_error_mode=mode):
                    graph_loss = step()
            except Exception as exc:                   # noqa: BLE001 - told to the other ranks below, then handled
                captured, why = False, exc
            torch.cuda.synchronize()
            if not agree(captured):
                raise why if why is not None else RuntimeError('another rank could not capture the step')
            fence()
            if eager_ms is not None:
                probe = {'eager_ms': round(dp.max_over_ranks(eager_ms, dev), 3),
                         'graph_ms': round(dp.max_over_ranks(1e3 * wall(graph.replay), dev), 3)}
                if probe['eager_ms'] <= probe['graph_ms']:          # the same numbers on every rank: the same decision
                    graph, use_graph = None, False          # no replay from here on
        except Exception as exc:                       # noqa: BLE001 - any capture failure: the eager loop below
            if args.hip_graph == 'on':
                raise
            print('bench.py: HIP graph capture failed (%s: %s), timing the eager loop' % (type(exc).__name__, exc), file=sys.stderr)
            graph, use_graph = None, False
            torch.cuda.synchronize()
    if graph is not None:
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            graph.replay()
        host_dt = time.perf_counter() - t0
        fence()
        dt = time.perf_counter() - t0
        loss = graph_loss
        # per-kernel rows: HIP events cannot be read back from inside a graph, so K eager steps, untimed
        _vah.prof_enable(True, args.profile_kernels)
        for _ in range(args.steps):
            step()
        fence()
        _vah.prof_enable(False)
    else:
        for _ in range(args.warmup):
            step()
        fence()
        _vah.prof_enable(True, args.profile_kernels)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss = step()
        host_dt = time.perf_counter() - t0           # the host has enqueued every step; the GPU may still be running
        fence()
        dt = time.perf_counter() - t0
        _vah.prof_enable(False)
    prof = _vah.prof_report()
    assert torch.isfinite(loss).item(), 'loss is not finite'
    # the rest of the step by operator family (55 % of the step had no row): a few more eager steps, untimed,
    # with HIP events on every library entry point of the families below (event pairs on ~900 launches per step cost
    # host time, so never inside the timed region)
    fam_prof = {}
    if args.family_steps > 0 and world == 1:
        _vah.prof_enable(True, FAMILY_PREFIXES)
        for _ in range(args.family_steps):
            step()
        fence()
        _vah.prof_enable(False)
        fam_prof = _vah.prof_report()

    dt = dp.max_over_ranks(dt, dev)

    if rank == 0:
        ips = world * args.batch * args.steps / dt
        kernels = {name: kernel_row(r, args.steps) for name, r in prof.items() if r['calls']}
        msda = {k: v for k, v in kernels.items() if k.startswith('msda_')}
        roofline = None
        if msda:
            dom = max(msda, key=lambda k: msda[k]['ms_per_step'])
            a = msda[dom]['achieved_GBps']
            roofline = {'kernel': dom, 'bound': 'hbm', 'achieved': a, 'peak': HBM_PEAK / 1e9,
                        'unit': 'GB/s', 'frac': round(a / (HBM_PEAK / 1e9), 4),
                        'traffic': pmc_traffic(dom, args, preset_kw),
                        'avg_launch_us': msda[dom]['avg_us'],
                        'algorithmic_bytes_per_launch': msda[dom]['bytes_per_launch'],
                        'bytes_are': 'moved for the IO dtypes of the launch: value, out / grad_out and grad_value in %s (the tile pass '
                                     'stores grad_value in the value dtype); under autocast the offsets / logits rows are the fp32 '
                                     'output of their GEMM read in place, their gradients bf16 rows' % args.dtype,
                        'frac_fp32_definition': msda[dom].get('frac_fp32_definition')}
        for name, r in fam_prof.items():
            if r['calls'] and name not in kernels:
                kernels[name] = kernel_row(r, args.family_steps)
        gemm = [r for n_, r in fam_prof.items() if n_.startswith('gemm_') and r['calls']]
        if gemm:        # the Linear layers as one family: all matrix-core flops over all their launch time (split-K reductions included)
            fl, ms = sum(r['flops'] for r in gemm), sum(r['total_ms'] for r in gemm)
            kernels['gemm_family'] = {'calls_per_step': sum(r['calls'] for r in gemm) / args.family_steps,
                                      'ms_per_step': round(ms / args.family_steps, 3), 'bound': 'mfma',
                                      'achieved_TFLOPs': round(fl / (ms * 1e-3) / 1e12, 1),
                                      'frac_of_mfma_peak': round(fl / (ms * 1e-3) / MFMA_PEAK, 4)}
        if args.boundary_iters > 0 and world == 1:
            kernels.update(boundary_kernels(dev, args.boundary_iters))
            if args.consumer_iters > 0:
                kernels.update(consumer_rows(dev, args.consumer_iters))
        line = {
            'metric': ('images/sec ViT-Adapter-B 1024x1024 fwd+bwd (+AdamW step)' if (args.preset, H, W) == ('base_det', 1024, 1024)
                       else 'images/sec %s %dx%d fwd+bwd (+AdamW step)' % (args.preset, H, W)),
            'value': round(ips, 3), 'unit': 'images/sec', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(1e3 * dt / args.steps, 3),
            'host_enqueue_ms_per_step': round(1e3 * host_dt / args.steps, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': args.dtype, 'data': 'synthetic',
            'config': {'workload': 'ViT-Adapter-%s (%s) %dx%d, per-GPU batch %d, train mode, '
                                   'drop_path %.1f, SyncBN, %s' % (
                                       args.preset, preset_kw['flavour'], H, W, args.batch,
                                       preset_kw['drop_path_rate'],
                                       ('step = fwd+bwd' if args.no_optimizer else 'step = fwd+bwd+AdamW')
                                       + (', activations kept (with_cp off)' if args.no_checkpoint else '')),
                       'global_batch': world * args.batch, 'params_M': round(n_params / 1e6, 2),
                       'parallelism': 'dp%d' % world, 'dp_mode': (args.dp_mode if dist.is_initialized() else None),
                       'hip_graph': use_graph, 'hip_graph_probe': probe,
                       'kernel_rows': ('HIP events over %d eager steps right after the timed region (events cannot be read '
                                       'back from inside a graph)' % args.steps) if use_graph else 'HIP events inside the timed region',
                       'rccl_ranks': dist.get_world_size() if dist.is_initialized() else 1,
                       'backend': args.backend if dist.is_initialized() else None,
                       # hipBLASLt algorithms the dispatcher's whole-output check refused (csrc/gemm.hip)
                       'gemm_candidates_rejected': int(_vah.lib.vah_gemm_rejected_candidates())},
            'roofline': roofline,
            'kernels': kernels,
        }
        mode = args.cpu_baseline
        if mode == 'auto':
            mode = 'full' if world == 1 else 'none'
        if mode != 'none':
            args.cpu_baseline = mode
            line['cpu_baseline'] = cpu_baseline(args, preset_kw)
        print(json.dumps(line), flush=True)
    if world > 1:
        dp.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
