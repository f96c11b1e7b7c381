"""GPU, 2 processes on the one card (gloo process group): the fused output tail under SyncBatchNorm
shares its statistics across ranks exactly as nn.SyncBatchNorm does - every rank's output and
gradients equal those of one process running plain BatchNorm over the whole batch."""
import os
import socket

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _make(seed, N, C, H, W, scale):
    g = torch.Generator().manual_seed(seed)
    a = torch.randn(N, C, H, W, generator=g).to(torch.bfloat16)
    b = (torch.randn(N, C, H, W, generator=g) + 0.3).to(torch.bfloat16)
    x = torch.randn(N, C, H // scale, W // scale, generator=g)
    dy = torch.randn(N, C, H, W, generator=g)
    w = torch.randn(C, generator=g) * 0.3 + 1
    bias = torch.randn(C, generator=g) * 0.3
    return a, b, x, dy, w, bias


def _worker(rank, world, port, shape, out_path):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, 'vit-adapter_amd'))
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from vitadapter import fused
        N, C, H, W, scale = shape
        a, b, x, dy, w, bias = (t.cuda() for t in _make(11, N, C, H, W, scale))
        per = N // world
        sl = slice(rank * per, (rank + 1) * per)
        bn = torch.nn.SyncBatchNorm(C).cuda().train()
        with torch.no_grad():
            bn.weight.copy_(w)
            bn.bias.copy_(bias)
        al, bl, xl = (t[sl].clone().requires_grad_(True) for t in (a, b, x))
        with torch.autocast('cuda', dtype=torch.bfloat16):
            y = fused.bn_tail(bn, al, bl, xl, scale)
        assert type(y.grad_fn).__name__ == '_BNTailBackward', 'fused path not taken'
        y.backward(dy[sl])
        torch.save(dict(y=y.detach().cpu(), da=al.grad.float().cpu(), db=bl.grad.float().cpu(), dx=xl.grad.cpu(),
                        dw=bn.weight.grad.cpu(), dbias=bn.bias.grad.cpu(), rm=bn.running_mean.cpu(),
                        rv=bn.running_var.cpu()), out_path % rank)
    finally:
        dist.destroy_process_group()


def test_bn_tail_syncbn_two_ranks(tmp_path):
    import torch.multiprocessing as mp
    shape = (4, 12, 32, 32, 4)
    world = 2
    out_path = str(tmp_path / 'rank%d.pt')
    mp.spawn(_worker, args=(world, _free_port(), shape, out_path), nprocs=world, join=True)
    N, C, H, W, scale = shape
    a, b, x, dy, w, bias = _make(11, N, C, H, W, scale)
    a2, b2, x2 = (t.float().requires_grad_(True) for t in (a, b, x))
    bn = torch.nn.BatchNorm2d(C).train()
    with torch.no_grad():
        bn.weight.copy_(w)
        bn.bias.copy_(bias)
    y = bn(a2 + b2 + F.interpolate(x2, scale_factor=scale, mode='bilinear', align_corners=False))
    y.backward(dy)
    per = N // world
    dw = torch.zeros(C)
    dbias = torch.zeros(C)
    for r in range(world):
        got = torch.load(out_path % r, weights_only=True)
        sl = slice(r * per, (r + 1) * per)
        for name, ref, tol in (('y', y.detach()[sl], 3e-5), ('da', a2.grad[sl], 1e-2), ('db', b2.grad[sl], 1e-2),
                               ('dx', x2.grad[sl], 1e-4), ('rm', bn.running_mean, 1e-5), ('rv', bn.running_var, 1e-5)):
            err = (got[name] - ref).abs().max().item()
            assert err <= tol * max(1.0, ref.abs().max().item()), (r, name, err)
        dw += got['dw']
        dbias += got['dbias']
    # weight / bias gradients are per-rank sums (DDP averages them afterwards)
    assert (dw - bn.weight.grad).abs().max().item() <= 1e-3 * bn.weight.grad.abs().max().item()
    assert (dbias - bn.bias.grad).abs().max().item() <= 1e-3 * max(1.0, bn.bias.grad.abs().max().item())
