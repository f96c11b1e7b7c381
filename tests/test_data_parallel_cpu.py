"""CPU, world_size 2 over gloo: the data-parallel path (vitadapter.data_parallel: batch sharding +
gradient averaging, by DDP's buckets or by one flat all-reduce) gives the single-process gradients of the full batch.

No GPU in this tier, so (for this test only) deformable attention inside the product modules is
routed to the oracle's torch restatement and SyncBatchNorm runs in eval mode (torch's
SyncBatchNorm has no CPU collective path); everything else is the product's wiring."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _build():
    for p in (ROOT, os.path.join(ROOT, 'vit-adapter_amd')):
        if p not in sys.path:
            sys.path.insert(0, p)
    import ops.modules.ms_deform_attn as mod
    from oracle import backbone_cases as bc
    from oracle import msda as oracle_msda
    from oracle import seeded

    class _OracleFunction:
        @staticmethod
        def apply(value, shapes, lsi, loc, attn, step):
            return oracle_msda.core_torch(value, shapes, loc, attn)

    mod.MSDeformAttnFunction = _OracleFunction
    from vitadapter.backbones import ViTAdapter
    case = bc.FULL_CASES['seg_glob_64']
    model = ViTAdapter(**case['cfg'])
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict(seeded.seeded_state_dict(shapes, 5))
    from vitadapter.data_parallel import revert_sync_batchnorm
    model = revert_sync_batchnorm(model)       # DDP refuses SyncBatchNorm on CPU modules
    model.eval()
    x = seeded.randn('dp/x', (4, 3, 64, 64), 3)
    return model, x


def _loss(model, x):
    return sum(f.float().mean() for f in model(x))


def _worker(rank, world, port, out_dir, mode):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.set_num_threads(2)
    model, x = _build()
    from vitadapter import data_parallel as dp
    r, lr, w = dp.init_from_env(backend='gloo')
    assert (r, w) == (rank, world)
    xs = dp.shard(x, rank, world)
    assert xs.shape[0] == 2
    if mode == 'ddp':
        net = dp.wrap(model)
        assert isinstance(net, torch.nn.parallel.DistributedDataParallel)
        _loss(net, xs).backward()
    else:                                   # one flat all-reduce after the backward (what bench.py runs)
        dp.broadcast_parameters(model)
        sync = dp.FlatGradSync(model.parameters())
        _loss(model, xs).backward()
        before = {k: p.grad for k, p in model.named_parameters() if p.grad is not None}
        sync()
        flat_base = next(iter(before.values()))
        assert all(p.grad is not before[k] and p.grad.shape == p.shape and p.grad._base is not None
                   for k, p in model.named_parameters() if p.grad is not None), 'gradients are views of the flat buffer'
        del flat_base
    t = dp.max_over_ranks(float(rank + 1), torch.device('cpu'))
    assert t == float(world)
    grads = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
    torch.save(grads, os.path.join(out_dir, 'grads_rank%d.pt' % rank))
    dp.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize('mode', ['ddp', 'flat'])
def test_two_rank_gloo_matches_single_process(tmp_path, mode):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), mode), nprocs=world, join=True)
    model, x = _build()
    _loss(model, x).backward()        # full batch, one process: mean over 4 images
    g0 = torch.load(os.path.join(tmp_path, 'grads_rank0.pt'), weights_only=True)
    g1 = torch.load(os.path.join(tmp_path, 'grads_rank1.pt'), weights_only=True)
    checked = 0
    for k, p in model.named_parameters():
        if p.grad is None:
            continue
        assert torch.equal(g0[k], g1[k]), k                      # all-reduced: identical on ranks
        scale = max(1.0, float(p.grad.abs().max()))
        assert float((g0[k] - p.grad).abs().max()) <= 1e-5 * scale, k
        checked += 1
    assert checked > 100
