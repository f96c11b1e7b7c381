"""Data-parallel harness of the backbone: one process per GPU, gradients all-reduced by torch
DDP over RCCL (backend "nccl" on ROCm), SyncBatchNorm statistics exchanged per layer.

The reference trains the same way (torch.distributed.launch + mm* DDP wrapper,
/root/reference/segmentation/dist_train.sh:8-9, segmentation/train.py:133,204; NCCL backend
/root/reference/segmentation/configs/_base_/default_runtime.py:9): the global batch is sharded
over the ranks, every rank runs the whole backbone, gradients are averaged.  Nothing else is
exchanged, so throughput scaling is "weak" (fixed per-GPU batch).

xGMI note: gradients travel in a few large buckets (default 64 MB) so that RCCL can spread each
all-reduce over all 7 links of a GPU; gradient_as_bucket_view avoids a copy per bucket.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Process-group setup from torchrun's environment.  Returns (rank, local_rank, world)."""
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if (world > 1 or one_rank_group()) and not dist.is_initialized():
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')     # dmabuf IPC only on this stack
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        kw = {}
        if backend == 'nccl':
            kw['device_id'] = torch.device('cuda', local_rank % max(1, torch.cuda.device_count()))
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, local_rank, world


def one_rank_group():
    """VAH_ONE_RANK_GROUP=1: build the process group, the DDP wrapper and the SyncBatchNorm collectives even for ONE rank.
    A one-GPU box can then run everything the N > 1 path runs - RCCL initialisation, DDP's bucket hooks, the statistics
    all-reduces, their capture in a HIP graph - except the traffic between GPUs (tests/test_data_parallel_gpu.py)."""
    return os.environ.get('VAH_ONE_RANK_GROUP', '0') == '1'


def wrap(model, device=None, bucket_cap_mb=64):
    """DDP wrapper (identity when there is one process)."""
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not one_rank_group()):
        return model
    ids = [device.index] if device is not None and device.type == 'cuda' else None
    return torch.nn.parallel.DistributedDataParallel(
        model, device_ids=ids, bucket_cap_mb=bucket_cap_mb, gradient_as_bucket_view=True,
        broadcast_buffers=False)


def revert_sync_batchnorm(module):
    """SyncBatchNorm -> BatchNorm2d in place of the same parameters / statistics, for runs
    without a GPU process group (the reference does this for non-distributed training,
    /root/reference/segmentation/train.py:178-183, via mmcv's helper of the same name)."""
    out = module
    if isinstance(module, torch.nn.SyncBatchNorm):
        out = torch.nn.BatchNorm2d(module.num_features, module.eps, module.momentum, module.affine,
                                   module.track_running_stats)
        if module.affine:
            out.weight, out.bias = module.weight, module.bias
        out.running_mean, out.running_var = module.running_mean, module.running_var
        out.num_batches_tracked = module.num_batches_tracked
        out.training = module.training
    for name, child in module.named_children():
        out.add_module(name, revert_sync_batchnorm(child))
    return out


def shard(batch, rank, world):
    """The slice of a global batch that belongs to ``rank`` (contiguous, equal sizes)."""
    per = batch.shape[0] // world
    assert per * world == batch.shape[0], 'global batch must divide by the world size'
    return batch[rank * per:(rank + 1) * per]


def max_over_ranks(seconds, device):
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
