"""Data-parallel harness of the backbone: one process per GPU, gradients averaged over RCCL (backend "nccl" on ROCm) -
one flat all-reduce per step (FlatGradSync, what bench.py runs) or torch DDP's overlapped buckets (wrap) -, SyncBatchNorm
statistics exchanged per layer.

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


class FlatGradSync:
    """Gradient averaging as ONE collective per step: after the backward every gradient is gathered into one flat buffer
    (one multi-tensor copy), all-reduced in place (average), and the parameters' ``.grad`` are re-pointed at their slices
    of it (no copy back; the optimizer reads the views).

    Why not DDP here: its bucket views make autograd COPY each of the ~300 gradients into a bucket (this code's weight
    gradients come straight out of GEMMs as fresh tensors that AccumulateGrad would otherwise just keep), its all-reduces
    sit on ProcessGroupNCCL's side stream, and a captured step with that many cross-stream edges is launched in pieces -
    measured with a one-rank RCCL group on ViT-Adapter-B 1024^2: 26.7 ms without a group, 29.9 ms under DDP, see DESIGN 6.
    xGMI is point to point and ring collectives are bound per link: one 400 MB all-reduce uses every link for its whole
    duration, which is what the hardware guide asks for (few, large collectives).  What this form gives up is the overlap
    of communication with the backward.  ``torch.nn.parallel.DistributedDataParallel`` (``wrap``) stays available.

    Every rank must build its model from the same seed or call ``broadcast_parameters`` first."""

    def __init__(self, parameters, group=None):
        self.params = [p for p in parameters if p.requires_grad]
        self.group = group

    def __call__(self):
        world = dist.get_world_size(self.group)
        by_dtype = {}
        for p in self.params:
            if p.grad is not None:
                by_dtype.setdefault((p.grad.dtype, p.grad.device), []).append(p)
        for params in by_dtype.values():
            grads = [p.grad.reshape(-1) for p in params]
            flat = torch.cat(grads)
            if dist.get_backend(self.group) == 'nccl':
                dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=self.group)
            else:                                   # gloo has no AVG
                dist.all_reduce(flat, group=self.group)
                flat.div_(world)
            off = 0
            for p, g in zip(params, grads):
                n = g.numel()
                p.grad = flat[off:off + n].view_as(p)
                off += n


def broadcast_parameters(module, src=0, group=None):
    """Parameters and buffers of rank ``src`` to every rank (what DDP's constructor does)."""
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t, src=src, group=group)


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
