"""Data-parallel plumbing: one process per GPU, torch.distributed (backend 'nccl' = RCCL over xGMI).

The reference has no distributed code: training is single process / single device.
The train step shards naturally over the batch -- every sample is independent and the only batch
coupling is the reduce_mean in the loss (tf_utils.py:19) -- so each rank runs the unchanged
forward/backward on its shard and the flat fp32 gradient buffer is SUM-all-reduced, then scaled
by 1/world inside the fused Adam kernel (grad_scale).  That equals the gradient of the global-batch
mean loss; Adam then runs redundantly on every rank from identical weights.
"""
import os

import torch


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torchrun contract).
    Returns (rank, world_size, local_rank)."""
    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local_rank


def bucket_views(flat, bucket_elems):
    """Contiguous views of the flat gradient buffer, largest-first friendly: [flat[a:b], ...]."""
    n = flat.numel()
    return [flat[a:min(a + bucket_elems, n)] for a in range(0, n, bucket_elems)]


def allreduce_sum_(flat, group=None, bucket_elems=None):
    """In-place SUM all-reduce of the flat gradient buffer (optionally in buckets)."""
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return flat
    if bucket_elems is None:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    else:
        works = [dist.all_reduce(v, op=dist.ReduceOp.SUM, group=group, async_op=True) for v in bucket_views(flat, bucket_elems)]
        for w in works:
            w.wait()
    return flat


def shard_batch(global_batch, rank, world):
    """Even split of the global batch (config 4: 512 = 8 x 64).  Returns (start, stop)."""
    if global_batch % world:
        raise ValueError("global batch %d not divisible by world size %d" % (global_batch, world))
    per = global_batch // world
    return rank * per, (rank + 1) * per
