"""Data-parallel plumbing: equal contiguous shards of every global batch, ONE flat fp32 gradient
all-reduced (sum) per step over the process group (RCCL over xGMI on the GPUs; gloo in the CPU tests), scaled 1/N.
The reference has no distributed code at all; this is the only parallelism the path needs (patches are independent,
the update is the only coupling)."""
import torch.distributed as dist


def shard_batch(n, rank, world):
    """[lo, hi) of rank's contiguous shard of a global batch of n (n must divide evenly: equal shards keep the
    mean-of-means equal to the global CrossEntropyLoss mean, utils/utils.py:29)."""
    if n % world:
        raise ValueError('global batch %d is not divisible by world size %d' % (n, world))
    per = n // world
    return rank * per, (rank + 1) * per


def allreduce_mean_(flat_grad, group=None):
    dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=group)
    flat_grad.mul_(1.0 / dist.get_world_size(group))
    return flat_grad
