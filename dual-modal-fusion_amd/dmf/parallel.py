"""Data-parallel plumbing.
Training: equal contiguous shards of every global batch, ONE flat fp32 gradient exchanged per step (dmf/xgmi.py, or an
all-reduce(sum) over the process group: RCCL over xGMI on the GPUs, gloo in the CPU tests), scaled 1/N.
Evaluation / colouring: the pixel list is cut into near-equal contiguous shards, every rank classifies its shard, the
K x K int64 confusion matrix is all-reduced (sum) and label-map tiles are merged by an all-reduce(max) (every pixel is
written by exactly one rank, unwritten pixels are 0).
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


def shard_range(n, rank, world):
    """[lo, hi) of rank's contiguous shard of n items, sizes differing by at most one (evaluation: any n)."""
    per, rem = divmod(n, world)
    lo = rank * per + min(rank, rem)
    return lo, lo + per + (1 if rank < rem else 0)


def allreduce_sum_(t, group=None):
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def allreduce_max_(t, group=None):
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return t
