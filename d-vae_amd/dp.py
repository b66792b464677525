"""Data-parallel helpers: graphs are independent, so ranks take contiguous batch
shards and the only exchange is a SUM all-reduce of the gradients (the reference's
losses are reduction='sum', dvae_model.py:405,429,431, so SUM -- not mean --
reproduces the single-device big-batch gradient) plus 4 logged loss scalars."""
import torch


def shard_range(n, rank, world):
    """Contiguous shard [lo, hi) of n items for `rank`; the first n % world ranks get one more."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_batch(batch, rank, world):
    n = batch['graph'].shape[0]
    lo, hi = shard_range(n, rank, world)
    return {k: (v[lo:hi] if torch.is_tensor(v) and v.dim() > 0 and v.shape[0] == n else v) for k, v in batch.items()}


def make_grad_sync(dist, group=None, force=False):
    """Callable for Dvae.loss_and_grad(grad_sync=...): asynchronous SUM all-reduce of one
    flat gradient buffer on whatever stream is current (RCCL on GPUs, gloo on CPU)."""
    if dist is None or not dist.is_initialized() or (dist.get_world_size(group) == 1 and not force):
        return None

    def sync(flat):
        return dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=True)
    return sync


def allreduce_losses(dist, out4, group=None):
    if dist is not None and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(out4, op=dist.ReduceOp.SUM, group=group)
    return out4
