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


def balanced_ranges(cost, world):
    """Contiguous shards [lo, hi) per rank with near-equal summed `cost` (1-D tensor, one entry per
    graph): boundary k sits where the running cost passes k/world of the total.  For ragged batches the
    work of a graph grows with its nodes + edges, so equal COUNTS leave ranks idle at the all-reduce
    (SURVEY.md 8e)."""
    n = int(cost.numel())
    cum = torch.cumsum(cost.double().flatten().cpu(), 0)
    total = float(cum[-1]) if n else 0.0
    bounds = [0]
    for k in range(1, world):
        if total > 0:
            target = total * k / world
            b = int(torch.searchsorted(cum, torch.tensor(target, dtype=torch.float64)))   # cum[b] >= target
            # cut before or after graph b, whichever lands closer to the target
            before = float(cum[b - 1]) if b > 0 else 0.0
            if b < n and float(cum[b]) - target <= target - before:
                b += 1
        else:
            b = n * k // world
        # boundaries stay ordered, and every rank keeps at least one graph when there are enough
        b = max(b, bounds[-1] + (1 if n >= world else 0))
        bounds.append(min(b, n - (world - k) if n >= world else n))
    bounds.append(n)
    return [(bounds[r], bounds[r + 1]) for r in range(world)]


def graph_cost(batch):
    """Per-graph work estimate of the level-synchronous step: nodes + edges (rows that enter a GEMM)."""
    g = batch['graph']
    nodes = batch['sizes'].to(g.device).double() if batch.get('sizes') is not None else torch.full(
        (g.shape[0],), float(g.shape[1]), dtype=torch.float64, device=g.device)
    return nodes + g.double().sum(dim=(1, 2))


def shard_batch(batch, rank, world, balance=None):
    """This rank's contiguous shard of a batch dict.  balance: None = by cost when the batch is ragged
    (has 'sizes'), by count otherwise; True/False force it."""
    n = batch['graph'].shape[0]
    if balance is None:
        balance = batch.get('sizes') is not None
    lo, hi = balanced_ranges(graph_cost(batch), world)[rank] if balance else shard_range(n, rank, world)
    return {k: (v[lo:hi] if torch.is_tensor(v) and v.dim() > 0 and v.shape[0] == n else v) for k, v in batch.items()}


def sharded_map(fn, dist, *tensors, group=None):
    """Decode-style data parallelism (BASELINE config 5: 100k latent samples over 8 GPUs): rows are
    independent, so every rank applies `fn` to its contiguous shard of the inputs' rows and the
    results are all-gathered back in row order.  `fn` returns a tensor or a tuple of tensors whose
    first dimension is the shard's row count.  No other collective."""
    n = tensors[0].shape[0]
    if dist is None or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return fn(*tensors)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    lo, hi = shard_range(n, rank, world)
    out = fn(*[t[lo:hi] for t in tensors])
    single = torch.is_tensor(out)
    outs = (out,) if single else tuple(out)
    cap = shard_range(n, 0, world)[1]          # the largest shard
    gathered = []
    for o in outs:
        pad = torch.zeros((cap,) + tuple(o.shape[1:]), dtype=o.dtype, device=o.device)
        pad[:hi - lo] = o
        parts = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(parts, pad, group=group)
        rows = [parts[r][:shard_range(n, r, world)[1] - shard_range(n, r, world)[0]] for r in range(world)]
        gathered.append(torch.cat(rows, 0))
    return gathered[0] if single else tuple(gathered)


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


def predict_sharded(decoder, z, dist, is_stochastic=False, uniforms=None, group=None):
    """`Decoder.predict` over the GPUs of one node: every rank decodes its contiguous shard of the
    latent rows `z` (and of the injected uniforms, if any) and receives all (edges, onehots) in row
    order.  Stochastic decode without injected uniforms draws them per rank on the device."""
    if uniforms is None:
        return sharded_map(lambda zz: decoder.predict(zz, is_stochastic=is_stochastic), dist, z, group=group)
    return sharded_map(lambda zz, ut, ue: decoder.predict(zz, is_stochastic=is_stochastic, uniforms=(ut, ue)),
                       dist, z, uniforms[0], uniforms[1], group=group)
