"""world_size-2 gloo test of the data-parallel rule: contiguous batch shards +
SUM all-reduce of the flat gradient reproduce the single-process gradient of the
concatenated batch.  The per-rank compute here is the CPU oracle;
the sharding / collective code is the product's (dvae_amd/dp.py)."""
import os
import sys
from argparse import Namespace

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import dvae_amd
    from dvae_amd import dp
    from helpers import fill_weights, oracle_sd
    from oracle import dense_ref as O
    torch.set_num_threads(1)
    hp = Namespace(dataset_file=None, mod=None, bidirectional=True, num_classes=5, max_seq=6, hidden_state_size=24)
    model = fill_weights(dvae_amd.Dvae(hp), 7)
    g, x = dvae_amd.synth.random_dags(10, 6, 5, 0.4, 3)
    eps = torch.randn(10, 56, generator=torch.Generator().manual_seed(0))   # drawn per GLOBAL graph index
    full = {'graph': g, 'node_encoding': x, 'eps': eps}
    mine = dp.shard_batch(full, rank, world)
    sd = oracle_sd(model, requires_grad=True)
    out = O.train_loss(sd, mine['graph'], mine['node_encoding'], mine['eps'], None, True)
    out[0].backward()
    keys = [k for k, _ in model.named_parameters()]
    flat = torch.cat([sd[k].grad.flatten() for k in keys])
    sync = dp.make_grad_sync(dist)
    sync(flat).wait()
    losses = dp.allreduce_losses(dist, torch.stack([out[1], out[2], out[3], out[0]]).detach())
    if rank == 0:
        sd1 = oracle_sd(model, requires_grad=True)
        ref = O.train_loss(sd1, g, x, eps, None, True)
        ref[0].backward()
        flat1 = torch.cat([sd1[k].grad.flatten() for k in keys])
        ret['grad_err'] = float((flat - flat1).abs().max() / flat1.abs().max())
        ret['loss_err'] = float(abs(losses[3] - ref[0]) / abs(ref[0]))
    dist.barrier()
    dist.destroy_process_group()


def test_dp_sum_allreduce_equals_big_batch():
    world = 2
    port = 29500 + os.getpid() % 2000
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
        assert ret['grad_err'] < 1e-5, ret['grad_err']
        assert ret['loss_err'] < 1e-6, ret['loss_err']


def test_balanced_ranges():
    """Ragged batches are split by summed nodes + edges, contiguously, every rank non-empty."""
    sys.path.insert(0, ROOT)
    from dvae_amd import dp
    cost = torch.tensor([1.0, 1, 1, 1, 10, 1, 1, 1, 1, 10])
    r = dp.balanced_ranges(cost, 2)
    assert r[0][0] == 0 and r[-1][1] == 10 and r[0][1] == r[1][0]
    s0, s1 = float(cost[r[0][0]:r[0][1]].sum()), float(cost[r[1][0]:r[1][1]].sum())
    assert abs(s0 - s1) <= 10.0 and min(s0, s1) >= 8.0
    # more ranks than cost mass: still ordered, covering, non-empty
    r = dp.balanced_ranges(torch.tensor([100.0, 1, 1, 1]), 4)
    assert [hi - lo for lo, hi in r] == [1, 1, 1, 1]
    r = dp.balanced_ranges(torch.tensor([1.0, 1, 1, 100]), 4)
    assert [hi - lo for lo, hi in r] == [1, 1, 1, 1]
    g = torch.rand(64, generator=torch.Generator().manual_seed(0)) * 100
    r = dp.balanced_ranges(g, 8)
    sums = [float(g[lo:hi].sum()) for lo, hi in r]
    assert r[0][0] == 0 and r[-1][1] == 64 and all(a[1] == b[0] for a, b in zip(r, r[1:]))
    assert max(sums) - min(sums) <= 2.0 * float(g.max())
    # the batch helper: ragged -> by cost, dense -> by count
    import dvae_amd
    gg, xx, sizes = dvae_amd.synth.ragged_dags(12, 4, 3, nmin=3, nmax=20)
    full = {'graph': gg, 'node_encoding': xx, 'sizes': sizes}
    parts = [dp.shard_batch(full, k, 3) for k in range(3)]
    assert sum(p['graph'].shape[0] for p in parts) == 12
    assert torch.equal(torch.cat([p['sizes'] for p in parts]), sizes)
    costs = [float(dp.graph_cost(p).sum()) for p in parts]
    assert max(costs) - min(costs) <= float(dp.graph_cost(full).max()) * 2
    dense = {'graph': gg, 'node_encoding': xx}
    assert [dp.shard_batch(dense, k, 3)['graph'].shape[0] for k in range(3)] == [4, 4, 4]


def _map_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dvae_amd import dp
    z = torch.arange(7 * 3, dtype=torch.float32).view(7, 3)          # 7 rows over 2 ranks: 4 + 3
    seen = []

    def fn(zz):
        seen.append(zz.shape[0])
        return zz * 2.0, zz.sum(1).long()
    a, b = dp.sharded_map(fn, dist, z)
    one = dp.sharded_map(lambda zz: zz + 1.0, dist, z)
    ret[rank] = (seen[0], bool(torch.equal(a, z * 2.0)), bool(torch.equal(b, z.sum(1).long())),
                 bool(torch.equal(one, z + 1.0)))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_map_gathers_in_row_order():
    world = 2
    port = 31500 + os.getpid() % 2000
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_map_worker, args=(world, port, ret), nprocs=world, join=True)
        assert ret[0] == (4, True, True, True), ret[0]
        assert ret[1] == (3, True, True, True), ret[1]


class _StubModel(torch.nn.Module):
    """Lightning-style hooks over a 1-parameter model: loss = w^2 * (graphs in the shard) / 2."""

    def __init__(self, w0, data):
        super().__init__()
        self.w = torch.nn.Parameter(torch.tensor([float(w0)]))
        self.data = data
        self.seen = []

    def training_step(self, batch, i):
        self.seen.append(int(batch['graph'].shape[0]))
        loss = (self.w ** 2).sum() * batch['graph'].shape[0] / 2
        v = float(loss)
        return {'loss': loss, 'log': {'train/edge_loss': v, 'train/vertex_loss': 0.0, 'train/kl_loss': 0.0, 'train/loss': v}}

    def configure_optimizers(self):
        opt = torch.optim.SGD(self.parameters(), lr=0.01)
        return [opt], [torch.optim.lr_scheduler.ReduceLROnPlateau(opt)]

    def train_dataloader(self):
        return self.data

    def val_dataloader(self):
        return None


def _trainer_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dvae_amd import trainer
    torch.manual_seed(100 + rank)          # ranks start with DIFFERENT generator states and weights
    # 5 graphs, 4 graphs, then a tail batch of ONE graph: fewer graphs than ranks
    data = [{'graph': torch.ones(5, 2, 2)}, {'graph': torch.ones(4, 2, 2)}, {'graph': torch.ones(1, 2, 2)}]
    m = _StubModel(3.0 + rank, data)
    t = trainer.Trainer(max_epochs=1, dist=dist, fused=False, log=lambda r: None)
    hist = t.fit(m)
    ret[rank] = (float(m.w), list(m.seen), hist[0]['steps'], torch.initial_seed() if False else int(torch.randint(0, 1 << 30, (1,))))
    dist.barrier()
    dist.destroy_process_group()


def test_trainer_data_parallel_broadcast_seed_and_tail_batch():
    """Trainer under data parallelism: rank 0's weights are broadcast, the shuffle seed is shared (ranks
    draw the same numbers afterwards), every rank takes its shard of each batch, and a tail batch with fewer
    graphs than ranks is dropped by ALL ranks instead of leaving one of them out of the all-reduce."""
    world = 2
    port = 31500 + os.getpid() % 2000
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_trainer_worker, args=(world, port, ret), nprocs=world, join=True)
        (w0, seen0, steps0, r0), (w1, seen1, steps1, r1) = ret[0], ret[1]
        assert steps0 == steps1 == 2                        # the 1-graph tail batch was dropped everywhere
        assert seen0 == [3, 2] and seen1 == [2, 2]          # contiguous shards of 5 and 4 graphs
        assert r0 == r1                                     # same generator state after the shared epoch seed
        # both started from rank 0's w = 3 and applied the SUM-reduced gradient: w <- w - lr * w * (graphs in batch)
        want = 3.0
        for n in (5, 4):
            want -= 0.01 * want * n
        assert abs(w0 - want) < 1e-6 and abs(w1 - want) < 1e-6, (w0, w1, want)
