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
