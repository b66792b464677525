"""Throughput of BASELINE.json's other configurations on one GPU (per-GPU shard sizes).
Not the headline bench (bench.py); numbers go to DESIGN.md / profiles/."""
import os, sys, time
from argparse import Namespace
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dvae_amd
dev = torch.device('cuda')


def timed(fn, warm=3, reps=8):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def train_case(name, B, N, C, mod, bidir, make):
    model = dvae_amd.Dvae(Namespace(dataset_file=None, mod=mod, bidirectional=bidir, num_classes=C, max_seq=N)).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-4, fused=True)
    X = {k: v.to(dev) for k, v in make().items()}
    e, depth = dvae_amd.synth.graph_stats(X['graph'].cpu()) if N <= 16 else (float(X['graph'].sum() / B), float('nan'))

    def step():
        model.loss_and_grad(X)
        opt.step()
    dt = timed(step)
    print('%-44s B=%5d  %8.2f ms/step  %10.0f DAGs/s  (mean E %.1f, depth %.1f)' % (name, B, dt * 1e3, B / dt, e, depth), flush=True)


def main():
    S = dvae_amd.synth
    train_case('c1 ENAS-8 (reference CPU config)', 32, 8, 8, None, False,
               lambda: dict(zip(('graph', 'node_encoding'), S.enas_like(32, 1))))
    train_case('c2 bn 10-node, bidir, batch 1024', 1024, 10, 10, 'bn', True,
               lambda: dict(zip(('graph', 'node_encoding'), S.random_dags(1024, 10, 10, 0.3, 2, fix_orphans=False))))
    train_case('c3 64-node p=0.1, 512/GPU shard', 512, 64, 8, None, False,
               lambda: dict(zip(('graph', 'node_encoding'), S.random_dags(512, 64, 8, 0.1, 3))))
    train_case('c3 64-node p=0.1, batch 4096 on ONE GPU', 4096, 64, 8, None, False,
               lambda: dict(zip(('graph', 'node_encoding'), S.random_dags(4096, 64, 8, 0.1, 3))))
    train_case('c4 ragged 8-128 bidir, batch 512', 512, 128, 8, None, True,
               lambda: dict(zip(('graph', 'node_encoding', 'sizes'), S.ragged_dags(512, 8, 4))))
    model = dvae_amd.Dvae(Namespace(dataset_file=None, mod=None, bidirectional=False)).to(dev)
    for M in (12500, 100000):
        z = torch.randn(M, 56, device=dev)
        u = (torch.rand(M, 8, device=dev), torch.rand(M, 8, 8, device=dev))
        dt = timed(lambda: model.decoder.predict(z), 1, 3)
        print('%-44s M=%6d  %8.2f ms  %10.0f DAGs/s' % ('c5 greedy decode', M, dt * 1e3, M / dt), flush=True)
        dt = timed(lambda: model.decoder.predict(z, uniforms=u), 1, 3)
        print('%-44s M=%6d  %8.2f ms  %10.0f DAGs/s' % ('c5 stochastic decode', M, dt * 1e3, M / dt), flush=True)


if __name__ == '__main__':
    main()
