"""Encoder forward: one 4096-graph batch on one stream vs two 2048-graph halves on two streams."""
import os, sys, time
from argparse import Namespace
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dvae_amd
from dvae_amd.engine import side_stream
dev = torch.device('cuda')
model = dvae_amd.Dvae(Namespace(dataset_file=None, mod=None, bidirectional=False)).to(dev)
g, x = dvae_amd.synth.enas_like(4096, 1)
full = {'graph': g.to(dev), 'node_encoding': x.to(dev)}
h0 = {'graph': g[:2048].to(dev), 'node_encoding': x[:2048].to(dev)}
h1 = {'graph': g[2048:].to(dev), 'node_encoding': x[2048:].to(dev)}
enc = model.encoder
side = side_stream(dev)
main = torch.cuda.current_stream()
s_full = model._schedule(full) if hasattr(model, '_schedule') else None
s0 = model._schedule(h0); s1 = model._schedule(h1)
torch.cuda.synchronize()

def one():
    with torch.no_grad():
        enc(full, sched=s_full)

def two():
    with torch.no_grad():
        side.wait_stream(main)
        enc(h0, sched=s0)
        with torch.cuda.stream(side):
            enc(h1, sched=s1)
        main.wait_stream(side)

def timed(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3

def half_only():
    with torch.no_grad():
        enc(h0, sched=s0)

for r in range(2):
    print('full batch, one stream   %.3f ms' % timed(one), flush=True)
    print('two halves, two streams  %.3f ms' % timed(two), flush=True)
    print('one half alone           %.3f ms' % timed(half_only), flush=True)
