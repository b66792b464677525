"""The reference-compatible call sequence (training_step -> loss.backward() -> Adam) against the fused
loss_and_grad step, same batches (one GPU)."""
import os, sys, time
from argparse import Namespace
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dvae_amd
dev = torch.device('cuda')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
model = dvae_amd.Dvae(Namespace(dataset_file=None, mod=None, bidirectional=False)).to(dev)
opt = torch.optim.Adam(model.parameters(), lr=1e-4, fused=True)
batches = []
for i in range(4):
    g, x = dvae_amd.synth.enas_like(B, 100 + i)
    batches.append({'graph': g.to(dev), 'node_encoding': x.to(dev)})

def api(i):
    opt.zero_grad(set_to_none=True)
    out = model.training_step(batches[i % 4], i)
    out['loss'].backward()
    opt.step()

def fused(i):
    model.loss_and_grad(batches[i % 4])
    opt.step()

def timed(fn, reps=20):
    for i in range(3): fn(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(reps): fn(i)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3

for r in range(2):
    print('B=%d training_step + backward + Adam  %.3f ms/step' % (B, timed(api)), flush=True)
    print('B=%d loss_and_grad + Adam             %.3f ms/step' % (B, timed(fused)), flush=True)
