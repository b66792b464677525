"""Feasibility probe: capture one fused train step (fixed batch, schedule built beforehand) in a HIP
graph through torch.cuda.graph and compare replay time with the eager step (tuning aid)."""
import os, sys, time
from argparse import Namespace
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dvae_amd
from dvae_amd import engine as E

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
if len(sys.argv) > 2 and sys.argv[2] == 'one':      # everything on ONE stream: a linear graph
    E.side_stream = lambda device: torch.cuda.current_stream()
    import dvae_amd.model as _m
    print('single-stream mode', flush=True)
dev = torch.device('cuda')
torch.manual_seed(0)
model = dvae_amd.Dvae(Namespace(dataset_file=None, mod=None, bidirectional=False)).to(dev)
opt = torch.optim.Adam(model.parameters(), lr=1e-4, fused=True, capturable=True)
g, x = dvae_amd.synth.enas_like(B, 1)
X = {'graph': g.to(dev), 'node_encoding': x.to(dev)}
eps = torch.randn(B, 56, device=dev)


def eager(n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        model.loss_and_grad(X, eps=eps)
        opt.step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


eager(5)
print('B=%d eager (schedule rebuilt every step): %.3f ms/step' % (B, eager(30)), flush=True)
sched = model._schedule(X)
ref = model.loss_and_grad(X, eps=eps, sched=sched).clone()
ref_g = torch.cat([p.grad.flatten() for p in model.parameters()]).clone()

s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2):
        model.loss_and_grad(X, eps=eps, sched=sched)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    out4 = model.loss_and_grad(X, eps=eps, sched=sched)
torch.cuda.synchronize()
print('captured', flush=True)
graph.replay()
torch.cuda.synchronize()
got_g = torch.cat([p.grad.flatten() for p in model.parameters()])
print('replay vs eager: loss', float(out4[3]), float(ref[3]), 'grad max abs diff', float((got_g - ref_g).abs().max()), flush=True)


def replay(n, with_opt):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        graph.replay()
        if with_opt:
            opt.step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


replay(5, False)
print('graph replay, no optimizer: %.3f ms/step' % replay(50, False), flush=True)
print('graph replay + eager fused Adam: %.3f ms/step' % replay(50, True), flush=True)
