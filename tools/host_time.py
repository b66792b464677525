"""Host-side enqueue time vs GPU time of one train step (tuning aid)."""
import sys, os, time
from argparse import Namespace
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dvae_amd
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device('cuda')
model = dvae_amd.Dvae(Namespace(dataset_file=None, mod=None, bidirectional=False)).to(dev)
opt = torch.optim.Adam(model.parameters(), lr=1e-4, fused=True)
g, x = dvae_amd.synth.enas_like(B, 1)
X = {'graph': g.to(dev), 'node_encoding': x.to(dev)}
for _ in range(5):
    model.loss_and_grad(X); opt.step()
torch.cuda.synchronize()
import cProfile, pstats
host = tot = 0.0
for _ in range(20):
    t0 = time.perf_counter()
    model.loss_and_grad(X); opt.step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host += t1 - t0; tot += t2 - t0
print('B=%d host enqueue %.2f ms, total %.2f ms' % (B, host / 20 * 1e3, tot / 20 * 1e3))
pr = cProfile.Profile(); pr.enable()
for _ in range(10):
    model.loss_and_grad(X); opt.step()
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
