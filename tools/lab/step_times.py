"""Lab: duration of each of the first steps of a bench.py-like run (events between steps, no host sync inside the run)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

# reuse bench.py's own set-up by timing inside its step(): monkeypatch time-critical loop via the --steps / --warmup path
N = int(sys.argv[1]) if len(sys.argv) > 1 else 80
evs = []
orig_barrier = None
import dvae_amd
from argparse import Namespace
dev = torch.device('cuda')
model = dvae_amd.Dvae(Namespace(dataset_file=None, mod=None, bidirectional=False)).to(dev)
opt = model.configure_optimizers(flat=True)[0][0] if hasattr(model, 'configure_optimizers') else None
batches = []
for i in range(4):
    g, x = dvae_amd.synth.enas_like(4096, 100 + i)
    batches.append({'graph': g.to(dev), 'node_encoding': x.to(dev)})
if len(sys.argv) > 3:                                          # learning rate override (0: the weights never move)
    for gq in opt.param_groups: gq['lr'] = float(sys.argv[3])
prepared = {}
def step(i):
    sched = prepared.pop(i, None)
    out = model.loss_and_grad(batches[i % 4], sched=sched)
    prepared[i + 1] = model.prepare(batches[(i + 1) % 4], resident=True)
    opt.step()
    return out
SPIN = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0     # seconds of plain GEMM load before the first step
for i in range(3): step(1000 + i)                               # (first-call costs out of the way)
torch.cuda.synchronize()
if SPIN > 0:
    import time
    a = torch.randn(8192, 8192, device=dev); t0 = time.time()
    while time.time() - t0 < SPIN:
        for _ in range(10): a @ a
        torch.cuda.synchronize()
elif SPIN < 0:                                                  # -n: n steps on the fp32-MFMA core in front (bench.py's comparison leg)
    from dvae_amd import _lib
    lib = _lib.load(); lib.dvae_gemm_x6(0)
    import time; time.sleep(1.0)
    for i in range(int(-SPIN)): step(2000 + i)
    torch.cuda.synchronize(); lib.dvae_gemm_x6(1); prepared.clear()
else:
    import time; time.sleep(1.0)                                # an idle second, as between set-up and the timed loop
for i in range(N):
    e = torch.cuda.Event(enable_timing=True); e.record(); evs.append(e)
    step(i)
e = torch.cuda.Event(enable_timing=True); e.record(); evs.append(e)
torch.cuda.synchronize()
t = [evs[i].elapsed_time(evs[i + 1]) for i in range(N)]
print(' '.join('%.2f' % v for v in t))
for a in range(0, N, 10): print('steps %3d-%3d mean %.3f ms' % (a, a + 9, sum(t[a:a + 10]) / 10))
