"""Where the chain spends a fused step (DVAE_PHASES=1: timing events at the phase boundaries, dvae_phase_read):
    DVAE_PHASES=1 B=512 python tools/lab/phases.py
prints the mean time between consecutive marks over the measured steps."""
import ctypes, os, sys, time
from argparse import Namespace
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import dvae_amd
from dvae_amd import engine as E

assert os.environ.get("DVAE_PHASES"), "run with DVAE_PHASES=1"
B, steps = int(os.environ.get("B", 512)), int(os.environ.get("STEPS", 50))
dev = torch.device("cuda:0")
model = dvae_amd.Dvae(Namespace(dataset_file=None, mod=None, bidirectional=False, num_classes=8, max_seq=8)).to(dev)
opt = dvae_amd.optim.FlatAdam(model)
batches = []
for k in range(4):
    g, x = dvae_amd.synth.enas_like(B, 100 + k)
    batches.append({'graph': g.to(dev), 'node_encoding': x.to(dev)})
lib = E.load()
prepared = {}
def step(i):
    sched = prepared.pop(i, None)
    model.loss_and_grad(batches[i % 4], sched=sched)
    prepared[i + 1] = model.prepare(batches[(i + 1) % 4], resident=True)
    opt.step()
NAMES = {0: "start", 1: "encoder forward", 2: "decoder sweep", 3: "wait: side add_edge products", 4: "pair scorer fwd",
         5: "wait: side add_vertex", 6: "ELBO", 7: "pair scorer bwd", 8: "add_edge input grads + graph state",
         9: "decoder bwd sweep", 10: "rest of part 1 (lin1, reparam)", 11: "encoder backward (chain)", 12: "wait: side weight grads"}
for i in range(10):
    step(i)
buf, tags = (ctypes.c_double * 4096)(), (ctypes.c_int * 4096)()
lib.dvae_phase_read(ctypes.cast(buf, ctypes.c_void_p), ctypes.cast(tags, ctypes.c_void_p), 4096)     # drop the warm-up marks
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(10, 10 + steps):
    step(i)
torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / steps * 1e3
n = lib.dvae_phase_read(ctypes.cast(buf, ctypes.c_void_p), ctypes.cast(tags, ctypes.c_void_p), 4096)
t, g = np.array(buf[:n]), np.array(tags[:n])
per = {}
starts = np.where(g == 0)[0]
for a, b in zip(starts, list(starts[1:]) + [n]):
    for k in range(a + 1, b):
        per.setdefault(int(g[k]), []).append(t[k] - t[k - 1])
    if b < n:
        per.setdefault(99, []).append(t[b] - t[b - 1])
print("B=%d: %.3f ms per step (wall, with the marks)" % (B, wall))
for k in sorted(per):
    print("  %-42s %7.1f us" % (NAMES.get(k, "between steps (Adam, python, zero, pack)"), 1e3 * float(np.mean(per[k]))))
print("  %-42s %7.1f us" % ("sum", 1e3 * sum(float(np.mean(v)) for v in per.values())))
