"""How much of the headline step is the encoder?  Encoder forward + backward alone (autograd path, two streams)
against the fused step, and what fraction of the encoder's node rows a value-numbered schedule would keep."""
import os, sys, time
from argparse import Namespace
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import dvae_amd
from dvae_amd import engine as E

B, N, C = int(os.environ.get("B", 4096)), 8, 8
dev = torch.device("cuda:0")
g, x = dvae_amd.synth.enas_like(B, 1234)
# distinct (index, type, predecessor values) per node index
vals = {}
ids = torch.zeros(B, N, dtype=torch.long)
t = x.argmax(2)
for i in range(N):
    for b in range(B):
        key = (i, int(t[b, i]), tuple(int(ids[b, u]) for u in range(i - 1, -1, -1) if g[b, i, u] != 0))
        ids[b, i] = vals.setdefault(key, len(vals))
print("distinct values per node index:", [len(set(ids[:, i].tolist())) for i in range(N)], "of", B)
model = dvae_amd.Dvae(Namespace(dataset_file=None, mod=None, bidirectional=False, num_classes=C, max_seq=N)).to(dev)
g, x = g.to(dev), x.to(dev)
X = {'graph': g, 'node_encoding': x}
def enc_only():
    hv, mu, lv = model.encoder(X)
    (mu.sum() + lv.sum()).backward()
def full():
    model.loss_and_grad(X)
for name, fn in (("encoder fwd+bwd (autograd path)", enc_only), ("full fused step", full)):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): fn()
    torch.cuda.synchronize(); print(name, "%.3f ms" % ((time.perf_counter() - t0) / 30 * 1e3))
