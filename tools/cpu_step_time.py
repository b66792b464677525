"""the CPU baseline (oracle/dense_ref.py train step) must take the reference's own time.
Runs in the BUILD container only (imports the real reference from /root/reference, as
tests/golden/make_golden.py does): train step = forward + backward + Adam at c1 (B=32, N=8, uni) on the
same weights and batch, `torch.set_num_threads(8)`."""
import os, sys, time
from argparse import Namespace
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import make_golden as MG            # installs the two arithmetic-free stand-ins and the reference path
from helpers import fill_weights, oracle_sd
from oracle import dense_ref as O
import dvae_amd
torch.set_num_threads(8)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
g, x = dvae_amd.synth.enas_like(B, 1)
hp = Namespace(dataset_file=None, mod=None, bidirectional=False, num_classes=8, max_seq=8)
ref = fill_weights(MG.RefDvae(hp), 1)
sd = oracle_sd(ref, requires_grad=True)
X = {'graph': g, 'node_encoding': x}
opt_r = torch.optim.Adam(ref.parameters(), lr=1e-4)
opt_o = torch.optim.Adam(list(sd.values()), lr=1e-4)
eps = torch.randn(B, 56)
def ref_step():
    opt_r.zero_grad(); ref.training_step(X, 0)['loss'].backward(); opt_r.step()
def ora_step():
    opt_o.zero_grad(); O.train_loss(sd, g, x, eps, None, False)[0].backward(); opt_o.step()
def timed(fn, n=5):
    fn(); t0 = time.perf_counter()
    for _ in range(n): fn()
    return (time.perf_counter() - t0) / n
tr, to = timed(ref_step), timed(ora_step)
tr2, to2 = timed(ref_step), timed(ora_step)
tr, to = min(tr, tr2), min(to, to2)
print('B=%d  reference %.3f s/step (%.1f DAGs/s)   oracle %.3f s/step (%.1f DAGs/s)   oracle/reference %.2f' % (
    B, tr, B / tr, to, B / to, to / tr))
