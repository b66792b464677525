import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from argparse import Namespace
import dvae_amd
from dvae_amd import _lib
from helpers import fill_weights
lib = _lib.load()
cuda = torch.device('cuda')
hp = Namespace(dataset_file=None, mod=None, bidirectional=False, num_classes=8, max_seq=8, hidden_state_size=501)
model = fill_weights(dvae_amd.Dvae(hp), 3).to(cuda)
g, x = dvae_amd.synth.random_dags(96, 8, 8, 0.35, seed=21, chain=True)
eps = torch.randn(96, 56, generator=torch.Generator().manual_seed(4)).to(cuda)
X = {'graph': g.to(cuda), 'node_encoding': x.to(cuda)}
def step():
    model.zero_grad(set_to_none=True)
    out = model.loss_and_grad(X, eps=eps)
    return out.clone(), {n: p.grad.clone() for n, p in model.named_parameters()}
lib.dvae_level_tune(-1, 0); a = step(); a2 = step()
lib.dvae_level_tune(0, 0); b = step()
print(a[0], b[0])
for n in a[1]:
    d = (a[1][n] - b[1][n]).abs().max().item(); d2 = (a[1][n] - a2[1][n]).abs().max().item()
    print('%-40s max|ref| %.3e  fused diff %.3e  rerun diff %.3e' % (n, a[1][n].abs().max().item(), d, d2))
