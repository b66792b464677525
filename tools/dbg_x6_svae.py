import os, sys
from argparse import Namespace
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dvae_amd
from dvae_amd import _lib
lib = _lib.load()
cuda = torch.device('cuda')
torch.manual_seed(0)
model = dvae_amd.Svae(Namespace(dataset_file=None, bidirectional=True)).to(cuda)
gen = torch.Generator().manual_seed(11)
B, C, N, T = 4096, 8, 8, 7
x = torch.cat((torch.nn.functional.one_hot(torch.randint(0, C, (B, T), generator=gen), C).float(),
               (torch.rand(B, T, N, generator=gen) < 0.5).float()), dim=-1).to(cuda)
eps = torch.randn(B, 56, generator=gen).to(cuda)
def run(core, shards):
    lib.dvae_gemm_x6(core)
    model.zero_grad()
    n = B // shards
    outs = []
    for s in range(shards):
        sl = slice(n * s, n * (s + 1))
        o = model.training_step({'graph': x[sl]}, 0, eps=eps[sl]); o['loss'].backward()
        outs.append(model(x[sl], eps=eps[sl])[0].detach())
    return {k: p.grad.clone() for k, p in model.named_parameters()}, torch.cat(outs)
ref, s0 = run(0, 1)
for core, sh in ((0, 4), (1, 1), (1, 4), (1, 2)):
    g, s = run(core, sh)
    print('core', core, 'shards', sh, 'scores bitwise == f32 whole:', bool(torch.equal(s, s0)), 'max diff', float((s - s0).abs().max()))
    for k in g:
        e = float((g[k] - ref[k]).abs().max()); sc = float(ref[k].abs().max())
        if e > 1e-4 * sc: print('   %-32s err %.3e scale %.3e rel %.2e  L2rel %.2e' % (k, e, sc, e / sc, float((g[k]-ref[k]).norm()/ref[k].norm())))
