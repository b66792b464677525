"""One configuration of tests/test_gpu_parity.py's random sweep (usage: parity_fp64.py <seed>): gradients of the
HIP path and of the fp32 oracle, each against the oracle run in fp64 -- tells fp32 noise of the oracle's
dense-loop formulation from an error of the HIP path."""
import os, sys, random, torch
from argparse import Namespace
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import dvae_amd
from helpers import fill_weights, oracle_sd
from oracle import dense_ref as O
cuda = torch.device('cuda')
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 198
rnd = random.Random(1000 + seed)
mod = rnd.choice([None, None, "bn"]); bidir = rnd.random() < 0.4
N = rnd.randint(1, 13); C = rnd.randint(2, 12); H = rnd.choice([24, 40, 64, 100, 200]); B = rnd.randint(1, 10)
p = rnd.choice([0.15, 0.3, 0.5, 0.8]); chain = rnd.random() < 0.4; fix_orphans = rnd.random() < 0.6
hp = Namespace(dataset_file=None, mod=mod, bidirectional=bidir, num_classes=C, max_seq=N, same_mapper_gater=False, hidden_state_size=H)
model = fill_weights(dvae_amd.Dvae(hp), seed=50 + seed).to(cuda)
g, x = dvae_amd.synth.random_dags(B, N, C, p, seed=70 + seed, chain=chain, fix_orphans=fix_orphans)
eps = torch.randn(B, 56, generator=torch.Generator().manual_seed(90 + seed))
print(dict(mod=mod, bidir=bidir, N=N, C=C, H=H, B=B, p=p, chain=chain))
sd32 = oracle_sd(model, requires_grad=True)
r32 = O.train_loss(sd32, g, x, eps, mod, bidir); r32[0].backward()
sd64 = {k: v.detach().double().requires_grad_(True) for k, v in oracle_sd(model).items()}
r64 = O.train_loss(sd64, g.double(), x.double(), eps.double(), mod, bidir); r64[0].backward()
X = {'graph': g.to(cuda), 'node_encoding': x.to(cuda)}
out = model.training_step(X, 0, eps=eps.to(cuda)); out['loss'].backward()
worst = []
for k, pp in model.named_parameters():
    g64 = sd64[k].grad
    if g64 is None: continue
    sc = float(g64.abs().max()) + 1e-30
    e_hip = float((pp.grad.cpu().double() - g64).abs().max()) / sc
    e_o32 = float((sd32[k].grad.double() - g64).abs().max()) / sc
    worst.append((e_hip, e_o32, k))
for e_hip, e_o32, k in sorted(worst, reverse=True)[:6]:
    print('%-40s HIP vs fp64 %.2e   fp32 oracle vs fp64 %.2e' % (k, e_hip, e_o32))
