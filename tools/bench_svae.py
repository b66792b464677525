"""S-VAE train step (svae_model.py:163-182 + backward + Adam, lr 1e-3) at the ENAS-8 shape:
the HIP path against the same model written with torch.nn.GRU / nn.Linear (MIOpen + rocBLAS)
on the same GPU, and the CPU oracle on a bounded sample.  Prints one JSON line."""
import json, os, sys, time
from argparse import Namespace
import torch
import torch.nn as nn
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import dvae_amd

B = int(os.environ.get('SVAE_B', 4096)); C = N = 8; T = N - 1; H, Z = 501, 56
STEPS, WARM = 20, 5
dev = torch.device('cuda')
gen = torch.Generator().manual_seed(0)
x = torch.cat((F.one_hot(torch.randint(0, C, (B, T), generator=gen), C).float(),
               (torch.rand(B, T, N, generator=gen) < 0.5).float()), dim=-1)
xd = x.to(dev)


class TorchSvae(nn.Module):          # the library baseline: MIOpen GRU + rocBLAS linears
    def __init__(self, bidir):
        super().__init__()
        self.bidir = bidir
        self.enc = nn.GRU(C + N, H, batch_first=True, bidirectional=bidir)
        self.mu, self.var = nn.Linear(H, Z), nn.Linear(H, Z)
        self.unify = nn.Linear(2 * H, H) if bidir else None
        self.h0 = nn.Linear(Z, H)
        self.dec = nn.GRU(H, H, batch_first=True)
        self.vert = nn.Sequential(nn.Linear(H, H), nn.ReLU(), nn.Linear(H, C))
        self.edge = nn.Sequential(nn.Linear(H, H), nn.ReLU(), nn.Linear(H, N))

    def loss(self, x):
        _, h = self.enc(x)
        h = h.transpose(0, 1).contiguous().view(x.shape[0], -1)
        if self.bidir:
            h = self.unify(h)
        mu, lv = self.mu(h), self.var(h)
        z = torch.randn_like(mu) * torch.exp(0.5 * lv) + mu
        out, _ = self.dec(torch.relu(self.h0(z)).unsqueeze(1).expand(-1, T, -1))
        ts, es = self.vert(out), self.edge(out)
        e = F.binary_cross_entropy_with_logits(es, x[:, :, C:], reduction='sum')
        v = F.cross_entropy(ts.reshape(-1, C), torch.argmax(x[:, :, :C].reshape(-1, C), 1), reduction='sum')
        return e + v + 0.005 * (-0.5 * torch.sum(1 + lv - mu ** 2 - lv.exp()))


def timed(step):
    for _ in range(WARM): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(STEPS): step()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / STEPS * 1e3


res = {}
for bidir in (False, True):
    torch.manual_seed(0)
    m = dvae_amd.Svae(Namespace(dataset_file=None, bidirectional=bidir)).to(dev)
    (opt,), _ = m.configure_optimizers()
    def hip_step():
        opt.zero_grad(set_to_none=True)
        out = m.training_step({'graph': xd}, 0)
        out['loss'].backward(); opt.step()
    t_hip = timed(hip_step)
    def fused_step():
        m.loss_and_grad({'graph': xd}); opt.step()
    t_fused = timed(fused_step)
    tm = TorchSvae(bidir).to(dev)
    topt = torch.optim.Adam(tm.parameters(), lr=1e-3)
    def torch_step():
        topt.zero_grad(set_to_none=True)
        tm.loss(xd).backward(); topt.step()
    t_torch = timed(torch_step)
    res['bidir' if bidir else 'uni'] = {'hip_ms': round(t_hip, 3), 'strings_per_s': round(B / t_hip * 1e3),
                                        'hip_fused_step_ms': round(t_fused, 3),
                                        'torch_miopen_ms': round(t_torch, 3), 'speedup': round(t_torch / t_hip, 2)}
    print(res, flush=True)

# CPU oracle (torch CPU, explicit loops) on a bounded sample
from helpers import oracle_sd
from oracle import svae_ref as O
torch.set_num_threads(16)
Bc = 256
sd = oracle_sd(dvae_amd.Svae(Namespace(dataset_file=None, bidirectional=False)), requires_grad=True)
ps = list(sd.values()); copt = torch.optim.Adam(ps, lr=1e-3)
eps = torch.randn(Bc, Z)
def cpu_step():
    copt.zero_grad(); O.train_loss(sd, x[:Bc], eps, False, C)[0].backward(); copt.step()
cpu_step(); t0 = time.perf_counter()
for _ in range(5): cpu_step()
t_cpu = (time.perf_counter() - t0) / 5
flops = 3 * (T * (6 * H * (C + N) + 6 * H * H) + 6 * H * H + T * 6 * H * H + T * (4 * H * H + 2 * H * (C + N)) + 4 * H * Z + 2 * Z * H)
print(json.dumps({'workload': 'S-VAE train step, ENAS-8 strings (T=7, I=16), hs=501, batch %d' % B, 'results': res,
                  'algorithmic_gflop_per_string': round(flops / 1e9, 4),
                  'hip_tflops_uni': round(flops * B / res['uni']['hip_ms'] / 1e9, 1),
                  'cpu_oracle': {'strings_per_s': round(Bc / t_cpu, 1), 'threads': 16, 'sample': '5 steps of %d strings' % Bc}}))
