"""Per-tensor error of the S-VAE HIP path against the oracle for a few batch sizes."""
import os, sys
from argparse import Namespace
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import dvae_amd
from dvae_amd import svae as S
from helpers import fill_weights, oracle_sd
from oracle import svae_ref as O
dev = torch.device('cuda')
def run(B, C, N, H, bidir, dbl=False):
    model = dvae_amd.Svae(Namespace(dataset_file=None, bidirectional=bidir, num_classes=C, max_seq=N))
    model.encoder = S.Encoder(C, N, hidden_size=H, bidir=bidir); model.decoder = S.Decoder(C, N, hidden_size=H)
    fill_weights(model, 17); model.to(dev)
    gen = torch.Generator().manual_seed(B); T = N - 1
    x = torch.cat((torch.nn.functional.one_hot(torch.randint(0, C, (B, T), generator=gen), C).float(),
                   (torch.rand(B, T, N, generator=gen) < 0.4).float()), dim=-1)
    eps = torch.randn(B, 56, generator=gen)
    sd = oracle_sd(model, requires_grad=True)
    if dbl:
        sd = {k: v.detach().double().requires_grad_(True) for k, v in sd.items()}
        ref = O.train_loss(sd, x.double(), eps.double(), bidir, C)
    else:
        ref = O.train_loss(sd, x, eps, bidir, C)
    ref[0].backward()
    out = model.training_step({'graph': x.to(dev)}, 0, eps=eps.to(dev)); out['loss'].backward()
    print('B=%d C=%d N=%d H=%d bidir=%d dbl=%d loss %.6f vs %.6f' % (B, C, N, H, bidir, dbl, float(out['loss']), float(ref[0])))
    worst = []
    for k, p in model.named_parameters():
        g = sd[k].grad.float()
        err = float((p.grad.cpu() - g).abs().max()); sc = float(g.abs().max())
        worst.append((err / max(sc, 1e-3), k, err, sc))
    for w in sorted(worst, reverse=True)[:6]:
        print('   %.2e  %-40s err %.2e scale %.2e' % w)
for B in (8, 32, 64):
    run(B, 8, 8, 501, True)
run(64, 8, 8, 501, True, dbl=True)
run(64, 8, 8, 501, False, dbl=True)

print('--- pieces at B=64 bidir')
B, C, N, H, bidir = 64, 8, 8, 501, True
model = dvae_amd.Svae(Namespace(dataset_file=None, bidirectional=bidir, num_classes=C, max_seq=N))
fill_weights(model, 17); model.to(dev)
gen = torch.Generator().manual_seed(B); T = N - 1
x = torch.cat((torch.nn.functional.one_hot(torch.randint(0, C, (B, T), generator=gen), C).float(),
               (torch.rand(B, T, N, generator=gen) < 0.4).float()), dim=-1)
sd = oracle_sd(model, requires_grad=True)
z = torch.randn(B, 56, generator=gen)
wt, we = torch.randn(B, T, C, generator=gen), torch.randn(B, T, N, generator=gen)
ts, es = O.decoder(sd, z, T)
((ts * wt).sum() + (es * we).sum()).backward()
def dec_once():
    model.zero_grad()
    hts, hes = model.decoder(z.to(dev))
    ((hts * wt.to(dev)).sum() + (hes * we.to(dev)).sum()).backward()
    return hts.detach().cpu(), {k: p.grad.cpu().clone() for k, p in model.decoder.named_parameters()}
a_ts, a = dec_once(); b_ts, b = dec_once()
print('fwd err', float((a_ts - ts).abs().max()), 'run-to-run', float((a_ts - b_ts).abs().max()))
for k in a:
    g = sd['decoder.' + k].grad
    print('  %-28s err %.2e  run-to-run %.2e  scale %.2e' % (k, float((a[k] - g).abs().max()), float((a[k] - b[k]).abs().max()), float(g.abs().max())))

print('--- encoder alone')
for Bq in (32, 64):
    xq = x[:Bq]
    for k in sd: sd[k].grad = None
    wm, wl = torch.randn(Bq, 56, generator=gen), torch.randn(Bq, 56, generator=gen)
    mu, lv = O.encoder(sd, xq, True)
    ((mu * wm).sum() + (lv * wl).sum()).backward()
    model.zero_grad()
    hmu, hlv = model.encoder(xq.to(dev))
    ((hmu * wm.to(dev)).sum() + (hlv * wl.to(dev)).sum()).backward()
    print('B=%d mu err %.2e logvar err %.2e (scale %.2e)' % (Bq, float((hmu.cpu() - mu).abs().max()), float((hlv.cpu() - lv).abs().max()), float(mu.abs().max())))
    for k, p in model.encoder.named_parameters():
        g = sd['encoder.' + k].grad
        print('  %-28s err %.2e  scale %.2e' % (k, float((p.grad.cpu() - g).abs().max()), float(g.abs().max())))
