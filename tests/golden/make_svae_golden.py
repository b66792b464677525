"""Generates tests/golden/svae_*.npz by running the REAL reference S-VAE
(/root/reference/dvae/models/svae_model.py, torch.nn.GRU included) in this container.

Same arrangement as make_golden.py: the reference cannot travel, so its outputs on small seeded
inputs are committed as data; `pytorch_lightning` and `toposort` (absent here, arithmetic-free)
get in-memory stand-ins.  Before anything is written the oracle (oracle/svae_ref.py) must agree
with the reference on outputs, losses and every gradient.

Run:  python tests/golden/make_svae_golden.py
"""
import os
import sys
import types
from argparse import Namespace

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import dvae_amd  # noqa: E402  (ours: key-name comparison only)
from helpers import fill_weights, oracle_sd  # noqa: E402
from oracle import svae_ref as O  # noqa: E402
from make_golden import grad_summary  # noqa: E402  (also installs the stand-ins and the reference path)

from dvae.models.svae_model import Svae as RefSvae  # noqa: E402
from dvae.utils.dataloader import NASDataset  # noqa: E402


def run_case(name, x, C, N, bidir, seed):
    hp = Namespace(dataset_file=None, bidirectional=bidir, num_classes=C, max_seq=N)
    ref = fill_weights(RefSvae(hp), seed)
    ours = dvae_amd.Svae(Namespace(**vars(hp)))
    assert list(ref.state_dict().keys()) == list(ours.state_dict().keys()), "state_dict keys differ"
    for k, v in ref.state_dict().items():
        assert v.shape == ours.state_dict()[k].shape, k
    B = x.shape[0]
    eps = torch.randn(B, 56, generator=torch.Generator().manual_seed(seed + 1))
    orig = torch.randn_like
    torch.randn_like = lambda t, **kw: eps.clone()          # inject eps into reparamterize (:109)
    try:
        out = ref.training_step({'graph': x}, 0)
    finally:
        torch.randn_like = orig
    ref.zero_grad()
    out['loss'].backward()
    with torch.no_grad():
        mu, logvar = ref.encoder(x)
        z = eps * torch.exp(0.5 * logvar) + mu
        ts, es = ref.decoder(z)
        onehot, flags = ref._predict_from_logits(ts, es)
    log = out['log']
    data = dict(x=x.numpy(), eps=eps.numpy(), seed=np.int64(seed), bidir=np.bool_(bidir), C=np.int64(C), N=np.int64(N),
                mu=mu.numpy(), logvar=logvar.numpy(), z=z.numpy(), type_scores=ts.numpy(), edge_scores=es.numpy(),
                onehot=onehot.numpy(), flags=flags.numpy(), margin=np.float64(O.min_margin(ts, es)),
                losses=np.array([float(log['train/edge_loss']), float(log['train/vertex_loss']),
                                 float(log['train/kl_loss']), float(log['train/loss'])], dtype=np.float64),
                acc=np.float64(float(log['train/acc'])))
    for k, p in ref.named_parameters():
        idx, vals, s, a = grad_summary(k, p.grad)
        data['gidx/' + k], data['gval/' + k] = idx, vals
        data['gsum/' + k] = np.array([s, a])
    sd = oracle_sd(ref, requires_grad=True)
    o = O.train_loss(sd, x, eps, bidir, C)
    o[0].backward()
    worst = max(float((o[6] - ts).abs().max()), float((o[7] - es).abs().max()), float((o[4] - mu).abs().max()),
                float((o[5] - logvar).abs().max()))
    gworst = max(float((p.grad - sd[k].grad).abs().max() / (p.grad.abs().max() + 1e-12))
                 for k, p in ref.named_parameters())
    assert worst < 1e-5 and gworst < 1e-4, (worst, gworst)
    assert abs(float(o[0]) - float(out['loss'])) < 1e-5 * abs(float(out['loss']))
    oh, fl = O.predict_from_logits(ts, es)
    assert torch.equal(oh, onehot) and torch.equal(fl, flags)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **data)
    print("%-22s loss %.4f acc %.3f  oracle-vs-reference max abs %.2e  grads rel %.2e  margin %.1e" % (
        name, float(out['loss']), float(log['train/acc']), worst, gworst, data['margin']))


def main():
    # (i) real ENAS rows in string form: first 8 structures of the reference's validation split
    ds = NASDataset('/root/reference/dvae/data/final_structures6.txt', split='val', fmt='str')
    x = torch.stack([ds[i]['graph'] for i in range(8)])
    run_case("svae_enas_val8_uni", x, 8, 8, False, seed=201)
    run_case("svae_enas_val8_bidir", x, 8, 8, True, seed=202)
    # (ii) random strings of the Bayesian-network shape (10 types, 10 nodes -> 9 steps of width 20)
    gen = torch.Generator().manual_seed(7)
    B, C, N = 6, 10, 10
    types = torch.randint(0, C, (B, N - 1), generator=gen)
    flags = (torch.rand(B, N - 1, N, generator=gen) < 0.3).float()
    x = torch.cat((torch.nn.functional.one_hot(types, C).float(), flags), dim=-1)
    run_case("svae_rand10_bidir", x, C, N, True, seed=203)


if __name__ == "__main__":
    main()
