"""Generates tests/golden/*.npz by running the REAL reference
(/root/reference/dvae/models/dvae_model.py) in this container.

The reference cannot travel to the GPU box, so its outputs on small seeded inputs
are committed as data.  Two arithmetic-free imports the reference needs are absent
here and get in-memory stand-ins: `pytorch_lightning` (only `LightningModule` as a
base class and the `data_loader` decorator are used, dvae_model.py:3,365,477) and
`toposort` (imported at dataloader.py:6, never called).

Run:  python tests/golden/make_golden.py
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

import dvae_amd  # noqa: E402  (ours: only for key-name comparison and synthetic graphs)
from helpers import fill_weights, oracle_sd  # noqa: E402
from oracle import dense_ref as O  # noqa: E402
from oracle import levels as OL  # noqa: E402

pl = types.ModuleType('pytorch_lightning')
pl.LightningModule = nn.Module
pl.data_loader = lambda f: f
sys.modules['pytorch_lightning'] = pl
ts = types.ModuleType('toposort')
ts.toposort = lambda *a, **k: None
sys.modules['toposort'] = ts
sys.path.insert(0, '/root/reference')
from dvae.models.dvae_model import Dvae as RefDvae  # noqa: E402
from dvae.utils.dataloader import NASDataset  # noqa: E402

GRAD_SAMPLES = 64


def grad_summary(name, g):
    flat = g.detach().flatten().double()
    gen = torch.Generator().manual_seed(abs(hash(name)) % (2 ** 31))
    idx = torch.randperm(flat.numel(), generator=gen)[:GRAD_SAMPLES]
    return idx.numpy().astype(np.int64), flat[idx].numpy(), float(flat.sum()), float(flat.abs().sum())


def run_case(name, mod, bidir, shared, N, C, g, x, seed):
    torch.manual_seed(0)
    hp = Namespace(dataset_file=None, mod=mod, bidirectional=bidir, num_classes=C, max_seq=N,
                   same_mapper_gater=shared)
    ref = fill_weights(RefDvae(hp), seed)
    ours = dvae_amd.Dvae(Namespace(**vars(hp)))
    assert list(ref.state_dict().keys()) == list(ours.state_dict().keys()), "state_dict keys differ"
    for k, v in ref.state_dict().items():
        assert v.shape == ours.state_dict()[k].shape, k
    B = g.shape[0]
    X = {'graph': g, 'node_encoding': x}
    # per-node states through forward hooks on the GRU cells
    cap = {'enc': [], 'back': [], 'dec': []}
    hooks = [ref.encoder.gru.register_forward_hook(lambda m, i, o: cap['enc'].append(o.detach().clone())),
             ref.decoder.gru.register_forward_hook(lambda m, i, o: cap['dec'].append(o.detach().clone()))]
    if bidir:
        hooks.append(ref.encoder.back_gru.register_forward_hook(lambda m, i, o: cap['back'].append(o.detach().clone())))
    gen = torch.Generator().manual_seed(seed + 1)
    eps = torch.randn(B, 56, generator=gen)
    orig = torch.randn_like
    torch.randn_like = lambda t, **kw: eps.clone()          # inject eps into reparamterize (:412)
    try:
        out = ref.training_step(X, 0)
    finally:
        torch.randn_like = orig
    for h in hooks:
        h.remove()
    ref.zero_grad()
    out['loss'].backward()
    fw = torch.stack(cap['enc'], dim=1)
    bw = torch.stack(cap['back'][::-1], dim=1) if bidir else None
    finals, pos = [], 0
    for i in range(N):                  # decoder.gru is called 1 + i times for node i
        pos += 1 + i
        finals.append(cap['dec'][pos - 1])
    hd = torch.stack(finals, dim=1)
    with torch.no_grad():
        hv, mu, logvar = ref.encoder(X)
        z = eps * torch.exp(0.5 * logvar) + mu
        e, t = ref.decoder(z, X)
    data = dict(g=g.numpy(), x=x.numpy(), eps=eps.numpy(), seed=np.int64(seed),
                mod=np.str_(str(mod)), bidir=np.bool_(bidir), shared=np.bool_(shared),
                enc_states=fw.numpy(), dec_states=hd.numpy(), hv=hv.numpy(), mu=mu.numpy(), logvar=logvar.numpy(),
                z=z.numpy(), edge_logits=e.numpy(), type_logits=t.numpy(),
                losses=np.array([out['log']['train/edge_loss'], out['log']['train/vertex_loss'],
                                 out['log']['train/kl_loss'], out['log']['train/loss']], dtype=np.float64),
                levels_fwd=OL.forward_levels(g.numpy()), levels_rev=OL.reverse_levels(g.numpy()))
    if bidir:
        data['enc_states_rev'] = bw.numpy()
    for k, p in ref.named_parameters():
        idx, vals, s, a = grad_summary(k, p.grad)
        data['gidx/' + k], data['gval/' + k] = idx, vals
        data['gsum/' + k] = np.array([s, a])
    # the oracle must agree with the reference before anything is written
    sd = oracle_sd(ref, requires_grad=True)
    o = O.train_loss(sd, g, x, eps, mod, bidir)
    o[0].backward()
    worst = max(float((o[6] - e).abs().max()), float((o[7] - t).abs().max()), float((o[4] - mu).abs().max()))
    gworst = max(float((p.grad - sd[k].grad).abs().max() / (p.grad.abs().max() + 1e-12))
                 for k, p in ref.named_parameters())
    assert worst < 1e-5 and gworst < 1e-4, (worst, gworst)
    assert abs(float(o[0]) - float(out['loss'])) < 1e-5 * abs(float(out['loss']))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **data)
    print("%-28s loss %.4f  oracle-vs-reference max abs %.2e  grads rel %.2e" % (name, float(out['loss']), worst, gworst))
    return ref


def predict_case(name, mod, N, C, seed, M=64):
    hp = Namespace(dataset_file=None, mod=mod, bidirectional=False, num_classes=C, max_seq=N)
    ref = fill_weights(RefDvae(hp), seed)
    z = 3.0 * torch.randn(M, 56, generator=torch.Generator().manual_seed(seed + 7))
    with torch.no_grad():
        edges, types = ref.decoder.predict(z)
        sd = oracle_sd(ref)
        o_edges, o_types = O.decoder_predict(sd, z, N, C, mod)
        assert torch.equal(edges, o_edges) and torch.equal(types, o_types)
        margins = np.array([O.decoder_predict(sd, z[m:m + 1], N, C, mod, return_margin=True)[2] for m in range(M)])
    np.savez_compressed(os.path.join(HERE, name + ".npz"), z=z.numpy(), edges=edges.numpy(), types=types.numpy(),
                        margins=margins, seed=np.int64(seed), mod=np.str_(str(mod)), N=np.int64(N), C=np.int64(C))
    print("%-28s greedy decode of %d latents, min margin %.2e, %d with margin > 2e-3" % (
        name, M, margins.min(), int((margins > 2e-3).sum())))


def main():
    # (i) real ENAS rows: first 8 graphs of the reference's validation split
    ds = NASDataset('/root/reference/dvae/data/final_structures6.txt', split='val')
    items = [ds[i] for i in range(8)]
    g = torch.stack([it['graph'] for it in items])
    x = torch.stack([it['node_encoding'] for it in items])
    run_case("enas_val8_none_bidir", None, True, False, 8, 8, g, x, seed=101)
    # the same 8 rows as raw text + their row numbers: pins the parser / split of d-vae_amd/data.py
    lines = [ln.strip() for ln in open('/root/reference/dvae/data/final_structures6.txt') if ln.strip()]
    idx = ds.samples.index.to_numpy()[:8]
    np.savez_compressed(os.path.join(HERE, "enas_rows.npz"), n_rows=np.int64(len(lines)), val_index=idx,
                        lines=np.array([lines[i] for i in idx]), acc=np.array([it['acc'] for it in items]),
                        n_train=np.int64(len(NASDataset('/root/reference/dvae/data/final_structures6.txt', split='train'))),
                        n_test=np.int64(len(NASDataset('/root/reference/dvae/data/final_structures6.txt', split='test'))))
    # (ii) random lower-triangular DAGs, uni-directional
    g, x = dvae_amd.synth.random_dags(8, 8, 8, 0.4, seed=5)
    run_case("rand8_none_uni", None, False, False, 8, 8, g, x, seed=102)
    # (iii) Bayesian-network mode, shared gate/map networks
    g, x = dvae_amd.synth.random_dags(6, 10, 10, 0.3, seed=6, fix_orphans=False)
    run_case("rand10_bn_bidir_shared", "bn", True, True, 10, 10, g, x, seed=103)
    # (v) greedy decode
    predict_case("predict_none", None, 8, 8, seed=104)
    predict_case("predict_bn", "bn", 10, 10, seed=105)


if __name__ == "__main__":
    main()
