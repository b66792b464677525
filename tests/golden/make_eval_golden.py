"""Generates the model-level decode fixtures and the Bayesian-network data fixture by running the
REAL reference in this container (same two arithmetic-free stand-ins as make_golden.py):

  eval_none_bidir.npz   Dvae.predict (dvae_model.py:492-496) and greedy Dvae.model_eval (:498-524) of the
                        reference on real ENAS validation rows, with the reparameterisation draw
                        (torch.randn_like, :412) replaced by recorded eps; decoded tensors, recon_acc and
                        the per-row decision margins (from the oracle, which must reproduce the decodes).
                        A second model_eval pass feeds the first pass's own decodes back in as the "true"
                        graphs, so that recon_acc is not trivially 0 with untrained weights.
  eval_bn.npz           the same for mod='bn' on random 10-node DAGs.
  bn_rows.npz           synthetic Bayesian-network text rows pushed through the reference's
                        NASDataset(task_type='bn') (dataloader.py:17-26,57-88): lines, split indices,
                        encoded graphs.  The reference ships no BN data (asia_200k.txt is absent), so the
                        ROWS are synthetic; the parsing / encoding / split is the reference's.

Run:  python tests/golden/make_eval_golden.py
"""
import os
import sys
import tempfile
import types
from argparse import Namespace

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import dvae_amd  # noqa: E402
from helpers import fill_weights, oracle_sd  # noqa: E402
from oracle import dense_ref as O  # noqa: E402

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


class inject_randn_like:
    """torch.randn_like -> the recorded draw (Dvae.reparamterize, :412)."""

    def __init__(self, eps):
        self.eps = eps

    def __enter__(self):
        self.orig = torch.randn_like
        torch.randn_like = lambda t, **kw: self.eps.clone().reshape(t.shape)

    def __exit__(self, *a):
        torch.randn_like = self.orig


def margins_of(sd, z, N, C, mod):
    return np.array([O.decoder_predict(sd, z[m:m + 1], N, C, mod, return_margin=True)[2] for m in range(z.shape[0])])


def eval_case(name, mod, bidir, N, C, g, x, seed, S=3, D=2):
    hp = Namespace(dataset_file=None, mod=mod, bidirectional=bidir, num_classes=C, max_seq=N)
    ref = fill_weights(RefDvae(hp), seed)
    sd = oracle_sd(ref)
    B = g.shape[0]
    gen = torch.Generator().manual_seed(seed + 11)
    eps_p = torch.randn(B, 56, generator=gen)
    eps_e = torch.randn(B, S, 56, generator=gen)
    X = {'graph': g, 'node_encoding': x}
    with torch.no_grad():
        # Dvae.predict
        with inject_randn_like(eps_p):
            p_edges, p_types = ref.predict(X)
        o_edges, o_types = O.model_predict(sd, g, x, eps_p, mod, bidir)
        assert torch.equal(p_edges, o_edges) and torch.equal(p_types, o_types), "oracle model_predict != reference"
        _, mu, logvar = ref.encoder(X)
        z_p = eps_p * torch.exp(0.5 * logvar) + mu
        p_marg = margins_of(sd, z_p, N, C, mod)
        # Dvae.model_eval, greedy; the reference returns only recon_acc, so the decoded tensors come from a
        # recording wrapper around its decoder.predict
        rec = {}
        orig_predict = ref.decoder.predict

        def recording(z, is_stochastic=False):
            out = orig_predict(z, is_stochastic)
            rec['z'], rec['edges'], rec['types'] = z.clone(), out[0].clone(), out[1].clone()
            return out
        ref.decoder.predict = recording
        with inject_randn_like(eps_e):
            acc1 = ref.model_eval(X, S, D, is_stochastic=False)['recon_acc']
        e1 = rec['edges'].view(B, D * S, N, N)
        t1 = rec['types'].view(B, D * S, N, C)
        z1 = rec['z']
        o_acc, o_e, o_t, o_z = O.model_eval(sd, g, x, eps_e, S, D, mod, bidir)
        assert o_acc == acc1 and torch.equal(o_e, e1) and torch.equal(o_t, t1), "oracle model_eval != reference"
        assert float((o_z - z1).abs().max()) < 1e-6
        m1 = margins_of(sd, z1, N, C, mod)
        # second pass: the first pass's first decode of every graph is the new "true" graph
        g2, x2 = e1[:, 0].clone(), t1[:, 0].clone()
        X2 = {'graph': g2, 'node_encoding': x2}
        with inject_randn_like(eps_e):
            acc2 = ref.model_eval(X2, S, D, is_stochastic=False)['recon_acc']
        e2 = rec['edges'].view(B, D * S, N, N)
        t2 = rec['types'].view(B, D * S, N, C)
        z2 = rec['z']
        o_acc2, o_e2, o_t2, _ = O.model_eval(sd, g2, x2, eps_e, S, D, mod, bidir)
        assert o_acc2 == acc2 and torch.equal(o_e2, e2) and torch.equal(o_t2, t2)
        m2 = margins_of(sd, z2, N, C, mod)
        ref.decoder.predict = orig_predict
    np.savez_compressed(
        os.path.join(HERE, name + ".npz"), g=g.numpy(), x=x.numpy(), seed=np.int64(seed), mod=np.str_(str(mod)),
        bidir=np.bool_(bidir), S=np.int64(S), D=np.int64(D), eps_predict=eps_p.numpy(), eps_eval=eps_e.numpy(),
        predict_edges=p_edges.numpy(), predict_types=p_types.numpy(), predict_margins=p_marg,
        eval1_acc=np.float64(acc1), eval1_edges=e1.numpy(), eval1_types=t1.numpy(), eval1_margins=m1,
        g2=g2.numpy(), x2=x2.numpy(),
        eval2_acc=np.float64(acc2), eval2_edges=e2.numpy(), eval2_types=t2.numpy(), eval2_margins=m2)
    print("%-18s predict: %d/%d rows with margin > 2e-3 | model_eval: recon_acc %.4f (%d/%d safe) | fed back: recon_acc %.4f (%d/%d safe)"
          % (name, int((p_marg > 2e-3).sum()), B, acc1, int((m1 > 2e-3).sum()), len(m1), acc2,
             int((m2 > 2e-3).sum()), len(m2)))


def bn_rows_case():
    """Synthetic BN structures in the reference's text format: 8 variables, row k = [variable id, k parent
    flags], then a score.  Every variable id appears once per structure (a BN over 8 variables)."""
    rnd = np.random.RandomState(7)
    lines = []
    for _ in range(60):
        perm = rnd.permutation(8)
        rows = []
        for k in range(8):
            flags = (rnd.rand(k) < 0.3).astype(int).tolist()
            rows.append([int(perm[k])] + flags)
        score = -11000.0 - 1000.0 * rnd.rand()
        lines.append("%s, %.6f" % (str(rows).replace(' ', ''), score))
        # the reference splits a line at ']],' (dataloader.py:29-33): the text must contain it exactly once
        assert lines[-1].count(']],') == 1
    with tempfile.NamedTemporaryFile('w', suffix='.txt', delete=False) as f:
        f.write("\n".join(lines) + "\n")
        path = f.name
    out = {'lines': np.array(lines)}
    for split in ('train', 'val', 'test'):
        ds = NASDataset(path, split=split, task_type='bn')
        out[split + '_index'] = ds.samples.index.to_numpy()
        if split == 'val':
            items = [ds[i] for i in range(len(ds))]
            out['val_graph'] = torch.stack([it['graph'] for it in items]).numpy()
            out['val_node_encoding'] = torch.stack([it['node_encoding'] for it in items]).numpy()
            out['val_acc'] = np.array([it['acc'] for it in items])
            ds_str = NASDataset(path, split=split, fmt='str', task_type='bn')
            out['val_str'] = torch.stack([ds_str[i]['graph'] for i in range(len(ds_str))]).numpy()
    os.unlink(path)
    np.savez_compressed(os.path.join(HERE, "bn_rows.npz"), **out)
    print("bn_rows            %d synthetic rows -> train/val/test %d/%d/%d, val graphs %s" % (
        len(lines), len(out['train_index']), len(out['val_index']), len(out['test_index']), out['val_graph'].shape))


def main():
    ds = NASDataset('/root/reference/dvae/data/final_structures6.txt', split='val')
    items = [ds[i] for i in range(8, 20)]          # 12 validation rows (make_golden.py holds rows 0..7)
    g = torch.stack([it['graph'] for it in items])
    x = torch.stack([it['node_encoding'] for it in items])
    eval_case("eval_none_bidir", None, True, 8, 8, g, x, seed=201)
    g, x = dvae_amd.synth.random_dags(10, 10, 10, 0.3, seed=16, fix_orphans=False)
    eval_case("eval_bn", "bn", False, 10, 10, g, x, seed=202)
    bn_rows_case()


if __name__ == "__main__":
    main()
