"""GPU: BASELINE.json's configurations at (per-GPU) full size, checked through
size-independent properties -- the CPU oracle cannot run them in reasonable time
(31 s/step at B=1024, N=8; 67 s at N=64, B=2):
  * additivity: every loss is a sum over graphs, so loss/grad of a batch equal the
    sum over its halves (this is also what makes SUM all-reduce data parallelism exact);
  * permutation equivariance of per-graph outputs;
  * bitwise run-to-run reproducibility (no atomics anywhere);
  * spot parity: a few graphs of the big batch against the oracle run on them alone.
"""
from argparse import Namespace

import pytest
import torch

from helpers import close_scaled, fill_weights, oracle_sd

pytestmark = pytest.mark.gpu


def _model(cuda, mod=None, bidir=False, N=8, C=8, seed=3):
    import dvae_amd
    hp = Namespace(dataset_file=None, mod=mod, bidirectional=bidir, num_classes=C, max_seq=N)
    return fill_weights(dvae_amd.Dvae(hp), seed).to(cuda)


def _grads(model):
    return torch.cat([p.grad.flatten() for p in model.parameters()]).clone()


def _step(model, X, eps):
    model.zero_grad(set_to_none=True)
    out4 = model.loss_and_grad(X, eps=eps)
    return out4.clone(), _grads(model)


def test_config2_enas_b4096_additivity_and_reproducibility(cuda):
    """Headline shape: ENAS-like 8-node DAGs, hs=501, batch 4096."""
    import dvae_amd
    from oracle import dense_ref as O
    B = 4096
    model = _model(cuda)
    g, x = dvae_amd.synth.enas_like(B, 11)
    eps = torch.randn(B, 56, generator=torch.Generator().manual_seed(2))
    X = {'graph': g.to(cuda), 'node_encoding': x.to(cuda)}
    out, gr = _step(model, X, eps.to(cuda))
    out2, gr2 = _step(model, X, eps.to(cuda))
    assert torch.equal(out, out2) and torch.equal(gr, gr2), "train step is not bitwise reproducible"
    h = B // 2
    parts = [_step(model, {k: v[s] for k, v in X.items()}, eps[s].to(cuda)) for s in (slice(0, h), slice(h, B))]
    close_scaled(parts[0][0] + parts[1][0], out, tol=2e-5, what="loss additivity")
    close_scaled(parts[0][1] + parts[1][1], gr, tol=2e-4, what="gradient additivity")
    # spot parity: 6 graphs of the big batch against the oracle
    idx = torch.tensor([0, 1, 777, 2048, 4000, 4095])
    sd = oracle_sd(model)
    with torch.no_grad():
        hv, mu, lv = model.encoder(X)
        o_hv, o_mu, o_lv = O.encoder_forward(sd, g[idx], x[idx])
        close_scaled(mu[idx.to(cuda)], o_mu, what="mu (spot)")
        close_scaled(lv[idx.to(cuda)], o_lv, what="logvar (spot)")
        z = O.reparameterize(o_mu, o_lv, eps[idx])
        zfull = torch.zeros(B, 56)
        zfull[idx] = z
        e, t = model.decoder(zfull.to(cuda), X)
        o_e, o_t = O.decoder_forward(sd, z, g[idx], x[idx])
        close_scaled(e[idx.to(cuda)], o_e, what="edge logits (spot)")
        close_scaled(t[idx.to(cuda)], o_t, what="type logits (spot)")


def test_config2_bn_b1024(cuda):
    """Bayesian-network mode, 10 nodes, batch 1024: permutation equivariance + additivity."""
    import dvae_amd
    B, N = 1024, 10
    model = _model(cuda, mod='bn', N=N, C=N)
    g, x = dvae_amd.synth.random_dags(B, N, N, 0.3, 21, fix_orphans=False)
    eps = torch.randn(B, 56, generator=torch.Generator().manual_seed(5))
    X = {'graph': g.to(cuda), 'node_encoding': x.to(cuda)}
    out, gr = _step(model, X, eps.to(cuda))
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(1))
    Xp = {k: v[perm.to(cuda)] for k, v in X.items()}
    outp, grp = _step(model, Xp, eps[perm].to(cuda))
    close_scaled(outp, out, tol=2e-5, what="loss under permutation")
    close_scaled(grp, gr, tol=2e-4, what="gradient under permutation")
    with torch.no_grad():
        _, mu, _ = model.encoder(X)
        _, mup, _ = model.encoder(Xp)
        assert torch.equal(mup, mu[perm.to(cuda)]), "per-graph encoder output depends on batch position"


def test_config3_n64_per_gpu_shard(cuda):
    """64-node random DAGs; 512 graphs = one GPU's shard of the 4096 batch at 8 GPUs."""
    import dvae_amd
    from oracle import dense_ref as O
    B, N = 512, 64
    model = _model(cuda, N=N)
    g, x = dvae_amd.synth.random_dags(B, N, 8, 0.1, 31)
    eps = torch.randn(B, 56, generator=torch.Generator().manual_seed(6))
    X = {'graph': g.to(cuda), 'node_encoding': x.to(cuda)}
    out, gr = _step(model, X, eps.to(cuda))
    assert torch.isfinite(out).all() and torch.isfinite(gr).all()
    h = B // 2
    parts = [_step(model, {k: v[s] for k, v in X.items()}, eps[s].to(cuda)) for s in (slice(0, h), slice(h, B))]
    close_scaled(parts[0][0] + parts[1][0], out, tol=2e-5, what="loss additivity")
    close_scaled(parts[0][1] + parts[1][1], gr, tol=3e-4, what="gradient additivity")
    # one 64-node graph against the oracle (encoder only: the dense decoder is O(N^3))
    sd = oracle_sd(model)
    with torch.no_grad():
        _, mu, lv = model.encoder(X)
        _, o_mu, o_lv = O.encoder_forward(sd, g[:1], x[:1])
    close_scaled(mu[:1], o_mu, what="mu of a 64-node graph")
    close_scaled(lv[:1], o_lv, what="logvar of a 64-node graph")


def test_config4_ragged_bidirectional(cuda):
    """Bidirectional encoder on mixed sizes 8..128 (ragged levels)."""
    import dvae_amd
    from oracle import dense_ref as O
    B, Nmax = 96, 128
    model = _model(cuda, bidir=True, N=Nmax)
    g, x, sizes = dvae_amd.synth.ragged_dags(B, 8, 41, nmin=8, nmax=Nmax)
    eps = torch.randn(B, 56, generator=torch.Generator().manual_seed(7))
    X = {'graph': g.to(cuda), 'node_encoding': x.to(cuda), 'sizes': sizes.to(cuda)}
    out, gr = _step(model, X, eps.to(cuda))
    assert torch.isfinite(out).all() and torch.isfinite(gr).all()
    h = B // 2
    parts = [_step(model, {k: v[s] for k, v in X.items()}, eps[s].to(cuda)) for s in (slice(0, h), slice(h, B))]
    close_scaled(parts[0][0] + parts[1][0], out, tol=2e-5, what="loss additivity")
    close_scaled(parts[0][1] + parts[1][1], gr, tol=3e-4, what="gradient additivity")
    # the two smallest graphs against the oracle (bidirectional encoder incl. the lin01 chain)
    small = torch.argsort(sizes)[:2]
    sd = oracle_sd(model)
    with torch.no_grad():
        _, mu, _ = model.encoder(X)
        for b in small.tolist():
            n = int(sizes[b])
            _, o_mu, _ = O.encoder_forward(sd, g[b:b + 1, :n, :n], x[b:b + 1, :n], None, True)
            close_scaled(mu[b:b + 1], o_mu, what="mu of ragged graph %d (n=%d)" % (b, n))


def test_config5_decode_12500(cuda):
    """One GPU's shard (100k / 8) of the decode workload: determinism, validity of the
    output encoding, and agreement of greedy decode with a small-batch decode."""
    M = 12500
    model = _model(cuda)
    z = torch.randn(M, 56, generator=torch.Generator().manual_seed(8)).to(cuda)
    e1, t1, m1 = model.decoder.predict(z, return_margin=True)
    e2, t2 = model.decoder.predict(z)
    assert torch.equal(e1, e2) and torch.equal(t1, t2)
    assert ((e1 == 0) | (e1 == 1)).all() and torch.equal(e1, torch.tril(e1, -1))
    assert (t1.sum(-1) == 1).all() and ((t1 == 0) | (t1 == 1)).all()
    es, ts = model.decoder.predict(z[:64])
    assert torch.equal(es, e1[:64]) and torch.equal(ts, t1[:64]), "decode depends on batch size"
    u = (torch.rand(M, 8, device=cuda), torch.rand(M, 8, 8, device=cuda))
    s1 = model.decoder.predict(z, uniforms=u)
    s2 = model.decoder.predict(z, uniforms=u)
    assert torch.equal(s1[0], s2[0]) and torch.equal(s1[1], s2[1])
    acc = model.model_eval({'graph': e1[:32], 'node_encoding': t1[:32]}, 2, 2, is_stochastic=False)['recon_acc']
    assert 0.0 <= acc <= 1.0
