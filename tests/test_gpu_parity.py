"""Parity of the HIP path (through the C ABI) against the CPU oracle on the same
seeded inputs: per-node hidden states, mu/logvar, logits, the four loss
scalars and every parameter gradient.  Tolerance: 1e-4 relative (fp32), the bar
BASELINE.json's north_star states; gradients 3e-4 (longer reduction chains)."""
from argparse import Namespace

import pytest
import torch

from helpers import close_scaled, fill_weights, oracle_sd

pytestmark = pytest.mark.gpu

CASES = {
    # name: (mod, bidir, shared, H, N, C, B, p, chain[, fix_orphans])
    "none_uni_h64": (None, False, False, 64, 8, 8, 12, 0.4, False),
    "none_bidir_h64": (None, True, False, 64, 7, 5, 9, 0.4, False),
    "bn_bidir_shared_h64": ("bn", True, True, 64, 10, 10, 6, 0.3, False),
    "none_uni_h501_enas": (None, False, False, 501, 8, 8, 16, 1.0 / 3, True),
    "none_bidir_h501": (None, True, False, 501, 8, 8, 6, 0.4, False),
    "bn_uni_h501": ("bn", False, False, 501, 10, 10, 5, 0.3, False),
    "none_uni_h128_n20": (None, False, False, 128, 20, 6, 5, 0.15, False),
    # fused pair backward, LDS accumulators (9..16 nodes at add_edge width 1024)
    "none_uni_h200_n12": (None, False, False, 200, 12, 6, 6, 0.3, False),
    # fused pair backward, register accumulators at widths 1024 and 3072
    "none_uni_h200_n8": (None, False, False, 200, 8, 6, 7, 0.4, True),
    "bn_uni_h501_n8": ("bn", False, False, 501, 8, 8, 5, 0.35, False),
    # degenerate sizes: two nodes (no scored edge row at all), a single node (no edge, no pair)
    "none_uni_h64_n2": (None, False, False, 64, 2, 4, 5, 0.5, False),
    "none_bidir_h64_n1": (None, True, False, 64, 1, 3, 4, 0.5, False),
    # limits: the maximum number of node types (31: the one-hot tables have 32 columns), batch of one
    "none_uni_h64_c31": (None, False, False, 64, 6, 31, 9, 0.4, False),
    "bn_bidir_h64_c31_b1": ("bn", True, False, 64, 7, 31, 1, 0.4, False),
    # source nodes other than node 0 (level 0 holds several nodes of a graph; their k = 0 rows are scored)
    "none_uni_h64_orphans": (None, False, False, 64, 9, 6, 10, 0.25, False, False),
    "none_bidir_h64_orphans": (None, True, False, 64, 8, 5, 7, 0.2, False, False),
}


def _make(case, cuda):
    import dvae_amd
    mod, bidir, shared, H, N, C, B, p, chain = CASES[case][:9]
    fix_orphans = CASES[case][9] if len(CASES[case]) > 9 else True
    hp = Namespace(dataset_file=None, mod=mod, bidirectional=bidir, num_classes=C, max_seq=N,
                   same_mapper_gater=shared, hidden_state_size=H)
    model = fill_weights(dvae_amd.Dvae(hp), seed=11).to(cuda)
    g, x = dvae_amd.synth.random_dags(B, N, C, p, seed=5, chain=chain, fix_orphans=fix_orphans)
    gen = torch.Generator().manual_seed(9)
    eps = torch.randn(B, 56, generator=gen)
    return model, g, x, eps


@pytest.mark.parametrize("case", list(CASES))
def test_forward_parity(cuda, case):
    from oracle import dense_ref as O
    mod, bidir = CASES[case][0], CASES[case][1]
    model, g, x, eps = _make(case, cuda)
    sd = oracle_sd(model)
    X = {'graph': g.to(cuda), 'node_encoding': x.to(cuda)}
    with torch.no_grad():
        hv, mu, lv = model.encoder(X)
        fw, bw = model.encoder.node_states(X)
        o_hv, o_mu, o_lv, o_fw, o_bw = O.encoder_forward(sd, g, x, mod, bidir, return_states=True)
        close_scaled(fw, o_fw, what="encoder forward states")
        if bidir:
            close_scaled(bw, o_bw, what="encoder reverse states")
        close_scaled(hv, o_hv, what="hv")
        close_scaled(mu, o_mu, what="mu")
        close_scaled(lv, o_lv, what="logvar")
        z = model.reparamterize(mu, lv, eps.to(cuda))
        o_z = O.reparameterize(o_mu, o_lv, eps)
        close_scaled(z, o_z, what="z")
        e, t = model.decoder(o_z.to(cuda), X)
        hd = model.decoder.node_states(o_z.to(cuda), X)
        o_e, o_t, o_hd, _ = O.decoder_forward(sd, o_z, g, x, mod, return_states=True)
        close_scaled(hd, o_hd, what="decoder states")
        close_scaled(t, o_t, what="type logits")
        close_scaled(e, o_e, what="edge logits")
        assert torch.equal(e.cpu() != 0, torch.tril(torch.ones_like(g), -1).bool() & (e.cpu() != 0))


@pytest.mark.parametrize("case", list(CASES))
def test_loss_and_grad_parity(cuda, case):
    from oracle import dense_ref as O
    mod, bidir = CASES[case][0], CASES[case][1]
    model, g, x, eps = _make(case, cuda)
    sd = oracle_sd(model, requires_grad=True)
    o = O.train_loss(sd, g, x, eps, mod, bidir, beta=0.005)
    o[0].backward()
    X = {'graph': g.to(cuda), 'node_encoding': x.to(cuda)}
    out = model.training_step(X, 0, eps=eps.to(cuda))
    logs = out['log']
    for k, ref in zip(('train/loss', 'train/edge_loss', 'train/vertex_loss', 'train/kl_loss'), o[:4]):
        assert abs(logs[k] - float(ref)) <= 1e-4 * abs(float(ref)) + 1e-5, (k, logs[k], float(ref))
    model.zero_grad()
    out['loss'].backward()
    params = dict(model.named_parameters())
    checked = 0
    unused = 0
    for k, ref in sd.items():
        if k not in params:
            continue
        if ref.grad is None:
            # a parameter the reference's graph never reaches (degenerate sizes only: e.g. with two
            # nodes nothing consumes the decoder's gate/map output): here its gradient is exactly 0
            assert not params[k].grad.any(), "grad %s should be zero" % k
            unused += 1
        else:
            close_scaled(params[k].grad, ref.grad, tol=3e-4, what="grad " + k)
        checked += 1
    assert checked == len(params)
    assert unused == 0 or CASES[case][4] <= 2
    # validation_step uses self.beta (dvae_model.py:454)
    model.beta = 0.3
    v = model.validation_step(X, 0, eps=eps.to(cuda))
    ov = O.elbo(o[6].detach(), o[7].detach(), g, x, o[4].detach(), o[5].detach(), beta=0.3)
    assert abs(float(v['val_loss']) - float(ov[0])) <= 1e-4 * abs(float(ov[0]))
    # fused no-autograd step gives the same gradients (shared gate/map networks: the two modules' parts summed)
    if True:
        ref_grads = {k: p.grad.clone() for k, p in params.items()}
        model.zero_grad()
        out4 = model.loss_and_grad(X, eps=eps.to(cuda))
        assert abs(float(out4[3]) - logs['train/loss']) <= 1e-6 * abs(logs['train/loss'])
        for k, p in params.items():
            assert torch.equal(p.grad, ref_grads[k]), k


def test_ragged_parity(cuda):
    """Mixed-size batch (beyond the reference, which forces N == max_seq); oracle
    = per-size groups through the dense path."""
    import dvae_amd
    from oracle import dense_ref as O
    H, C, Nmax, B = 64, 6, 16, 14
    hp = Namespace(dataset_file=None, mod=None, bidirectional=True, num_classes=C, max_seq=Nmax,
                   hidden_state_size=H)
    model = fill_weights(dvae_amd.Dvae(hp), seed=3).to(cuda)
    g, x, sizes = dvae_amd.synth.ragged_dags(B, C, 7, nmin=3, nmax=Nmax)
    eps = torch.randn(B, 56, generator=torch.Generator().manual_seed(1))
    sd = oracle_sd(model, requires_grad=True)
    o = O.train_loss_ragged(sd, g, x, sizes, eps, None, True)
    o[0].backward()
    X = {'graph': g.to(cuda), 'node_encoding': x.to(cuda), 'sizes': sizes.to(cuda)}
    out = model.training_step(X, 0, eps=eps.to(cuda))
    for k, ref in zip(('train/loss', 'train/edge_loss', 'train/vertex_loss', 'train/kl_loss'), o[:4]):
        assert abs(out['log'][k] - float(ref)) <= 1e-4 * abs(float(ref)) + 1e-5, (k, out['log'][k], float(ref))
    out['loss'].backward()
    for k, p in model.named_parameters():
        close_scaled(p.grad, sd[k].grad, tol=3e-4, what="grad " + k)


@pytest.mark.parametrize("mod", [None, "bn"])
def test_predict_parity(cuda, mod):
    """Greedy decode must be bit-exact on graphs whose decision margins are well
    above the fp32 noise; stochastic decode with injected uniforms likewise."""
    import dvae_amd
    from oracle import dense_ref as O
    H, N, C, M = 501, 8, 8, 48
    if mod == "bn":
        N = C = 10
    hp = Namespace(dataset_file=None, mod=mod, bidirectional=False, num_classes=C, max_seq=N, hidden_state_size=H)
    model = fill_weights(dvae_amd.Dvae(hp), seed=21, scale=None).to(cuda)
    sd = oracle_sd(model)
    z = 3.0 * torch.randn(M, 56, generator=torch.Generator().manual_seed(4))
    with torch.no_grad():
        o_edges, o_types = O.decoder_predict(sd, z, N, C, mod)
        margins = torch.tensor([O.decoder_predict(sd, z[m:m + 1], N, C, mod, return_margin=True)[2] for m in range(M)])
    edges, types, margin = model.decoder.predict(z.to(cuda), return_margin=True)
    keep = margins > 2e-3
    assert int(keep.sum()) >= M // 4, "too few graphs with a safe margin"
    assert torch.equal(edges.cpu()[keep], o_edges[keep])
    assert torch.equal(types.cpu()[keep], o_types[keep])
    assert abs(float(margin) - float(margins.min())) < 1e-3
    # stochastic, injected uniforms: compare away from decision boundaries by re-deciding
    gen = torch.Generator().manual_seed(8)
    u = {'type': torch.rand(M, N, generator=gen), 'edge': torch.rand(M, N, N, generator=gen)}
    with torch.no_grad():
        s_edges, s_types = O.decoder_predict(sd, z, N, C, mod, uniforms=u)
    edges, types = model.decoder.predict(z.to(cuda), uniforms=(u['type'].to(cuda), u['edge'].to(cuda)))
    same = (edges.cpu() == s_edges).flatten(1).all(1) & (types.cpu() == s_types).flatten(1).all(1)
    assert int(same.sum()) >= M - 2, "stochastic decode differs on %d/%d graphs" % (M - int(same.sum()), M)


@pytest.mark.parametrize("seed", list(range(40)))
def test_random_configurations_against_oracle(cuda, seed):
    """Seeded random model / graph configurations (sizes, widths, modes, densities, orphan sources):
    the four losses and every parameter gradient against the oracle."""
    import random
    import dvae_amd
    from oracle import dense_ref as O
    rnd = random.Random(1000 + seed)
    mod = rnd.choice([None, None, "bn"])
    bidir = rnd.random() < 0.4
    N = rnd.randint(1, 13)
    C = rnd.randint(2, 12)
    H = rnd.choice([24, 40, 64, 100, 200])
    B = rnd.randint(1, 10)
    p = rnd.choice([0.15, 0.3, 0.5, 0.8])
    chain = rnd.random() < 0.4
    fix_orphans = rnd.random() < 0.6
    hp = Namespace(dataset_file=None, mod=mod, bidirectional=bidir, num_classes=C, max_seq=N,
                   same_mapper_gater=False, hidden_state_size=H)
    model = fill_weights(dvae_amd.Dvae(hp), seed=50 + seed).to(cuda)
    g, x = dvae_amd.synth.random_dags(B, N, C, p, seed=70 + seed, chain=chain, fix_orphans=fix_orphans)
    eps = torch.randn(B, 56, generator=torch.Generator().manual_seed(90 + seed))
    sd = oracle_sd(model, requires_grad=True)
    o = O.train_loss(sd, g, x, eps, mod, bidir, beta=0.005)
    o[0].backward()
    X = {'graph': g.to(cuda), 'node_encoding': x.to(cuda)}
    out4 = model.loss_and_grad(X, eps=eps.to(cuda))
    what = "mod=%s bidir=%s N=%d C=%d H=%d B=%d p=%.2f chain=%s orphans_fixed=%s" % (mod, bidir, N, C, H, B, p, chain,
                                                                                      fix_orphans)
    for got, ref in zip(out4.tolist(), (o[1], o[2], o[3], o[0])):
        assert abs(got - float(ref)) <= 1e-4 * abs(float(ref)) + 1e-5, (what, got, float(ref))
    params = dict(model.named_parameters())
    for k, ref in sd.items():
        if k not in params:
            continue
        if ref.grad is None:
            assert not params[k].grad.any(), (what, k)
        else:
            close_scaled(params[k].grad, ref.grad, tol=3e-4, what="%s: grad %s" % (what, k))
