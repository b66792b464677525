"""GPU parity at the sizes of BASELINE configs 3 / 4 (64-node and 100+-node graphs) against the dense
oracle, element-wise tolerances on the reference's golden vectors, and the fp64 argument for the
gradient bar.  The dense oracle is O(N^3) per graph, so these cases hold one or two graphs.

 * 64 nodes at hs=501: the N > 32 scheduler masks, 63-level sweeps, the two-kernel / blocked pair
   backward at its real width (dvae_model.py:196-259).
 * ragged, 110 + 37 nodes, bidirectional: sizes beyond the reference (oracle = per-size dense runs).
"""
from argparse import Namespace

import numpy as np
import pytest
import torch

from helpers import assert_close, close_scaled, fill_weights, oracle_sd

pytestmark = pytest.mark.gpu


def _worst_rel(a, b, floor):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float(((a - b).abs() / b.abs().clamp_min(floor)).max())


def test_n64_hs501_decoder_against_oracle(cuda):
    import dvae_amd
    from oracle import dense_ref as O
    H, N, C, B = 501, 64, 8, 2
    hp = Namespace(dataset_file=None, mod=None, bidirectional=False, num_classes=C, max_seq=N, hidden_state_size=H)
    model = fill_weights(dvae_amd.Dvae(hp), seed=31).to(cuda)
    g, x = dvae_amd.synth.random_dags(B, N, C, 0.1, seed=64)
    eps = torch.randn(B, 56, generator=torch.Generator().manual_seed(2))
    sd = oracle_sd(model, requires_grad=True)
    # oracle: one pass, states kept
    _, o_mu, o_lv, o_fw, _ = O.encoder_forward(sd, g, x, None, False, return_states=True)
    o_z = O.reparameterize(o_mu, o_lv, eps)
    o_e, o_t, o_hd, _ = O.decoder_forward(sd, o_z, g, x, None, return_states=True)
    o = O.elbo(o_e, o_t, g, x, o_mu, o_lv, 0.005)
    o[0].backward()
    X = {'graph': g.to(cuda), 'node_encoding': x.to(cuda)}
    with torch.no_grad():
        fw, _ = model.encoder.node_states(X)
        close_scaled(fw, o_fw, what="encoder states (64 nodes)")
        zc = o_z.detach().to(cuda)
        e, t = model.decoder(zc, X)
        close_scaled(model.decoder.node_states(zc, X), o_hd, what="decoder states (64 nodes)")
        close_scaled(t, o_t, what="type logits (64 nodes)")
        close_scaled(e, o_e, what="edge logits (64 nodes)")
        assert torch.equal(e.cpu() != 0, torch.tril(torch.ones_like(g), -1).bool() & (e.cpu() != 0))
    out = model.training_step(X, 0, eps=eps.to(cuda))
    for k, ref in zip(('train/loss', 'train/edge_loss', 'train/vertex_loss', 'train/kl_loss'), o):
        assert abs(out['log'][k] - float(ref)) <= 1e-4 * abs(float(ref)) + 1e-5, (k, out['log'][k], float(ref))
    out['loss'].backward()
    for k, p in model.named_parameters():
        close_scaled(p.grad, sd[k].grad, tol=3e-4, what="grad " + k)
    # the fused no-autograd step walks the same kernels
    auto = {k: p.grad.clone() for k, p in model.named_parameters()}
    out4 = model.loss_and_grad(X, eps=eps.to(cuda))
    assert abs(float(out4[3]) - out['log']['train/loss']) <= 1e-6 * abs(out['log']['train/loss'])
    for k, p in model.named_parameters():
        assert torch.equal(p.grad, auto[k]), k


def test_ragged_110_nodes_bidirectional_against_oracle(cuda):
    import dvae_amd
    from oracle import dense_ref as O
    H, C, Nmax = 64, 6, 128
    sizes = torch.tensor([110, 37], dtype=torch.int32)
    B = int(sizes.numel())
    g = torch.zeros(B, Nmax, Nmax)
    x = torch.zeros(B, Nmax, C)
    for b, n in enumerate(sizes.tolist()):
        gb, xb = dvae_amd.synth.random_dags(1, n, C, min(0.5, 4.0 / n), seed=900 + b)
        g[b, :n, :n], x[b, :n] = gb[0], xb[0]
    hp = Namespace(dataset_file=None, mod=None, bidirectional=True, num_classes=C, max_seq=Nmax, hidden_state_size=H)
    model = fill_weights(dvae_amd.Dvae(hp), seed=33).to(cuda)
    eps = torch.randn(B, 56, generator=torch.Generator().manual_seed(3))
    sd = oracle_sd(model, requires_grad=True)
    o = O.train_loss_ragged(sd, g, x, sizes, eps, None, True)
    o[0].backward()
    X = {'graph': g.to(cuda), 'node_encoding': x.to(cuda), 'sizes': sizes.to(cuda)}
    # logits of the 110-node graph against its own dense run
    with torch.no_grad():
        n = int(sizes[0])
        sd0 = {k: v.detach() for k, v in sd.items()}
        _, o_mu, o_lv = O.encoder_forward(sd0, g[:1, :n, :n], x[:1, :n], None, True)
        _, mu, lv = model.encoder(X)
        close_scaled(mu[:1], o_mu, what="mu of the 110-node graph")
        close_scaled(lv[:1], o_lv, what="logvar of the 110-node graph")
        o_z = O.reparameterize(o_mu, o_lv, eps[:1])
        o_e, o_t, o_hd, _ = O.decoder_forward(sd0, o_z, g[:1, :n, :n], x[:1, :n], None, return_states=True)
        z = torch.zeros(B, 56)
        z[:1] = o_z
        e, t = model.decoder(z.to(cuda), X)
        hd = model.decoder.node_states(z.to(cuda), X)
        close_scaled(hd[:1, :n], o_hd, what="decoder states (110 nodes)")
        close_scaled(e[:1, :n, :n], o_e, what="edge logits (110 nodes)")
        close_scaled(t[:1, :n], o_t, what="type logits (110 nodes)")
        assert not e[0, n:].any() and not e[0, :, n:].any() and not t[0, n:].any()
    out = model.training_step(X, 0, eps=eps.to(cuda))
    for k, ref in zip(('train/loss', 'train/edge_loss', 'train/vertex_loss', 'train/kl_loss'), o[:4]):
        assert abs(out['log'][k] - float(ref)) <= 1e-4 * abs(float(ref)) + 1e-5, (k, out['log'][k], float(ref))
    out['loss'].backward()
    for k, p in model.named_parameters():
        close_scaled(p.grad, sd[k].grad, tol=3e-4, what="grad " + k)


def test_ragged_90_nodes_hs256_multiwindow_pair_kernels_against_oracle(cuda):
    """Config 4's PRODUCTION pair kernels against the dense oracle (dvae_model.py:196-259): at hidden_state_size = 256
    the add_edge width is EHP = 1024, which selects pair_fwd_slice_kernel -- with max_seq > 64 in windows of 56
    candidates, so the 90-node graph takes two windows -- and pair_bwd_win_kernel (whole groups of 1024 columns),
    where the 110-node hs = 64 case above runs the fallback scorers.  Ragged, bidirectional, sizes {90, 23}:
    decoder states, edge and type logits, the four losses and every gradient."""
    import dvae_amd
    from dvae_amd import _lib
    from oracle import dense_ref as O
    H, C, Nmax = 256, 6, 96
    dims = _lib.make_dims(2, Nmax, C, H, 56, None, False, Nmax)
    assert dims.EHP == 1024
    path = _lib.load().dvae_pair_path(Nmax, dims.EHP)
    assert path == (1 | 2 | 8), "expected sliced multi-window forward + windowed backward, got %d" % path
    sizes = torch.tensor([90, 23], dtype=torch.int32)
    B = int(sizes.numel())
    g = torch.zeros(B, Nmax, Nmax)
    x = torch.zeros(B, Nmax, C)
    for b, n in enumerate(sizes.tolist()):
        gb, xb = dvae_amd.synth.random_dags(1, n, C, min(0.5, 4.0 / n), seed=700 + b)
        g[b, :n, :n], x[b, :n] = gb[0], xb[0]
    # candidates beyond the first window must carry edges, or the second window is never exercised
    assert g[0, :90, 56:90].sum() > 0
    hp = Namespace(dataset_file=None, mod=None, bidirectional=True, num_classes=C, max_seq=Nmax, hidden_state_size=H)
    model = fill_weights(dvae_amd.Dvae(hp), seed=35).to(cuda)
    eps = torch.randn(B, 56, generator=torch.Generator().manual_seed(4))
    sd = oracle_sd(model, requires_grad=True)
    o = O.train_loss_ragged(sd, g, x, sizes, eps, None, True)
    o[0].backward()
    X = {'graph': g.to(cuda), 'node_encoding': x.to(cuda), 'sizes': sizes.to(cuda)}
    with torch.no_grad():
        sd0 = {k: v.detach() for k, v in sd.items()}
        _, mu, lv = model.encoder(X)
        z = torch.zeros(B, 56)
        per = []
        for b, n in enumerate(sizes.tolist()):
            _, o_mu, o_lv = O.encoder_forward(sd0, g[b:b + 1, :n, :n], x[b:b + 1, :n], None, True)
            close_scaled(mu[b:b + 1], o_mu, what="mu of graph %d" % b)
            close_scaled(lv[b:b + 1], o_lv, what="logvar of graph %d" % b)
            z[b:b + 1] = O.reparameterize(o_mu, o_lv, eps[b:b + 1])
            per.append(O.decoder_forward(sd0, z[b:b + 1], g[b:b + 1, :n, :n], x[b:b + 1, :n], None, return_states=True))
        e, t = model.decoder(z.to(cuda), X)
        hd = model.decoder.node_states(z.to(cuda), X)
        for b, n in enumerate(sizes.tolist()):
            o_e, o_t, o_hd, _ = per[b]
            close_scaled(hd[b:b + 1, :n], o_hd, what="decoder states (%d nodes)" % n)
            close_scaled(e[b:b + 1, :n, :n], o_e, what="edge logits (%d nodes)" % n)
            close_scaled(t[b:b + 1, :n], o_t, what="type logits (%d nodes)" % n)
            assert not e[b, n:].any() and not e[b, :, n:].any() and not t[b, n:].any()
    out = model.training_step(X, 0, eps=eps.to(cuda))
    for k, ref in zip(('train/loss', 'train/edge_loss', 'train/vertex_loss', 'train/kl_loss'), o[:4]):
        assert abs(out['log'][k] - float(ref)) <= 1e-4 * abs(float(ref)) + 1e-5, (k, out['log'][k], float(ref))
    out['loss'].backward()
    for k, p in model.named_parameters():
        close_scaled(p.grad, sd[k].grad, tol=3e-4, what="grad " + k)


@pytest.mark.parametrize("name", ["enas_val8_none_bidir", "rand8_none_uni", "rand10_bn_bidir_shared"])
def test_golden_elementwise_rtol(cuda, name, record_property, gemm_core):
    """north_star's bar read element by element: |hip - reference| <= 1e-6 + 1e-4 |reference| for every
    per-node hidden state, mu, logvar and loss of the reference's golden vectors (the other tests take
    1e-4 of the tensor's scale).  The worst element-wise relative error is recorded with the result."""
    from test_oracle_golden import build_model, load_case
    d, mod, bidir, shared = load_case(name)
    model = build_model(d, mod, bidir, shared, cuda)
    g, x, eps = (torch.from_numpy(d[k]).to(cuda) for k in ('g', 'x', 'eps'))
    X = {'graph': g, 'node_encoding': x}
    worst = {}
    with torch.no_grad():
        hv, mu, lv = model.encoder(X)
        fw, bw = model.encoder.node_states(X)
        zt = torch.from_numpy(d['z']).to(cuda)
        hd = model.decoder.node_states(zt, X)
        pairs = [("enc_states", fw), ("hv", hv), ("mu", mu), ("logvar", lv), ("dec_states", hd)]
        if bidir:
            pairs.append(("enc_states_rev", bw))
        for key, got in pairs:
            ref = torch.from_numpy(d[key])
            assert_close(got, ref, rtol=1e-4, atol=1e-6, what="%s/%s" % (name, key))
            worst[key] = _worst_rel(got, ref, 1e-3)
    out = model.training_step(X, 0, eps=eps)
    got = np.array([out['log'][k] for k in ('train/edge_loss', 'train/vertex_loss', 'train/kl_loss', 'train/loss')])
    np.testing.assert_allclose(got, d['losses'], rtol=1e-4, atol=1e-6)
    worst["losses"] = float(np.abs(got / d['losses'] - 1).max())
    record_property("worst_elementwise_rel_err", worst)
    print("%s worst element-wise relative error (|ref| floored at 1e-3): %s" % (
        name, {k: "%.1e" % v for k, v in worst.items()}))


def test_gradients_against_fp64_oracle(cuda):
    """Why the gradient bar against the fp32 oracle is 3e-4 of scale: against the same oracle run in
    fp64 the HIP gradients are within 1e-5 of scale, while the fp32 dense-loop oracle is itself up to
    3e-4 away (the reference re-sums every gate/map row at every step, dvae_model.py:77-80,230-241).
    Case = seed 198 of the random sweep (13 nodes, p=0.8, hs=200), the one configuration of 200 that
    missed the fp32 bar, plus the headline shape."""
    import dvae_amd
    from oracle import dense_ref as O
    for (N, C, H, B, p, chain, wseed, gseed, eseed) in [(13, 4, 200, 10, 0.8, True, 248, 268, 288),
                                                        (8, 8, 501, 4, 1.0 / 3, True, 11, 5, 9)]:
        hp = Namespace(dataset_file=None, mod=None, bidirectional=False, num_classes=C, max_seq=N,
                       same_mapper_gater=False, hidden_state_size=H)
        model = fill_weights(dvae_amd.Dvae(hp), seed=wseed).to(cuda)
        g, x = dvae_amd.synth.random_dags(B, N, C, p, seed=gseed, chain=chain)
        eps = torch.randn(B, 56, generator=torch.Generator().manual_seed(eseed))
        sd64 = {k: v.double().requires_grad_(True) for k, v in oracle_sd(model).items()}
        r64 = O.train_loss(sd64, g.double(), x.double(), eps.double(), None, False)
        r64[0].backward()
        X = {'graph': g.to(cuda), 'node_encoding': x.to(cuda)}
        out4 = model.loss_and_grad(X, eps=eps.to(cuda))
        assert abs(float(out4[3]) - float(r64[0])) <= 2e-6 * abs(float(r64[0]))
        for k, prm in model.named_parameters():
            g64 = sd64[k].grad
            scale = max(1e-3, float(g64.abs().max()))
            err = float((prm.grad.cpu().double() - g64).abs().max())
            assert err <= 1e-5 * scale, "grad %s: %.2e of scale against the fp64 oracle" % (k, err / scale)


def test_prepare_converts_on_its_own_stream(cuda):
    """Dvae.prepare with inputs that need a conversion (bool / non-contiguous adjacency, float64
    one-hots, int64 sizes): the conversions must be ordered before the build kernels on the build
    stream, resident or not."""
    import dvae_amd
    H, C, Nmax, B = 64, 6, 16, 300
    hp = Namespace(dataset_file=None, mod=None, bidirectional=True, num_classes=C, max_seq=Nmax, hidden_state_size=H)
    model = fill_weights(dvae_amd.Dvae(hp), seed=3).to(cuda)
    g, x, sizes = dvae_amd.synth.ragged_dags(B, C, 7, nmin=3, nmax=Nmax)
    eps = torch.randn(B, 56, generator=torch.Generator().manual_seed(1)).to(cuda)
    plain = {'graph': g.to(cuda), 'node_encoding': x.to(cuda), 'sizes': sizes.to(cuda)}
    want = model.loss_and_grad(plain, eps=eps).clone()
    want_g = {k: p.grad.clone() for k, p in model.named_parameters()}
    # transposed storage (non-contiguous view), bool dtype, float64 encodings, int64 sizes
    g_nc = g.to(cuda).transpose(1, 2).contiguous().transpose(1, 2).bool()
    assert not g_nc.is_contiguous()
    odd = {'graph': g_nc, 'node_encoding': x.to(cuda).double(), 'sizes': sizes.to(cuda).long()}
    for resident in (False, True):
        torch.cuda.synchronize()
        big = torch.randn(4096, 4096, device=cuda)
        for _ in range(4):
            big = big @ big.clamp(-1e-3, 1e-3)          # keep the main stream busy while prepare() queues its work
        sched = model.prepare(odd, resident=resident)
        out4 = model.loss_and_grad(odd, eps=eps, sched=sched)
        assert torch.equal(out4, want), (resident, out4, want)
        for k, p in model.named_parameters():
            assert torch.equal(p.grad, want_g[k]), (resident, k)


def test_backward_after_weight_change_is_refused(cuda):
    """The autograd wrappers read the module's one packed parameter copy in backward; an in-place
    weight change between forward and backward must trip autograd's version check, not silently mix
    old activations with new weights."""
    import dvae_amd
    hp = Namespace(dataset_file=None, mod=None, bidirectional=False, num_classes=5, max_seq=6, hidden_state_size=40)
    model = fill_weights(dvae_amd.Dvae(hp), seed=4).to(cuda)
    g, x = dvae_amd.synth.random_dags(4, 6, 5, 0.4, seed=1)
    X = {'graph': g.to(cuda), 'node_encoding': x.to(cuda)}
    out = model.training_step(X, 0)
    with torch.no_grad():
        model.decoder.gru.weight_hh.mul_(1.01)
    with pytest.raises(RuntimeError, match="modified by an inplace operation"):
        out['loss'].backward()
    # two forwards before the first backward (gradient accumulation) stay legal while weights are unchanged
    model.zero_grad()
    a = model.training_step(X, 0, eps=torch.zeros(4, 56, device=cuda))['loss']
    b = model.training_step(X, 0, eps=torch.zeros(4, 56, device=cuda))['loss']
    a.backward()
    g1 = {k: p.grad.clone() for k, p in model.named_parameters()}
    b.backward()
    for k, p in model.named_parameters():
        close_scaled(p.grad, 2 * g1[k], tol=1e-6, what="accumulated grad " + k)
