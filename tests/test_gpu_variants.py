"""GPU: the alternative execution forms of the train step give the results of the
default form -- fused level kernels (same formulas; the only difference is that
the plain GEMM may split K for small launches, i.e. fp32 summation order), a
schedule built ahead of time (bitwise)."""
from argparse import Namespace

import pytest
import torch

from helpers import close_scaled, fill_weights

pytestmark = pytest.mark.gpu


def _setup(cuda, B, bidir=False, N=8, C=8, H=501, seed=3, mod=None):
    import dvae_amd
    hp = Namespace(dataset_file=None, mod=mod, bidirectional=bidir, num_classes=C, max_seq=N, hidden_state_size=H)
    model = fill_weights(dvae_amd.Dvae(hp), seed).to(cuda)
    g, x = dvae_amd.synth.random_dags(B, N, C, 0.35, seed=21, chain=True)
    eps = torch.randn(B, 56, generator=torch.Generator().manual_seed(4))
    X = {'graph': g.to(cuda), 'node_encoding': x.to(cuda)}
    return model, X, eps.to(cuda)


def _step(model, X, eps, **kw):
    model.zero_grad(set_to_none=True)
    out4 = model.loss_and_grad(X, eps=eps, **kw)
    return out4.clone(), torch.cat([p.grad.flatten() for p in model.parameters()]).clone()


@pytest.mark.parametrize("bidir", [False, True])
def test_fused_level_kernels(cuda, bidir):
    """level.hip (hidden GEMM + GRU gates, gate/map GEMM + sigmoid*map in one launch each)
    against the GEMM + epilogue-kernel form: same formulas, fp32 summation order may differ."""
    from dvae_amd import _lib
    lib = _lib.load()
    model, X, eps = _setup(cuda, 96, bidir=bidir, H=501 if not bidir else 128)
    try:
        lib.dvae_level_tune(-1, 0, 1)
        ref = _step(model, X, eps)
        with torch.no_grad():
            ref_states = model.encoder.node_states(X)[0].clone()
        # gru = 2: node sweeps run CSR aggregation + hidden GEMM + GRU gates as ONE launch (the gather loader)
        for gm_min, gru in ((0, 0), (0, 1), (-1, 1), (-1, 2), (0, 2)):
            lib.dvae_level_tune(gm_min, gru, 1)
            out = _step(model, X, eps)
            with torch.no_grad():
                states = model.encoder.node_states(X)[0]
            close_scaled(states, ref_states, tol=1e-5, what="states %d/%d" % (gm_min, gru))
            close_scaled(out[0], ref[0], tol=1e-5, what="losses %d/%d" % (gm_min, gru))
            # gradients: a 1e-7 change of a pre-activation can flip one ReLU of add_vertex /
            # add_edge (a kink, not an error), which moves single entries by O(1e-2): L2 norm
            rel = float((out[1] - ref[1]).norm() / ref[1].norm())
            assert rel <= 1e-3, "gradients %d/%d: relative L2 error %.3e" % (gm_min, gru, rel)
    finally:
        lib.dvae_level_tune(-2, 0, 1)


@pytest.mark.parametrize("mod,bidir,B", [(None, False, 96), (None, True, 70), ("bn", True, 45), (None, False, 1500)])
def test_small_level_kernels(cuda, mod, bidir, B):
    """level.hip small_kernel (32-row tiles, K split over the waves of a workgroup; gates fused forward, plain
    accumulate backward) against the GEMM + epilogue-kernel form on the same levels: dvae_small_rows_tune(0)
    = never, (1 << 30) = every level.  Same formulas, fp32 summation order differs."""
    from dvae_amd import _lib
    lib = _lib.load()
    model, X, eps = _setup(cuda, B, bidir=bidir, mod=mod, H=501 if B < 100 and not bidir else 128)
    try:
        lib.dvae_small_rows_tune(0)
        ref = _step(model, X, eps)
        with torch.no_grad():
            ref_states = model.encoder.node_states(X)[0].clone()
            ref_logits = [t.clone() for t in model.decoder(eps, X)]
        for rows in (1 << 30, 1024, 768):
            lib.dvae_small_rows_tune(rows)
            out = _step(model, X, eps)
            with torch.no_grad():
                states = model.encoder.node_states(X)[0]
                logits = model.decoder(eps, X)
            close_scaled(states, ref_states, tol=1e-5, what="states, small rows %d" % rows)
            for a, r in zip(logits, ref_logits):
                close_scaled(a, r, tol=1e-5, what="logits, small rows %d" % rows)
            close_scaled(out[0], ref[0], tol=1e-5, what="losses, small rows %d" % rows)
            rel = float((out[1] - ref[1]).norm() / ref[1].norm())      # ReLU kinks: see test_fused_level_kernels
            assert rel <= 1e-3, "gradients, small rows %d: relative L2 error %.3e" % (rows, rel)
    finally:
        lib.dvae_small_rows_tune(768)     # the library's default


def test_prepared_schedule_bitwise(cuda):
    model, X, eps = _setup(cuda, 64)
    ref = _step(model, X, eps)
    sched = model.prepare(X)
    out = _step(model, X, eps, sched=sched)
    assert torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1])
    sched = model.prepare(X, resident=True)
    out = _step(model, X, eps, sched=sched)
    assert torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1])


def test_prepared_schedule_rejects_bad_graph(cuda):
    from dvae_amd import _lib
    model, X, eps = _setup(cuda, 8)
    bad = {k: v.clone() for k, v in X.items()}
    bad['graph'][0, 2, 5] = 1.0          # edge from a later node: not lower-triangular
    sched = model.prepare(bad)
    with pytest.raises(_lib.GraphError):
        model.loss_and_grad(bad, eps=eps, sched=sched)


@pytest.mark.parametrize("mod", [None, "bn"])
def test_type_shared_rows_match_per_row(cuda, mod):
    """One-hot node encodings: the decoder's (type, h_in = 0) prefix rows come from a C-row table
    (dvae_sched_types / TypeShare), and add_edge's first layer runs on the scored prefix rows only
    -- same results as computing every row (dvae_level_tune(.., 0) turns both off)."""
    from dvae_amd import _lib
    lib = _lib.load()
    model, X, eps = _setup(cuda, 300, mod=mod, H=501 if mod is None else 128)
    assert model._schedule(X).onehot
    try:
        lib.dvae_level_tune(-2, 0, 0)
        ref = _step(model, X, eps)
        lib.dvae_level_tune(-2, 0, 1)
        out = _step(model, X, eps)
    finally:
        lib.dvae_level_tune(-2, 0, 1)
    close_scaled(out[0], ref[0], tol=1e-5, what="losses")
    rel = float((out[1] - ref[1]).norm() / ref[1].norm())      # ReLU kinks: see test_fused_level_kernels
    assert rel <= 1e-3, "gradients: relative L2 error %.3e" % rel


def test_soft_node_encodings_fall_back_to_per_row(cuda):
    """Encodings that are not one-hot (the API takes any float rows): the schedule says so and
    every row is computed; checked against the oracle."""
    from oracle import dense_ref as O
    from helpers import oracle_sd
    model, X, eps = _setup(cuda, 6, H=64)
    gen = torch.Generator().manual_seed(8)
    xs = torch.rand(X['node_encoding'].shape, generator=gen)
    Xs = {'graph': X['graph'], 'node_encoding': xs.to(cuda)}
    assert not model._schedule(Xs).onehot
    sd = oracle_sd(model, requires_grad=True)
    ref = O.train_loss(sd, X['graph'].cpu(), xs, eps.cpu(), None, False)
    ref[0].backward()
    out = _step(model, Xs, eps)
    assert abs(float(out[0][3]) - float(ref[0])) <= 1e-4 * abs(float(ref[0]))
    params = dict(model.named_parameters())
    for k, r in sd.items():
        if k in params and r.grad is not None:
            close_scaled(params[k].grad, r.grad, tol=3e-4, what="grad " + k)


def test_generation_eval_on_device(cuda):
    """Decode from the prior on the GPU and score validity / uniqueness / novelty there; the same
    decoded tensors scored on the CPU give the same report (metrics.py is device-agnostic)."""
    import dvae_amd
    from dvae_amd import metrics
    hp = Namespace(dataset_file=None, mod=None, bidirectional=False, num_classes=8, max_seq=8, hidden_state_size=64)
    model = fill_weights(dvae_amd.Dvae(hp), seed=3).to(cuda)
    z = torch.randn(300, 56, generator=torch.Generator().manual_seed(1)).to(cuda)
    g, x = dvae_amd.synth.enas_like(64, 5)
    train = {'graph': g.to(cuda), 'node_encoding': x.to(cuda)}
    rep = model.generation_eval(z=z, train_batch=train)
    assert set(rep) == {'validity', 'uniqueness', 'novelty'} and all(0.0 <= v <= 1.0 for v in rep.values())
    edges, onehots = model.decoder.predict(z)
    ref = metrics.generation_report(edges.cpu(), onehots.cpu(), g, x)
    assert rep == ref
    assert model.generation_eval(num_samples=50, is_stochastic=True).keys() == {'validity', 'uniqueness'}
    # synthetic ENAS-like graphs pass the validity rules once start / end tokens are in place
    xs = x.clone()
    xs[:, 0] = 0.0; xs[:, 0, 6] = 1.0
    xs[:, -1] = 0.0; xs[:, -1, 7] = 1.0
    inner = xs[:, 1:-1].argmax(2) % 6
    xs[:, 1:-1] = torch.nn.functional.one_hot(inner, 8).float()
    gs = g.clone()
    childless = gs[:, :, 1:7].sum(1) == 0
    gs[:, 7, 1:7] = torch.where(childless, torch.ones_like(gs[:, 7, 1:7]), gs[:, 7, 1:7])
    assert metrics.dag_validity(gs.to(cuda), xs.to(cuda)).all()


@pytest.mark.parametrize("fused", [True, False])
def test_trainer_fits_real_enas_rows(cuda, tmp_path, fused):
    """train.py's flow end to end on the GPU: dataset file -> loaders -> Trainer.fit (fused step or
    autograd training_step) -> validation -> checkpoint -> reload gives the same validation loss."""
    import os
    import numpy as np
    import dvae_amd
    from dvae_amd import trainer
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    lines = [str(s) for s in np.load(os.path.join(gold, "enas_rows.npz"))["lines"]]
    path = tmp_path / "structures.txt"
    path.write_text("\n".join(lines * 5) + "\n")                       # 40 rows -> 32 / 4 / 4
    hp = Namespace(dataset_file=str(path), batch_size=8, mod=None, bidirectional=False, hidden_state_size=64)
    torch.manual_seed(0)
    model = dvae_amd.Dvae(hp).to(cuda)
    assert len(model.train_loader.dataset) == 32 and len(model.val_loader[0].dataset) == 4
    logs = []
    t = trainer.Trainer(max_epochs=3, save_dir=str(tmp_path / "ckpt"), fused=fused, log=logs.append)
    hist = t.fit(model)
    assert len(hist) == 3 and all(h['steps'] == 4 for h in hist)
    assert hist[2]['train/loss'] < hist[0]['train/loss']                 # Adam at 1e-4 already moves it
    assert all(np.isfinite(h['val_loss']) for h in hist)
    ck = os.path.join(str(tmp_path / "ckpt"), "_ckpt_epoch_2.ckpt")
    assert os.listdir(str(tmp_path / "ckpt")) == ["_ckpt_epoch_2.ckpt"]
    again = dvae_amd.Dvae(Namespace(dataset_file=str(path), batch_size=8, mod=None, bidirectional=False,
                                    hidden_state_size=64)).to(cuda)
    trainer.load_checkpoint(ck, again)
    torch.manual_seed(5); v1 = float(t.validate(model)['avg_val_loss'])
    torch.manual_seed(5); v2 = float(t.validate(again)['avg_val_loss'])
    assert abs(v1 - v2) <= 1e-6 * abs(v1)


@pytest.mark.parametrize("bidir,B,p", [(False, 300, 0.35), (True, 90, 0.6), (False, 40, 0.1)])
def test_recomputed_gates_are_bit_identical(cuda, gemm_core, bidir, B, p):
    """Decoder levels that rebuild the GRU gates of their prefix rows in the backward pass (dvae_recompute_tune: by default
    levels of 1024 node rows or more) against levels that save them in the forward pass: the same operations in the same
    order, so losses and every gradient agree bit for bit.  Dense and sparse graphs (long and short predecessor lists)."""
    import os
    import dvae_amd
    from dvae_amd import _lib
    lib = _lib.load()
    model, X, eps = _setup(cuda, B, bidir=bidir)
    g, x = dvae_amd.synth.random_dags(B, 8, 8, p, seed=33, chain=True)
    X = {'graph': g.to(cuda), 'node_encoding': x.to(cuda)}
    default = -1 if os.environ.get("DVAE_NO_RECOMPUTE") else int(os.environ.get("DVAE_RECOMPUTE_MIN", 1024))
    try:
        lib.dvae_recompute_tune(-1)
        ref = _step(model, X, eps)
        lib.dvae_recompute_tune(0)
        out = _step(model, X, eps)
    finally:
        lib.dvae_recompute_tune(default)
    assert torch.equal(out[0], ref[0])
    assert torch.equal(out[1], ref[1])


@pytest.mark.parametrize("shared_rows", [False, True])
@pytest.mark.parametrize("bidir", [False, True])
def test_early_weight_gradient_chunk(cuda, gemm_core, bidir, shared_rows):
    """Weight-gradient products of the upper levels issued while the backward sweep still runs
    (forced here at a small size; by default only sweeps of >= 16384 rows do it): the gradients
    are those of the single pass after the sweep -- the loss is bit-identical (forward untouched),
    the gradients differ by the order of two partial sums only.  On both GEMM cores, with and without shared
    encoder rows: the encoder's last products are split between the two streams behind the early chunk (the chain's
    half accumulates into what the side stream's early chunk set), and a repeated step must give the same bits."""
    import os
    from dvae_amd import _lib
    lib = _lib.load()
    model, X, eps = _setup(cuda, 200, bidir=bidir)
    slots0 = -1 if os.environ.get("DVAE_NO_MERGE") else int(os.environ.get("DVAE_MERGE_MIN", 8192))
    try:
        lib.dvae_enc_merge_tune(0 if shared_rows else -1)
        lib.dvae_wgrad_split_tune(0.0, 16384)
        ref = _step(model, X, eps)
        lib.dvae_wgrad_split_tune(0.5, 0)
        out = _step(model, X, eps)
        for _ in range(3):
            again = _step(model, X, eps)
            assert torch.equal(again[0], out[0]) and torch.equal(again[1], out[1]), "the step is not reproducible"
        lib.dvae_wgrad_split_tune(0.2, 0)
        out2 = _step(model, X, eps)
    finally:
        lib.dvae_wgrad_split_tune(0.35, 16384)
        lib.dvae_enc_merge_tune(slots0)
    assert torch.equal(out[0], ref[0]) and torch.equal(out2[0], ref[0])
    for o in (out, out2):
        rel = float((o[1] - ref[1]).norm() / ref[1].norm())
        assert rel <= 1e-5, "gradients: relative L2 error %.3e" % rel
        close_scaled(o[1], ref[1], tol=1e-4, what="gradients")


def test_folded_splitk_combine_is_bit_identical(cuda, tmp_path):
    """Small levels: the split-K slabs of the gate/map and hidden products are summed by the consumer kernel
    (default) or by a combine launch (DVAE_NO_FOLD=1, read once per process): same sums in the same order,
    so losses and gradients agree bit for bit.  The other mode runs in a child process."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "run.py"
    script.write_text('''
import sys, torch
from argparse import Namespace
sys.path.insert(0, %r); sys.path.insert(0, %r)
import dvae_amd
from helpers import fill_weights
dev = torch.device("cuda")
out = {}
for bidir in (False, True):
    model = fill_weights(dvae_amd.Dvae(Namespace(dataset_file=None, mod=None, bidirectional=bidir)), 3).to(dev)
    g, x = dvae_amd.synth.enas_like(300, 5)
    eps = torch.randn(300, 56, generator=torch.Generator().manual_seed(1))
    o4 = model.loss_and_grad({"graph": g.to(dev), "node_encoding": x.to(dev)}, eps=eps.to(dev))
    torch.cuda.synchronize()
    out[bidir] = (o4.cpu(), [p.grad.cpu().clone() for p in model.parameters()])
torch.save(out, sys.argv[1])
''' % (root, os.path.join(root, "tests")))
    assert "DVAE_NO_FOLD" not in os.environ          # this process runs the default (folded) form
    path = str(tmp_path / "nofold.pt")
    subprocess.run([sys.executable, str(script), path], check=True, env=dict(os.environ, DVAE_NO_FOLD="1"), timeout=400)
    other = torch.load(path, weights_only=True)
    import dvae_amd
    from helpers import fill_weights
    for bidir in (False, True):
        model = fill_weights(dvae_amd.Dvae(Namespace(dataset_file=None, mod=None, bidirectional=bidir)), 3).to(cuda)
        g, x = dvae_amd.synth.enas_like(300, 5)
        eps = torch.randn(300, 56, generator=torch.Generator().manual_seed(1))
        o4 = model.loss_and_grad({"graph": g.to(cuda), "node_encoding": x.to(cuda)}, eps=eps.to(cuda))
        torch.cuda.synchronize()
        assert torch.equal(o4.cpu(), other[bidir][0])
        assert all(torch.equal(p.grad.cpu(), v) for p, v in zip(model.parameters(), other[bidir][1]))


def test_windowed_pair_backward_matches_two_kernel_form(cuda, tmp_path):
    """Graphs beyond 8 nodes: the one-pass windowed pair backward (default, 8 candidates per window)
    against the same kernel with 16 per window and against the two-kernel form (DVAE_NO_PAIR_WIN=1), each in
    a child process (the switches are read once per process): ragged 30-node graphs at hs=501 and
    Bayesian-network graphs (three column groups), every gradient within fp32 reorder noise; dL/dBq and
    dL/dA sums are per-thread and fixed-order in all forms, only the order within a row differs."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    body = '''
import sys, torch
from argparse import Namespace
sys.path.insert(0, %r); sys.path.insert(0, %r)
import dvae_amd
from helpers import fill_weights
dev = torch.device("cuda")
def run():
    out = {}
    for name, mod, N, C in (("none", None, 30, 6), ("bn", "bn", 10, 10)):
        hp = Namespace(dataset_file=None, mod=mod, bidirectional=False, num_classes=C, max_seq=N)
        model = fill_weights(dvae_amd.Dvae(hp), 5).to(dev)
        if mod is None:
            g, x, sizes = dvae_amd.synth.ragged_dags(24, C, 3, nmin=9, nmax=N)
            X = {"graph": g.to(dev), "node_encoding": x.to(dev), "sizes": sizes.to(dev)}
        else:
            g, x = dvae_amd.synth.random_dags(40, N, C, 0.3, 4, fix_orphans=False)
            X = {"graph": g.to(dev), "node_encoding": x.to(dev)}
        eps = torch.randn(g.shape[0], 56, generator=torch.Generator().manual_seed(1))
        o4 = model.loss_and_grad(X, eps=eps.to(dev))
        torch.cuda.synchronize()
        out[name] = (o4.cpu(), [p.grad.cpu().clone() for p in model.parameters()])
    return out
''' % (root, os.path.join(root, "tests"))
    script = tmp_path / "run.py"
    script.write_text(body + "torch.save(run(), sys.argv[1])\n")
    assert "DVAE_NO_PAIR_WIN" not in os.environ and "DVAE_PAIR_WIN" not in os.environ
    others = {}
    for tag, env in (("two_kernel", {"DVAE_NO_PAIR_WIN": "1"}), ("win16", {"DVAE_PAIR_WIN": "16"})):
        path = str(tmp_path / (tag + ".pt"))
        subprocess.run([sys.executable, str(script), path], check=True, env=dict(os.environ, **env), timeout=400)
        others[tag] = torch.load(path, weights_only=True)
    ns = {}
    exec(compile(body, "run_body", "exec"), ns)
    mine = ns["run"]()
    for tag, other in others.items():
        for name in ("none", "bn"):
            assert torch.allclose(mine[name][0], other[name][0], rtol=1e-6), (tag, name)
            for a, b in zip(mine[name][1], other[name][1]):
                close_scaled(a, b, tol=2e-5, what="%s/%s gradient" % (tag, name))


def test_decoder_optimizer_step_under_the_encoder_backward(cuda):
    """loss_and_grad(on_decoder_grads=...): the decoder's Adam step issued on the side stream as soon as its
    gradients are final gives the same parameters, bit for bit, as one Adam over everything after the step."""
    import dvae_amd
    hp = Namespace(dataset_file=None, mod=None, bidirectional=True, num_classes=8, max_seq=8, hidden_state_size=128)
    g, x = dvae_amd.synth.enas_like(200, 9)
    X = {'graph': g.to(cuda), 'node_encoding': x.to(cuda)}
    eps = [torch.randn(200, 56, generator=torch.Generator().manual_seed(s)).to(cuda) for s in (1, 2, 3)]
    a = fill_weights(dvae_amd.Dvae(hp), 6).to(cuda)
    b = fill_weights(dvae_amd.Dvae(hp), 6).to(cuda)
    opt_a = torch.optim.Adam(a.parameters(), lr=1e-3, fused=True)
    opt_be = torch.optim.Adam(b.encoder.parameters(), lr=1e-3, fused=True)
    opt_bd = torch.optim.Adam(b.decoder.parameters(), lr=1e-3, fused=True)
    for e in eps:
        la = a.loss_and_grad(X, eps=e)
        opt_a.step()
        lb = b.loss_and_grad(X, eps=e, on_decoder_grads=opt_bd.step)
        opt_be.step()
        assert torch.equal(la, lb)
    torch.cuda.synchronize()
    for (k, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
        assert torch.equal(pa, pb), k
