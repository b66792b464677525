"""Shared encoder rows (include/dvae_hip.h dvae_sched_types; sched.hip): the value numbering of the encoder's
forward sweep against its definition (oracle/levels.py shared_encoder_rows, integer: exact), the encoder on
shared rows against the encoder on one row per node (forward: bitwise; gradients: fp32 summation order), and a
batch full of repeated graphs against the dense oracle."""
from argparse import Namespace

import numpy as np
import pytest
import torch

from helpers import close_scaled, fill_weights, oracle_sd

pytestmark = pytest.mark.gpu


def _default_slots():
    """The library's start-up setting (sched.hip g_merge_min_slots), to put back after a test changed it."""
    import os
    if os.environ.get("DVAE_NO_MERGE"):
        return -1
    return int(os.environ.get("DVAE_MERGE_MIN", 8192))


@pytest.fixture
def share_always():
    """Share rows at any batch size (the default starts at 8192 node slots)."""
    from dvae_amd import _lib
    lib = _lib.load()
    lib.dvae_enc_merge_tune(0)
    yield lib
    lib.dvae_enc_merge_tune(_default_slots())


def _repeated(B, N, C, p, seed, distinct, chain=True, fix_orphans=True):
    """B graphs drawn from `distinct` templates whose first nodes are again shared between templates."""
    import dvae_amd
    g0, x0 = dvae_amd.synth.random_dags(distinct, N, C, p, seed, chain=chain, fix_orphans=fix_orphans)
    half = N // 2
    for k in range(1, distinct, 2):          # odd templates copy the first half of their left neighbour
        g0[k, :half, :half] = g0[k - 1, :half, :half]
        x0[k, :half] = x0[k - 1, :half]
    pick = torch.randint(0, distinct, (B,), generator=torch.Generator().manual_seed(seed + 1))
    return g0[pick].clone(), x0[pick].clone()


SCHED_CASES = {
    # name: (B, N, C, p, chain, fix_orphans, distinct templates or 0, ragged)
    "enas_like": (64, 8, 8, 1.0 / 3, True, True, 0, False),
    "two_types": (48, 9, 2, 0.3, True, True, 0, False),
    "repeated": (40, 8, 3, 0.4, True, True, 6, False),
    "orphans": (30, 10, 2, 0.2, False, False, 0, False),
    "wide": (6, 100, 2, 0.05, True, True, 0, False),
    "ragged": (24, 12, 2, 0.3, False, True, 0, True),
    "one_graph": (1, 5, 3, 0.5, True, True, 0, False),
}


@pytest.mark.parametrize("case", list(SCHED_CASES))
def test_value_rows_match_definition(cuda, share_always, case):
    import dvae_amd
    from dvae_amd import engine as E
    from oracle import levels as L
    B, N, C, p, chain, fix, distinct, ragged = SCHED_CASES[case]
    sizes = None
    if ragged:
        g, x, sizes = dvae_amd.synth.ragged_dags(B, C, 3, nmin=2, nmax=N)
    elif distinct:
        g, x = _repeated(B, N, C, p, 21, distinct, chain, fix)
    else:
        g, x = dvae_amd.synth.random_dags(B, N, C, p, 21, chain=chain, fix_orphans=fix)
    s = E.Schedule(g.to(cuda), None if sizes is None else sizes.to(cuda), want_reverse=False, x=x.to(cuda))
    assert s.shared_rows is not None
    rep, node2row, csr, csc = L.shared_encoder_rows(g.numpy(), x.argmax(2).numpy(), None if sizes is None else sizes.numpy())
    Rm, Em = len(csr), sum(len(c) for c in csr)
    assert s.shared_rows == (Rm, Em)
    RM = B * N
    ex = lambda which, n: s.export(which, n).cpu().numpy().astype(np.int64)
    assert np.array_equal(ex(17, RM), rep)
    assert np.array_equal(ex(10, RM), node2row)
    ptr = np.concatenate([[0], np.cumsum([len(c) for c in csr])])
    assert np.array_equal(ex(13, RM + 1)[:Rm + 1], ptr)
    assert np.array_equal(ex(14, max(Em, 1))[:Em], np.array([u for c in csr for u in c], dtype=np.int64))
    ptr = np.concatenate([[0], np.cumsum([len(c) for c in csc])])
    assert np.array_equal(ex(15, RM + 1)[:Rm + 1], ptr)
    assert np.array_equal(ex(16, max(Em, 1))[:Em], np.array([w for c in csc for w in c], dtype=np.int64))
    reps = sorted(set(int(r) for r in rep if r >= 0), key=lambda r: node2row[r])
    assert np.array_equal(ex(11, RM)[:Rm], np.array([r // N for r in reps]))
    assert np.array_equal(ex(12, RM)[:Rm], np.array([r % N for r in reps]))
    # the sharing is real in every case but the single graph
    assert Rm < int((node2row >= 0).sum()) or B == 1
    # the level summary: value rows per level
    from dvae_amd import _lib
    m = _lib.sched_merge_at(N)
    lev = L.forward_levels(g.numpy())
    starts = [sum(1 for r in reps if lev[r // N, r % N] < k) for k in range(N + 1)]
    assert list(s.info[m + 4:m + 5 + N]) == starts


def test_not_shared_without_types_or_below_threshold(cuda):
    import dvae_amd
    from dvae_amd import engine as E
    from dvae_amd import _lib
    lib = _lib.load()
    g, x = dvae_amd.synth.enas_like(16, 3)
    xs = x.clone()
    xs[3, 2] *= 0.5                                                        # not one-hot
    try:
        lib.dvae_enc_merge_tune(8192)                                     # the default
        assert E.Schedule(g.to(cuda)).shared_rows is None                  # no dvae_sched_types call
        assert E.Schedule(g.to(cuda), x=x.to(cuda)).shared_rows is None    # 128 node slots < 8192
        lib.dvae_enc_merge_tune(0)
        assert E.Schedule(g.to(cuda), x=xs.to(cuda)).shared_rows is None
        assert E.Schedule(g.to(cuda), x=x.to(cuda)).shared_rows is not None
    finally:
        lib.dvae_enc_merge_tune(_default_slots())


@pytest.mark.parametrize("mod,bidir", [(None, False), (None, True), ("bn", False), ("bn", True)])
def test_encoder_on_shared_rows_equals_one_row_per_node(cuda, mod, bidir):
    import dvae_amd
    from dvae_amd import _lib
    lib = _lib.load()
    B, N, C = 96, 9, 3
    hp = Namespace(dataset_file=None, mod=mod, bidirectional=bidir, num_classes=C, max_seq=N, hidden_state_size=96)
    model = fill_weights(dvae_amd.Dvae(hp), seed=4).to(cuda)
    g, x = _repeated(B, N, C, 0.35, 8, 10)
    X = {'graph': g.to(cuda), 'node_encoding': x.to(cuda)}
    w = torch.randn(B, 56, generator=torch.Generator().manual_seed(2)).to(cuda)
    res = []
    # the bitwise comparison below needs both runs on the same kernel family: shared rows change the row COUNT of a level
    # (bn: 864 node rows against ~a third of that), and the library picks small-level kernels or GEMM + epilogue by row count
    lib.dvae_small_rows_tune(1 << 30)
    try:
        for slots in (-1, 0):
            lib.dvae_enc_merge_tune(slots)
            try:
                model.zero_grad()
                hv, mu, lv = model.encoder(X)
                ((mu * w).sum() + (lv * w.flip(0)).sum() + hv.sum()).backward()
                fw, bw = model.encoder.node_states(X)
                res.append((hv.detach(), mu.detach(), lv.detach(), fw, bw,
                            {k: p.grad.clone() for k, p in model.encoder.named_parameters()}))
            finally:
                lib.dvae_enc_merge_tune(_default_slots())
    finally:
        lib.dvae_small_rows_tune(768)     # the library's default
    a, b = res
    for k in range(4):
        assert torch.equal(a[k], b[k]), "forward output %d differs" % k      # same arithmetic on the same operands
    if bidir:
        assert torch.equal(a[4], b[4])
    for k in a[5]:
        close_scaled(b[5][k], a[5][k], tol=2e-5, what="grad " + k)


@pytest.mark.parametrize("mod,bidir", [(None, False), (None, True), ("bn", True)])
def test_repeated_graphs_against_oracle(cuda, share_always, mod, bidir):
    """Loss and every gradient of a batch in which most encoder rows are shared (whole graphs repeat, too)."""
    import dvae_amd
    from oracle import dense_ref as O
    B, N, C = 28, 8, 3
    hp = Namespace(dataset_file=None, mod=mod, bidirectional=bidir, num_classes=C, max_seq=N, hidden_state_size=64)
    model = fill_weights(dvae_amd.Dvae(hp), seed=6).to(cuda)
    g, x = _repeated(B, N, C, 0.4, 12, 5)
    eps = torch.randn(B, 56, generator=torch.Generator().manual_seed(5))
    sd = oracle_sd(model, requires_grad=True)
    o = O.train_loss(sd, g, x, eps, mod, bidir, beta=0.005)
    o[0].backward()
    X = {'graph': g.to(cuda), 'node_encoding': x.to(cuda)}
    assert model._schedule(X).shared_rows is not None
    out4 = model.loss_and_grad(X, eps=eps.to(cuda))
    for k in range(3):
        ref = float(o[1 + k])
        assert abs(float(out4[k]) - ref) <= 1e-4 * abs(ref) + 1e-5
    assert abs(float(out4[3]) - float(o[0])) <= 1e-4 * abs(float(o[0]))
    for k, p in model.named_parameters():
        close_scaled(p.grad, sd[k].grad, tol=3e-4, what="grad " + k)
