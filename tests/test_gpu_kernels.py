"""GPU unit tests of the building blocks, each against a plain torch fp32
reference of the same op (run on the same device)."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(params=[0, 1], ids=["f32mfma", "bf16x6"])
def core(request):
    """Both arithmetic cores of the GEMM: v_mfma_f32_32x32x2_f32, and six bf16 MFMAs on operands split into three
    bf16 pieces (gemm.hip, the x6 core)."""
    from dvae_amd._lib import load
    lib = load()
    lib.dvae_gemm_x6(request.param)
    yield request.param
    lib.dvae_gemm_x6(int(__import__("os").environ.get("DVAE_GEMM_X6", "1")))


def _gemm(lib, A, a_km, Bm, b_km, M, N, K, bias=None, relu=0, acc=None, ws_mb=64):
    from dvae_amd import engine as E
    dev = A.device
    C = acc.clone() if acc is not None else torch.full((M, N), float("nan"), device=dev)
    ws = torch.empty(ws_mb << 20, dtype=torch.uint8, device=dev)
    E.check(lib.dvae_gemm_f32(E._ptr(A), A.stride(0), a_km, E._ptr(Bm), Bm.stride(0), b_km, E._ptr(C), C.stride(0),
                              M, N, K, E._ptr(bias), relu, int(acc is not None), E._ptr(ws), ws.numel(), E._stream()))
    torch.cuda.synchronize()
    return C


@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (4096, 1536, 512), (1000, 1024, 512), (37, 32, 1024),
                                   (300, 2048, 64), (5, 128, 2048), (64, 512, 1024), (100, 512, 2048),
                                   (64, 128, 1024), (48, 1536, 512),
                                   (3900, 1536, 512), (4096, 1440, 256)])   # the 128x96 tiles (x6 core), ragged rows / 15 column tiles
def test_gemm_nt(cuda, M, N, K, core):
    from dvae_amd._lib import load
    lib = load()
    torch.manual_seed(0)
    A = torch.randn(M, K, device=cuda)
    W = torch.randn(N, K, device=cuda)
    bias = torch.randn(N, device=cuda)
    ref = (A.double() @ W.double().t() + bias.double())
    out = _gemm(lib, A, 0, W, 0, M, N, K, bias=bias)
    assert torch.allclose(out.double(), ref, rtol=1e-5, atol=1e-4 * K ** 0.5)
    out = _gemm(lib, A, 0, W, 0, M, N, K, bias=bias, relu=1)
    assert torch.allclose(out.double(), ref.clamp(min=0), rtol=1e-5, atol=1e-4 * K ** 0.5)


def _gemm_mapped(lib, A, a_km, Bm, b_km, C, M, N, K, acc=0, row_map=None, k_map=None):
    from dvae_amd import engine as E
    ws = torch.empty(64 << 20, dtype=torch.uint8, device=A.device)
    E.check(lib.dvae_gemm_f32_mapped(E._ptr(A), A.stride(0), a_km, E._ptr(Bm), Bm.stride(0), b_km, E._ptr(C),
                                     C.stride(0), M, N, K, acc, E._ptr(row_map), E._ptr(k_map), E._ptr(ws),
                                     ws.numel(), E._stream()))
    torch.cuda.synchronize()


@pytest.mark.parametrize("rows,M,N,K,b_km", [(3000, 1000, 512, 2048, 1), (3000, 1000, 2048, 512, 0),
                                             (500, 77, 96, 36, 0), (200, 200, 512, 1536, 1), (64, 1, 128, 64, 1)])
def test_gemm_row_map(cuda, rows, M, N, K, b_km, core):
    """Product over a scattered subset of rows: reads A[map[m]], writes C[map[m]], leaves the other rows alone."""
    from dvae_amd._lib import load
    lib = load()
    torch.manual_seed(1)
    A = torch.randn(rows, K, device=cuda)
    W = torch.randn(K, N, device=cuda) if b_km else torch.randn(N, K, device=cuda)
    idx = torch.randperm(rows, device=cuda)[:M].sort().values.int()
    C = torch.full((rows, N), 7.0, device=cuda)
    _gemm_mapped(lib, A, 0, W, b_km, C, M, N, K, row_map=idx)
    Wd = W.double() if b_km else W.double().t()
    ref = torch.full((rows, N), 7.0, device=cuda, dtype=torch.float64)
    ref[idx.long()] = A[idx.long()].double() @ Wd
    assert torch.allclose(C.double(), ref, rtol=1e-5, atol=1e-4 * K ** 0.5)
    untouched = torch.ones(rows, dtype=torch.bool, device=cuda)
    untouched[idx.long()] = False
    assert torch.equal(C[untouched], torch.full_like(C[untouched], 7.0))
    _gemm_mapped(lib, A, 0, W, b_km, C, M, N, K, acc=1, row_map=idx)       # accumulate
    ref[idx.long()] += A[idx.long()].double() @ Wd
    assert torch.allclose(C.double(), ref, rtol=1e-5, atol=2e-4 * K ** 0.5)


@pytest.mark.parametrize("rows,K,M,N", [(9000, 5003, 2048, 512), (300, 123, 128, 64), (40, 16, 512, 512),
                                        (5000, 4096, 1536, 128), (100, 1, 64, 64)])
def test_gemm_k_map(cuda, rows, K, M, N, core):
    """Weight-gradient shape summed over a scattered subset of rows (ragged K, split K)."""
    from dvae_amd._lib import load
    lib = load()
    torch.manual_seed(2)
    dY = torch.randn(rows, M, device=cuda)
    X = torch.randn(rows, N, device=cuda)
    idx = torch.randperm(rows, device=cuda)[:K].sort().values.int()
    C = torch.full((M, N), float("nan"), device=cuda)
    _gemm_mapped(lib, dY, 1, X, 1, C, M, N, K, k_map=idx)
    ref = dY[idx.long()].double().t() @ X[idx.long()].double()
    assert torch.allclose(C.double(), ref, rtol=1e-5, atol=1e-4 * K ** 0.5)
    _gemm_mapped(lib, dY, 1, X, 1, C, M, N, K, acc=1, k_map=idx)
    assert torch.allclose(C.double(), 2 * ref, rtol=1e-5, atol=2e-4 * K ** 0.5)


def test_gemm_asymmetric_exact(cuda, core):
    """A = I with an asymmetric integer B catches swapped row/col maps."""
    from dvae_amd._lib import load
    lib = load()
    M = N = K = 128
    A = torch.eye(M, device=cuda)
    W = (torch.arange(N, device=cuda).view(N, 1) * 3 + torch.arange(K, device=cuda).view(1, K) * 7 % 11).float()
    out = _gemm(lib, A, 0, W, 0, M, N, K)
    assert torch.equal(out, W.t().contiguous())


@pytest.mark.parametrize("M,N,K", [(4096, 512, 1536), (777, 512, 1024), (64, 1024, 32), (130, 512, 2048), (24, 512, 1024), (32, 2048, 512)])
def test_gemm_nn(cuda, M, N, K, core):
    from dvae_amd._lib import load
    lib = load()
    torch.manual_seed(1)
    dY = torch.randn(M, K, device=cuda)
    W = torch.randn(K, N, device=cuda)
    base = torch.randn(M, N, device=cuda)
    ref = dY.double() @ W.double()
    out = _gemm(lib, dY, 0, W, 1, M, N, K)
    assert torch.allclose(out.double(), ref, rtol=1e-5, atol=1e-4 * K ** 0.5)
    out = _gemm(lib, dY, 0, W, 1, M, N, K, acc=base)
    assert torch.allclose(out.double(), ref + base.double(), rtol=1e-5, atol=1e-4 * K ** 0.5)


@pytest.mark.parametrize("M,N,K", [(1536, 512, 4096), (2048, 512, 90000), (32, 1024, 5000), (1024, 512, 333),
                                   (512, 64, 4096)])
def test_gemm_tn(cuda, M, N, K, core):
    from dvae_amd._lib import load
    lib = load()
    torch.manual_seed(2)
    dY = torch.randn(K, M, device=cuda)
    X = torch.randn(K, N, device=cuda)
    ref = dY.double().t() @ X.double()
    out = _gemm(lib, dY, 1, X, 1, M, N, K)
    assert torch.allclose(out.double(), ref, rtol=1e-5, atol=2e-4 * K ** 0.5)


@pytest.mark.parametrize("form", ["nt", "nn", "tn"])
@pytest.mark.parametrize("M,N", [(4096, 512), (300, 1536), (4096, 2048)])
def test_gemm_prefetch_forms_bit_identical(cuda, form, M, N):
    """The x6 main loop's prefetch forms (gemm.hip pf_of: 0 plain, 1 split held behind the products, 2 two k-tiles in
    flight) change the schedule, not the arithmetic: every form gives the same bits as the plain loop, on reductions of
    one to many k-tiles, ragged ones included (the two-deep form's prologue, unrolled pairs and drain), on 64x64, 64x128
    and 128x128 tiles, and the product is right against fp64."""
    from dvae_amd._lib import load
    lib = load()
    lib.dvae_gemm_x6(1)
    torch.manual_seed(5)
    try:
        for K in (32, 36, 64, 96, 100, 128, 160, 192, 224, 260, 512, 1028):
            if form == "nt":
                A, W, akm, bkm = torch.randn(M, K, device=cuda), torch.randn(N, K, device=cuda), 0, 0
                ref = A.double() @ W.double().t()
            elif form == "nn":
                A, W, akm, bkm = torch.randn(M, K, device=cuda), torch.randn(K, N, device=cuda), 0, 1
                ref = A.double() @ W.double()
            else:
                A, W, akm, bkm = torch.randn(K, M, device=cuda), torch.randn(K, N, device=cuda), 1, 1
                ref = A.double().t() @ W.double()
            outs = []
            for pf in (0, 1, 2, -1):
                lib.dvae_gemm_pf(pf)
                outs.append(_gemm(lib, A, akm, W, bkm, M, N, K))
            assert torch.allclose(outs[0].double(), ref, rtol=1e-5, atol=1e-4 * K ** 0.5), (form, K)
            for pf, o in zip((1, 2, -1), outs[1:]):
                assert torch.equal(o, outs[0]), (form, K, pf)
    finally:
        lib.dvae_gemm_pf(-1)
        lib.dvae_gemm_x6(int(__import__("os").environ.get("DVAE_GEMM_X6", "1")))


@pytest.mark.parametrize("form", ["nt", "nn", "tn"])
def test_gemm_x6_is_fp32_accurate(cuda, form):
    """The split-operand core drops only piece products below 2^-24 |a b|: against an fp64 product its error is that
    of the fp32 MFMA core (same fp32 accumulation), nowhere near bf16's 2^-8."""
    from dvae_amd._lib import load
    lib = load()
    torch.manual_seed(5)
    M, N, K = 1024, 512, 2048
    # wide dynamic range inside every row: pieces of very different magnitude meet in one accumulator
    scale = lambda *s: torch.randn(*s, device=cuda) * torch.exp2(torch.randint(-6, 7, s, device=cuda).float())
    if form == "nt":
        A, W = scale(M, K), scale(N, K)
        ref, args = A.double() @ W.double().t(), (A, 0, W, 0)
        mag = A.abs().double() @ W.abs().double().t()
    elif form == "nn":
        A, W = scale(M, K), scale(K, N)
        ref, args = A.double() @ W.double(), (A, 0, W, 1)
        mag = A.abs().double() @ W.abs().double()
    else:
        A, W = scale(K, M), scale(K, N)
        ref, args = A.double().t() @ W.double(), (A, 1, W, 1)
        mag = A.abs().double().t() @ W.abs().double()
    err = {}
    try:
        for core_id in (0, 1):
            lib.dvae_gemm_x6(core_id)
            out = _gemm(lib, *args, M, N, K)
            err[core_id] = float(((out.double() - ref).abs() / mag).max())     # relative to sum |a||b|: the fp32 bound's scale
    finally:
        lib.dvae_gemm_x6(int(__import__("os").environ.get("DVAE_GEMM_X6", "1")))
    print("max |err| / sum|a||b|: fp32 MFMA %.2e, bf16x6 %.2e" % (err[0], err[1]))
    assert err[0] < 2e-6 and err[1] < 2e-6, err
    assert err[1] < 4 * err[0] + 1e-7, err


def _both_cores(lib, args, M, N, K, ref, mag):
    err = {}
    try:
        for core_id in (0, 1):
            lib.dvae_gemm_x6(core_id)
            out = _gemm(lib, *args, M, N, K)
            assert torch.isfinite(out).all()
            err[core_id] = float(((out.double() - ref).abs() / mag).max())
    finally:
        lib.dvae_gemm_x6(int(__import__("os").environ.get("DVAE_GEMM_X6", "1")))
    return err


# x6.h's WORST-CASE bound: the three dropped piece products of one a*b are below (2^-22 + 2^-22 + 2^-30) |a b|, all with
# the sign of a*b, so a whole dot product is off by less than 2^-21 sum|a||b| on top of the fp32 accumulation error
X6_DROP = 2.0 ** -21


def test_gemm_x6_same_sign_weight_gradient_depth(cuda):
    """(a) the weight-gradient shape at the headline's depth (TN, K = 57344 rows) on SAME-SIGN operands: the dropped
    piece products all carry the sign of a*b, so with |randn| operands (post-ReLU activations, dW = dY^T X) their bias
    adds up linearly in K instead of averaging out -- it must still stay inside the worst-case bound."""
    from dvae_amd._lib import load
    lib = load()
    torch.manual_seed(11)
    M, N, K = 256, 128, 57344
    A, W = torch.randn(K, M, device=cuda).abs(), torch.randn(K, N, device=cuda).abs()
    ref = A.double().t() @ W.double()
    err = _both_cores(lib, (A, 1, W, 1), M, N, K, ref, ref)          # sum|a||b| = the product itself here
    print("same-sign TN K=57344: fp32 MFMA %.2e, bf16x6 %.2e of sum|a||b|" % (err[0], err[1]))
    assert err[1] <= 2 * err[0] + X6_DROP, err
    # and as a signed statement: the bias of the x6 result is one-sided (towards zero) and below the bound
    lib.dvae_gemm_x6(1)
    try:
        out = _gemm(lib, A, 1, W, 1, M, N, K)
    finally:
        lib.dvae_gemm_x6(int(__import__("os").environ.get("DVAE_GEMM_X6", "1")))
    bias = float(((out.double() - ref) / ref).mean())
    print("mean signed relative error %.2e" % bias)
    assert abs(bias) < X6_DROP


@pytest.mark.parametrize("form", ["nt", "nn", "tn"])
def test_gemm_x6_adversarial_mantissas(cuda, form):
    """(b) operands whose 24 significand bits are all ones (maximal second and third pieces: the dropped products are
    as large as they can get), mixed signs and exponents; (c) a row of denormal-range values beside normal ones and a
    row that is denormal throughout."""
    from dvae_amd._lib import load
    lib = load()
    torch.manual_seed(12)
    M, N, K = 256, 256, 2048
    ones = lambda *s: (torch.full(s, 2.0 - 2.0 ** -23, device=cuda) * torch.exp2(torch.randint(-3, 4, s, device=cuda).float())
                       * (torch.randint(0, 2, s, device=cuda).float() * 2 - 1))
    tiny = torch.finfo(torch.float32).tiny

    def build(rows, cols, kdim):   # kdim: which axis is the reduction
        X = ones(rows, cols)
        den = (torch.rand(K, device=cuda) * tiny)                      # denormal-range magnitudes
        if kdim == 1:
            X[1, ::2] = den[::2]                                       # denormals beside normal values
            X[2, :] = den                                              # a row that is denormal throughout
        else:
            X[::2, 1] = den[::2]
            X[:, 2] = den
        return X
    if form == "nt":
        A, W = build(M, K, 1), ones(N, K)
        ref, args, mag = A.double() @ W.double().t(), (A, 0, W, 0), A.abs().double() @ W.abs().double().t()
    elif form == "nn":
        A, W = build(M, K, 1), ones(K, N)
        ref, args, mag = A.double() @ W.double(), (A, 0, W, 1), A.abs().double() @ W.abs().double()
    else:
        A, W = build(K, M, 0), ones(K, N)
        ref, args, mag = A.double().t() @ W.double(), (A, 1, W, 1), A.abs().double().t() @ W.abs().double()
    keep = torch.ones(M, dtype=torch.bool, device=cuda)
    keep[2] = False                                                    # the all-denormal row: absolute check below
    try:
        outs = {}
        for core_id in (0, 1):
            lib.dvae_gemm_x6(core_id)
            outs[core_id] = _gemm(lib, *args, M, N, K)
    finally:
        lib.dvae_gemm_x6(int(__import__("os").environ.get("DVAE_GEMM_X6", "1")))
    err = {c: float(((o.double() - ref).abs() / mag)[keep].max()) for c, o in outs.items()}
    print("all-ones mantissas: fp32 MFMA %.2e, bf16x6 %.2e of sum|a||b|" % (err[0], err[1]))
    assert err[1] <= 2 * err[0] + X6_DROP, err
    for o in outs.values():
        assert torch.isfinite(o).all()
        # a dot product of denormals with values <= 16: exact result below K * 16 * tiny; either core may flush
        # denormal inputs or results to zero, neither may produce anything larger
        assert float(o[2].abs().max()) <= 16 * K * tiny
        assert float((o[2].double() - ref[2]).abs().max()) <= 16 * K * tiny


@pytest.mark.parametrize("form", ["nt", "nn", "tn"])
def test_gemm_x6_non_finite_operands(cuda, form, core):
    """(d) non-finite in, non-finite out, on both cores.  One +inf and one NaN operand: every output that meets them is
    non-finite and every other output is untouched.  The x6 core turns an inf operand into NaN (split3(inf) has
    second piece inf - inf = NaN; documented in x6.h), the fp32 MFMA keeps inf where no 0 * inf occurs."""
    from dvae_amd._lib import load
    lib = load()
    torch.manual_seed(13)
    M, N, K = 192, 160, 512
    if form == "tn":
        A, W = torch.randn(K, M, device=cuda), torch.randn(K, N, device=cuda)
        A[7, 3], A[400, 100] = float("inf"), float("nan")
        args, ref = (A, 1, W, 1), None
        clean = torch.nan_to_num(A, nan=0.0, posinf=0.0).double().t() @ W.double()
    else:
        A = torch.randn(M, K, device=cuda)
        W = torch.randn(N, K, device=cuda) if form == "nt" else torch.randn(K, N, device=cuda)
        A[3, 7], A[100, 400] = float("inf"), float("nan")
        args = (A, 0, W, 0 if form == "nt" else 1)
        Wd = W.double().t() if form == "nt" else W.double()
        clean = torch.nan_to_num(A, nan=0.0, posinf=0.0).double() @ Wd
    out = _gemm(lib, *args, M, N, K)
    bad = torch.zeros(M, dtype=torch.bool, device=cuda)
    bad[3] = bad[100] = True
    assert not torch.isfinite(out[bad]).any()
    assert torch.isnan(out[100]).all()
    if core == 0:
        assert torch.isinf(out[3]).all()          # fp32 MFMA: inf * finite = inf
    assert torch.isfinite(out[~bad]).all()
    assert torch.allclose(out[~bad].double(), clean[~bad], rtol=1e-5, atol=1e-4 * K ** 0.5)


@pytest.mark.parametrize("B,N,p,ragged", [(64, 8, 0.5, False), (33, 20, 0.2, False), (16, 128, 0.05, False),
                                          (50, 40, 0.15, True), (7, 1, 0.5, False)])
def test_schedule_bit_exact(cuda, B, N, p, ragged):
    """Levels (both directions), node rows and CSR against oracle/levels.py."""
    from dvae_amd import engine as E, synth
    from oracle import levels as OL
    if ragged:
        g, x, sizes = synth.ragged_dags(B, 4, 5, nmin=3, nmax=N)
    else:
        g, x = synth.random_dags(B, N, 4, p, 3, fix_orphans=(N > 1))
        sizes = None
    s = E.Schedule(g.to(cuda), None if sizes is None else sizes.to(cuda), want_reverse=True)
    lf = s.levels(False).cpu().numpy()
    lr = s.levels(True).cpu().numpy()
    ef, er = OL.forward_levels(g.numpy()), OL.reverse_levels(g.numpy())
    if sizes is not None:
        valid = np.arange(N)[None, :] < sizes.numpy()[:, None]
        ef = np.where(valid, ef, -1)
        er = np.where(valid, er, -1)
    assert np.array_equal(lf, ef)
    assert np.array_equal(lr, er)
    R = int((ef >= 0).sum())
    assert s.R == R and s.E == int(g.sum())
    assert s.nlev_fwd == int(ef.max()) + 1 and s.nlev_rev == int(er.max()) + 1
    # rows are ordered by (level, graph, node)
    n2r = s.export(2, B * N).cpu().numpy().reshape(B, N)
    order = sorted((int(ef[b, i]), b, i) for b in range(B) for i in range(N) if ef[b, i] >= 0)
    for row, (_, b, i) in enumerate(order):
        assert n2r[b, i] == row
    starts = s.info[8:8 + N + 1]
    for L in range(s.nlev_fwd):
        assert starts[L] == sum(1 for o in order if o[0] < L)
    # CSR: predecessors in descending index order
    ptr = s.export(4, R + 1).cpu().numpy()
    idx = s.export(5, max(s.E, 1)).cpu().numpy()
    preds = OL.pred_lists_desc(g.numpy())
    for b in range(B):
        for i in range(N):
            if ef[b, i] < 0:
                continue
            r = n2r[b, i]
            got = idx[ptr[r]:ptr[r + 1]].tolist()
            assert got == [n2r[b, u] for u in preds[b][i]]
    # scored / unscored edge prefix rows: add_edge scores pair (i, j) on the prefix holding the
    # predecessors above j (dvae_model.py:210-220), so the full prefix of a node (its last edge
    # row) is read only when node 0 is not among its predecessors
    scored, skipped = [], []
    for r in range(R):
        b, i = order[r][1], order[r][2]
        for k in range(ptr[r], ptr[r + 1]):
            last = k == ptr[r + 1] - 1
            (skipped if last and 0 in preds[b][i] else scored).append(R + k)
    if s.E:
        assert s.export(8, s.E).cpu().numpy()[:len(scored)].tolist() == scored
        assert s.export(9, s.E).cpu().numpy()[:len(skipped)].tolist() == skipped
    es = s.info[8 + 2 * (N + 1):8 + 3 * (N + 1)]
    cs = s.info[8 + 3 * (N + 1):8 + 4 * (N + 1)]
    for L in range(N + 1):
        assert cs[L] == sum(1 for p in scored if p - R < es[L])
    sources_are_node0 = all(o[2] == 0 for o in order if o[0] == 0)
    assert int(s.info[7]) == int(sources_are_node0)


def test_schedule_rejects_bad_graph(cuda):
    from dvae_amd import engine as E
    from dvae_amd._lib import GraphError
    g = torch.zeros(4, 6, 6)
    g[1, 2, 3] = 1.0       # upper triangle
    with pytest.raises(GraphError):
        E.Schedule(g.to(cuda))
    g = torch.zeros(4, 6, 6)
    g[0, 3, 1] = 0.5       # not 0/1
    with pytest.raises(AssertionError):
        E.Schedule(g.to(cuda))


@pytest.mark.parametrize("M,G,HP,K,km,acc", [(512, 3, 512, 512, 0, 0), (512, 2, 512, 512, 0, 0), (512, 1, 512, 1536, 1, 1),
                                             (37, 3, 64, 64, 0, 0), (1, 1, 32, 16, 0, 1), (100, 1, 128, 384, 1, 0),
                                             (1000, 2, 256, 272, 0, 1), (33, 1, 512, 1024, 1, 1), (700, 3, 192, 192, 0, 0)])
def test_small_gemm(cuda, M, G, HP, K, km, acc):
    """small_kernel's plain form (32-row tiles, K split over the four waves, LDS combine) against torch fp64:
    forward layout (weights [G*HP][K]) and backward layout (weights [K][HP]), ragged M, K not a multiple of 64,
    accumulate on / off; rows beyond M untouched."""
    from dvae_amd import engine as E
    from dvae_amd._lib import load
    lib = load()
    torch.manual_seed(1)
    A = torch.randn(M, K, device=cuda)
    N = G * HP
    W = torch.randn(K, HP, device=cuda) if km else torch.randn(N, K, device=cuda)
    C0 = torch.randn(M + 3, N, device=cuda)
    C = C0.clone()
    E.check(lib.dvae_small_gemm(E._ptr(A), A.stride(0), E._ptr(W), W.stride(0), km, E._ptr(C), C.stride(0), M, G, HP, K,
                                acc, E._stream()))
    torch.cuda.synchronize()
    ref = A.double() @ (W.double() if km else W.double().t())
    if acc:
        ref = ref + C0[:M].double()
    assert torch.allclose(C[:M].double(), ref, rtol=1e-5, atol=1e-4 * K ** 0.5)
    assert torch.equal(C[M:], C0[M:])


@pytest.mark.parametrize("rows,W,C,ld_pad,ldo,acc", [(28672, 2048, 8, 0, 32, 0), (1000, 1024, 10, 64, 32, 0), (37, 64, 1, 0, 32, 0),
                                                     (5000, 3072, 31, 0, 32, 1), (0, 256, 8, 0, 32, 0), (513, 2048, 17, 0, 32, 0),
                                                     (61440, 1536, 8, 0, 32, 1), (4096, 1024, 8, 0, 12, 0), (3, 100, 9, 0, 16, 1),
                                                     (0, 64, 4, 0, 8, 1)])
def test_typed_colsum(cuda, rows, W, C, ld_pad, ldo, acc):
    """Per-type column sums (X^T . [onehot(type) | 1]: the decoder's DTT, the GRU cells' input-weight gradients, the gate/map
    position columns) against torch: every accumulator count (8 / 16 / 32 registers), a padded leading dimension, a single
    row chunk, fewer rows than waves, no rows at all, narrow outputs (ldo < 32), set and accumulate."""
    from dvae_amd import engine as E
    from dvae_amd._lib import load
    lib = load()
    torch.manual_seed(7)
    ld = W + ld_pad
    X = torch.randn(max(rows, 1), ld, device=cuda)
    t = torch.randint(0, C, (max(rows, 1),), device=cuda, dtype=torch.int32)
    before = torch.randn(W, ldo, device=cuda) if acc else torch.full((W, ldo), float("nan"), device=cuda)
    out = before.clone()
    part = torch.empty(8 << 20, device=cuda)
    E.check(lib.dvae_typed_colsum(E._ptr(X), ld, rows, W, E._ptr(t), C, E._ptr(out), ldo, acc, E._ptr(part), part.numel(),
                                  E._stream()))
    torch.cuda.synchronize()
    onehot = torch.zeros(max(rows, 1), ldo, device=cuda, dtype=torch.float64)
    onehot[torch.arange(max(rows, 1), device=cuda), t.long()] = 1.0
    onehot[:, C] = 1.0
    ref = X[:rows, :W].double().t() @ onehot[:rows]
    if acc:
        ref = ref + before.double()
    assert torch.allclose(out.double(), ref, rtol=1e-5, atol=1e-4 * max(rows, 1) ** 0.5)
    if acc:
        assert torch.equal(out[:, C + 1:], before[:, C + 1:])          # untouched
    else:
        assert torch.equal(out[:, C + 1:], torch.zeros_like(out[:, C + 1:]))
