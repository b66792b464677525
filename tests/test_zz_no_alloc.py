"""GPU: dvae_elbo_static -- the one entry point that allocates nothing (partial sums in a file-scope
device array inside the library) -- against dvae_elbo through the C ABI: two calls in a row on the
same GPU, the second with a smaller batch, outputs and gradients bit-identical; the size limit is
refused.  Collected last on purpose (file name): a failure here must not hide the other tests."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _inputs(B, N, C, Z, seed, dev, ragged):
    gen = torch.Generator().manual_seed(seed)
    el = torch.tril(torch.randn(B, N, N, generator=gen), -1)
    tl = torch.randn(B, N, C, generator=gen)
    g = torch.tril((torch.rand(B, N, N, generator=gen) < 0.4).float(), -1)
    x = torch.nn.functional.one_hot(torch.randint(0, C, (B, N), generator=gen), C).float()
    mu = torch.randn(B, Z, generator=gen)
    lv = 0.3 * torch.randn(B, Z, generator=gen)
    sizes = torch.randint(1, N + 1, (B,), generator=gen).to(torch.int32) if ragged else None
    return [t if t is None else t.to(dev) for t in (el, tl, g, x, sizes, mu, lv)]


def _run(lib, fn_static, dims, inp, beta, dev):
    from dvae_amd.engine import _ptr, _stream
    from dvae_amd._lib import check
    el, tl, g, x, sizes, mu, lv = inp
    outs = []
    for static in (False, True):
        out4 = torch.full((4,), float('nan'), device=dev)
        grads = [torch.full_like(t, float('nan')) for t in (el, tl, mu, lv)]
        common = [ctypes.byref(dims), _ptr(el), _ptr(tl), _ptr(g), _ptr(x), _ptr(sizes), _ptr(mu), _ptr(lv),
                  ctypes.c_float(beta), _ptr(out4)]
        if static:
            check(fn_static(*common, *[_ptr(t) for t in grads], _stream()))
        else:
            part = torch.empty(lib.dvae_elbo_ws_bytes(dims.B), dtype=torch.uint8, device=dev)
            check(lib.dvae_elbo(*common, _ptr(part), *[_ptr(t) for t in grads], _stream()))
        torch.cuda.synchronize()
        outs.append((out4, grads))
    return outs


def test_elbo_static_matches_elbo_twice_second_smaller(cuda):
    from dvae_amd import _lib
    lib = _lib.load()
    N, C, Z = 8, 8, 56
    # first call large, second smaller: whatever the first left in the library's array must not leak
    for B, seed, ragged, beta in ((4096, 1, False, 0.005), (37, 2, True, 0.3)):
        dims = _lib.make_dims(B, N, C, 501, Z, None, False, N)
        inp = _inputs(B, N, C, Z, seed, cuda, ragged)
        (want4, wantg), (got4, gotg) = _run(lib, lib.dvae_elbo_static, dims, inp, beta, cuda)
        assert torch.isfinite(want4).all() and torch.equal(got4, want4), (B, got4, want4)
        for a, b in zip(gotg, wantg):
            assert torch.equal(a, b)
        # against plain torch as well (the loss of dvae_model.py:425-432)
        el, tl, g, x, sizes, mu, lv = inp
        if not ragged:
            ref_e = torch.nn.functional.binary_cross_entropy_with_logits(el.double(), g.double(), reduction='sum')
            ref_v = (-torch.log_softmax(tl.double(), 2) * x.double()).sum()
            ref_k = -0.5 * torch.sum(1 + lv.double() - mu.double() ** 2 - lv.double().exp())
            ref = torch.stack([ref_e, ref_v, ref_k, ref_e + ref_v + beta * ref_k]).float()
            assert torch.allclose(got4, ref, rtol=1e-5), (got4, ref)


def test_elbo_static_allocates_nothing(cuda):
    """The property the entry point is named for: the device's free memory (hipMemGetInfo, which sees every
    hipMalloc of the process, the library's included -- torch's own counters would not) is the same before and
    after a run of calls.  One warm-up call first: it resolves the array's address and may load the code object."""
    from dvae_amd import _lib
    from dvae_amd.engine import _ptr, _stream
    from dvae_amd._lib import check
    lib = _lib.load()
    B, N, C, Z = 512, 8, 8, 56
    dims = _lib.make_dims(B, N, C, 501, Z, None, False, N)
    el, tl, g, x, sizes, mu, lv = _inputs(B, N, C, Z, 3, cuda, False)
    out4 = torch.empty(4, device=cuda)
    grads = [torch.empty_like(t) for t in (el, tl, mu, lv)]
    args = [ctypes.byref(dims), _ptr(el), _ptr(tl), _ptr(g), _ptr(x), None, _ptr(mu), _ptr(lv), ctypes.c_float(0.005),
            _ptr(out4), *[_ptr(t) for t in grads], _stream()]
    check(lib.dvae_elbo_static(*args))                 # warm-up
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info(cuda)
    for _ in range(50):
        check(lib.dvae_elbo_static(*args))
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info(cuda)
    assert free1 == free0, "free device memory moved by %d bytes over 50 calls" % (free0 - free1)
    assert torch.isfinite(out4).all()


def test_elbo_static_rejects_batches_beyond_its_array(cuda):
    from dvae_amd import _lib
    from dvae_amd.engine import _ptr, _stream
    lib = _lib.load()
    B = 65537
    dims = _lib.make_dims(B, 2, 2, 64, 8, None, False, 2)
    t = torch.zeros(8, device=cuda)
    rc = lib.dvae_elbo_static(ctypes.byref(dims), _ptr(t), _ptr(t), _ptr(t), _ptr(t), None, _ptr(t), _ptr(t),
                              ctypes.c_float(0.005), _ptr(t), None, None, None, None, _stream())
    assert rc == -1 and b"at most 65536" in lib.dvae_last_error()
