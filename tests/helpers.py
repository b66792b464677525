"""Shared test helpers: deterministic weights, oracle state dicts, synthetic batches."""
import zlib

import torch


def fill_weights(module, seed, scale=None):
    """Deterministic weights that do not depend on module construction order:
    each state_dict key gets its own generator seeded from (seed, crc32(key)).
    Shared tensors (same_mapper_gater) are written once per storage."""
    sd = module.state_dict()
    seen = {}
    with torch.no_grad():
        for key in sorted(sd.keys()):
            t = sd[key]
            ptr = t.data_ptr()
            if ptr in seen:
                continue
            seen[ptr] = key
            canon = key
            gen = torch.Generator().manual_seed((int(seed) * 1000003 + zlib.crc32(canon.encode())) % (2 ** 31))
            s = scale if scale is not None else 1.0 / (t.shape[-1] ** 0.5 if t.dim() > 1 else 22.0)
            t.copy_(((torch.rand(t.shape, generator=gen) * 2 - 1) * s).to(t.device))
    return module


def cpu_state_dict(module):
    return {k: v.detach().cpu().float().clone() for k, v in module.state_dict().items()}


def rel_err(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def assert_close(a, b, rtol=1e-4, atol=1e-5, what=""):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    assert a.shape == b.shape, "%s: shape %s vs %s" % (what, tuple(a.shape), tuple(b.shape))
    err = (a - b).abs()
    tol = atol + rtol * b.abs()
    bad = err > tol
    assert not bad.any(), "%s: %d/%d outside rtol=%g atol=%g, max abs err %.3e (ref max %.3e)" % (
        what, int(bad.sum()), bad.numel(), rtol, atol, float(err.max()), float(b.abs().max()))


def oracle_sd(module, requires_grad=False):
    """CPU copy of the state dict for oracle/dense_ref.py; tensors that share
    storage in the module (same_mapper_gater) stay one tensor here."""
    out, by_ptr = {}, {}
    for k, v in module.state_dict().items():
        ptr = v.data_ptr()
        if ptr not in by_ptr:
            t = v.detach().cpu().float().clone()
            if requires_grad:
                t.requires_grad_(True)
            by_ptr[ptr] = t
        out[k] = by_ptr[ptr]
    return out


def close_scaled(a, b, tol=1e-4, what=""):
    """max|a-b| <= tol * max(1e-3, max|b|): the rtol-1e-4 bar of BASELINE.json,
    taken against the tensor's scale so that near-zero entries do not dominate."""
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    assert a.shape == b.shape, "%s: shape %s vs %s" % (what, tuple(a.shape), tuple(b.shape))
    err = float((a - b).abs().max()) if a.numel() else 0.0
    scale = max(1e-3, float(b.abs().max())) if b.numel() else 1.0
    assert err <= tol * scale, "%s: max abs err %.3e > %g * %.3e" % (what, err, tol, scale)
