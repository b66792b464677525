"""Is the x6 GEMM limited by what the data costs to move and multiply (power), or by its instruction stream?  The same
launches on random operands, on operands of one repeated value, and on zeros."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dvae_amd
from dvae_amd import _lib, engine as E
lib = _lib.load()
dev = torch.device('cuda')
ws = torch.empty(96 << 20, dtype=torch.uint8, device=dev)
def timed(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for form, M, N, K in [("nt", 57344, 1536, 512), ("nn", 57344, 512, 1536), ("tn", 1536, 512, 57344)]:
    for core in (1, 0):
        lib.dvae_gemm_x6(core)
        line = '%s M=%6d N=%5d K=%6d core %s:' % (form, M, N, K, 'x6 ' if core else 'f32')
        for kind in ('random', 'constant 1.2345', 'zeros'):
            mk = (lambda *s: torch.randn(*s, device=dev)) if kind == 'random' else (lambda *s: torch.full(s, 1.2345, device=dev)) if kind != 'zeros' else (lambda *s: torch.zeros(*s, device=dev))
            if form == "nt": A, W, akm, bkm = mk(M, K), mk(N, K), 0, 0
            elif form == "nn": A, W, akm, bkm = mk(M, K), mk(K, N), 0, 1
            else: A, W, akm, bkm = mk(K, M), mk(K, N), 1, 1
            C = torch.empty(M, N, device=dev)
            t = timed(lambda: E.check(lib.dvae_gemm_f32(E._ptr(A), A.stride(0), akm, E._ptr(W), W.stride(0), bkm, E._ptr(C), N, M, N, K, None, 0, 0, E._ptr(ws), ws.numel(), E._stream())))
            line += '  %s %.1f us %.1f TF' % (kind, t, 2.0 * M * N * K / t / 1e6)
        print(line, flush=True)
lib.dvae_gemm_x6(1)
