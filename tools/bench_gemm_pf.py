"""Same-box A/B of the x6 GEMM's prefetch variants (dvae_gemm_pf: 0 plain, 1 split held behind the products, 2 two k-tiles in flight)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dvae_amd
from dvae_amd import _lib, engine as E
lib = _lib.load()
dev = torch.device('cuda')
ws = torch.empty(96 << 20, dtype=torch.uint8, device=dev)
VARS = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "0,1,2").split(",")]

def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

shapes = [("nt", 57344, 1536, 512), ("nt", 32768, 512, 512), ("nt", 4096, 1536, 512), ("nt", 4096, 1024, 512), ("nt", 28672, 2048, 512),
          ("nn", 4096, 512, 1536), ("nn", 4096, 512, 1024), ("nn", 57344, 512, 1536), ("nn", 28672, 512, 2048), ("nn", 90000, 512, 2048),
          ("tn", 1536, 512, 57344), ("tn", 2048, 512, 28672), ("tn", 1024, 512, 28672),
          ("nt", 2048, 1536, 512), ("nn", 2048, 512, 1536), ("tn", 1536, 512, 4096), ("tn", 1024, 512, 4096), ("tn", 1536, 512, 2048),
          ("nt", 4096, 512, 1536), ("nn", 4096, 512, 1536), ("nt", 4096, 512, 1024), ("nn", 4096, 512, 1024)]
if len(sys.argv) > 2: shapes = shapes[-int(sys.argv[2]):]
for form, M, N, K in shapes:
    if form == "nt": A, W, akm, bkm = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev), 0, 0
    elif form == "nn": A, W, akm, bkm = torch.randn(M, K, device=dev), torch.randn(K, N, device=dev), 0, 1
    else: A, W, akm, bkm = torch.randn(K, M, device=dev), torch.randn(K, N, device=dev), 1, 1
    C = torch.empty(M, N, device=dev)
    call = lambda: E.check(lib.dvae_gemm_f32(E._ptr(A), A.stride(0), akm, E._ptr(W), W.stride(0), bkm, E._ptr(C), N, M, N, K, None, 0, 0,
                                             E._ptr(ws), ws.numel(), E._stream()))
    out, ref = [], None
    for rep in range(2):
        for v in VARS:
            lib.dvae_gemm_pf(v % 10); lib.dvae_gemm_force_tile(v // 10)
            t = timed(call)
            if rep == 0:
                if ref is None: ref = C.clone()
                if v < 10: assert torch.equal(C, ref), "variant %d differs" % v      # (a forced tile may change the split-K factor)
                else: assert torch.allclose(C, ref, rtol=1e-4, atol=1e-3 * K ** 0.5), "variant %d is wrong" % v
            out.append((v, t))
    lib.dvae_gemm_pf(-1); lib.dvae_gemm_force_tile(0)
    fl = 2.0 * M * N * K
    print('%s M=%6d N=%5d K=%6d  ' % (form, M, N, K) + '  '.join('v%d %7.1f us %5.1f TF' % (v, t, fl / t / 1e6) for v, t in out), flush=True)
