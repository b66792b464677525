"""Same-box A/B of the two GEMM cores on the shapes of the headline step."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dvae_amd
from dvae_amd import _lib, engine as E
lib = _lib.load()
dev = torch.device('cuda')
ws = torch.empty(96 << 20, dtype=torch.uint8, device=dev)

def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

shapes = [("nt", 57344, 1536, 512), ("nt", 32768, 512, 512), ("nt", 4096, 1536, 512), ("nt", 4096, 1024, 512), ("nt", 28672, 2048, 512),
          ("nn", 4096, 512, 1536), ("nn", 4096, 512, 1024), ("nn", 57344, 512, 1536), ("nn", 28672, 512, 2048),
          ("tn", 1536, 512, 57344), ("tn", 2048, 512, 28672), ("tn", 1024, 512, 28672), ("tn", 1536, 32, 57344),
          ("nt", 512, 1536, 512), ("nn", 512, 512, 1536)]
for form, M, N, K in shapes:
    if form == "nt": A, W, akm, bkm = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev), 0, 0
    elif form == "nn": A, W, akm, bkm = torch.randn(M, K, device=dev), torch.randn(K, N, device=dev), 0, 1
    else: A, W, akm, bkm = torch.randn(K, M, device=dev), torch.randn(K, N, device=dev), 1, 1
    C = torch.empty(M, N, device=dev)
    call = lambda: E.check(lib.dvae_gemm_f32(E._ptr(A), A.stride(0), akm, E._ptr(W), W.stride(0), bkm, E._ptr(C), N, M, N, K, None, 0, 0,
                                             E._ptr(ws), ws.numel(), E._stream()))
    t = []
    for core in (0, 1):
        lib.dvae_gemm_x6(core)
        t.append(timed(call))
    fl = 2.0 * M * N * K
    print('%s M=%6d N=%5d K=%6d  f32 %8.1f us %6.1f TF   x6 %8.1f us %6.1f TF   x%.2f' % (
        form, M, N, K, t[0], fl / t[0] / 1e6, t[1], fl / t[1] / 1e6, t[0] / t[1]), flush=True)
