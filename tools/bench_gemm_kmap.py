"""The gathered weight-gradient product (reduction rows through k_map, ascending) against the plain TN product of the same size."""
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

M, N, rows = 2048, 512, 57344
A = torch.randn(rows, M, device=dev); X = torch.randn(rows, N, device=dev); C = torch.empty(M, N, device=dev)
for keep in (1.0, 0.8, 0.5):
    K = int(rows * keep)
    idx = torch.sort(torch.randperm(rows, device=dev)[:K]).values.int().contiguous()
    plain = lambda: E.check(lib.dvae_gemm_f32(E._ptr(A), M, 1, E._ptr(X), N, 1, E._ptr(C), N, M, N, K, None, 0, 0, E._ptr(ws), ws.numel(), E._stream()))
    mapped = lambda: E.check(lib.dvae_gemm_f32_mapped(E._ptr(A), M, 1, E._ptr(X), N, 1, E._ptr(C), N, M, N, K, 0, None, E._ptr(idx), E._ptr(ws), ws.numel(), E._stream()))
    for rep in range(2):
        tp = timed(plain)
        lib.dvae_gemm_pf(0); tm0 = timed(mapped); c0 = C.clone()
        lib.dvae_gemm_pf(1); tm1 = timed(mapped)
        lib.dvae_gemm_pf(-1); tm = timed(mapped)
        assert torch.equal(C, c0)
    print('   form 1 (also the split behind the products): %7.1f us' % tm1)
    fl = 2.0 * M * N * K
    print('K=%6d of %d rows: plain %7.1f us %5.1f TF   gathered, plain loop %7.1f us %5.1f TF   gathered, loads at the top (default) %7.1f us %5.1f TF' % (
        K, rows, tp, fl / tp / 1e6, tm0, fl / tm0 / 1e6, tm, fl / tm / 1e6), flush=True)
