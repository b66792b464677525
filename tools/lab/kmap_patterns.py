"""Lab: which row-list patterns slow the gathered TN product down (all keep 80 % of 57344 rows)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
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
r = torch.arange(rows, device=dev)
pats = {
    "identity (first 80 %)": r[: rows * 4 // 5],
    "drop every 5th row": r[r % 5 != 4],
    "drop every 5th block of 32": r[(r // 32) % 5 != 4],
    "drop every 5th block of 4": r[(r // 4) % 5 != 4],
    "random 80 %": torch.sort(torch.randperm(rows, device=dev)[: rows * 4 // 5]).values,
}
for name, idx in pats.items():
    idx = idx.int().contiguous(); K = idx.numel()
    mapped = lambda: E.check(lib.dvae_gemm_f32_mapped(E._ptr(A), M, 1, E._ptr(X), N, 1, E._ptr(C), N, M, N, K, 0, None, E._ptr(idx), E._ptr(ws), ws.numel(), E._stream()))
    for rep in range(2): t = timed(mapped)
    print('%-28s K=%d %7.1f us %5.1f TF' % (name, K, t, 2.0 * M * N * K / t / 1e6), flush=True)
