"""Per-type column sums alone: time and bandwidth on the shapes the step uses (dvae_typed_colsum through the C ABI)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dvae_amd import engine as E
from dvae_amd._lib import load
lib = load()
dev = torch.device("cuda:0")
a = torch.randn(8192, 8192, device=dev)
for _ in range(30): a @ a          # clocks up
part = torch.empty(9 << 20, device=dev)
for rows, W, C in [(28672, 2048, 8), (61440, 1536, 8), (61440, 512, 8), (21211, 1536, 8), (28672, 1024, 9), (61440, 1536, 8)]:
    X = torch.randn(rows, W, device=dev)
    t = torch.randint(0, C, (rows,), device=dev, dtype=torch.int32)
    out = torch.empty(W, 32, device=dev)
    call = lambda: E.check(lib.dvae_typed_colsum(E._ptr(X), W, rows, W, E._ptr(t), C, E._ptr(out), 32, 0, E._ptr(part), part.numel(), E._stream()))
    for _ in range(5): call()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 50
    for _ in range(n): call()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print("rows %6d W %5d C %2d: %.1f us  %.2f TB/s" % (rows, W, C, dt * 1e6, rows * W * 4 / dt / 1e12))
