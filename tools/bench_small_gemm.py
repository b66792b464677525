"""small_kernel (plain form) against gemm_f32 on the per-level shapes of a small shard (tuning aid)."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dvae_amd import _lib, engine as E
lib = _lib.load()
dev = torch.device('cuda')
ws = torch.empty(96 << 20, dtype=torch.uint8, device=dev)
def timed(run):
    for _ in range(3): run()
    best = 1e30
    for trial in range(3):
        s, e = torch.cuda.Event(True), torch.cuda.Event(True)
        s.record()
        for _ in range(20): run()
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) * 50)
    return best
for (M, G, HP, K, km) in [(512, 3, 512, 512, 0), (512, 2, 512, 512, 0), (512, 1, 512, 1536, 1), (512, 1, 512, 1024, 1),
                          (1024, 3, 512, 512, 0), (1024, 1, 512, 1536, 1), (2048, 3, 512, 512, 0), (2048, 1, 512, 1536, 1),
                          (32, 3, 512, 512, 0), (32, 1, 512, 1536, 1), (128, 3, 512, 512, 0), (256, 3, 512, 512, 0)]:
    N = G * HP
    A = torch.randn(M, K, device=dev)
    W = torch.randn(K, HP, device=dev) if km else torch.randn(N, K, device=dev)
    C = torch.empty(M, N, device=dev)
    t_small = timed(lambda: E.check(lib.dvae_small_gemm(E._ptr(A), K, E._ptr(W), W.stride(0), km, E._ptr(C), N, M, G, HP, K, 0, E._stream())))
    t_gemm = timed(lambda: E.check(lib.dvae_gemm_f32(E._ptr(A), K, 0, E._ptr(W), W.stride(0), km, E._ptr(C), N, M, N, K, None, 0, 0,
                                                     E._ptr(ws), ws.numel(), E._stream())))
    print('%s M=%5d N=%5d K=%5d  small %6.1f us   gemm_f32 (auto split, + combine launch) %6.1f us' % ('nn' if km else 'nt', M, N, K, t_small, t_gemm), flush=True)
