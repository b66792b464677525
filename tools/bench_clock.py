"""Core clock held while the fp32 GEMM runs (probe wave on a second stream)."""
import ctypes, sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dvae_amd import _lib, engine as E
lib = _lib.load()
clk = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'lab', 'libclock.so'))
clk.clock_probe_launch.argtypes = [ctypes.c_longlong, ctypes.c_void_p, ctypes.c_void_p]
dev = torch.device('cuda')
ws = torch.empty(96 << 20, dtype=torch.uint8, device=dev)
M, N, K = 90000, 2048, 512
A = torch.randn(M, K, device=dev); B = torch.randn(N, K, device=dev); C = torch.empty(M, N, device=dev)
out = torch.zeros(2, dtype=torch.int64, device=dev)
side = torch.cuda.Stream()
def gemm():
    E.check(lib.dvae_gemm_f32(E._ptr(A), K, 0, E._ptr(B), K, 0, E._ptr(C), N, M, N, K, None, 0, 0, E._ptr(ws), ws.numel(), E._stream()))
def probe(cycles):
    clk.clock_probe_launch(cycles, ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(side.cuda_stream))
probe(2000000); torch.cuda.synchronize(); o = out.tolist()
print('idle: %.0f MHz' % (o[0] / o[1] * 100))
for rep in range(3):
    for _ in range(300): gemm()          # ~0.5 s of load before the probe
    s, e = torch.cuda.Event(True), torch.cuda.Event(True)
    s.record()
    probe(20000000)
    for _ in range(20): gemm()
    e.record(); torch.cuda.synchronize(); o = out.tolist()
    us = s.elapsed_time(e) * 1000 / 20
    print('under fp32 MFMA GEMM load: %.0f MHz; gemm %.1f us = %.1f TF/s -> %.3f of the 64 FLOP/clk/SIMD rate at that clock' % (
        o[0] / o[1] * 100, us, 2.0 * M * N * K / us / 1e6, 2.0 * M * N * K / us / 1e6 / (1024 * 64 * o[0] / o[1] * 100e6 / 1e12)))
