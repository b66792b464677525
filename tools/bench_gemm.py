"""GEMM shape x tile sweep on the GPU (tuning aid, not a test)."""
import ctypes, sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dvae_amd import _lib, engine as E
lib = _lib.load()
dev = torch.device('cuda')
ws = torch.empty(96 << 20, dtype=torch.uint8, device=dev)
shapes = [  # (layout, M, N, K)
    ('nt', 4096, 1536, 512), ('nt', 4096, 1024, 512), ('nt', 8192, 1536, 512), ('nt', 90000, 2048, 512),
    ('nt', 32768, 2048, 512), ('nt', 32768, 1024, 512),
    ('nn', 4096, 512, 1536), ('nn', 4096, 512, 1024), ('nn', 8192, 512, 1536), ('nn', 90000, 512, 2048),
    ('nn', 32768, 512, 2048),
    ('tn', 2048, 512, 90000), ('tn', 1536, 512, 57000), ('tn', 1024, 512, 28672), ('tn', 1536, 32, 90000),
]
for lay, M, N, K in shapes:
    if lay == 'nt':
        A = torch.randn(M, K, device=dev); B = torch.randn(N, K, device=dev); akm, bkm = 0, 0
    elif lay == 'nn':
        A = torch.randn(M, K, device=dev); B = torch.randn(K, N, device=dev); akm, bkm = 0, 1
    else:
        A = torch.randn(K, M, device=dev); B = torch.randn(K, N, device=dev); akm, bkm = 1, 1
    C = torch.empty(M, N, device=dev)
    line = '%s M=%6d N=%5d K=%6d ' % (lay, M, N, K)
    for mode in (0, 3):
        lib.dvae_gemm_variant(mode)
        def run():
            E.check(lib.dvae_gemm_f32(E._ptr(A), A.stride(0), akm, E._ptr(B), B.stride(0), bkm, E._ptr(C), N, M, N, K,
                                      None, 0, 0, E._ptr(ws), ws.numel(), E._stream()))
        for _ in range(3): run()
        best = 1e30
        for trial in range(4):
            s, e = torch.cuda.Event(True), torch.cuda.Event(True)
            s.record()
            for _ in range(10): run()
            e.record(); torch.cuda.synchronize()
            best = min(best, s.elapsed_time(e) * 100)
        us = best
        line += ' | v%d %7.1f us %5.1f TF' % (mode, us, 2.0 * M * N * K / us / 1e6)
    print(line, flush=True)
lib.dvae_gemm_variant(0)
