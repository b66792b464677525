"""Skinny-output GEMMs of the backward level chain (N=512): tile x split-K sweep (tuning aid)."""
import ctypes, sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dvae_amd import _lib, engine as E
lib = _lib.load()
dev = torch.device('cuda')
ws = torch.empty(96 << 20, dtype=torch.uint8, device=dev)
shapes = [('nn', 4096, 512, 1536), ('nn', 4096, 512, 1024), ('nn', 5600, 512, 2048), ('nn', 512, 512, 1536),
          ('nn', 1024, 512, 1536), ('nt', 4096, 1536, 512), ('nt', 512, 1536, 512), ('nt', 512, 1024, 512)]
for lay, M, N, K in shapes:
    if lay == 'nt':
        A = torch.randn(M, K, device=dev); B = torch.randn(N, K, device=dev); akm, bkm = 0, 0
    else:
        A = torch.randn(M, K, device=dev); B = torch.randn(K, N, device=dev); akm, bkm = 0, 1
    C = torch.empty(M, N, device=dev)
    line = '%s M=%6d N=%5d K=%6d ' % (lay, M, N, K)
    for tile in (0, 1, 2, 3):
        for sp in (0, 1, 2, 4):
            lib.dvae_gemm_force_tile(tile); lib.dvae_gemm_force_splits(sp)
            def run():
                E.check(lib.dvae_gemm_f32(E._ptr(A), A.stride(0), akm, E._ptr(B), B.stride(0), bkm, E._ptr(C), N, M, N, K,
                                          None, 0, 0, E._ptr(ws), ws.numel(), E._stream()))
            for _ in range(3): run()
            best = 1e30
            for trial in range(3):
                s, e = torch.cuda.Event(True), torch.cuda.Event(True)
                s.record()
                for _ in range(10): run()
                e.record(); torch.cuda.synchronize()
                best = min(best, s.elapsed_time(e) * 100)
            line += ' | t%d s%d %6.1f us' % (tile, sp, best)
    print(line, flush=True)
lib.dvae_gemm_force_tile(0); lib.dvae_gemm_force_splits(0)
