"""Per-level NN/NT shapes: tile x split-K sweep (tuning aid)."""
import ctypes, sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dvae_amd import _lib, engine as E
lib = _lib.load()
dev = torch.device('cuda')
ws = torch.empty(96 << 20, dtype=torch.uint8, device=dev)
shapes = [('nn', 4096, 512, 1536), ('nn', 4096, 512, 1024), ('nn', 8192, 512, 1536), ('nn', 12288, 512, 1536),
          ('nt', 4096, 1536, 512), ('nt', 4096, 1024, 512), ('nn', 2048, 512, 1536), ('nn', 1024, 512, 1536)]
for lay, M, N, K in shapes:
    if lay == 'nt':
        A = torch.randn(M, K, device=dev); B = torch.randn(N, K, device=dev); akm, bkm = 0, 0
    else:
        A = torch.randn(M, K, device=dev); B = torch.randn(K, N, device=dev); akm, bkm = 0, 1
    C = torch.empty(M, N, device=dev)
    print('%s M=%6d N=%5d K=%6d' % (lay, M, N, K), flush=True)
    for tile in (0, 1, 2, 3):
        line = '   tile %d:' % tile
        for sp in (0, 1, 2, 3, 4, 6):
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
            line += '  s%d %6.1f us %5.1f TF |' % (sp, best, 2.0 * M * N * K / best / 1e6)
        print(line, flush=True)
lib.dvae_gemm_force_tile(0); lib.dvae_gemm_force_splits(0)
