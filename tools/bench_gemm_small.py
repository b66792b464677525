"""Per-level GEMMs of a small shard (M = 512 / 1024 rows): k-tile depth of the 64x64 tile (tuning aid)."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dvae_amd import _lib, engine as E
lib = _lib.load()
dev = torch.device('cuda')
ws = torch.empty(96 << 20, dtype=torch.uint8, device=dev)
shapes = [('nt', 512, 1536, 512), ('nt', 512, 1024, 512), ('nn', 512, 512, 1536), ('nn', 512, 512, 1024),
          ('nt', 1024, 1536, 512), ('nn', 1024, 512, 1536), ('nt', 32, 1536, 512), ('nt', 4096, 1536, 512), ('nn', 4096, 512, 1536)]
for lay, M, N, K in shapes:
    if lay == 'nt':
        A = torch.randn(M, K, device=dev); B = torch.randn(N, K, device=dev); akm, bkm = 0, 0
    else:
        A = torch.randn(M, K, device=dev); B = torch.randn(K, N, device=dev); akm, bkm = 0, 1
    C = torch.empty(M, N, device=dev)
    line = '%s M=%5d N=%5d K=%5d' % (lay, M, N, K)
    for var in (0, 4, 5):
        for sp in (0, 1):
            lib.dvae_gemm_variant(var); lib.dvae_gemm_force_tile(3); lib.dvae_gemm_force_splits(sp)
            def run():
                E.check(lib.dvae_gemm_f32(E._ptr(A), A.stride(0), akm, E._ptr(B), B.stride(0), bkm, E._ptr(C), N, M, N, K,
                                          None, 0, 0, E._ptr(ws), ws.numel(), E._stream()))
            for _ in range(3): run()
            best = 1e30
            for trial in range(3):
                s, e = torch.cuda.Event(True), torch.cuda.Event(True)
                s.record()
                for _ in range(20): run()
                e.record(); torch.cuda.synchronize()
                best = min(best, s.elapsed_time(e) * 50)
            line += ' | bk%s s%d %5.1f' % ({0: 32, 4: 64, 5: 96}[var], sp, best)
    print(line, flush=True)
