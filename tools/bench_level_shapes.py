"""Round 3: the per-level product shapes (M = rows of a level) over tile x split-K choices of the in-tree x6 kernel and the
in-workgroup split-K lab kernel; min over interleaved rounds (the clock moves by +-10 % from launch to launch)."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dvae_amd
from dvae_amd import _lib, engine as E
lib = _lib.load()
lab = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'lab', 'libx6p_lab.so'))
lab.x6_wsk_gemm.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                            ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
dev = torch.device('cuda')
ws = torch.empty(96 << 20, dtype=torch.uint8, device=dev)
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

_w = torch.randn(8192, 8192, device=dev)
for _ in range(60): _w @ _w
torch.cuda.synchronize()
lib.dvae_gemm_x6(1)
shapes = [("nn", 4096, 512, 1536), ("nn", 4096, 512, 1024), ("nt", 4096, 1536, 512), ("nt", 4096, 1024, 512),
          ("nn", 2048, 512, 1536), ("nn", 2048, 512, 1024), ("nt", 2048, 1536, 512), ("nn", 1536, 512, 1536), ("nt", 1536, 1536, 512)]
tn = {0: 'auto', 3: '64x64', 2: '64x128', 1: '128x128'}
for form, M, N, K in shapes:
    A = torch.randn(M, K, device=dev)
    W = torch.randn(N, K, device=dev) if form == "nt" else torch.randn(K, N, device=dev)
    bkm = 0 if form == "nt" else 1
    C = torch.empty(M, N, device=dev)
    cands = {}
    def intree(tile, splits):
        def f():
            lib.dvae_gemm_force_tile(tile); lib.dvae_gemm_force_splits(splits)
            E.check(lib.dvae_gemm_f32(E._ptr(A), A.stride(0), 0, E._ptr(W), W.stride(0), bkm, E._ptr(C), N, M, N, K, None, 0, 0,
                                      E._ptr(ws), ws.numel(), E._stream()))
        return f
    for tile in (0, 3, 2, 1):
        for sp in ((0,) if tile == 0 else (0, 2, 3, 4, 6)):
            cands['%s/s%d' % (tn[tile], sp)] = intree(tile, sp)
    for v, nm in ((0, 'wsk128x64'), (1, 'wsk64x128'), (2, 'wsk128x128')):
        cands[nm] = (lambda v: (lambda: lab.x6_wsk_gemm(v, A.data_ptr(), A.stride(0), W.data_ptr(), W.stride(0), bkm, C.data_ptr(), M, N, K, st())))(v)
    best = {k: 1e9 for k in cands}
    for rnd in range(3):
        for k, f in cands.items():
            best[k] = min(best[k], timed(f))
    lib.dvae_gemm_force_tile(0); lib.dvae_gemm_force_splits(0)
    order = sorted(best.items(), key=lambda kv: kv[1])
    print('%s M=%5d N=%5d K=%5d  auto %.1f us | ' % (form, M, N, K, best['auto/s0']) + '  '.join('%s %.1f' % kv for kv in order[:8]), flush=True)
