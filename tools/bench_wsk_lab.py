"""Round 3 lab: in-workgroup split-K x6 GEMM (tools/lab/x6p_lab.hip x6_wsk) against the in-tree kernel on the per-level shapes."""
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

def timed(fn, n=30):
    for _ in range(5): fn()
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
names = {0: '128x64', 1: '64x128', 2: '128x128'}
shapes = [("nt", 4096, 1536, 512), ("nt", 4096, 1024, 512), ("nn", 4096, 512, 1536), ("nn", 4096, 512, 1024),
          ("nt", 2048, 1536, 512), ("nn", 2048, 512, 1536), ("nt", 8192, 1536, 512), ("nn", 8192, 512, 1536), ("nn", 28672, 512, 2048)]
for form, M, N, K in shapes:
    A = torch.randn(M, K, device=dev)
    W = torch.randn(N, K, device=dev) if form == "nt" else torch.randn(K, N, device=dev)
    bkm = 0 if form == "nt" else 1
    C = torch.empty(M, N, device=dev)
    ref = A[:256].double() @ (W.double().t() if form == "nt" else W.double())
    line = '%s M=%6d N=%5d K=%5d' % (form, M, N, K)
    fl = 2.0 * M * N * K
    for tile in (0, 3, 2, 1):
        lib.dvae_gemm_force_tile(tile)
        call = lambda: E.check(lib.dvae_gemm_f32(E._ptr(A), A.stride(0), 0, E._ptr(W), W.stride(0), bkm, E._ptr(C), N, M, N, K, None, 0, 0,
                                                 E._ptr(ws), ws.numel(), E._stream()))
        t = timed(call)
        line += ' | in-tree %s %6.1f us %5.1f TF' % ({0: 'auto', 3: '64x64', 2: '64x128', 1: '128x128'}[tile], t, fl / t / 1e6)
    lib.dvae_gemm_force_tile(0)
    for v in (0, 1, 2):
        C.zero_()
        f = lambda: lab.x6_wsk_gemm(v, A.data_ptr(), A.stride(0), W.data_ptr(), W.stride(0), bkm, C.data_ptr(), M, N, K, st())
        t = timed(f)
        err = ((C[:256].double() - ref).abs().max() / ref.abs().max()).item()
        line += ' | wsk %s %6.1f us %5.1f TF (err %.0e)' % (names[v], t, fl / t / 1e6, err)
    print(line, flush=True)
