"""Round 3 lab: x6 GEMM on pre-split operands (tools/lab/x6p_lab.hip) against the in-tree x6 kernel, same box."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dvae_amd
from dvae_amd import _lib, engine as E
lib = _lib.load()
lab = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'lab', 'libx6p_lab.so'))
lab.x6p_split.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
lab.x6p_gemm_hybrid.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
lab.x6p_gemm.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
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

def split(X):
    R, K = X.shape
    out = torch.empty(R * K * 6, dtype=torch.uint8, device=dev)
    assert lab.x6p_split(X.data_ptr(), X.stride(0), R, K, out.data_ptr(), st()) == 0
    return out

names = {0: 'v1 128x128', 1: 'v1 64x128', 2: 'v1 64x64', 3: 'v2 256x128 glds', 4: 'v2 128x128 glds'}
variants = [int(v) for v in os.environ.get('VARIANTS', '0,1,2,3,4').split(',')]
hybrids = [int(v) for v in os.environ.get('HYBRIDS', '0,1,2').split(',') if v != '']
shapes = [(57344, 1536, 512), (32768, 512, 512), (28672, 2048, 512), (4096, 1536, 512), (4096, 1024, 512), (4096, 512, 1536), (4096, 512, 1024),
          (57344, 512, 1536)]
lib.dvae_gemm_x6(1)
# the first seconds of a fresh process measure low (clock ramp): warm the chip up before the first shape
_w = torch.randn(8192, 8192, device=dev)
for _ in range(60): _w @ _w
torch.cuda.synchronize()
shapes = shapes + shapes[:1]
for M, N, K in shapes:
    A, W = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev)
    C = torch.empty(M, N, device=dev)
    ref = (A[:256].double() @ W.double().t())
    call = lambda: E.check(lib.dvae_gemm_f32(E._ptr(A), A.stride(0), 0, E._ptr(W), W.stride(0), 0, E._ptr(C), N, M, N, K, None, 0, 0,
                                             E._ptr(ws), ws.numel(), E._stream()))
    t0 = timed(call)
    e0 = ((C[:256].double() - ref).abs().max() / ref.abs().max()).item()
    fl = 2.0 * M * N * K
    line = 'nt M=%6d N=%5d K=%5d  x6 %7.1f us %6.1f TF (err %.1e)' % (M, N, K, t0, fl / t0 / 1e6, e0)
    A3, W3 = split(A), split(W)
    for v in variants:
        C.zero_()
        f = lambda: lab.x6p_gemm(v, A3.data_ptr(), W3.data_ptr(), C.data_ptr(), M, N, K, st())
        t = timed(f)
        err = ((C[:256].double() - ref).abs().max() / ref.abs().max()).item()
        tail = ((C[-64:].double() - A[-64:].double() @ W.double().t()).abs().max() / ref.abs().max()).item()
        line += ' | %s %7.1f us %6.1f TF (err %.1e %.1e)' % (names[v], t, fl / t / 1e6, err, tail)
    for v in hybrids:
        C.zero_()
        f = lambda: lab.x6p_gemm_hybrid(v, A.data_ptr(), A.stride(0), W3.data_ptr(), C.data_ptr(), M, N, K, st())
        t = timed(f)
        err = ((C[:256].double() - ref).abs().max() / ref.abs().max()).item()
        line += ' | hybrid(B pre-split) %s %7.1f us %6.1f TF (err %.1e)' % (names[v][3:], t, fl / t / 1e6, err)
    print(line, flush=True)
