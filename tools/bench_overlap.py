"""How a short 'chain' GEMM fares when a long bulk split-K GEMM is already resident on another stream."""
import ctypes, sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dvae_amd
from dvae_amd import _lib
from dvae_amd.engine import side_stream
lib = _lib.load()
dev = torch.device('cuda')
def P(t): return ctypes.c_void_p(t.data_ptr())
K = 90000
dA = torch.randn(K, 2048, device=dev); X = torch.randn(K, 512, device=dev); dW = torch.empty(2048, 512, device=dev)
ws1 = torch.empty(96 << 20, dtype=torch.uint8, device=dev); ws2 = torch.empty(96 << 20, dtype=torch.uint8, device=dev)
Yc = torch.randn(4096, 2048, device=dev); Wc = torch.randn(2048, 512, device=dev); Oc = torch.empty(4096, 512, device=dev)
side = side_stream(dev); main = torch.cuda.current_stream()
def bulk():
    _lib.check(lib.dvae_gemm_f32(P(dA), 2048, 1, P(X), 512, 1, P(dW), 512, 2048, 512, K, None, 0, 0, P(ws2), 96 << 20, ctypes.c_void_p(side.cuda_stream)))
def chain():
    _lib.check(lib.dvae_gemm_f32(P(Yc), 2048, 0, P(Wc), 512, 1, P(Oc), 512, 4096, 512, 2048, None, 0, 0, P(ws1), 96 << 20, ctypes.c_void_p(main.cuda_stream)))
def timed(fn, stream):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(stream); fn(); b.record(stream)
    return a, b
for blocks in (768, 512, 256):
    for prio in (0, 1):
        lib.dvae_gemm_tune(blocks, -1)
        for _ in range(2): bulk(); chain()
        torch.cuda.synchronize()
        a, b = timed(chain, main); torch.cuda.synchronize(); alone = a.elapsed_time(b)
        lib.dvae_gemm_tune(blocks, 0)
        ba, bb = timed(bulk, side); torch.cuda.synchronize(); bulk_alone = ba.elapsed_time(bb)
        ba, bb = timed(bulk, side)
        time.sleep(0.0003)     # let the bulk launch become resident
        lib.dvae_gemm_tune(blocks, prio)
        evs = [timed(chain, main) for _ in range(4)]
        torch.cuda.synchronize()
        print('bulk blocks %d prio %d: chain alone %.3f ms, bulk alone %.3f ms, bulk with chains %.3f ms, chains under bulk: %s' % (
            blocks, prio, alone, bulk_alone, ba.elapsed_time(bb), ' '.join('%.3f' % x.elapsed_time(y) for x, y in evs)), flush=True)
