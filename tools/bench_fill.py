"""Does a bulk weight-gradient GEMM on the side stream fit into the encoder's forward sweep for free?"""
import ctypes, os, sys, time
from argparse import Namespace
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dvae_amd
from dvae_amd import _lib, engine as E
from dvae_amd.engine import side_stream
lib = _lib.load()
dev = torch.device('cuda')
model = dvae_amd.Dvae(Namespace(dataset_file=None, mod=None, bidirectional=False)).to(dev)
g, x = dvae_amd.synth.enas_like(4096, 1)
full = {'graph': g.to(dev), 'node_encoding': x.to(dev)}
sched = model._schedule(full)
enc = model.encoder
side = side_stream(dev); main = torch.cuda.current_stream()
K = 45056
dA = torch.randn(K, 2048, device=dev); X = torch.randn(K, 512, device=dev); dW = torch.empty(2048, 512, device=dev)
dB = torch.randn(28672, 2048, device=dev); Xb = torch.randn(28672, 512, device=dev); dWb = torch.empty(2048, 512, device=dev)
ws = torch.empty(96 << 20, dtype=torch.uint8, device=dev)
def P(t): return ctypes.c_void_p(t.data_ptr())
def bulk(stream):
    s = ctypes.c_void_p(stream.cuda_stream)
    _lib.check(lib.dvae_gemm_f32(P(dA), 2048, 1, P(X), 512, 1, P(dW), 512, 2048, 512, K, None, 0, 0, P(ws), 96 << 20, s))
    _lib.check(lib.dvae_gemm_f32(P(dB), 2048, 1, P(Xb), 512, 1, P(dWb), 512, 2048, 512, 28672, None, 0, 0, P(ws), 96 << 20, s))
def encf():
    with torch.no_grad():
        enc(full, sched=sched)
def both():
    side.wait_stream(main)
    bulk(side)
    encf()
    main.wait_stream(side)
def timed(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
for r in range(2):
    print('encoder forward alone            %.3f ms' % timed(encf), flush=True)
    print('dW1a + dW1b products alone       %.3f ms' % timed(lambda: bulk(main)), flush=True)
    print('both, products on the side stream %.3f ms' % timed(both), flush=True)
