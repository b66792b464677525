"""One GEMM shape, one core, a handful of launches (for counter passes). usage: x6_one.py form M N K core"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dvae_amd
from dvae_amd import _lib, engine as E
lib = _lib.load()
dev = torch.device('cuda')
form, M, N, K, core = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
ws = torch.empty(96 << 20, dtype=torch.uint8, device=dev)
if form == "nt": A, W, akm, bkm = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev), 0, 0
elif form == "nn": A, W, akm, bkm = torch.randn(M, K, device=dev), torch.randn(K, N, device=dev), 0, 1
else: A, W, akm, bkm = torch.randn(K, M, device=dev), torch.randn(K, N, device=dev), 1, 1
C = torch.empty(M, N, device=dev)
lib.dvae_gemm_x6(core)
for _ in range(5):
    E.check(lib.dvae_gemm_f32(E._ptr(A), A.stride(0), akm, E._ptr(W), W.stride(0), bkm, E._ptr(C), N, M, N, K, None, 0, 0,
                              E._ptr(ws), ws.numel(), E._stream()))
torch.cuda.synchronize()
