"""Training sanity of the default GEMM core: the same 300 Adam steps (same init, same batches, same eps) on the x6 core
and on the fp32-MFMA core; the loss curves must track each other (fp32 rounding noise amplified by training only)."""
import os, sys
from argparse import Namespace
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dvae_amd
from dvae_amd import _lib
lib = _lib.load()
dev = torch.device('cuda')
B, STEPS = 512, 300
batches = []
for i in range(8):
    g, x = dvae_amd.synth.enas_like(B, 100 + i)
    batches.append({'graph': g.to(dev), 'node_encoding': x.to(dev)})
gen = torch.Generator().manual_seed(7)
eps = [torch.randn(B, 56, generator=gen).to(dev) for _ in range(STEPS)]
curves = {}
# third run: the fp32-MFMA core again with another summation order (small-level kernels off): how far two fp32 runs drift apart
for core, small in ((1, 768), (0, 768), (2, 0)):
    lib.dvae_gemm_x6(core & 1)
    lib.dvae_small_rows_tune(small)
    torch.manual_seed(0)
    model = dvae_amd.Dvae(Namespace(dataset_file=None, mod=None, bidirectional=False)).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True)
    losses = []
    for s in range(STEPS):
        out4 = model.loss_and_grad(batches[s % len(batches)], eps=eps[s])
        opt.step()
        losses.append(out4[3].clone())
    curves[core] = torch.stack(losses).cpu() / B
    assert torch.isfinite(curves[core]).all()
lib.dvae_gemm_x6(1); lib.dvae_small_rows_tune(768)
a, b, c = curves[1], curves[0], curves[2]
for s in (0, 1, 10, 50, 100, 200, 299):
    print('step %3d  loss/graph  x6 %.5f   fp32 MFMA %.5f   fp32 MFMA, other summation order %.5f   rel diff x6-fp32 %.2e   fp32-fp32 %.2e'
          % (s, a[s], b[s], c[s], abs(a[s] - b[s]) / abs(b[s]), abs(c[s] - b[s]) / abs(b[s])))
print('max rel diff over %d steps: x6 vs fp32 %.2e, fp32 vs fp32 (other order) %.2e;  first -> last loss/graph: %.3f -> %.3f'
      % (STEPS, float(((a - b).abs() / b.abs()).max()), float(((c - b).abs() / b.abs()).max()), b[0], b[-1]))
