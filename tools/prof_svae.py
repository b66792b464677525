"""Short S-VAE train loop for rocprofv3 --kernel-trace --stats (uni, B=4096)."""
import os, sys
from argparse import Namespace
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dvae_amd
B, C, N, T = 4096, 8, 8, 7
dev = torch.device('cuda')
gen = torch.Generator().manual_seed(0)
x = torch.cat((F.one_hot(torch.randint(0, C, (B, T), generator=gen), C).float(),
               (torch.rand(B, T, N, generator=gen) < 0.5).float()), dim=-1).to(dev)
m = dvae_amd.Svae(Namespace(dataset_file=None, bidirectional=False)).to(dev)
(opt,), _ = m.configure_optimizers()
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    opt.zero_grad(set_to_none=True)
    out = m.training_step({'graph': x}, 0)
    out['loss'].backward(); opt.step()
torch.cuda.synchronize()
