"""c2 (bn, bidirectional, 10 nodes, B=1024) train steps for a kernel-trace profile."""
import os, sys
from argparse import Namespace
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dvae_amd
dev = torch.device('cuda')
model = dvae_amd.Dvae(Namespace(dataset_file=None, mod='bn', bidirectional=True, num_classes=10, max_seq=10)).to(dev)
opt = torch.optim.Adam(model.parameters(), lr=1e-4, fused=True)
g, x = dvae_amd.synth.random_dags(1024, 10, 10, 0.3, 2, fix_orphans=False)
X = {'graph': g.to(dev), 'node_encoding': x.to(dev)}
for i in range(8):
    model.loss_and_grad(X); opt.step()
torch.cuda.synchronize()
print('done')
