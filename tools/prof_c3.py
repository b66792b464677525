"""c3 (64-node DAGs) train steps for a kernel-trace profile.  usage: prof_c3.py [B]"""
import os, sys
from argparse import Namespace
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dvae_amd
dev = torch.device('cuda')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
model = dvae_amd.Dvae(Namespace(dataset_file=None, mod=None, bidirectional=False, num_classes=8, max_seq=64)).to(dev)
opt = torch.optim.Adam(model.parameters(), lr=1e-4, fused=True)
g, x = dvae_amd.synth.random_dags(B, 64, 8, 0.1, 3)
X = {'graph': g.to(dev), 'node_encoding': x.to(dev)}
for i in range(6):
    model.loss_and_grad(X); opt.step()
torch.cuda.synchronize()
print('done')
