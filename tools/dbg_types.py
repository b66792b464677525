import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from argparse import Namespace
import dvae_amd
from helpers import fill_weights
cuda = torch.device('cuda')
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
hp = Namespace(dataset_file=None, mod=None, bidirectional=False, num_classes=8, max_seq=8, hidden_state_size=501)
model = fill_weights(dvae_amd.Dvae(hp), 3).to(cuda)
g, x = dvae_amd.synth.enas_like(B, 11)
eps = torch.randn(B, 56, generator=torch.Generator().manual_seed(4)).to(cuda)
X = {'graph': g.to(cuda), 'node_encoding': x.to(cuda)}
out = model.loss_and_grad(X, eps=eps)
res = {'out': out.cpu()}
for n, p in model.named_parameters(): res[n] = p.grad.cpu().clone()
torch.save(res, 'gpurun_out/grads_%s.pt' % sys.argv[1])
