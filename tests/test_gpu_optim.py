"""dvae_adam_step / optim.FlatAdam against torch.optim.Adam (the reference's optimizer, dvae_model.py:471-475):
the update itself on a flat buffer, a model trained by either for a few steps, gradients that come from autograd instead of
the fused step, and state_dict exchange in both directions.  Tolerance: fp32 rounding of a different operation order."""
from argparse import Namespace

import pytest
import torch

from helpers import close_scaled, fill_weights

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,wd", [(1000003, 0.0), (4096, 0.01), (3, 0.0), (1, 0.0)])
def test_adam_step_matches_torch(cuda, n, wd):
    from dvae_amd import engine as E
    lib = E.load()
    gen = torch.Generator().manual_seed(n)
    p0 = torch.randn(n, generator=gen)
    ref = torch.nn.Parameter(p0.clone().to(cuda))
    opt = torch.optim.Adam([ref], lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=wd, foreach=False, fused=False)
    p, m, v = p0.clone().to(cuda), torch.zeros(n, device=cuda), torch.zeros(n, device=cuda)
    for step in range(1, 7):
        g = (torch.randn(n, generator=gen) * (0.1 if step % 2 else 10.0)).to(cuda)
        ref.grad = g.clone()
        opt.step()
        keep = g.clone()
        E.check(lib.dvae_adam_step(E._ptr(p), E._ptr(g), E._ptr(m), E._ptr(v), n, 1e-3, 0.9, 0.999, 1e-8, wd, step, E._stream()))
        assert torch.equal(g, keep)                                   # the gradient is read only
        st = opt.state[ref]
        assert torch.allclose(p, ref.detach(), rtol=2e-6, atol=1e-7), step
        assert torch.allclose(m, st['exp_avg'], rtol=2e-6, atol=1e-9)
        assert torch.allclose(v, st['exp_avg_sq'], rtol=2e-6, atol=1e-12)


def _model(cuda, mod=None, bidir=False, H=64):
    import dvae_amd
    hp = Namespace(dataset_file=None, mod=mod, bidirectional=bidir, num_classes=6, max_seq=8, hidden_state_size=H)
    return fill_weights(dvae_amd.Dvae(hp), seed=5).to(cuda)


@pytest.mark.parametrize("mod,bidir", [(None, False), ("bn", True)])
def test_flat_adam_trains_like_torch_adam(cuda, mod, bidir):
    import dvae_amd
    from dvae_amd.optim import FlatAdam
    a, b = _model(cuda, mod, bidir), _model(cuda, mod, bidir)
    before = {k: v.clone() for k, v in b.state_dict().items()}
    oa = torch.optim.Adam(a.parameters(), lr=1e-3)
    ob = FlatAdam(b, lr=1e-3)
    for k, v in b.state_dict().items():
        assert torch.equal(v, before[k]), k                       # flattening moves storage, not values
    for step in range(5):
        g, x = dvae_amd.synth.random_dags(24, 8, 6, 0.4, seed=step, chain=True)
        X = {'graph': g.to(cuda), 'node_encoding': x.to(cuda)}
        eps = torch.randn(24, 56, generator=torch.Generator().manual_seed(step)).to(cuda)
        la = a.loss_and_grad(X, eps=eps)
        oa.step()
        lb = b.loss_and_grad(X, eps=eps)
        ob.step()
        assert abs(float(la[3]) - float(lb[3])) <= 1e-4 * abs(float(la[3]))
    for (k, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
        close_scaled(pb, pa, tol=2e-4, what=k)
    # the same state in torch's layout: each optimizer continues from the other's checkpoint
    sa, sb = oa.state_dict(), ob.state_dict()
    assert sa['state'].keys() == sb['state'].keys()
    for i in sa['state']:
        assert float(sa['state'][i]['step']) == float(sb['state'][i]['step']) == 5.0
        close_scaled(sb['state'][i]['exp_avg'], sa['state'][i]['exp_avg'], tol=2e-3, what="exp_avg %d" % i)
    ob.load_state_dict(sa)
    oa.load_state_dict(sb)
    assert ob._t == 5
    ob.step()          # (gradients still installed by the last loss_and_grad)
    oa.step()


def test_flat_adam_takes_autograd_gradients(cuda):
    """training_step(...)['loss'].backward() leaves p.grad in tensors of their own: they are copied into the flat buffer."""
    import dvae_amd
    from dvae_amd.optim import FlatAdam
    a, b = _model(cuda), _model(cuda)
    oa, ob = torch.optim.Adam(a.parameters(), lr=1e-3), FlatAdam(b, lr=1e-3)
    g, x = dvae_amd.synth.random_dags(10, 8, 6, 0.4, seed=3, chain=True)
    X = {'graph': g.to(cuda), 'node_encoding': x.to(cuda)}
    eps = torch.randn(10, 56, generator=torch.Generator().manual_seed(2)).to(cuda)
    for m, o in ((a, oa), (b, ob)):
        for _ in range(2):
            o.zero_grad(set_to_none=True)
            m.training_step(X, 0, eps=eps)['loss'].backward()
            o.step()
    for (k, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
        close_scaled(pb, pa, tol=1e-4, what=k)


def test_flat_adam_refuses_shared_parameters(cuda):
    import dvae_amd
    from dvae_amd.optim import FlatAdam
    hp = Namespace(dataset_file=None, mod=None, bidirectional=False, num_classes=6, max_seq=8, hidden_state_size=64,
                   same_mapper_gater=True)
    m = dvae_amd.Dvae(hp).to(cuda)
    with pytest.raises(ValueError):
        FlatAdam(m)
    (opt,), _ = m.configure_optimizers(flat=True)      # falls back to torch's
    assert type(opt) is torch.optim.Adam
