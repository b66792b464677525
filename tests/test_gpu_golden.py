"""GPU: the HIP path (through the C ABI) against the golden vectors of the real
reference.  rtol 1e-4 for states / mu / logvar / logits / losses (north_star);
bit-exact levels; bit-exact greedy decode on graphs whose recorded decision
margin is far above fp32 noise."""
import os

import numpy as np
import pytest
import torch

from helpers import close_scaled
from test_oracle_golden import GOLD, TRAIN_CASES, build_model, load_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", TRAIN_CASES)
def test_hip_matches_reference(cuda, name):
    from dvae_amd import engine as E
    d, mod, bidir, shared = load_case(name)
    model = build_model(d, mod, bidir, shared, cuda)
    g, x, eps = (torch.from_numpy(d[k]).to(cuda) for k in ('g', 'x', 'eps'))
    X = {'graph': g, 'node_encoding': x}
    s = E.Schedule(g, None, want_reverse=True)
    assert np.array_equal(s.levels(False).cpu().numpy(), d['levels_fwd'])
    assert np.array_equal(s.levels(True).cpu().numpy(), d['levels_rev'])
    with torch.no_grad():
        hv, mu, lv = model.encoder(X)
        fw, bw = model.encoder.node_states(X)
        close_scaled(fw, torch.from_numpy(d['enc_states']), what="encoder states")
        if bidir:
            close_scaled(bw, torch.from_numpy(d['enc_states_rev']), what="reverse states")
        close_scaled(hv, torch.from_numpy(d['hv']), what="hv")
        close_scaled(mu, torch.from_numpy(d['mu']), what="mu")
        close_scaled(lv, torch.from_numpy(d['logvar']), what="logvar")
        zt = torch.from_numpy(d['z']).to(cuda)
        e, t = model.decoder(zt, X)
        close_scaled(model.decoder.node_states(zt, X), torch.from_numpy(d['dec_states']), what="decoder states")
        close_scaled(e, torch.from_numpy(d['edge_logits']), what="edge logits")
        close_scaled(t, torch.from_numpy(d['type_logits']), what="type logits")
    out = model.training_step(X, 0, eps=eps)
    got = [out['log'][k] for k in ('train/edge_loss', 'train/vertex_loss', 'train/kl_loss', 'train/loss')]
    np.testing.assert_allclose(got, d['losses'], rtol=1e-4)
    out['loss'].backward()
    for k, p in model.named_parameters():
        flat = p.grad.flatten().double().cpu()
        scale = max(1e-3, float(np.abs(d['gval/' + k]).max()))
        err = np.abs(flat[torch.from_numpy(d['gidx/' + k])].numpy() - d['gval/' + k]).max()
        assert err <= 3e-4 * max(scale, d['gsum/' + k][1] / flat.numel()), (k, err, scale)
        assert abs(float(flat.sum()) - d['gsum/' + k][0]) <= 3e-4 * max(1e-3, d['gsum/' + k][1]), k


@pytest.mark.parametrize("name", ["predict_none", "predict_bn"])
def test_hip_greedy_decode_bit_exact(cuda, name):
    from argparse import Namespace
    import dvae_amd
    from helpers import fill_weights
    d = np.load(os.path.join(GOLD, name + ".npz"))
    mod = None if str(d['mod']) == 'None' else str(d['mod'])
    N, C = int(d['N']), int(d['C'])
    hp = Namespace(dataset_file=None, mod=mod, bidirectional=False, num_classes=C, max_seq=N)
    model = fill_weights(dvae_amd.Dvae(hp), int(d['seed'])).to(cuda)
    edges, types = model.decoder.predict(torch.from_numpy(d['z']).to(cuda))
    keep = d['margins'] > 2e-3          # decisions closer than this to a tie are fp32 noise
    assert keep.sum() >= 40
    assert np.array_equal(edges.cpu().numpy()[keep], d['edges'][keep])
    assert np.array_equal(types.cpu().numpy()[keep], d['types'][keep])
