"""CPU: the oracle (oracle/dense_ref.py, oracle/levels.py) against the golden
vectors produced by the real reference (tests/golden/make_golden.py)."""
import glob
import os
from argparse import Namespace

import numpy as np
import pytest
import torch

from helpers import fill_weights, oracle_sd

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TRAIN_CASES = ["enas_val8_none_bidir", "rand8_none_uni", "rand10_bn_bidir_shared"]


def load_case(name):
    d = np.load(os.path.join(GOLD, name + ".npz"))
    mod = None if str(d['mod']) == 'None' else str(d['mod'])
    return d, mod, bool(d['bidir']), bool(d['shared'])


def build_model(d, mod, bidir, shared, device=None):
    import dvae_amd
    N, C = d['x'].shape[1], d['x'].shape[2]
    hp = Namespace(dataset_file=None, mod=mod, bidirectional=bidir, num_classes=C, max_seq=N,
                   same_mapper_gater=shared)
    model = fill_weights(dvae_amd.Dvae(hp), int(d['seed']))
    return model.to(device) if device is not None else model


EVAL_CASES = ["eval_none_bidir", "eval_bn"]       # tests/golden/make_eval_golden.py


def eval_model(d, device=None):
    import dvae_amd
    mod = None if str(d['mod']) == 'None' else str(d['mod'])
    N, C = d['x'].shape[1], d['x'].shape[2]
    hp = Namespace(dataset_file=None, mod=mod, bidirectional=bool(d['bidir']), num_classes=C, max_seq=N)
    model = fill_weights(dvae_amd.Dvae(hp), int(d['seed']))
    return (model.to(device) if device is not None else model), mod, bool(d['bidir'])


def test_fixture_inventory():
    names = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLD, "*.npz")))
    svae = ["svae_enas_val8_uni", "svae_enas_val8_bidir", "svae_rand10_bidir"]      # test_svae_oracle.py
    assert names == sorted(TRAIN_CASES + ["predict_none", "predict_bn", "enas_rows", "bn_rows"] + EVAL_CASES + svae)


@pytest.mark.parametrize("name", TRAIN_CASES)
def test_oracle_matches_reference(name):
    from oracle import dense_ref as O
    from oracle import levels as OL
    d, mod, bidir, shared = load_case(name)
    model = build_model(d, mod, bidir, shared)
    sd = oracle_sd(model, requires_grad=True)
    g, x, eps = (torch.from_numpy(d[k]) for k in ('g', 'x', 'eps'))
    hv, mu, lv, fw, bw = O.encoder_forward(sd, g, x, mod, bidir, return_states=True)
    tol = dict(rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(fw.detach().numpy(), d['enc_states'], **tol)
    if bidir:
        np.testing.assert_allclose(bw.detach().numpy(), d['enc_states_rev'], **tol)
    np.testing.assert_allclose(hv.detach().numpy(), d['hv'], **tol)
    np.testing.assert_allclose(mu.detach().numpy(), d['mu'], **tol)
    np.testing.assert_allclose(lv.detach().numpy(), d['logvar'], **tol)
    z = O.reparameterize(mu, lv, eps)
    np.testing.assert_allclose(z.detach().numpy(), d['z'], **tol)
    e, t, hd, _ = O.decoder_forward(sd, torch.from_numpy(d['z']), g, x, mod, return_states=True)
    np.testing.assert_allclose(hd.detach().numpy(), d['dec_states'], **tol)
    np.testing.assert_allclose(e.detach().numpy(), d['edge_logits'], **tol)
    np.testing.assert_allclose(t.detach().numpy(), d['type_logits'], **tol)
    o = O.train_loss(sd, g, x, eps, mod, bidir)
    np.testing.assert_allclose([float(o[1]), float(o[2]), float(o[3]), float(o[0])], d['losses'], rtol=1e-5)
    o[0].backward()
    seen = set()
    for k in sd:
        if 'gval/' + k not in d or id(sd[k]) in seen:
            continue
        seen.add(id(sd[k]))
        got = sd[k].grad.flatten().double()
        np.testing.assert_allclose(got[torch.from_numpy(d['gidx/' + k])].numpy(), d['gval/' + k],
                                   rtol=2e-4, atol=1e-6 * max(1.0, float(d['gsum/' + k][1])))
        assert abs(float(got.sum()) - d['gsum/' + k][0]) <= 2e-4 * max(1e-3, d['gsum/' + k][1])
    # level definition
    assert np.array_equal(OL.forward_levels(d['g']), d['levels_fwd'])
    assert np.array_equal(OL.reverse_levels(d['g']), d['levels_rev'])


def test_enas_fixture_is_a_depth7_chain():
    d, *_ = load_case("enas_val8_none_bidir")
    assert (d['levels_fwd'].max(axis=1) == 7).all()          # every ENAS graph is a depth-7 chain


@pytest.mark.parametrize("name", ["predict_none", "predict_bn"])
def test_oracle_predict_matches_reference(name):
    import dvae_amd
    from oracle import dense_ref as O
    d = np.load(os.path.join(GOLD, name + ".npz"))
    mod = None if str(d['mod']) == 'None' else str(d['mod'])
    N, C = int(d['N']), int(d['C'])
    hp = Namespace(dataset_file=None, mod=mod, bidirectional=False, num_classes=C, max_seq=N)
    model = fill_weights(dvae_amd.Dvae(hp), int(d['seed']))
    with torch.no_grad():
        edges, types = O.decoder_predict(oracle_sd(model), torch.from_numpy(d['z']), N, C, mod)
    assert np.array_equal(edges.numpy(), d['edges']) and np.array_equal(types.numpy(), d['types'])


def test_dataset_parser_and_split(tmp_path):
    """d-vae_amd/data.py against the reference's NASDataset (dataloader.py:10-108) on real ENAS rows."""
    from dvae_amd import data
    r = np.load(os.path.join(GOLD, "enas_rows.npz"))
    d = np.load(os.path.join(GOLD, "enas_val8_none_bidir.npz"))
    for k, line in enumerate(r['lines']):
        rows, acc = data.parse_line(str(line))
        g, x = data.encode_structure(rows, 'enas')
        assert np.array_equal(g.numpy(), d['g'][k]) and np.array_equal(x.numpy(), d['x'][k])
        assert acc == r['acc'][k]
    idx = data.split_indices(int(r['n_rows']))
    assert np.array_equal(idx['val'][:8], r['val_index'])
    assert len(idx['train']) == int(r['n_train']) and len(idx['test']) == int(r['n_test'])
    # loaders: same tuple shape as the reference (val is a 1-tuple)
    f = tmp_path / "rows.txt"
    f.write_text("\n".join(str(l) for l in r['lines']) + "\n")
    tr, va, te = data.get_dataloaders(2, str(f))
    assert isinstance(va, tuple) and len(va) == 1
    b = next(iter(tr))
    assert b['graph'].shape == (2, 8, 8) and b['node_encoding'].shape == (2, 8, 8) and b['acc'].dtype == torch.float64


@pytest.mark.parametrize("name", EVAL_CASES)
def test_oracle_model_predict_and_eval_match_reference(name):
    """Dvae.predict (:492-496) and greedy Dvae.model_eval (:498-524) of the reference, eps injected:
    decoded tensors and recon_acc, on real rows and on the fed-back decodes (non-zero accuracy)."""
    from oracle import dense_ref as O
    d = np.load(os.path.join(GOLD, name + ".npz"))
    model, mod, bidir = eval_model(d)
    sd = oracle_sd(model)
    g, x = torch.from_numpy(d['g']), torch.from_numpy(d['x'])
    S, D = int(d['S']), int(d['D'])
    with torch.no_grad():
        e, t = O.model_predict(sd, g, x, torch.from_numpy(d['eps_predict']), mod, bidir)
        assert np.array_equal(e.numpy(), d['predict_edges']) and np.array_equal(t.numpy(), d['predict_types'])
        acc, ee, tt, _ = O.model_eval(sd, g, x, torch.from_numpy(d['eps_eval']), S, D, mod, bidir)
        assert acc == float(d['eval1_acc'])
        assert np.array_equal(ee.numpy(), d['eval1_edges']) and np.array_equal(tt.numpy(), d['eval1_types'])
        acc2, ee2, tt2, _ = O.model_eval(sd, torch.from_numpy(d['g2']), torch.from_numpy(d['x2']),
                                         torch.from_numpy(d['eps_eval']), S, D, mod, bidir)
        assert acc2 == float(d['eval2_acc']) and 0.0 < acc2 < 1.0
        assert np.array_equal(ee2.numpy(), d['eval2_edges']) and np.array_equal(tt2.numpy(), d['eval2_types'])
    # the product's metric (torch, not a kernel) on the reference's decodes gives the reference's number
    from dvae_amd import reconstruction_accuracy
    assert reconstruction_accuracy(torch.from_numpy(d['x2']), torch.from_numpy(d['g2']),
                                   torch.from_numpy(d['eval2_types']), torch.from_numpy(d['eval2_edges'])) == float(d['eval2_acc'])


def test_bn_dataset_parser_and_split(tmp_path):
    """d-vae_amd/data.py in task_type='bn' against the reference's NASDataset(task_type='bn')
    (dataloader.py:17-26,57-88) on synthetic BN rows (the reference ships no BN data): every split index,
    adjacency / one-hot encoding (no chain edge in 'bn', :82-84), score column, string form."""
    from dvae_amd import data
    r = np.load(os.path.join(GOLD, "bn_rows.npz"))
    f = tmp_path / "bn.txt"
    f.write_text("\n".join(str(l) for l in r['lines']) + "\n")
    for split in ('train', 'val', 'test'):
        ds = data.NASDataset(str(f), split=split, task_type='bn')
        assert np.array_equal(ds.index, r[split + '_index'])
    ds = data.NASDataset(str(f), split='val', task_type='bn')
    for i in range(len(ds)):
        it = ds[i]
        assert it['graph'].shape == (10, 10) and it['node_encoding'].shape == (10, 10)
        assert np.array_equal(it['graph'].numpy(), r['val_graph'][i])
        assert np.array_equal(it['node_encoding'].numpy(), r['val_node_encoding'][i])
        assert it['acc'] == r['val_acc'][i]
    ds = data.NASDataset(str(f), split='val', fmt='str', task_type='bn')
    for i in range(len(ds)):
        assert np.array_equal(ds[i]['graph'].numpy(), r['val_str'][i])


def test_nodes_with_one_value_have_one_encoder_state():
    """The premise of the library's shared encoder rows, on the restated reference itself: nodes that
    oracle/levels.py shared_encoder_rows gives one value hold bit-identical encoder states in the dense walk
    (dvae_model.py:63-87 computes a state from the node's index, type and predecessor states only)."""
    import numpy as np
    from oracle import dense_ref as O
    from oracle import levels as L
    import dvae_amd
    from argparse import Namespace
    from helpers import fill_weights, oracle_sd
    B, N, C = 40, 7, 3
    g, x = dvae_amd.synth.random_dags(B, N, C, 0.4, 17, chain=True)
    for mod in (None, "bn"):
        model = fill_weights(dvae_amd.Dvae(Namespace(dataset_file=None, mod=mod, bidirectional=False, num_classes=C,
                                                     max_seq=N, hidden_state_size=32)), seed=2)
        with torch.no_grad():
            states = O.encoder_forward(oracle_sd(model), g, x, mod, False, return_states=True)[3]
        rep, node2row, _, _ = L.shared_encoder_rows(g.numpy(), x.argmax(2).numpy())
        flat = states.reshape(B * N, -1)
        shared = 0
        for i, r in enumerate(rep):
            if r != i:
                shared += 1
                assert torch.equal(flat[i], flat[int(r)])
        assert shared > B          # most nodes of the first levels repeat
