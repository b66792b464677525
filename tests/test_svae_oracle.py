"""CPU: oracle/svae_ref.py against the golden vectors of the real reference S-VAE
(tests/golden/svae_*.npz, written by tests/golden/make_svae_golden.py)."""
import os
from argparse import Namespace

import numpy as np
import pytest
import torch

from helpers import close_scaled, fill_weights, oracle_sd

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SVAE_CASES = ["svae_enas_val8_uni", "svae_enas_val8_bidir", "svae_rand10_bidir"]


def load_svae_case(name):
    d = np.load(os.path.join(GOLD, name + ".npz"))
    return d, bool(d['bidir']), int(d['C']), int(d['N'])


def build_svae(d, bidir, C, N, device=None):
    import dvae_amd
    hp = Namespace(dataset_file=None, bidirectional=bidir, num_classes=C, max_seq=N)
    model = fill_weights(dvae_amd.Svae(hp), int(d['seed']))
    return model if device is None else model.to(device)


def check_grads_against_golden(named_grads, d, tol=3e-4):
    for k, g in named_grads:
        flat = g.flatten().double().cpu()
        scale = max(1e-3, float(np.abs(d['gval/' + k]).max()))
        err = np.abs(flat[torch.from_numpy(d['gidx/' + k])].numpy() - d['gval/' + k]).max()
        assert err <= tol * max(scale, d['gsum/' + k][1] / flat.numel()), (k, err, scale)
        assert abs(float(flat.sum()) - d['gsum/' + k][0]) <= tol * max(1e-3, d['gsum/' + k][1]), k


@pytest.mark.parametrize("name", SVAE_CASES)
def test_oracle_matches_reference(name):
    from oracle import svae_ref as O
    d, bidir, C, N = load_svae_case(name)
    model = build_svae(d, bidir, C, N)
    sd = oracle_sd(model, requires_grad=True)
    x, eps = torch.from_numpy(d['x']), torch.from_numpy(d['eps'])
    o = O.train_loss(sd, x, eps, bidir, C)
    close_scaled(o[4], torch.from_numpy(d['mu']), tol=1e-5, what="mu")
    close_scaled(o[5], torch.from_numpy(d['logvar']), tol=1e-5, what="logvar")
    close_scaled(o[6], torch.from_numpy(d['type_scores']), tol=1e-5, what="type scores")
    close_scaled(o[7], torch.from_numpy(d['edge_scores']), tol=1e-5, what="edge scores")
    np.testing.assert_allclose([float(o[1]), float(o[2]), float(o[3]), float(o[0])], d['losses'], rtol=1e-5)
    o[0].backward()
    check_grads_against_golden([(k, sd[k].grad) for k in sd], d, tol=1e-4)
    onehot, flags = O.predict_from_logits(torch.from_numpy(d['type_scores']), torch.from_numpy(d['edge_scores']))
    assert np.array_equal(onehot.numpy(), d['onehot']) and np.array_equal(flags.numpy(), d['flags'])


def test_string_batches_from_the_reference_file_format(tmp_path):
    """fmt='str' rows (dataloader.py:96-103) = [one-hot type | parent flags] without the start row,
    and the ENAS golden case holds exactly those rows."""
    from dvae_amd.data import encode_structure, parse_line
    rows_file = np.load(os.path.join(GOLD, "enas_rows.npz"))
    d = np.load(os.path.join(GOLD, "svae_enas_val8_uni.npz"))
    for i, line in enumerate(rows_file['lines']):
        rows, _ = parse_line(str(line))
        g, x = encode_structure(rows, 'enas')
        assert np.array_equal(torch.cat((x[1:], g[1:]), -1).numpy(), d['x'][i])


def test_svae_mirror_has_the_reference_surface():
    import dvae_amd
    m = dvae_amd.Svae(Namespace(dataset_file=None, bidirectional=True))
    for name in ("reparamterize", "_predict_from_logits", "predict", "forward", "compute_accuracy", "training_step",
                 "validation_step", "validation_end", "configure_optimizers", "train_dataloader", "val_dataloader",
                 "test_dataloader", "model_eval"):
        assert callable(getattr(m, name)), name
    (opt,), (sched,) = m.configure_optimizers()
    assert opt.defaults['lr'] == 1e-3 and sched.min_lrs == [1e-6]          # svae_model.py:254-256
    assert m.encoder.gru_layer.weight_ih_l0.shape == (3 * 501, 16)
    assert m.encoder.hg_unify.weight.shape == (501, 1002)
    assert m.decoder.add_edges[2].weight.shape == (8, 501)
    with pytest.raises(Exception):           # no CPU fallback
        m(torch.zeros(2, 7, 16))


def test_svae_abi_host_side():
    """dims validation, parameter counts and the pack table (host-only entry points: no GPU needed)."""
    import ctypes
    import dvae_amd
    from dvae_amd import _lib
    lib = _lib.load()
    for bad in (dict(C=33), dict(N=0), dict(T=0), dict(H=4096), dict(B=0)):
        kw = dict(B=4, T=7, C=8, N=8, H=501, Z=56, bidir=1)
        kw.update(bad)
        with pytest.raises(_lib.DvaeError):
            _lib.make_svae_dims(**kw)
    d = _lib.make_svae_dims(4, 7, 8, 8, 501, 56, True)
    assert (d.I, d.IP, d.HP, d.ZP) == (16, 32, 512, 64)
    assert lib.dvae_svae_enc_param_count(ctypes.byref(d)) == 14 and lib.dvae_svae_dec_param_count(ctypes.byref(d)) == 14
    d.bidir = 0
    assert lib.dvae_svae_enc_param_count(ctypes.byref(d)) == 8
    HP, IP, ZP = 512, 32, 64
    gru = lambda kp: 3 * HP * kp + 3 * HP * HP + 6 * HP
    assert lib.dvae_svae_enc_param_floats(ctypes.byref(d)) == gru(IP) + 2 * ZP * HP + 2 * ZP
    assert lib.dvae_svae_dec_param_floats(ctypes.byref(d)) == HP * ZP + HP + gru(HP) + 2 * HP * HP + 2 * HP + 2 * (32 * HP + 32)
    # the table covers every element of every tensor exactly once
    m = dvae_amd.Svae(Namespace(dataset_file=None, bidirectional=True))
    for mod, fn, kind in ((m.encoder, lib.dvae_svae_enc_pack_table, 1), (m.decoder, lib.dvae_svae_dec_pack_table, 0)):
        ps = mod.hip_params()
        dd = _lib.make_svae_dims(1, 1, 8, 8, 501, 56, kind)
        raw = np.array([1 << 20] * len(ps), dtype=np.uint64)          # fake addresses: the table is built on the host
        offs = np.cumsum([0] + [p.numel() for p in ps])[:-1].astype(np.int64)
        host = (_lib.PackEntry * 256)()
        n = fn(ctypes.byref(dd), ctypes.c_void_p(raw.ctypes.data), ctypes.c_void_p(offs.ctypes.data),
               ctypes.cast(host, ctypes.c_void_p), 256)
        assert n > 0
        assert sum(host[i].rows * host[i].cols for i in range(n)) == sum(p.numel() for p in ps)
        small = (_lib.PackEntry * 4)()
        assert fn(ctypes.byref(dd), ctypes.c_void_p(raw.ctypes.data), ctypes.c_void_p(offs.ctypes.data),
                  ctypes.cast(small, ctypes.c_void_p), 4) < 0
