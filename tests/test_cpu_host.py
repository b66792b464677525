"""CPU-only checks: the C ABI library loads and exports every declared symbol,
host-side logic (dims, pack tables, state_dict surface, synthetic data, metric,
data-parallel sharding) -- no GPU compute."""
import ctypes
import os
import re
from argparse import Namespace

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "dvae_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dvae_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from dvae_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    syms = declared_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), "libdvae_hip.so does not export " + s
        assert s in _lib.SIGNATURES, "no ctypes signature for " + s
    assert _lib.load().dvae_abi_version() == 1


def test_dims_and_errors():
    from dvae_amd import _lib
    d = _lib.make_dims(4096, 8, 8, 501, 56, None, False, 8)
    assert (d.HP, d.ZP, d.EH, d.EHP, d.VHP, d.CP) == (512, 64, 2004, 2048, 1024, 32)
    d = _lib.make_dims(16, 10, 10, 501, 56, 'bn', True, 10)
    assert (d.EH, d.EHP, d.npos) == (3006, 3072, 0)
    with pytest.raises(_lib.DvaeError):
        _lib.make_dims(4, 200, 8, 501, 56, None, False, 200)      # N > 128
    with pytest.raises(_lib.DvaeError):
        _lib.make_dims(4, 8, 40, 501, 56, None, False, 8)         # too many classes
    with pytest.raises(KeyError):
        _lib.make_dims(4, 8, 8, 501, 56, 'nn', False, 8)


def test_batches_beyond_int32_row_indices_are_rejected():
    """B*N*(N-1)/2 candidate pairs (and the decoder's prefix rows) are int32 indices: a batch that
    overflows them is refused by every size query instead of being clamped (no GPU needed)."""
    from dvae_amd import _lib
    lib = _lib.load()
    assert lib.dvae_sched_bytes(4096, 128) > 0
    assert lib.dvae_sched_bytes(270000, 128) == 0            # 270000*128*127/2 > 2^31
    assert lib.dvae_sched_bytes(1 << 24, 64) == 0            # B*N >= 2^30
    assert lib.dvae_sched_bytes(0, 8) == 0 and lib.dvae_sched_bytes(8, 129) == 0
    with pytest.raises(_lib.DvaeError, match="int32 row indices"):
        _lib.make_dims(270000, 128, 8, 501, 56, None, False, 128)
    dummy = ctypes.c_void_p(256)                             # never dereferenced: the size check comes first
    rc = lib.dvae_sched_build(dummy, None, 270000, 128, 0, dummy, dummy, None)
    assert rc == -1 and b"int32 row indices" in lib.dvae_last_error()
    _lib.make_dims(100000, 8, 8, 501, 56, None, False, 8)   # config 5's decode batch is fine


@pytest.mark.parametrize("mod,bidir,N,C,count", [(None, False, 8, 8, 5165932), (None, True, 8, 8, 6947488),
                                                  ("bn", False, 10, 10, 6685968), ("bn", True, 10, 10, 7970532)])
def test_state_dict_surface(mod, bidir, N, C, count):
    """Key names / shapes / parameter counts of SURVEY.md 2.1 (captured from the reference)."""
    import dvae_amd
    m = dvae_amd.Dvae(Namespace(dataset_file=None, mod=mod, bidirectional=bidir, num_classes=C, max_seq=N))
    assert sum(p.numel() for p in m.parameters()) == count
    sd = m.state_dict()
    gm_in = 501 + N if mod is None else C
    assert sd['encoder.gating_network.0.weight'].shape == (501, gm_in)
    assert sd['encoder.gru.weight_ih'].shape == (1503, C) and sd['encoder.gru.weight_hh'].shape == (1503, 501)
    assert sd['encoder.lin11.weight'].shape == (56, 501) and sd['decoder.lin1.weight'].shape == (501, 56)
    assert sd['decoder.add_vertex.0.weight'].shape == (1002, 501) and sd['decoder.add_vertex.2.weight'].shape == (C, 1002)
    nin = 3 if mod == 'bn' else 2
    assert sd['decoder.add_edge.0.weight'].shape == (2 * nin * 501, nin * 501)
    assert sd['decoder.add_edge.2.weight'].shape == (1, 2 * nin * 501)
    assert ('encoder.lin01.weight' in sd) == bidir
    if bidir:
        assert sd['encoder.lin01.weight'].shape == (501, 1002)


def test_shared_gate_map_keys():
    import dvae_amd
    m = dvae_amd.Dvae(Namespace(dataset_file=None, mod='bn', bidirectional=True, num_classes=10, max_seq=10,
                                same_mapper_gater=True))
    sd = m.state_dict()
    for k in ('gating_network.0.weight', 'mapping_network.weight'):
        assert sd[k].data_ptr() == sd['encoder.' + k].data_ptr() == sd['decoder.' + k].data_ptr()
    assert sd['encoder.back_gating_network.0.weight'].data_ptr() != sd['gating_network.0.weight'].data_ptr()


def test_pack_table_covers_every_parameter():
    """Every raw element is written to the packed buffer by exactly one entry and
    every gradient element is produced by exactly one entry."""
    import dvae_amd
    from dvae_amd import _lib
    lib = _lib.load()
    for mod, bidir, N, C in [(None, True, 8, 8), ('bn', False, 10, 10)]:
        m = dvae_amd.Dvae(Namespace(dataset_file=None, mod=mod, bidirectional=bidir, num_classes=C, max_seq=N,
                                    hidden_state_size=40))
        for module, fn, nfl, gfl in ((m.encoder, lib.dvae_enc_pack_table, lib.dvae_enc_param_floats, lib.dvae_enc_grad_floats),
                                     (m.decoder, lib.dvae_dec_pack_table, lib.dvae_dec_param_floats, lib.dvae_dec_grad_floats)):
            params = module.hip_params()
            dims = module._dims
            base = 1 << 40
            raw, offs, o = [], [], 0
            for p in params:
                raw.append(base + o * 4)
                offs.append(o)
                o += p.numel()
            host = (_lib.PackEntry * 256)()
            raw_a, off_a = np.array(raw, dtype=np.uint64), np.array(offs, dtype=np.int64)
            n = fn(ctypes.byref(dims), ctypes.c_void_p(raw_a.ctypes.data), ctypes.c_void_p(off_a.ctypes.data),
                   ctypes.cast(host, ctypes.c_void_p), 256)
            assert 0 < n <= 256
            pfl, gtot = nfl(ctypes.byref(dims)), gfl(ctypes.byref(dims))
            read = np.zeros(o, dtype=np.int32)
            grad = np.zeros(o, dtype=np.int32)
            for e in list(host)[:n]:
                if e.kind == 2:      # bias folded into every position column
                    a0 = (e.aux_raw - base) // 4
                    read[a0:a0 + e.rows] = 1
                for r in range(e.rows):
                    lo = (e.raw - base) // 4 + r * e.raw_ld + e.raw_col0
                    if e.kind >= 0:
                        read[lo:lo + e.cols] += 1
                        assert 0 <= e.dst_off and e.dst_off + (e.rows if e.kind == 0 else e.cols) * e.dst_ld <= pfl + e.dst_ld
                    if e.g_kind >= 0:
                        glo = e.grad + r * e.raw_ld + e.raw_col0
                        grad[glo:glo + e.cols] += 1
                        assert 0 <= e.g_off < gtot
            assert (read == 1).all(), "a parameter element is packed %s times" % set(read.tolist())
            assert (grad == 1).all(), "a gradient element is produced %s times" % set(grad.tolist())


def test_product_path_refuses_cpu_tensors():
    import dvae_amd
    from dvae_amd._lib import DvaeError
    m = dvae_amd.Dvae(Namespace(dataset_file=None, mod=None, bidirectional=False))
    g, x = dvae_amd.synth.enas_like(2, 0)
    with pytest.raises(DvaeError):
        m.encoder({'graph': g, 'node_encoding': x})
    with pytest.raises(DvaeError):
        m.decoder.predict(torch.zeros(2, 56))


def test_missing_library_fails_loudly(monkeypatch):
    from dvae_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libdvae_hip.so")
    with pytest.raises(_lib.DvaeError):
        _lib.load()


def test_synthetic_generators():
    from dvae_amd import synth
    g, x = synth.enas_like(256, 3)
    assert g.shape == (256, 8, 8) and x.shape == (256, 8, 8)
    assert torch.equal(g, torch.tril(g, -1)) and set(g.unique().tolist()) <= {0.0, 1.0}
    assert (x.sum(-1) == 1).all()
    e, depth = synth.graph_stats(g)
    assert 13.0 < e < 15.0 and depth == 7.0
    g, _ = synth.random_dags(64, 12, 5, 0.1, 1)
    assert (g[:, 1:].sum(-1) >= 1).all()                    # orphans get the 0 -> i edge
    g, x, sizes = synth.ragged_dags(20, 6, 2, nmin=3, nmax=17)
    for b in range(20):
        n = int(sizes[b])
        assert g[b, n:].abs().sum() == 0 and g[b, :, n:].abs().sum() == 0 and x[b, n:].abs().sum() == 0


def test_reconstruction_accuracy():
    """dvae/utils/metrics.py:4-28 semantics."""
    from dvae_amd import reconstruction_accuracy
    B, S, N, C = 3, 4, 5, 6
    t = torch.nn.functional.one_hot(torch.randint(0, C, (B, N)), C).float()
    g = torch.tril((torch.rand(B, N, N) < 0.5).float(), -1)
    pt = t.unsqueeze(1).repeat(1, S, 1, 1).clone()
    pg = g.unsqueeze(1).repeat(1, S, 1, 1).clone()
    assert reconstruction_accuracy(t, g, pt, pg) == 1.0
    pg[0, 1, 3, 0] = 1 - pg[0, 1, 3, 0]
    pt[2, 0] = pt[2, 0].roll(1, dims=-1)
    assert abs(reconstruction_accuracy(t, g, pt, pg) - 10.0 / 12.0) < 1e-12


def test_shard_range():
    from dvae_amd.dp import shard_range
    for n, w in [(4096, 8), (10, 3), (2, 4)]:
        got = [shard_range(n, r, w) for r in range(w)]
        assert got[0][0] == 0 and got[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(got, got[1:]))
        assert max(h - l for l, h in got) - min(h - l for l, h in got) <= 1


def test_generation_metrics_on_real_and_broken_graphs():
    """validity / uniqueness / novelty (additions, d-vae_amd/metrics.py): every real ENAS row of
    the fixture is valid by the loader's own construction rules; targeted defects are caught."""
    import os
    import numpy as np
    from dvae_amd import data, metrics
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    r = np.load(os.path.join(gold, "enas_rows.npz"))
    enc = [data.encode_structure(data.parse_line(str(ln))[0], 'enas') for ln in r["lines"]]
    E = torch.stack([g for g, _ in enc])
    X = torch.stack([x for _, x in enc])
    assert metrics.dag_validity(E, X, 'enas').all()
    assert metrics.uniqueness(E, X) == 1.0
    assert metrics.novelty(E, X, E, X) == 0.0
    assert metrics.novelty(E, X, E[:4], X[:4]) == 0.5
    N, C = X.shape[1], X.shape[2]
    bad_e, bad_x = E[:6].clone(), X[:6].clone()
    bad_e[0, 3, 2] = 0.0                                   # chain edge 2->3 missing
    bad_x[1, 4] = 0.0; bad_x[1, 4, C - 2] = 1.0            # a second start token
    bad_e[2, 2, 5] = 1.0                                   # entry above the diagonal
    bad_e[3, N - 1, :] = 0.0                               # end node without predecessor
    bad_x[4, 2] = 0.0                                      # node without a type
    col = bad_e[5, :, 6].clone(); bad_e[5, :, 6] = 0.0     # node 6 feeds nobody
    assert col.sum() > 0
    ok = metrics.dag_validity(bad_e, bad_x, 'enas')
    assert ok.tolist() == [False] * 6
    rep = metrics.generation_report(torch.cat([E, bad_e]), torch.cat([X, bad_x]), E[:2], X[:2])
    assert abs(rep['validity'] - 8 / 14) < 1e-6 and rep['uniqueness'] == 1.0 and abs(rep['novelty'] - 0.75) < 1e-6
    # duplicates lower uniqueness; keys do not depend on the device or on batch position
    assert metrics.uniqueness(torch.cat([E, E[:2]]), torch.cat([X, X[:2]])) == 0.8
    assert torch.equal(metrics.graph_keys(E[3:5], X[3:5]), metrics.graph_keys(E, X)[3:5])
    # bn: variable types must be distinct
    g = torch.zeros(1, 5, 5); x = torch.zeros(1, 5, 5)
    for i in range(1, 5):
        g[0, i, i - 1] = 1.0
    x[0, 0, 3] = 1.0; x[0, 4, 4] = 1.0; x[0, 1, 0] = 1.0; x[0, 2, 1] = 1.0; x[0, 3, 2] = 1.0
    assert metrics.dag_validity(g, x, 'bn').all()
    x[0, 3] = 0.0; x[0, 3, 1] = 1.0
    assert not metrics.dag_validity(g, x, 'bn').any()


def test_checkpoint_roundtrip_by_key_name(tmp_path):
    """Lightning-layout checkpoints ({'state_dict': ...}, prediction.py:14-22): saved, loaded with
    weights_only=True, matched by the reference's parameter names."""
    from argparse import Namespace
    import dvae_amd
    from dvae_amd import trainer
    hp = Namespace(dataset_file=None, mod=None, bidirectional=True, num_classes=5, max_seq=6, hidden_state_size=16)
    a, b = dvae_amd.Dvae(hp), dvae_amd.Dvae(hp)
    (opt,), (sched,) = a.configure_optimizers()
    path = trainer.save_checkpoint(str(tmp_path / "run" / "_ckpt_epoch_3.ckpt"), a, opt, sched, epoch=3)
    ck = trainer.load_checkpoint(path, b)
    assert ck['epoch'] == 3 and set(ck) >= {'state_dict', 'optimizer_states', 'lr_schedulers'}
    for (ka, va), (kb, vb) in zip(a.state_dict().items(), b.state_dict().items()):
        assert ka == kb and torch.equal(va, vb)
    # a bare state_dict loads too; a missing key is an error when strict
    torch.save(a.state_dict(), str(tmp_path / "bare.pt"))
    trainer.load_checkpoint(str(tmp_path / "bare.pt"), b)
    sd = dict(a.state_dict()); sd.pop(next(iter(sd)))
    torch.save({'state_dict': sd}, str(tmp_path / "short.ckpt"))
    with pytest.raises(RuntimeError):
        trainer.load_checkpoint(str(tmp_path / "short.ckpt"), b)


def test_trainer_loop_on_a_stub_model(tmp_path):
    """The epoch loop over the Lightning-style hooks (train.py:39-47): optimizer steps, validation
    mean, ReduceLROnPlateau on the validation loss, one checkpoint per epoch, resume."""
    from dvae_amd import trainer

    class Stub(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.tensor([4.0]))
            self.data = [{'graph': torch.ones(2, 1)} for _ in range(3)]

        def training_step(self, batch, i):
            loss = (self.w ** 2).sum() * batch['graph'].sum() / 2
            v = float(loss)
            return {'loss': loss, 'log': {'train/edge_loss': v, 'train/vertex_loss': 0.0, 'train/kl_loss': 0.0,
                                          'train/loss': v}}

        def validation_step(self, batch, i):
            v = (self.w.detach() ** 2).sum()
            return {'val_loss': v, 'val/edge_loss': v, 'val/vertex_loss': v * 0, 'val/kl_loss': v * 0}

        def validation_end(self, outs):
            m = torch.stack([o['val_loss'] for o in outs]).mean()
            return {'avg_val_loss': m, 'log': {'val_loss': m}}

        def configure_optimizers(self):
            opt = torch.optim.SGD(self.parameters(), lr=0.1)
            return [opt], [torch.optim.lr_scheduler.ReduceLROnPlateau(opt, patience=0, factor=0.5)]

        def train_dataloader(self):
            return self.data

        def val_dataloader(self):
            return self.data[:2]

    logs = []
    m = Stub()
    t = trainer.Trainer(max_epochs=3, save_dir=str(tmp_path), log=logs.append)
    hist = t.fit(m)
    assert [h['epoch'] for h in hist] == [0, 1, 2] and all(h['steps'] == 3 for h in hist)
    assert hist[0]['train/loss'] > hist[1]['train/loss'] > hist[2]['train/loss']     # w shrinks every step
    assert abs(hist[0]['val_loss'] - float(m.w ** 2) ) > 0 and hist[2]['val_loss'] < hist[0]['val_loss']
    assert sorted(os.listdir(str(tmp_path))) == ['_ckpt_epoch_2.ckpt']              # older epochs removed
    w_end = float(m.w)
    m2 = Stub()
    t2 = trainer.Trainer(max_epochs=4, save_dir=str(tmp_path), log=logs.append)
    h2 = t2.fit(m2, resume_from=str(tmp_path / '_ckpt_epoch_2.ckpt'))
    assert [h['epoch'] for h in h2] == [3] and float(m2.w) < w_end
