"""Drop-in mirror of the reference's S-VAE baseline (dvae/models/svae_model.py):
``Encoder``, ``Decoder``, ``Svae`` with the same constructor arguments, method names,
return values and ``state_dict()`` keys/shapes, backed by libdvae_hip.so
(csrc/svae.hip).  SURVEY.md 8 f4: the comparison model of the reference's
experiments on the same boundary.

A batch is ``{'graph': [B, max_seq_len-1, node_type+max_seq_len]}``: one row per
node after the start node, ``[one-hot type | parent flags]`` (dataloader.py:96-103,
``fmt='str'``).  The torch modules below only own parameters; ``GruWeights`` has
the parameter names and initialisation of ``torch.nn.GRU`` (the reference's layer,
svae_model.py:24,58) and no forward of its own -- MIOpen is not involved.
"""
import ctypes
import math

import torch
import torch.nn as nn

from . import _lib
from . import engine as E
from ._lib import check, load
from .engine import _f32c, _ptr, _stream


class GruWeights(nn.Module):
    """Parameters of a one-layer ``nn.GRU`` under its names (``weight_ih_l0`` ...,
    gates stacked r|z|n), initialised like it: U(-1/sqrt(H), 1/sqrt(H))."""

    def __init__(self, input_size, hidden_size, bidirectional=False):
        super(GruWeights, self).__init__()
        self.input_size, self.hidden_size, self.bidirectional = input_size, hidden_size, bidirectional
        for suffix in ([''] + (['_reverse'] if bidirectional else [])):
            self.register_parameter('weight_ih_l0' + suffix, nn.Parameter(torch.empty(3 * hidden_size, input_size)))
            self.register_parameter('weight_hh_l0' + suffix, nn.Parameter(torch.empty(3 * hidden_size, hidden_size)))
            self.register_parameter('bias_ih_l0' + suffix, nn.Parameter(torch.empty(3 * hidden_size)))
            self.register_parameter('bias_hh_l0' + suffix, nn.Parameter(torch.empty(3 * hidden_size)))
        k = 1.0 / math.sqrt(hidden_size)
        for p in self.parameters():
            nn.init.uniform_(p, -k, k)

    def direction(self, reverse=False):
        s = '_reverse' if reverse else ''
        return [getattr(self, n + s) for n in ('weight_ih_l0', 'weight_hh_l0', 'bias_ih_l0', 'bias_hh_l0')]


TWO_STREAMS = True      # weight-gradient products on the library's side stream (False: one stream; tests)


def _two_streams(pack, G, run):
    if TWO_STREAMS:
        return E._bwd_on_two_streams(pack, G, run)
    res = run(None)
    return pack.unpack(G), res


def _dims(d0, B, T):
    return _lib.make_svae_dims(B, T, d0.C, d0.N, d0.H, d0.Z, d0.bidir)


# raw forward / backward calls (no autograd): used by the Functions below and by Svae.loss_and_grad
def _enc_fwd(d, x, P):
    lib = load()
    ws = torch.empty(lib.dvae_svae_enc_ws_bytes(ctypes.byref(d)), dtype=torch.uint8, device=x.device)
    mu = torch.empty(d.B, d.Z, device=x.device)
    logvar = torch.empty(d.B, d.Z, device=x.device)
    check(lib.dvae_svae_enc_fwd(ctypes.byref(d), _ptr(x), _ptr(P), _ptr(ws), _ptr(mu), _ptr(logvar), _stream()))
    return ws, mu, logvar


def _enc_bwd(d, P, ws, d_mu, d_logvar, G, side):
    check(load().dvae_svae_enc_bwd(ctypes.byref(d), _ptr(P), _ptr(ws), _ptr(d_mu), _ptr(d_logvar), _ptr(G), _stream(),
                                   E._side(side)))


def _dec_fwd(d, z, P):
    lib = load()
    ws = torch.empty(lib.dvae_svae_dec_ws_bytes(ctypes.byref(d)), dtype=torch.uint8, device=z.device)
    ts = torch.empty(d.B, d.T, d.C, device=z.device)
    es = torch.empty(d.B, d.T, d.N, device=z.device)
    check(lib.dvae_svae_dec_fwd(ctypes.byref(d), _ptr(z), _ptr(P), _ptr(ws), _ptr(ts), _ptr(es), _stream()))
    return ws, ts, es


def _dec_bwd(d, P, ws, d_ts, d_es, G, side):
    d_z = torch.empty(d.B, d.Z, device=P.device)
    check(load().dvae_svae_dec_bwd(ctypes.byref(d), _ptr(P), _ptr(ws), _ptr(d_ts), _ptr(d_es), _ptr(G), _ptr(d_z),
                                   _stream(), E._side(side)))
    return d_z


def _elbo(d, ts, es, x, mu, logvar, beta):
    lib = load()
    out4 = torch.empty(4, device=x.device)
    part = torch.empty(lib.dvae_elbo_ws_bytes(d.B), dtype=torch.uint8, device=x.device)
    grads = (torch.empty_like(ts), torch.empty_like(es), torch.empty_like(mu), torch.empty_like(logvar))
    check(lib.dvae_svae_elbo(ctypes.byref(d), _ptr(ts), _ptr(es), _ptr(x), _ptr(mu), _ptr(logvar),
                             ctypes.c_float(beta), _ptr(out4), _ptr(part), _ptr(grads[0]), _ptr(grads[1]),
                             _ptr(grads[2]), _ptr(grads[3]), _stream()))
    return out4, grads


class _EncFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, x, *params):
        B, T, _ = x.shape
        d = _dims(module._dims, B, T)
        P = module._pack.pack(d, params)
        ws, mu, logvar = _enc_fwd(d, x, P)
        ctx.d, ctx.P, ctx.ws, ctx.module = d, P, ws, module
        ctx.save_for_backward(*params)     # autograd's version check guards the shared packed copy P (engine.EncoderFn)
        return mu, logvar

    @staticmethod
    def backward(ctx, d_mu, d_logvar):
        pack = ctx.module._pack
        G = torch.empty(pack.gfloats, dtype=torch.float32, device=ctx.P.device)
        cont = lambda t: None if t is None else t.contiguous()
        d_mu, d_logvar = cont(d_mu), cont(d_logvar)
        flat = _two_streams(pack, G, lambda side: _enc_bwd(ctx.d, ctx.P, ctx.ws, d_mu, d_logvar, G, side))[0]
        grads = pack.views(flat, ctx.saved_tensors)
        ctx.ws = None
        return (None, None) + tuple(grads)


class _DecFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, z, *params):
        d = _dims(module._dims, z.shape[0], module.max_seq_len - 1)
        P = module._pack.pack(d, params)
        ws, ts, es = _dec_fwd(d, _f32c(z), P)
        ctx.d, ctx.P, ctx.ws, ctx.module = d, P, ws, module
        ctx.save_for_backward(*params)     # autograd's version check guards the shared packed copy P (engine.EncoderFn)
        return ts, es

    @staticmethod
    def backward(ctx, d_ts, d_es):
        pack = ctx.module._pack
        G = torch.empty(pack.gfloats, dtype=torch.float32, device=ctx.P.device)
        cont = lambda t: None if t is None else t.contiguous()
        d_ts, d_es = cont(d_ts), cont(d_es)
        flat, d_z = _two_streams(pack, G, lambda side: _dec_bwd(ctx.d, ctx.P, ctx.ws, d_ts, d_es, G, side))
        grads = pack.views(flat, ctx.saved_tensors)
        ctx.ws = None
        return (None, d_z) + tuple(grads)


class _ElboFn(torch.autograd.Function):
    """(type_scores, edge_scores, mu, logvar) -> [edge_loss, vertex_loss, kl, loss], gradients of
    `loss` from the same kernel pass (svae_model.py:171-182)."""

    @staticmethod
    def forward(ctx, d, ts, es, x, mu, logvar, beta):
        out4, grads = _elbo(d, ts.contiguous(), es.contiguous(), x, mu.contiguous(), logvar.contiguous(), beta)
        ctx.save_for_backward(*grads)
        ctx.beta = beta
        return out4

    @staticmethod
    def backward(ctx, d_out4):
        gt, ge, gm, gl = ctx.saved_tensors
        d = d_out4
        sk = d[3] + (d[2] / ctx.beta if ctx.beta != 0 else 0.0)
        return None, gt * (d[1] + d[3]), ge * (d[0] + d[3]), None, gm * sk, gl * sk, None


class Encoder(nn.Module):
    """svae_model.py:13-44."""

    def __init__(self, node_type=8, max_seq_len=8, hidden_size=501, latent_dim=56, bidir=False):
        super(Encoder, self).__init__()
        self.bidir = bidir
        self.node_type = node_type
        self.hidden_size = hidden_size
        self.latent_dim = latent_dim
        self.max_seq_len = max_seq_len
        self.inp_dim = node_type + max_seq_len
        self.gru_layer = GruWeights(self.inp_dim, hidden_size, bidirectional=bidir)
        self.fc_mu = nn.Linear(hidden_size, latent_dim)
        self.fc_var = nn.Linear(hidden_size, latent_dim)
        if self.bidir:
            self.hg_unify = nn.Linear(hidden_size * 2, hidden_size)
        self._dims = _lib.make_svae_dims(1, 1, node_type, max_seq_len, hidden_size, latent_dim, bidir)
        self._pack = E.ParamPack("svae_enc")

    def hip_params(self):
        """Canonical order of dvae_svae_enc_pack_table (pack.hip)."""
        ps = self.gru_layer.direction() + [self.fc_mu.weight, self.fc_mu.bias, self.fc_var.weight, self.fc_var.bias]
        if self.bidir:
            ps += self.gru_layer.direction(reverse=True) + [self.hg_unify.weight, self.hg_unify.bias]
        return ps

    def forward(self, x):
        assert x.dim() == 3 and x.shape[2] == self.inp_dim and x.shape[1] >= 1, \
            'expected [B, T, %d] strings, got %s' % (self.inp_dim, tuple(x.shape))
        E._require_cuda(x, "graph")
        return _EncFn.apply(self, _f32c(x), *self.hip_params())


class Decoder(nn.Module):
    """svae_model.py:47-79."""

    def __init__(self, node_type=8, max_seq_len=8, hidden_size=501, latent_dim=56):
        super(Decoder, self).__init__()
        self.node_type = node_type
        self.hidden_size = hidden_size
        self.latent_dim = latent_dim
        self.max_seq_len = max_seq_len
        self.inp_dim = node_type + max_seq_len
        self.gru_layer = GruWeights(hidden_size, hidden_size)
        self.fc_h0 = nn.Linear(latent_dim, hidden_size)
        self.add_vertex = nn.Sequential(nn.Linear(hidden_size, hidden_size), nn.ReLU(),
                                        nn.Linear(hidden_size, node_type))
        self.add_edges = nn.Sequential(nn.Linear(hidden_size, hidden_size), nn.ReLU(),
                                       nn.Linear(hidden_size, max_seq_len))
        self.relu = nn.ReLU()
        assert max_seq_len >= 2, 'max_seq_len must be at least 2'
        self._dims = _lib.make_svae_dims(1, 1, node_type, max_seq_len, hidden_size, latent_dim, False)
        self._pack = E.ParamPack("svae_dec")

    def hip_params(self):
        """Canonical order of dvae_svae_dec_pack_table (pack.hip)."""
        return self.gru_layer.direction() + [
            self.fc_h0.weight, self.fc_h0.bias,
            self.add_vertex[0].weight, self.add_vertex[0].bias, self.add_vertex[2].weight, self.add_vertex[2].bias,
            self.add_edges[0].weight, self.add_edges[0].bias, self.add_edges[2].weight, self.add_edges[2].bias]

    def forward(self, z):
        assert z.dim() == 2 and z.shape[1] == self.latent_dim, 'expected [B, %d] latents' % self.latent_dim
        E._require_cuda(z, "z")
        return _DecFn.apply(self, z, *self.hip_params())


class Svae(nn.Module):
    """svae_model.py:82-301 (a LightningModule there; hooks keep their names and return values)."""

    def __init__(self, hparams):
        super(Svae, self).__init__()
        self.hparams = hparams
        task_type = getattr(hparams, 'task_type', 'enas')
        if hparams.dataset_file:
            from .data import get_dataloaders
            self.train_loader, self.val_loader, self.test_loader = get_dataloaders(
                hparams.batch_size, hparams.dataset_file, fmt="str", task_type=task_type)
        else:
            self.train_loader, self.val_loader, self.test_loader = [None] * 3
        self.node_type = getattr(hparams, 'num_classes', 8)
        self.max_seq_len = getattr(hparams, 'max_seq', 8)
        self.bidir = self.hparams.bidirectional
        self.encoder = Encoder(self.node_type, self.max_seq_len, bidir=self.bidir)
        self.decoder = Decoder(self.node_type, self.max_seq_len)
        self.beta = getattr(hparams, "beta", 0.005)

    def reparamterize(self, mu, logvar, eps=None):
        if eps is None:
            eps = torch.randn_like(mu)
        return E.ReparamFn.apply(mu, logvar, eps)

    def _predict_from_logits(self, type_scores, edge_scores, is_stochastic=False, uniforms=None):
        """-> (one-hot types, parent flags), exact 0/1 (:112-131).  Stochastic draws use the
        uniforms given as (u_type [..], u_edge [.., N]) or fresh ``torch.rand`` ones, through the
        inverse CDF (the reference draws with torch.multinomial / torch.bernoulli)."""
        ts, es = _f32c(type_scores), _f32c(edge_scores)
        E._require_cuda(ts, "type_scores")
        rows = ts.numel() // ts.shape[-1]
        d = _dims(self.decoder._dims, 1, 1)
        ut = ue = None
        if is_stochastic:
            if uniforms is None:
                uniforms = (torch.rand(ts.shape[:-1], device=ts.device), torch.rand(es.shape, device=ts.device))
            ut, ue = _f32c(uniforms[0]), _f32c(uniforms[1])
            assert ut.numel() == rows and ue.numel() == es.numel()
        onehot, flags = torch.empty_like(ts), torch.empty_like(es)
        check(load().dvae_svae_predict(ctypes.byref(d), rows, _ptr(ts), _ptr(es), _ptr(ut), _ptr(ue), _ptr(onehot),
                                       _ptr(flags), _stream()))
        return onehot, flags

    def predict(self, X):
        with torch.no_grad():
            mu, logvar = self.encoder(X)
            z = self.reparamterize(mu, logvar)
            type_scores, edge_scores = self.decoder(z)
            # the reference unpacks the pair under swapped names and returns it swapped back
            # (:137-141), so the caller receives (parent flags, one-hot types)
            gen_node_encoding, gen_dep_graph = self._predict_from_logits(type_scores, edge_scores)
        return gen_dep_graph, gen_node_encoding

    def forward(self, X, eps=None):
        mu, logvar = self.encoder(X)
        z = self.reparamterize(mu, logvar, eps)
        type_scores, edge_scores = self.decoder(z)
        return type_scores, edge_scores, mu, logvar

    def compute_accuracy(self, gen_dep_graph, gen_node_encoding, true_dep_graph, is_logits=False):
        if is_logits:
            gen_node_encoding, gen_dep_graph = self._predict_from_logits(gen_node_encoding, gen_dep_graph)
        pred = torch.cat((gen_node_encoding, gen_dep_graph), dim=-1)
        is_equal = torch.eq(pred, true_dep_graph).all(dim=-1).all(dim=-1)
        return (is_equal.sum() / is_equal.shape[0]).float()

    def _losses(self, batch, beta, eps=None):
        x = _f32c(batch["graph"])
        ts, es, mu, logvar = self.forward(x, eps=eps)
        d = _dims(self.decoder._dims, x.shape[0], x.shape[1])
        out4 = _ElboFn.apply(d, ts, es, x, mu, logvar, float(beta))
        with torch.no_grad():
            acc = self.compute_accuracy(es.detach(), ts.detach(), x, is_logits=True)
        return out4, acc

    def loss_and_grad(self, batch, eps=None, beta=None, grad_sync=None):
        """Forward + backward of training_step's loss without an autograd graph (an addition, as
        ``Dvae.loss_and_grad``): same kernels, gradients written into two flat buffers whose views become
        ``p.grad``.  ``grad_sync(flat)`` (optional, e.g. ``dp.make_grad_sync(dist)``) is queued on the side
        stream as soon as a buffer is complete -- the decoder's while the encoder's backward still runs.
        Returns the device tensor [edge_loss, vertex_loss, kl_loss, loss]."""
        enc, dec = self.encoder, self.decoder
        x = _f32c(batch["graph"])
        E._require_cuda(x, "graph")
        assert x.dim() == 3 and x.shape[2] == enc.inp_dim, 'expected [B, T, %d] strings' % enc.inp_dim
        beta = self.beta if beta is None else beta
        enc_p, dec_p = enc.hip_params(), dec.hip_params()
        with torch.no_grad():
            main = torch.cuda.current_stream()
            side = E.side_stream(x.device)
            de, dd = _dims(enc._dims, x.shape[0], x.shape[1]), _dims(dec._dims, x.shape[0], x.shape[1])
            Pe = enc._pack.pack(de, enc_p)
            side.wait_stream(main)
            with torch.cuda.stream(side):       # the decoder's weights are not needed before the encoder is through
                Pd = dec._pack.pack(dd, dec_p)
                pd_ready = torch.cuda.Event()
                pd_ready.record(side)
            key = tuple(p.data_ptr() for p in enc_p + dec_p)
            if getattr(self, '_flat', None) is None or self._flat[0] != key:
                self._flat = (key, torch.zeros(sum(p.numel() for p in enc_p), device=x.device),
                              torch.zeros(sum(p.numel() for p in dec_p), device=x.device))
            fe, fd = self._flat[1], self._flat[2]
            ws_e, mu, logvar = _enc_fwd(de, x, Pe)
            if eps is None:
                eps = torch.randn_like(mu)
            z = E.reparam_fwd(mu, logvar, _f32c(eps))
            main.wait_event(pd_ready)
            ws_d, ts, es = _dec_fwd(dd, z, Pd)
            out4, (gt, ge, gmu, glv) = _elbo(dd, ts, es, x, mu, logvar, float(beta))
            Gd = torch.empty(dec._pack.gfloats, dtype=torch.float32, device=x.device)
            Ge = torch.empty(enc._pack.gfloats, dtype=torch.float32, device=x.device)
            handles = []
            side.wait_stream(main)
            d_z = _dec_bwd(dd, Pd, ws_d, gt, ge, Gd, side)
            with torch.cuda.stream(side):
                dec._pack.unpack(Gd, fd)
                if grad_sync is not None:
                    handles.append(grad_sync(fd))
            dmu, dlv = E.reparam_bwd(d_z, logvar, _f32c(eps))
            gmu.add_(dmu)
            glv.add_(dlv)
            _enc_bwd(de, Pe, ws_e, gmu, glv, Ge, side)
            with torch.cuda.stream(side):
                enc._pack.unpack(Ge, fe)
                if grad_sync is not None:
                    handles.append(grad_sync(fe))
            main.wait_stream(side)
            for p, v in zip(enc_p, enc._pack.views(fe, enc_p)):
                p.grad = v
            for p, v in zip(dec_p, dec._pack.views(fd, dec_p)):
                p.grad = v
            for h in handles:
                if h is not None:
                    h.wait()
        return out4

    def training_step(self, batch, batch_nb, eps=None):
        out4, acc = self._losses(batch, self.beta, eps)     # :182 uses self.beta
        logs = {"train/edge_loss": out4[0].detach(), "train/vertex_loss": out4[1].detach(),
                "train/kl_loss": out4[2].detach(), "train/loss": out4[3].detach(), "train/acc": acc}
        return {"loss": out4[3], "log": logs}

    def validation_step(self, batch, batch_nb, eps=None):
        with torch.no_grad():
            out4, acc = self._losses(batch, self.beta, eps)
            return {"val/edge_loss": out4[0], "val/vertex_loss": out4[1], "val/kl_loss": out4[2],
                    "val_loss": out4[3], "val/acc": acc}

    def validation_end(self, outputs):
        with torch.no_grad():
            keys = ["val_loss", "val/edge_loss", "val/vertex_loss", "val/kl_loss", "val/acc"]
            metrics = {k: torch.stack([o[k] for o in outputs]).mean() for k in keys}
        return {"avg_val_loss": metrics["val_loss"], "log": metrics}

    def configure_optimizers(self):
        optimizer = torch.optim.Adam(self.parameters(), lr=1e-3)
        lr_schedule = torch.optim.lr_scheduler.ReduceLROnPlateau(optimizer, min_lr=1e-6)
        return [optimizer], [lr_schedule]

    def train_dataloader(self):
        return self.train_loader

    def val_dataloader(self):
        return self.val_loader[0]

    def test_dataloader(self):
        return self.test_loader

    def model_eval(self, X, sample_num, decode_num, is_stochastic=True, eps=None, uniforms=None):
        with torch.no_grad():
            true_graph = X["graph"]
            mu, logvar = self.encoder(true_graph)
            stack_mu = torch.stack([mu] * sample_num, dim=1)
            stack_logvar = torch.stack([logvar] * sample_num, dim=1)
            stack_z = self.reparamterize(stack_mu, stack_logvar, eps)
            batch_size, samples, hidden_size = stack_z.shape
            assert samples == sample_num
            stack_z = stack_z.repeat(1, decode_num, 1)
            sampled_z = stack_z.view(batch_size * sample_num * decode_num, hidden_size)
            type_scores, edge_scores = self.decoder(sampled_z)
            onehot, flags = self._predict_from_logits(type_scores, edge_scores, is_stochastic=is_stochastic,
                                                      uniforms=uniforms)
            pred_graph = torch.cat((onehot, flags), dim=-1)
            pred_graph = pred_graph.view(batch_size, decode_num * samples, *pred_graph.shape[1:])
            correct = (true_graph.unsqueeze(1) == pred_graph).all(dim=-1).all(dim=-1)
            acc = correct.sum().item() / (batch_size * decode_num * samples)
            return {"recon_acc": acc}
