"""Drop-in mirror of the reference model API (dvae/models/dvae_model.py):
``Encoder``, ``Decoder``, ``Dvae`` with the same constructor arguments, method
names (including the ``reparamterize`` spelling, :410), return values and
``state_dict()`` keys/shapes -- backed by the gfx950 kernels in libdvae_hip.so.

The torch layers below only own the parameters (so checkpoints keep the
reference's key names); no torch op touches the hot path.
"""
import ctypes

import os
import torch
import torch.nn as nn

from . import _lib
from . import engine as E
from .metrics import reconstruction_accuracy


def _batch_parts(X):
    g, x = X['graph'], X['node_encoding']
    assert g.dim() == 3 and x.dim() == 3 and g.shape[0] == x.shape[0] and g.shape[1] == x.shape[1], \
        "graph must be [B,N,N] and node_encoding [B,N,C]"
    sizes = X.get('sizes', None) if isinstance(X, dict) else None
    return g, x, sizes


class Encoder(nn.Module):
    """dvae_model.py:13-126."""

    def __init__(self, num_classes, seq_len, hidden_state_size=501,
                 latent_space_dim=56, mod=None, is_bidirectional=False,
                 mapping_network=None, gating_network=None):
        super(Encoder, self).__init__()
        if mod not in (None, 'bn'):
            raise NotImplementedError
        self.seq_len = seq_len
        self.hidden_state_size = hidden_state_size
        self.num_classes = num_classes
        self.latent_space_dim = latent_space_dim
        gm_in = hidden_state_size + seq_len if mod is None else num_classes
        self.gating_network = gating_network
        if gating_network is None:
            self.gating_network = nn.Sequential(nn.Linear(gm_in, hidden_state_size), nn.Sigmoid())
        self.mapping_network = mapping_network
        if mapping_network is None:
            self.mapping_network = nn.Linear(gm_in, hidden_state_size)
        self.gru = nn.GRUCell(num_classes, hidden_state_size)
        self.lin11 = nn.Linear(hidden_state_size, latent_space_dim)
        self.lin12 = nn.Linear(hidden_state_size, latent_space_dim)
        self.mod = mod
        self.is_bidirectional = is_bidirectional
        if is_bidirectional:
            self.back_gru = nn.GRUCell(num_classes, hidden_state_size)
            self.back_mapping_network = nn.Linear(gm_in, hidden_state_size)
            self.back_gating_network = nn.Sequential(nn.Linear(gm_in, hidden_state_size), nn.Sigmoid())
            self.lin01 = nn.Linear(2 * hidden_state_size, hidden_state_size)
        self._dims = _lib.make_dims(1, seq_len, num_classes, hidden_state_size, latent_space_dim, mod,
                                    is_bidirectional, seq_len)
        self._pack = E.ParamPack("enc")

    def hip_params(self):
        """Parameters in the canonical order of dvae_enc_pack_table (pack.hip)."""
        ps = [self.gating_network[0].weight, self.gating_network[0].bias,
              self.mapping_network.weight, self.mapping_network.bias,
              self.gru.weight_ih, self.gru.weight_hh, self.gru.bias_ih, self.gru.bias_hh,
              self.lin11.weight, self.lin11.bias, self.lin12.weight, self.lin12.bias]
        if self.is_bidirectional:
            ps += [self.back_gru.weight_ih, self.back_gru.weight_hh, self.back_gru.bias_ih, self.back_gru.bias_hh,
                   self.back_mapping_network.weight, self.back_mapping_network.bias,
                   self.back_gating_network[0].weight, self.back_gating_network[0].bias,
                   self.lin01.weight, self.lin01.bias]
        return ps

    def _check(self, g, x):
        B, N, _ = g.shape
        if self.mod is None:
            assert N <= self.seq_len, \
                'graph has %d nodes but the position encoding has %d columns' % (N, self.seq_len)
        assert x.shape[2] == self.num_classes, 'node_encoding must have %d classes' % self.num_classes

    def forward(self, X, sched=None):
        g, x, sizes = _batch_parts(X)
        self._check(g, x)
        if sched is None:
            sched = E.Schedule(g, sizes, want_reverse=self.is_bidirectional, x=x)
        return E.EncoderFn.apply(self, sched, E._f32c(x), *self.hip_params())

    def predict(self, X):
        return self.forward(X)

    def node_states(self, X):
        """Per-node hidden states [B,N,H] (forward, reverse or None) -- parity tests."""
        g, x, sizes = _batch_parts(X)
        sched = E.Schedule(g, sizes, want_reverse=self.is_bidirectional, x=x)
        dims = E._dims_for(self._dims, sched.B, sched.N)
        with torch.no_grad():
            P = self._pack.pack(dims, self.hip_params())
            call = E.EncoderCall(dims, sched, E._f32c(x), P, want_states=True)
        return call.states, call.states_rev


class Decoder(nn.Module):
    """dvae_model.py:129-362."""

    def __init__(self, num_classes, seq_len, hidden_state_size=501,
                 latent_space_dim=56, mod=None,
                 mapping_network=None, gating_network=None):
        super(Decoder, self).__init__()
        if mod not in (None, 'bn'):
            raise NotImplementedError
        self.seq_len = seq_len
        self.hidden_state_size = hidden_state_size
        self.num_classes = num_classes
        self.latent_space_dim = latent_space_dim
        H = hidden_state_size
        gm_in = H + seq_len if mod is None else num_classes
        self.lin1 = nn.Linear(latent_space_dim, H)
        self.add_vertex = nn.Sequential(nn.Linear(H, 2 * H), nn.ReLU(), nn.Linear(2 * H, num_classes))
        nin = 3 if mod == 'bn' else 2
        self.add_edge = nn.Sequential(nn.Linear(nin * H, 2 * nin * H), nn.ReLU(), nn.Linear(2 * nin * H, 1))
        self.gating_network = gating_network
        if gating_network is None:
            self.gating_network = nn.Sequential(nn.Linear(gm_in, H), nn.Sigmoid())
        self.mapping_network = mapping_network
        if mapping_network is None:
            self.mapping_network = nn.Linear(gm_in, H)
        self.gru = nn.GRUCell(num_classes, H)
        self.mod = mod
        self._dims = _lib.make_dims(1, seq_len, num_classes, H, latent_space_dim, mod, False, seq_len)
        self._pack = E.ParamPack("dec")

    def hip_params(self):
        """Parameters in the canonical order of dvae_dec_pack_table (pack.hip)."""
        return [self.lin1.weight, self.lin1.bias,
                self.add_vertex[0].weight, self.add_vertex[0].bias,
                self.add_vertex[2].weight, self.add_vertex[2].bias,
                self.add_edge[0].weight, self.add_edge[0].bias,
                self.add_edge[2].weight, self.add_edge[2].bias,
                self.gating_network[0].weight, self.gating_network[0].bias,
                self.mapping_network.weight, self.mapping_network.bias,
                self.gru.weight_ih, self.gru.weight_hh, self.gru.bias_ih, self.gru.bias_hh]

    def forward(self, z, X, sched=None):
        g, x, sizes = _batch_parts(X)
        if self.mod is None:
            assert g.shape[1] <= self.seq_len, \
                'graph has %d nodes but the position encoding has %d columns' % (g.shape[1], self.seq_len)
        assert z.shape == (g.shape[0], self.latent_space_dim), 'z must be [B,%d]' % self.latent_space_dim
        if sched is None:
            sched = E.Schedule(g, sizes, want_reverse=False, x=x)
        return E.DecoderFn.apply(self, sched, z, E._f32c(x), *self.hip_params())

    def node_states(self, z, X):
        g, x, sizes = _batch_parts(X)
        sched = E.Schedule(g, sizes)
        dims = E._dims_for(self._dims, sched.B, sched.N)
        with torch.no_grad():
            P = self._pack.pack(dims, self.hip_params())
            call = E.DecoderCall(dims, sched, E._f32c(z), E._f32c(x), P, want_states=True)
        return call.states

    def predict(self, z, is_stochastic=False, uniforms=None, return_margin=False):
        """Autoregressive decode (:263-362).  Greedy by default; stochastic mode
        draws its uniforms on the device unless `uniforms` = (u_type [M,N],
        u_edge [M,N,N]) is injected (parity tests)."""
        E._require_cuda(z, "z")
        M = z.shape[0]
        dims = E._dims_for(self._dims, M, self.seq_len)
        u_type = u_edge = None
        if is_stochastic or uniforms is not None:
            if uniforms is None:
                u_type = torch.rand(M, self.seq_len, device=z.device)
                u_edge = torch.rand(M, self.seq_len, self.seq_len, device=z.device)
            else:
                u_type, u_edge = (E._f32c(u) for u in uniforms)
        with torch.no_grad():
            P = self._pack.pack(dims, self.hip_params())
            edges, onehot, margin = E.predict_call(dims, E._f32c(z), P, u_type, u_edge, return_margin)
        if return_margin:
            return edges, onehot, margin
        return edges, onehot


class Dvae(nn.Module):
    """dvae_model.py:365-524 (a LightningModule there; the hooks Lightning calls
    keep their names and return values)."""

    def __init__(self, hparams):
        super(Dvae, self).__init__()
        self.hparams = hparams
        task_type = getattr(hparams, 'task_type', 'enas')
        if hparams.dataset_file:
            from .data import get_dataloaders
            self.train_loader, self.val_loader, self.test_loader = get_dataloaders(
                hparams.batch_size, hparams.dataset_file, task_type=task_type)
        else:
            self.train_loader, self.val_loader, self.test_loader = [None] * 3
        self.num_classes = getattr(hparams, 'num_classes', 8)
        self.max_seq = getattr(hparams, 'max_seq', 8)
        self.hidden_state_size = getattr(hparams, 'hidden_state_size', 501)
        self.mod = hparams.mod
        kw = {}
        if getattr(hparams, 'same_mapper_gater', False):
            gm_in = self.hidden_state_size + self.max_seq if self.mod is None else self.num_classes
            self.gating_network = nn.Sequential(nn.Linear(gm_in, self.hidden_state_size), nn.Sigmoid())
            self.mapping_network = nn.Linear(gm_in, self.hidden_state_size)
            kw = dict(mapping_network=self.mapping_network, gating_network=self.gating_network)
        self.encoder = Encoder(num_classes=self.num_classes, seq_len=self.max_seq,
                               hidden_state_size=self.hidden_state_size, mod=self.mod,
                               is_bidirectional=hparams.bidirectional, **kw)
        self.decoder = Decoder(num_classes=self.num_classes, seq_len=self.max_seq,
                               hidden_state_size=self.hidden_state_size, mod=self.mod, **kw)
        self._shared_gm = bool(kw)
        self.beta = getattr(hparams, 'beta', 0.005)
        self._flat = None
        self._prep = None

    # -- reference API -------------------------------------------------------
    def reparamterize(self, mu, logvar, eps=None):
        if eps is None:
            eps = torch.randn_like(mu)
        return E.ReparamFn.apply(mu, logvar, eps)

    def _schedule(self, X):
        g, x, sizes = _batch_parts(X)
        return E.Schedule(g, sizes, want_reverse=self.encoder.is_bidirectional, x=x)

    def forward(self, X, eps=None):
        sched = self._schedule(X)
        _, mu, logvar = self.encoder(X, sched=sched)
        z = self.reparamterize(mu, logvar, eps)
        X_hat = self.decoder(z, X, sched=sched)
        return X_hat, mu, logvar

    def _losses(self, batch, beta, eps=None):
        g, x, sizes = _batch_parts(batch)
        (e, t), mu, logvar = self.forward(batch, eps=eps)
        dims = E._dims_for(self.decoder._dims, g.shape[0], g.shape[1])
        sz = None if sizes is None else sizes.to(device=g.device, dtype=torch.int32).contiguous()
        return E.ElboFn.apply(dims, e, t, E._f32c(g), E._f32c(x), sz, mu, logvar, float(beta))

    def training_step(self, batch, batch_nb, eps=None):
        out4 = self._losses(batch, 0.005, eps)           # :432 hard-codes 0.005
        vals = out4.detach().tolist()                    # one host sync for the four logged floats
        logs = {'train/edge_loss': vals[0], 'train/vertex_loss': vals[1],
                'train/kl_loss': vals[2], 'train/loss': vals[3]}
        return {'loss': out4[3], 'log': logs}

    def validation_step(self, batch, batch_nb, eps=None):
        with torch.no_grad():
            out4 = self._losses(batch, self.beta, eps)   # :454 uses self.beta
            return {'val/edge_loss': out4[0], 'val/vertex_loss': out4[1],
                    'val/kl_loss': out4[2], 'val_loss': out4[3]}

    def validation_end(self, outputs):
        with torch.no_grad():
            keys = ['val_loss', 'val/edge_loss', 'val/vertex_loss', 'val/kl_loss']
            metrics = {k: torch.stack([o[k] for o in outputs]).mean() for k in keys}
        return {'avg_val_loss': metrics['val_loss'], 'log': metrics}

    def configure_optimizers(self, flat=False):
        """dvae_model.py:471-475: Adam(lr=1e-4) + ReduceLROnPlateau.  ``flat=True``: the same update as one HIP launch
        per module over flat buffers (optim.FlatAdam: a torch Optimizer with torch Adam's state_dict layout); models
        whose modules share parameters (same_mapper_gater) keep torch's."""
        if flat and not self._shared_gm and next(self.parameters()).is_cuda:
            from .optim import FlatAdam
            optimizer = FlatAdam(self, lr=1e-4)
        else:
            optimizer = torch.optim.Adam(self.parameters(), lr=1e-4)
        lr_schedule = torch.optim.lr_scheduler.ReduceLROnPlateau(optimizer)
        return [optimizer], [lr_schedule]

    def train_dataloader(self):
        return self.train_loader

    def val_dataloader(self):
        return self.val_loader[0]

    def test_dataloader(self):
        return self.test_loader

    def predict(self, X, eps=None, return_margin=False):
        """:492-496.  ``eps`` [B,Z] (optional) replaces the reparameterisation draw (parity tests: device
        and CPU generators cannot match)."""
        with torch.no_grad():
            _, mu, logvar = self.encoder.predict(X)
            z = self.reparamterize(mu, logvar, eps)
        return self.decoder.predict(z, return_margin=return_margin)

    def model_eval(self, X, sample_num, decode_num, is_stochastic=True, eps=None, uniforms=None, return_decoded=False):
        """:498-524.  ``eps`` [B,sample_num,Z] replaces the reparameterisation draw, ``uniforms`` =
        (u_type [B*S*D,N], u_edge [B*S*D,N,N]) the decoder's draws (stochastic mode); both optional.
        ``return_decoded`` adds the decoded tensors [B, D*S, N, .] to the result."""
        with torch.no_grad():
            _, mu, logvar = self.encoder.predict(X)
            stack_mu = torch.stack([mu] * sample_num, dim=1)
            stack_logvar = torch.stack([logvar] * sample_num, dim=1)
            stack_z = self.reparamterize(stack_mu, stack_logvar, eps)
            batch_size, samples, hidden_size = stack_z.shape
            assert samples == sample_num
            stack_z = stack_z.repeat(1, decode_num, 1)
            sampled_z = stack_z.view(batch_size * sample_num * decode_num, hidden_size)
            gen_edges, gen_nodes = self.decoder.predict(sampled_z, is_stochastic, uniforms=uniforms)
            gen_edges = gen_edges.view(batch_size, decode_num * sample_num, *gen_edges.shape[1:])
            gen_nodes = gen_nodes.view(batch_size, decode_num * sample_num, *gen_nodes.shape[1:])
            recon_acc = reconstruction_accuracy(X['node_encoding'], X['graph'], gen_nodes, gen_edges)
            out = {'recon_acc': recon_acc}
            if return_decoded:
                out['gen_edges'], out['gen_nodes'] = gen_edges, gen_nodes
            return out

    def generation_eval(self, num_samples=None, z=None, train_batch=None, is_stochastic=False, task_type='enas',
                        dist=None):
        """Addition (not in the reference; BASELINE config 5's validity/novelty path): decode
        ``num_samples`` draws from the prior N(0,I) -- or the given latents ``z`` -- and report
        validity, uniqueness and (given the training graphs) novelty of the decoded DAGs
        (definitions in metrics.py).  ``dist``: shard the decode over the ranks (dp.predict_sharded)."""
        from . import dp, metrics
        dev = next(self.parameters()).device
        if z is None:
            z = torch.randn(int(num_samples), self.decoder.lin1.in_features, device=dev)
        with torch.no_grad():
            edges, onehots = dp.predict_sharded(self.decoder, z, dist, is_stochastic=is_stochastic)
        te = tx = None
        if train_batch is not None:
            te, tx = train_batch['graph'], train_batch['node_encoding']
        return metrics.generation_report(edges, onehots, te, tx, task_type)

    # -- fused train step (no autograd graph) --------------------------------------
    def _flat_grads(self):
        """Persistent flat gradient buffers (one per module) with p.grad installed
        as views, so a data-parallel all-reduce is one collective per buffer.
        Returns (fe, fd, encoder views, decoder views); the views are made once per buffer."""
        enc_p, dec_p = self.encoder.hip_params(), self.decoder.hip_params()
        key = tuple(p.data_ptr() for p in enc_p + dec_p)
        if self._flat is None or self._flat[0] != key:
            dev = enc_p[0].device
            fe = torch.zeros(sum(p.numel() for p in enc_p), device=dev)
            fd = torch.zeros(sum(p.numel() for p in dec_p), device=dev)
            ve, vd, o = [], [], 0
            for p in enc_p:
                ve.append(fe[o:o + p.numel()].view_as(p))
                o += p.numel()
            o = 0
            for p in dec_p:
                vd.append(fd[o:o + p.numel()].view_as(p))
                o += p.numel()
            self._flat = (key, fe, fd, ve, vd)
        return self._flat[1:]

    def _step_state(self, B, N, device):
        key = (int(B), int(N), str(device))
        if getattr(self, '_step', None) is None or self._step[0] != key:
            d_enc = E._dims_for(self.encoder._dims, B, N)
            d_dec = E._dims_for(self.decoder._dims, B, N)
            self._step = (key, E.StepState(d_enc, d_dec, device))
        return self._step[1]

    def prepare(self, batch, resident=False, on_side_stream=False):
        """Queue the level-schedule build of a FUTURE batch on a stream of its own and
        return at once; pass the result to ``loss_and_grad(batch, sched=...)``.  A data
        loader that calls this one batch ahead hides the schedule's host round trip
        (its build kernels run beside the previous step when ``resident``).
        ``on_side_stream``: queue it on the library's side stream instead of a third stream --
        call it just BEFORE the current step's ``loss_and_grad``: the side stream idles through the
        encoder's forward sweep, and a process that also runs RCCL stays within the 4 hardware
        queues the box gives it."""
        g, x, sizes = _batch_parts(batch)
        dev = g.device
        if on_side_stream:
            stream = E.side_stream(dev)
        else:
            if self._prep is None or self._prep.device != dev:
                self._prep = torch.cuda.Stream(device=dev)
            stream = self._prep
        if not resident:   # resident=True: the batch tensors are complete already, start at once
            stream.wait_stream(torch.cuda.current_stream())
        return E.Schedule(g, sizes, want_reverse=self.encoder.is_bidirectional, stream=stream, defer=True, x=x)

    def loss_and_grad(self, batch, eps=None, beta=0.005, grad_sync=None, sched=None, on_decoder_grads=None):
        """Forward + backward of training_step's loss without building an autograd
        graph: same kernels, gradients written straight into flat buffers whose
        views become ``p.grad``.  ``grad_sync(flat)`` (optional) is called as soon
        as a buffer is complete -- the decoder's while the encoder's backward is
        still running -- and may return a handle with ``.wait()``.  ``sched``: the
        result of ``prepare(batch)`` for this batch (built ahead of time).
        ``on_decoder_grads()`` (optional) runs on the library's side stream as soon as the decoder's
        gradients are final (all-reduced when ``grad_sync`` is given) and installed as ``p.grad`` -- e.g. the
        decoder's optimizer step, which then overlaps the encoder's backward (not with same_mapper_gater,
        whose shared gradients are complete only at the end).
        Returns the device tensor [edge_loss, vertex_loss, kl_loss, loss]."""
        g, x, sizes = _batch_parts(batch)
        self.encoder._check(g, x)
        enc, dec = self.encoder, self.decoder
        enc_p, dec_p = enc.hip_params(), dec.hip_params()
        with torch.no_grad():
            main = torch.cuda.current_stream()
            if sched is not None:
                assert sched.B == int(g.shape[0]) and sched.N == int(g.shape[1]), "sched was prepared for another batch"
                sched.join(main)
                sched.wait()
            else:
                sched = self._schedule(batch)
            xs = E._f32c(x)
            fe, fd, ve, vd = self._flat_grads()
            side = E.side_stream(main.device)
            st = self._step_state(sched.B, sched.N, main.device)
            enc._pack.ensure(st.d_enc, enc_p)
            dec._pack.ensure(st.d_dec, dec_p)
            if eps is None:
                eps = torch.randn(sched.B, enc.latent_space_dim, device=main.device)
            eps = E._f32c(eps)
            assert eps.shape == (sched.B, enc.latent_space_dim), "eps must be [B,%d]" % enc.latent_space_dim
            out4 = E._out(4, device=main.device)
            ws_enc, ws_dec = st.workspaces(sched.R, sched.E)
            c = st.step
            c.g, c.x = sched.g.data_ptr(), xs.data_ptr()
            c.sizes = None if sched.sizes is None else sched.sizes.data_ptr()
            c.sched, c.info_host = sched.buf.data_ptr(), sched.info.ctypes.data
            c.enc_table, c.enc_n = enc._pack.table.data_ptr(), enc._pack.n
            c.dec_table, c.dec_n = dec._pack.table.data_ptr(), dec._pack.n
            c.Pe, c.Pd = enc._pack.P.data_ptr(), dec._pack.P.data_ptr()
            c.ws_enc, c.ws_dec, c.ws_step = ws_enc.data_ptr(), ws_dec.data_ptr(), st.ws_step.data_ptr()
            c.eps, c.beta = eps.data_ptr(), float(beta)
            c.Ge, c.Gd, c.out4 = st.Ge.data_ptr(), st.Gd.data_ptr(), out4.data_ptr()
            handles = []
            raw_main, raw_side = E._stream(), E._side(side)
            side.wait_stream(main)
            # part 1: everything up to the decoder's backward (its weight-gradient products and the
            # decoder's unpack + all-reduce run on `side`, under the encoder's backward)
            E.check(E.load().dvae_step_part1(ctypes.byref(c), raw_main, raw_side))
            with torch.cuda.stream(side):
                dec._pack.unpack(st.Gd, fd)
                h = grad_sync(fd) if grad_sync is not None else None
                if on_decoder_grads is not None and not self._shared_gm:
                    if h is not None:
                        h.wait()              # orders the side stream after the collective
                    for p, v in zip(dec_p, vd):
                        p.grad = v
                    on_decoder_grads()
                elif h is not None:
                    handles.append(h)
            E.check(E.load().dvae_step_part2(ctypes.byref(c), raw_main, raw_side))
            with torch.cuda.stream(side):
                enc._pack.unpack(st.Ge, fe)
                if grad_sync is not None:
                    handles.append(grad_sync(fe))
            main.wait_stream(side)
            self._step_keep = (sched, xs, eps)       # inputs of the queued kernels stay alive until the next step
            for h in handles:
                if h is not None:
                    h.wait()
            for p, v in zip(enc_p, ve):
                p.grad = v
            owner = {id(p): v for p, v in zip(enc_p, ve)} if self._shared_gm else {}
            for p, v in zip(dec_p, vd):
                if id(p) in owner:
                    # same_mapper_gater (:382-399): one gate/map network serves both modules; its gradient is
                    # the sum of the two modules' parts (each already all-reduced with its buffer)
                    owner[id(p)].add_(v)
                else:
                    p.grad = v
        return out4
