"""Host-side glue between torch tensors and the C ABI: schedules, packed
parameters, workspaces and the autograd Functions behind Encoder / Decoder /
the ELBO.  torch is plumbing here (device memory, streams, autograd graph); all
arithmetic of the hot path happens in libdvae_hip.so.
"""
import ctypes
import os

import numpy as np
import torch

from . import _lib
from ._lib import check, load


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _side(side):
    """torch.cuda.Stream (or None) -> hipStream_t for the library's side stream."""
    return None if side is None else ctypes.c_void_p(side.cuda_stream)


_side_streams = {}


def side_stream(device):
    """The library's low-priority side stream of `device` (one per device per
    process), wrapped as a torch stream so torch ops / RCCL can be queued on it."""
    idx = torch.device(device).index
    if idx is None:
        idx = torch.cuda.current_device()
    if idx not in _side_streams:
        raw, prio = ctypes.c_void_p(), ctypes.c_int()
        with torch.cuda.device(idx):
            check(load().dvae_side_stream_create(ctypes.byref(raw), ctypes.byref(prio)))
        _side_streams[idx] = torch.cuda.ExternalStream(raw.value, device=torch.device("cuda", idx))
    return _side_streams[idx]


def _require_cuda(t, name):
    if not t.is_cuda:
        raise _lib.DvaeError(
            "%s is on %s: the D-VAE hot path runs only on a ROCm device "
            "(there is no CPU fallback)" % (name, t.device))


def _f32c(t):
    return t.detach().to(torch.float32).contiguous()


_POISON = bool(os.environ.get("DVAE_POISON_WS"))


def _workspace(nbytes, device):
    """Uninitialised device bytes for a *_ws_bytes workspace.  DVAE_POISON_WS=1 (test aid) fills it with
    0xFF bytes -- NaN as floats, -1 as ints -- so that a kernel reading a location nothing wrote shows
    up as NaN results instead of depending on what the allocator handed back."""
    ws = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
    if _POISON:
        ws.fill_(255)
    return ws


def _out(*shape, device):
    t = torch.empty(*shape, device=device)
    if _POISON:
        t.fill_(float('nan'))
    return t


# ---------------------------------------------------------------------------
# level schedule of one batch
# ---------------------------------------------------------------------------
_pinned = {}


def _pinned_info(dev_index, nint):
    """Small ring of pinned host buffers for schedule summaries (a schedule built ahead of
    time must not share its buffer with the one in use)."""
    key = (dev_index, nint)
    ring = _pinned.setdefault(key, [[], 0])
    if len(ring[0]) < 8:
        ring[0].append(torch.empty(nint, dtype=torch.int32).pin_memory())
        return ring[0][-1]
    ring[1] = (ring[1] + 1) % len(ring[0])
    return ring[0][ring[1]]


class Schedule:
    """Topological-level schedule of a batch of DAGs (device) + its host summary.

    g [B,N,N] fp32 0/1 strictly lower-triangular, g[b,i,j]=1 <=> edge j->i
    (reference layout, dataloader.py:75-86); sizes optional int32 [B].
    The per-level row counts are needed on the host to size the per-level
    GEMMs, so a schedule costs one wait for its build kernels.  With
    ``defer=True`` (``Dvae.prepare``) the build is queued on ``stream`` and the
    wait happens on first use -- built one batch ahead, it costs nothing.
    """

    def __init__(self, g, sizes=None, want_reverse=False, stream=None, defer=False, x=None):
        _require_cuda(g, "X['graph']")
        assert g.dim() == 3 and g.shape[1] == g.shape[2], "graph must be [B,N,N]"
        lib = load()
        self.B, self.N = int(g.shape[0]), int(g.shape[1])
        self.want_reverse = bool(want_reverse)
        self.stream = stream if stream is not None else torch.cuda.current_stream()
        self._x = None
        with torch.cuda.stream(self.stream):
            # dtype / layout conversions run on the build stream too: queued on the caller's stream they
            # would not be ordered before the build kernels (Dvae.prepare builds on a stream of its own),
            # and their results would belong to another stream's allocator pool
            self.g = _f32c(g)
            self.sizes = None if sizes is None else sizes.detach().to(device=g.device, dtype=torch.int32).contiguous()
            nbytes = lib.dvae_sched_bytes(self.B, self.N)
            self.buf = _workspace(max(int(nbytes), 256), g.device)
            self._pin = _pinned_info(g.device.index, _lib.sched_info_ints(self.N))
            check(lib.dvae_sched_build(_ptr(self.g), _ptr(self.sizes), self.B, self.N, int(self.want_reverse),
                                       _ptr(self.buf), ctypes.c_void_p(self._pin.data_ptr()),
                                       ctypes.c_void_p(self.stream.cuda_stream)))
            if x is not None:
                # node types: lets the decoder share its type-only prefix rows (dvae_sched_types)
                assert x.dim() == 3 and x.shape[0] == self.B and x.shape[1] == self.N, "node_encoding must be [B,N,C]"
                self._x = _f32c(x)
                check(lib.dvae_sched_types(_ptr(self._x), self.B, self.N, int(x.shape[2]), _ptr(self.buf),
                                           ctypes.c_void_p(self._pin.data_ptr()),
                                           ctypes.c_void_p(self.stream.cuda_stream)))
            self.event = torch.cuda.Event()
            self.event.record(self.stream)
        self._info = None
        if not defer:
            self.wait()

    def wait(self):
        """Host wait for the build (first call only); validates the adjacency."""
        if self._info is None:
            self.event.synchronize()
            self._info = self._pin.numpy().copy()
            self._pin = None
            if self._info[0] != 0:
                raise _lib.GraphError(
                    "X['graph'] must be a 0/1 strictly lower-triangular adjacency "
                    "(node index order = topological order) within each graph's size")
        return self

    def join(self, stream):
        """Order `stream` after the build (no-op on the build's own stream)."""
        if stream != self.stream:
            stream.wait_event(self.event)
            for t in (self.buf, self.g, self.sizes, self._x):   # built (and perhaps converted) on the build stream
                if t is not None:
                    t.record_stream(stream)

    @property
    def info(self):
        return self.wait()._info

    @property
    def R(self):
        return int(self.info[1])

    @property
    def E(self):
        return int(self.info[2])

    @property
    def nlev_fwd(self):
        return int(self.info[3])

    @property
    def nlev_rev(self):
        return int(self.info[4])

    @property
    def onehot(self):
        """True when dvae_sched_types ran and found every node encoding one-hot."""
        return int(self.info[5]) == 1

    @property
    def shared_rows(self):
        """(value rows, links between them) of the encoder's forward sweep when dvae_sched_types shared them, else None."""
        m = _lib.sched_merge_at(self.N)
        return (int(self.info[m + 1]), int(self.info[m + 2])) if int(self.info[m]) == 1 else None

    @property
    def info_ptr(self):
        return ctypes.c_void_p(self.info.ctypes.data)

    def export(self, which, n):
        out = torch.empty(n, dtype=torch.int32, device=self.buf.device)
        check(load().dvae_sched_export(_ptr(self.buf), self.B, self.N, which, _ptr(out), n, _stream()))
        return out

    def levels(self, reverse=False):
        return self.export(1 if reverse else 0, self.B * self.N).view(self.B, self.N)


# ---------------------------------------------------------------------------
# packed parameters
# ---------------------------------------------------------------------------
class ParamPack:
    """Zero-padded device copy of one module's parameters + the table that maps
    reference-shaped tensors to it (and packed gradients back)."""

    # kind -> C entry points (parameter count, pack table, packed floats, packed gradient floats)
    _FNS = {"enc": ("dvae_enc_param_count", "dvae_enc_pack_table", "dvae_enc_param_floats", "dvae_enc_grad_floats"),
            "dec": ("dvae_dec_param_count", "dvae_dec_pack_table", "dvae_dec_param_floats", "dvae_dec_grad_floats"),
            "svae_enc": ("dvae_svae_enc_param_count", "dvae_svae_enc_pack_table", "dvae_svae_enc_param_floats",
                         "dvae_svae_enc_param_floats"),
            "svae_dec": ("dvae_svae_dec_param_count", "dvae_svae_dec_pack_table", "dvae_svae_dec_param_floats",
                         "dvae_svae_dec_param_floats")}

    def __init__(self, kind):
        assert kind in self._FNS
        self.kind = kind
        self.key = None
        self.table = None
        self.n = 0
        self.P = None
        self.offsets = None
        self.total = 0

    def _build(self, dims, params):
        lib = load()
        f_count, f_table, f_floats, f_gfloats = (getattr(lib, n) for n in self._FNS[self.kind])
        count = f_count(ctypes.byref(dims))
        assert len(params) == count, "expected %d parameters, got %d" % (count, len(params))
        raw = np.array([p.data_ptr() for p in params], dtype=np.uint64)
        offs, o = [], 0
        for p in params:
            offs.append(o)
            o += p.numel()
        self.offsets, self.total = offs, o
        goff = np.array(offs, dtype=np.int64)
        host = (_lib.PackEntry * 256)()
        n = f_table(ctypes.byref(dims), ctypes.c_void_p(raw.ctypes.data), ctypes.c_void_p(goff.ctypes.data),
               ctypes.cast(host, ctypes.c_void_p), 256)
        if n < 0:
            check(n)
        self.n = n
        dev = params[0].device
        blob = np.frombuffer(bytes(host), dtype=np.uint8)[: n * ctypes.sizeof(_lib.PackEntry)].copy()
        self.table = torch.from_numpy(blob).to(dev)
        self.nfloats = int(f_floats(ctypes.byref(dims)))
        self.gfloats = int(f_gfloats(ctypes.byref(dims)))
        self.P = torch.zeros(self.nfloats, dtype=torch.float32, device=dev)

    def ensure(self, dims, params):
        """Table and packed buffer for these parameters (rebuilt when a storage or a shape changed)."""
        shape = tuple(getattr(dims, f) for f, _ in dims._fields_ if f not in ("B", "T"))   # batch-independent fields
        key = shape + tuple(p.data_ptr() for p in params)
        if key != self.key:
            for p in params:      # (a tensor keeps dtype and strides for as long as it keeps its storage)
                _require_cuda(p, "parameter")
                assert p.dtype == torch.float32 and p.is_contiguous(), "parameters must be contiguous fp32"
            self._build(dims, params)
            self.key = key
        return self.P

    def pack(self, dims, params):
        self.ensure(dims, params)
        check(load().dvae_pack_params(_ptr(self.table), self.n, _ptr(self.P), _stream()))
        return self.P

    def new_packed_grad(self):
        return torch.zeros(self.gfloats, dtype=torch.float32, device=self.P.device)

    def unpack(self, G, flat=None):
        """packed gradient buffer -> flat reference-shaped gradients (views returned)."""
        if flat is None:
            flat = _out(self.total, device=G.device)
        check(load().dvae_unpack_grads(_ptr(self.table), self.n, _ptr(G), _ptr(flat), _stream()))
        return flat

    def views(self, flat, params):
        return [flat[o:o + p.numel()].view_as(p) for o, p in zip(self.offsets, params)]


# ---------------------------------------------------------------------------
# raw forward / backward calls (no autograd) -- used by the Functions below and by
# the fused train step of model.Dvae
# ---------------------------------------------------------------------------
def _dims_for(d0, B, N):
    d = _lib.DvaeDims()
    ctypes.memmove(ctypes.byref(d), ctypes.byref(d0), ctypes.sizeof(d))
    d.B, d.N = int(B), int(N)
    check(load().dvae_dims_init(ctypes.byref(d)))
    return d


class EncoderCall:
    def __init__(self, dims, sched, x, P, want_states=False):
        lib = load()
        self.dims, self.sched, self.x, self.P = dims, sched, x, P
        dev = x.device
        B = dims.B
        self.ws = _workspace(lib.dvae_enc_ws_bytes(ctypes.byref(dims), sched.R, sched.E), dev)
        self.hv = _out(B, dims.H, device=dev)
        self.mu = _out(B, dims.Z, device=dev)
        self.logvar = _out(B, dims.Z, device=dev)
        self.states = torch.empty(B, dims.N, dims.H, device=dev) if want_states else None
        self.states_rev = torch.empty(B, dims.N, dims.H, device=dev) if (want_states and dims.bidir) else None
        check(lib.dvae_encoder_fwd(ctypes.byref(dims), _ptr(x), _ptr(sched.sizes), _ptr(sched.buf), sched.info_ptr,
                                   _ptr(P), _ptr(self.ws), _ptr(self.hv), _ptr(self.mu), _ptr(self.logvar),
                                   _ptr(self.states), _ptr(self.states_rev), _stream()))

    def backward(self, d_hv, d_mu, d_logvar, G, side=None):
        s = self.sched
        check(load().dvae_encoder_bwd(ctypes.byref(self.dims), _ptr(self.x), _ptr(s.sizes), _ptr(s.buf), s.info_ptr,
                                      _ptr(self.P), _ptr(self.ws), _ptr(d_hv), _ptr(d_mu), _ptr(d_logvar),
                                      _ptr(G), _stream(), _side(side)))


class DecoderCall:
    def __init__(self, dims, sched, z, x, P, want_states=False, side=None):
        lib = load()
        self.dims, self.sched, self.z, self.x, self.P = dims, sched, z, x, P
        dev = x.device
        B, N = dims.B, dims.N
        self.ws = _workspace(lib.dvae_dec_ws_bytes(ctypes.byref(dims), sched.R, sched.E), dev)
        self.edge_logits = _out(B, N, N, device=dev)
        self.type_logits = _out(B, N, dims.C, device=dev)
        self.states = torch.empty(B, N, dims.H, device=dev) if want_states else None
        check(lib.dvae_decoder_fwd(ctypes.byref(dims), _ptr(z), _ptr(x), _ptr(sched.sizes), _ptr(sched.buf),
                                   sched.info_ptr, _ptr(P), _ptr(self.ws), _ptr(self.edge_logits),
                                   _ptr(self.type_logits), _ptr(self.states), _stream(), _side(side)))

    def backward(self, d_edge, d_type, G, want_dz=True, side=None):
        s = self.sched
        d_z = _out(*self.z.shape, device=self.z.device) if want_dz else None
        check(load().dvae_decoder_bwd(ctypes.byref(self.dims), _ptr(self.z), _ptr(self.x), _ptr(s.sizes), _ptr(s.buf),
                                      s.info_ptr, _ptr(self.P), _ptr(self.ws), _ptr(d_edge), _ptr(d_type), _ptr(G),
                                      _ptr(d_z), _stream(), _side(side)))
        return d_z


class StepState:
    """Persistent device buffers of the fused train step (dvae_step_part1 / part2): nothing is allocated
    per step once the batch shape has been seen (workspaces grow to the largest schedule met)."""

    def __init__(self, d_enc, d_dec, device):
        lib = load()
        self.d_enc, self.d_dec, self.device = d_enc, d_dec, device
        self.ws_step = _workspace(lib.dvae_step_ws_bytes(ctypes.byref(d_dec)), device)
        self.Ge = torch.empty(lib.dvae_enc_grad_floats(ctypes.byref(d_enc)), device=device)
        self.Gd = torch.empty(lib.dvae_dec_grad_floats(ctypes.byref(d_dec)), device=device)
        self.ws_enc = self.ws_dec = None
        self.step = _lib.DvaeStep()
        self.step.d_enc, self.step.d_dec = ctypes.pointer(d_enc), ctypes.pointer(d_dec)

    def workspaces(self, R, E):
        lib = load()
        ne = lib.dvae_enc_ws_bytes(ctypes.byref(self.d_enc), R, E)
        nd = lib.dvae_dec_ws_bytes(ctypes.byref(self.d_dec), R, E)
        if self.ws_enc is None or self.ws_enc.numel() < ne:
            self.ws_enc = _workspace(int(ne * 1.1) + 256, self.device)
        if self.ws_dec is None or self.ws_dec.numel() < nd:
            self.ws_dec = _workspace(int(nd * 1.1) + 256, self.device)
        if _POISON:
            self.ws_enc.fill_(255); self.ws_dec.fill_(255); self.ws_step.fill_(255)
        return self.ws_enc, self.ws_dec


def elbo_call(dims, edge_logits, type_logits, g, x, sizes, mu, logvar, beta, want_grads):
    lib = load()
    dev = g.device
    out4 = _out(4, device=dev)
    part = _workspace(lib.dvae_elbo_ws_bytes(dims.B), dev)
    grads = (None, None, None, None)
    if want_grads:
        grads = tuple(_out(*t.shape, device=dev) for t in (edge_logits, type_logits, mu, logvar))
    check(lib.dvae_elbo(ctypes.byref(dims), _ptr(edge_logits), _ptr(type_logits), _ptr(g), _ptr(x), _ptr(sizes),
                        _ptr(mu), _ptr(logvar), ctypes.c_float(beta), _ptr(out4), _ptr(part), _ptr(grads[0]),
                        _ptr(grads[1]), _ptr(grads[2]), _ptr(grads[3]), _stream()))
    return out4, grads


def reparam_fwd(mu, logvar, eps):
    z = torch.empty_like(mu)
    check(load().dvae_reparam_fwd(_ptr(mu), _ptr(logvar), _ptr(eps), _ptr(z), mu.numel(), _stream()))
    return z


def reparam_bwd(d_z, logvar, eps):
    d_mu, d_lv = torch.empty_like(d_z), torch.empty_like(d_z)
    check(load().dvae_reparam_bwd(_ptr(d_z), _ptr(logvar), _ptr(eps), _ptr(d_mu), _ptr(d_lv), d_z.numel(), _stream()))
    return d_mu, d_lv


def predict_call(dims, z, P, u_type=None, u_edge=None, want_margin=False):
    lib = load()
    dev = z.device
    M, N, C = dims.B, dims.N, dims.C
    ws = _workspace(lib.dvae_predict_ws_bytes(ctypes.byref(dims)), dev)
    edges = _out(M, N, N, device=dev)
    onehot = _out(M, N, C, device=dev)
    margin = torch.empty(1, device=dev) if want_margin else None
    check(lib.dvae_decoder_predict(ctypes.byref(dims), _ptr(z), _ptr(P), _ptr(ws), _ptr(u_type), _ptr(u_edge),
                                   _ptr(edges), _ptr(onehot), _ptr(margin), _stream()))
    return edges, onehot, margin


# ---------------------------------------------------------------------------
# autograd Functions
# ---------------------------------------------------------------------------
def _bwd_on_two_streams(pack, G, run):
    """A module's backward as the fused step runs it: the level chain on the current stream, every
    weight-gradient product on the library's side stream, the unpack behind them there; the current
    stream continues once the side stream is through.  Returns (flat gradients, run's result)."""
    main = torch.cuda.current_stream()
    side = side_stream(G.device)
    side.wait_stream(main)
    flat = _out(pack.total, device=G.device)   # owned by `main`'s allocator pool
    res = run(side)
    with torch.cuda.stream(side):
        pack.unpack(G, flat)
    main.wait_stream(side)
    return flat, res


class EncoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, sched, x, *params):
        dims = _dims_for(module._dims, sched.B, sched.N)
        P = module._pack.pack(dims, params)
        call = EncoderCall(dims, sched, x, P)
        ctx.call, ctx.module = call, module
        # the backward reads the module's shared packed copy P: saving the parameters lets autograd's version
        # check refuse a backward after an in-place weight change (optimizer.step between forward and backward)
        ctx.save_for_backward(*params)
        return call.hv, call.mu, call.logvar

    @staticmethod
    def backward(ctx, d_hv, d_mu, d_logvar):
        pack = ctx.module._pack
        params = ctx.saved_tensors
        G = pack.new_packed_grad()
        cont = lambda t: None if t is None else t.contiguous()
        flat = _bwd_on_two_streams(
            pack, G, lambda side: ctx.call.backward(cont(d_hv), cont(d_mu), cont(d_logvar), G, side=side))[0]
        grads = pack.views(flat, params)
        ctx.call = None
        return (None, None, None) + tuple(grads)


class DecoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, sched, z, x, *params):
        dims = _dims_for(module._dims, sched.B, sched.N)
        P = module._pack.pack(dims, params)
        main = torch.cuda.current_stream()
        side = side_stream(x.device)       # add_edge's first layer and add_vertex run beside the level chain;
        side.wait_stream(main)             # the library orders the side stream back into `main` before it returns
        call = DecoderCall(dims, sched, _f32c(z), x, P, side=side)
        ctx.call, ctx.module = call, module
        ctx.save_for_backward(*params)      # version check, as in EncoderFn
        ctx.need_dz = z.requires_grad
        return call.edge_logits, call.type_logits

    @staticmethod
    def backward(ctx, d_edge, d_type):
        pack = ctx.module._pack
        params = ctx.saved_tensors
        G = pack.new_packed_grad()
        flat, d_z = _bwd_on_two_streams(
            pack, G, lambda side: ctx.call.backward(d_edge.contiguous(), d_type.contiguous(), G, want_dz=True, side=side))
        grads = pack.views(flat, params)
        ctx.call = None
        return (None, None, d_z, None) + tuple(grads)


class ReparamFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mu, logvar, eps):
        mu, logvar, eps = mu.contiguous(), logvar.contiguous(), eps.contiguous()
        ctx.save_for_backward(logvar, eps)
        return reparam_fwd(mu, logvar, eps)

    @staticmethod
    def backward(ctx, d_z):
        logvar, eps = ctx.saved_tensors
        d_mu, d_lv = reparam_bwd(d_z.contiguous(), logvar, eps)
        return d_mu, d_lv, None


class ElboFn(torch.autograd.Function):
    """(edge_logits, type_logits, mu, logvar) -> [edge_loss, vertex_loss, kl, loss]
    with the gradients of `loss` computed in the same kernel pass."""

    @staticmethod
    def forward(ctx, dims, edge_logits, type_logits, g, x, sizes, mu, logvar, beta):
        out4, grads = elbo_call(dims, edge_logits.contiguous(), type_logits.contiguous(), g, x, sizes,
                                mu.contiguous(), logvar.contiguous(), beta, True)
        ctx.save_for_backward(*grads)
        ctx.beta = beta
        return out4

    @staticmethod
    def backward(ctx, d_out4):
        ge, gt, gm, gl = ctx.saved_tensors
        # saved: d loss / d(.) with loss = edge + vertex + beta*kl, so each input's
        # gradient comes from exactly one part
        d = d_out4
        sk = d[3] + (d[2] / ctx.beta if ctx.beta != 0 else 0.0)
        return None, ge * (d[0] + d[3]), gt * (d[1] + d[3]), None, None, None, gm * sk, gl * sk, None
