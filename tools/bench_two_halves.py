"""Experiment: one 4096-graph step (two streams: chain + side) vs two 2048-graph half-steps, each a
single serial stream, running side by side (phase-shifted by the host's enqueue time)."""
import os, sys, time
from argparse import Namespace
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dvae_amd
from dvae_amd import engine as E
dev = torch.device('cuda')
model = dvae_amd.Dvae(Namespace(dataset_file=None, mod=None, bidirectional=False)).to(dev)
enc, dec = model.encoder, model.decoder
g, x = dvae_amd.synth.enas_like(4096, 1)
full = {'graph': g.to(dev), 'node_encoding': x.to(dev)}
halves = [{'graph': g[i:i + 2048].to(dev), 'node_encoding': x[i:i + 2048].to(dev)} for i in (0, 2048)]
s_full = model._schedule(full)
s_half = [model._schedule(h) for h in halves]
eps_half = [torch.randn(2048, 56, device=dev) for _ in range(2)]
streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
flat = [(torch.zeros_like(model._flat_grads()[0]), torch.zeros_like(model._flat_grads()[1])) for _ in range(2)]
torch.cuda.synchronize()

def half_step(i):
    batch, sched = halves[i], s_half[i]
    xs = E._f32c(batch['node_encoding'])
    d_enc = E._dims_for(enc._dims, sched.B, sched.N)
    d_dec = E._dims_for(dec._dims, sched.B, sched.N)
    Pe = enc._pack.pack(d_enc, enc.hip_params())
    Pd = dec._pack.pack(d_dec, dec.hip_params())
    ec = E.EncoderCall(d_enc, sched, xs, Pe)
    z = E.reparam_fwd(ec.mu, ec.logvar, eps_half[i])
    dc = E.DecoderCall(d_dec, sched, z, xs, Pd, side=None)
    out4, (ge, gt, gmu, glv) = E.elbo_call(d_dec, dc.edge_logits, dc.type_logits, sched.g, xs, sched.sizes,
                                           ec.mu, ec.logvar, 0.005, True)
    Gd = dec._pack.new_packed_grad(); Ge = enc._pack.new_packed_grad()
    d_z = dc.backward(ge, gt, Gd, side=None)
    dec._pack.unpack(Gd, flat[i][1])
    dmu, dlv = E.reparam_bwd(d_z, ec.logvar, eps_half[i])
    gmu.add_(dmu); glv.add_(dlv)
    ec.backward(None, gmu, glv, Ge, side=None)
    enc._pack.unpack(Ge, flat[i][0])
    return out4

def two():
    with torch.no_grad():
        main = torch.cuda.current_stream()
        for i in range(2):
            streams[i].wait_stream(main)
            with torch.cuda.stream(streams[i]):
                half_step(i)
        for i in range(2):
            main.wait_stream(streams[i])

def one_serial_half():
    with torch.no_grad():
        half_step(0)

def one():
    model.loss_and_grad(full, sched=None if True else s_full)

def timed(fn, reps=15):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3

for r in range(2):
    print('full step (chain + side)        %.3f ms' % timed(one), flush=True)
    print('two serial half-steps, 2 streams %.3f ms' % timed(two), flush=True)
    print('one serial half-step alone       %.3f ms' % timed(one_serial_half), flush=True)
