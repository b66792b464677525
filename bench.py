#!/usr/bin/env python
"""Headline benchmark: DAGs/sec of a full D-VAE train step (level schedule,
encode, reparameterise, teacher-forced decode, ELBO, backward, gradient
all-reduce, Adam) on ENAS-like 8-node DAGs, hs=501, nz=56, global batch 4096.

    python bench.py --gpus N --steps K --warmup W [--config headline|c1|c2|c3|c4|c5]

N > 1: one rank per GPU over RCCL.  Either the caller wraps this file in
torch.distributed.run (WORLD_SIZE is then set), or -- when WORLD_SIZE is unset --
this process starts the N ranks itself, as children, BEFORE it touches the GPU, and
exits with their status (the reference's analogue is `Trainer(gpus=...)`, train.py:46).
Rank 0 prints ONE JSON line.  Every step sees a different pre-generated batch (already
resident in HBM), so the schedule build is inside the timed region, as it would be in
training.

--config selects BASELINE.json's other configurations; the default is
the configuration the metric is quoted on.
"""
import argparse
import ctypes
import json
import math
import os
import socket
import subprocess
import sys
import time
from argparse import Namespace

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

H, Z = 501, 56
PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD
PEAK_BF16_MFMA_TFLOPS = 2500.0   # MI355X_MICROARCH.md: dense bf16, v_mfma_f32_32x32x16_bf16 at 32 cycles
# the default GEMM core forms every fp32 product from SIX bf16 MFMAs on exactly split operands (gemm.hip, "x6"):
# its roof in fp32-equivalent FLOP/s is the bf16 peak / 6
PEAK_X6_TFLOPS = PEAK_BF16_MFMA_TFLOPS / 6.0
PEAK_HBM_TBS = 8.0               # MI355X_MICROARCH.md: HBM3E
PROF_EVERY = 8                   # timed steps between two steps whose GEMM launches carry HIP events (4 in runs of fewer than 16 steps)

# BASELINE.json `configs` (c1..c5) + the headline shape of `metric`
CONFIGS = {
    "headline": dict(kind="train", mod=None, bidir=False, N=8, C=8, B=4096,
                     workload="ENAS-like 8-node DAGs (chain + p=1/3), nvt=8, hs=501, nz=56, uni-directional, "
                              "mod=None, fwd+bwd+all-reduce+Adam",
                     metric="DAGs/sec D-VAE train step (ENAS-8node, hs=501, batch 4096)"),
    "c1": dict(kind="train", mod=None, bidir=False, N=8, C=8, B=32,
               workload="config 1: ENAS-like 8-node DAGs, batch 32 (the reference's CPU-runnable case)",
               metric="DAGs/sec D-VAE train step (ENAS-8node, hs=501, batch 32)"),
    "c2": dict(kind="train", mod="bn", bidir=True, N=10, C=10, B=1024,
               workload="config 2: Bayesian-network mode (mod='bn', bidirectional, train.py:18-28), 8 variables + SOS/EOS "
                        "= 10 nodes / 10 types, p=0.3, batch 1024",
               metric="DAGs/sec D-VAE train step (bn 10-node, hs=501, batch 1024)"),
    "c3": dict(kind="train", mod=None, bidir=False, N=64, C=8, B=4096,
               workload="config 3: random 64-node DAGs (p=0.1, orphans -> node 0), uni-directional, batch 4096",
               metric="DAGs/sec D-VAE train step (64-node random DAGs, hs=501, batch 4096)"),
    "c4": dict(kind="train", mod=None, bidir=True, N=128, C=8, B=512,
               workload="config 4: bidirectional encoder, ragged DAGs with 8..128 nodes (p=min(0.5,4/n)), batch 512",
               metric="DAGs/sec D-VAE train step (ragged 8-128-node DAGs, bidirectional, hs=501, batch 512)"),
    "c5": dict(kind="decode", mod=None, bidir=False, N=8, C=8, B=100000,
               workload="config 5: greedy Decoder.predict of 100k latent samples z~N(0,I) to 8-node DAGs",
               metric="DAGs/sec D-VAE decode (100k latents, 8-node, hs=501)"),
    "c5gen": dict(kind="generate", mod=None, bidir=False, N=8, C=8, B=100000,
                  workload="config 5, evaluation path: decode 100k latent samples z~N(0,I) and score the decoded DAGs on the "
                           "device -- validity, uniqueness, novelty against a 4096-graph training batch (Dvae.generation_eval; "
                           "the reference implements only exact-match reconstruction accuracy, SURVEY.md 8 f2)",
                  metric="DAGs/sec D-VAE decode + validity/uniqueness/novelty (100k latents, 8-node, hs=501)"),
}


def algorithmic_flops(n, e, pairs, mod=None, bidir=False, c=8):
    """SURVEY.md 8(d), summed over graphs (n = nodes, e = edges, pairs = sum n_b(n_b-1)/2, all per graph
    on average): forward H^2(34N + 14E) + 12H pairs + 4HZ + 2ZH + 4HCN for mod=None; train step = 3 x
    forward.  Generalised by the same accounting to the other modes:
      per node:  encoder sweep + decoder sweep (GRU hidden 6H^2 [+ gate/map 4H^2, mod None only]),
                 add_vertex 4H^2, W1b 2*H*EH;  per distinct suffix (E+N): W1a 2*H*EH;  per extra suffix
                 (E-N): GRU 6H^2;  bn: W1c 2*H*EH per graph;  bidirectional: reverse sweep + lin01 4H^2 per node;
      EH = 4H (None) / 6H (bn)."""
    eh = 6 * H if mod == "bn" else 4 * H
    sweep = 6 * H * H if mod == "bn" else 10 * H * H
    fwd = n * (2 * sweep + 4 * H * H + 2 * H * eh) + (e + n) * 2 * H * eh + (e - n) * 6 * H * H
    fwd += 3 * eh * pairs + 4 * H * Z + 2 * Z * H + 4 * H * c * n
    if mod == "bn":
        fwd += 2 * H * eh
    if bidir:
        fwd += n * (sweep + 4 * H * H)
    return 3.0 * fwd


def decode_flops(n):
    """SURVEY.md 8(d) decode: H^2(10N + 4N + 8N + 14 N(N-1)/2) per decoded DAG."""
    return H * H * (22 * n + 14 * n * (n - 1) / 2)


def host_cores():
    """(threads to use, physical cores of the box).  The GPU box gives one job a 16-CPU share of a much
    larger machine (OMP_NUM_THREADS=16 there); the baseline runs on that share and says so."""
    phys = set()
    try:
        pid = cid = None
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("physical id"):
                pid = ln.split(":")[1].strip()
            elif ln.startswith("core id"):
                cid = ln.split(":")[1].strip()
            elif not ln.strip():
                if pid is not None and cid is not None:
                    phys.add((pid, cid))
                pid = cid = None
    except OSError:
        pass
    nphys = len(phys) or (os.cpu_count() or 1)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    share = int(os.environ.get("OMP_NUM_THREADS", "0")) or 16
    return max(1, min(share, avail, nphys)), nphys


def cpu_baseline(cfg_name, seconds_budget=20.0, stochastic=False):
    """The reference's CPU path, restated (oracle/dense_ref.py, checked against the real reference in
    tests/golden), timed on this box's host cores on a bounded sample of the configuration."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import dvae_amd
    from oracle import dense_ref as O
    threads, nphys = host_cores()
    torch.set_num_threads(threads)
    cfg = CONFIGS[cfg_name]
    S = dvae_amd.synth
    N, C, mod, bidir = cfg["N"], cfg["C"], cfg["mod"], cfg["bidir"]
    note = ""
    if cfg["kind"] in ("decode", "generate"):
        model = dvae_amd.Dvae(Namespace(dataset_file=None, mod=mod, bidirectional=False, num_classes=C, max_seq=N))
        sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
        M = 256
        z = torch.randn(M, Z, generator=torch.Generator().manual_seed(0))
        uni = None
        if stochastic:     # the reference draws with torch.multinomial / torch.bernoulli; the port takes uniforms
            gu = torch.Generator().manual_seed(1)
            uni = {'type': torch.rand(M, N, generator=gu), 'edge': torch.rand(M, N, N, generator=gu)}
        score = None
        if cfg["kind"] == "generate":      # the same report on the host: metrics.py on CPU tensors
            from dvae_amd import metrics
            tg, tx = S.enas_like(4096, 4242)
            score = lambda e, t: metrics.generation_report(e, t, tg, tx, 'enas')
        with torch.no_grad():
            O.decoder_predict(sd, z[:8], N, C, mod)
            t0 = time.perf_counter()
            n = 0
            while n < 2 or (time.perf_counter() - t0 < seconds_budget and n < 50):
                e, t = O.decoder_predict(sd, z, N, C, mod, uniforms=uni)[:2]
                if score is not None:
                    score(e, t)
                n += 1
        dt = (time.perf_counter() - t0) / n
        return {"value": M / dt, "unit": "DAGs/s", "cores": threads, "host_physical_cores": nphys, "kind": "port",
                "sample": "%d %s decodes of %d latents by oracle/dense_ref.py decoder_predict%s, %.0f ms each"
                          % (n, "stochastic (injected uniforms)" if stochastic else "greedy", M,
                             " + metrics.generation_report on the host" if score is not None else "", dt * 1e3)}
    # dense O(B N^3 H) autograd memory: micro-batches beyond config 1
    if cfg_name in ("headline", "c1"):
        B, N_run, make = 32, N, lambda: S.enas_like(32, 1234)
        note = "B=32, N=8, C=8 (config 1)"
    elif cfg_name == "c2":
        B, N_run, make = 128, N, lambda: S.random_dags(128, N, C, 0.3, 1234, fix_orphans=False)
        note = "micro-batch B=128, bn, bidirectional, N=C=10"
    elif cfg_name == "c3":
        B, N_run, make = 1, 64, lambda: S.random_dags(1, 64, C, 0.1, 1234)
        note = "micro-batch B=1 at N=64 (the dense path is O(N^3) per graph)"
    else:   # c4: one mid-size graph of the ragged mix (mean size 68), bidirectional
        B, N_run, make = 1, 68, lambda: S.random_dags(1, 68, C, 4.0 / 68, 1234)
        note = "micro-batch B=1 at N=68 (the mean size of the ragged mix), bidirectional"
    model = dvae_amd.Dvae(Namespace(dataset_file=None, mod=mod, bidirectional=bidir, num_classes=C, max_seq=N_run))
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    opt = torch.optim.Adam(list(sd.values()), lr=1e-4)
    g, x = make()
    gen = torch.Generator().manual_seed(0)

    def step():
        eps = torch.randn(B, Z, generator=gen)
        opt.zero_grad()
        loss = O.train_loss(sd, g, x, eps, mod, bidir)[0]
        loss.backward()
        opt.step()

    heavy = N_run > 16
    if not heavy:
        step()
    t0 = time.perf_counter()
    n = 0
    while n < (1 if heavy else 5) or (time.perf_counter() - t0 < seconds_budget and n < 200):
        step()
        n += 1
    dt = (time.perf_counter() - t0) / n
    return {"value": B / dt, "unit": "DAGs/s", "cores": threads, "host_physical_cores": nphys, "kind": "port",
            "sample": "%d train steps (fwd+bwd+Adam) of oracle/dense_ref.py, %s, hs=501, %.0f ms/step"
                      % (n, note, dt * 1e3)}


def kept_pmc_traffic(cfg_name):
    """HBM bytes measured by the kept rocprofv3 counter passes (tools/pmc.sh; profiles/r03_pmc_traffic.json is written
    by tools/pmc_traffic.py from the same pass whose summary is profiles/r03_pmc_summary.txt): the GEMM's bytes per
    launch and the per-step bytes of the HBM-bound kernel families.  None when the file has no entry for the config."""
    path = os.path.join(ROOT, "profiles", "r03_pmc_traffic.json")
    try:
        with open(path) as f:
            d = json.load(f)
    except (OSError, ValueError):
        return None
    e = d.get(cfg_name)
    if not e:
        return None
    e = dict(e)
    e["source"] = "profiles/r03_pmc_traffic.json (" + d.get("_how", "rocprofv3 --pmc, separate passes") + ")"
    return e


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(args):
    """WORLD_SIZE unset and --gpus N > 1: start the N ranks as children of this process, which has not
    touched the GPU (and never will); their stdout -- rank 0's JSON line -- is ours."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def make_batches(cfg, B_global, rank, world, dev, nb):
    import dvae_amd
    from dvae_amd import dp
    S = dvae_amd.synth
    N, C = cfg["N"], cfg["C"]
    batches, stats = [], []
    for i in range(nb):
        seed = 100 + i
        if cfg["N"] == 8:
            g, x = S.enas_like(B_global, seed)
            full = {'graph': g, 'node_encoding': x}
        elif cfg["mod"] == "bn":
            g, x = S.random_dags(B_global, N, C, 0.3, seed, fix_orphans=False)
            full = {'graph': g, 'node_encoding': x}
        elif cfg["bidir"]:
            g, x, sizes = S.ragged_dags(B_global, C, seed)
            full = {'graph': g, 'node_encoding': x, 'sizes': sizes}
        else:
            g, x = S.random_dags(B_global, N, C, 0.1, seed)
            full = {'graph': g, 'node_encoding': x}
        mine = dp.shard_batch(full, rank, world)
        gb = mine['graph']
        nodes = mine['sizes'].double() if 'sizes' in mine else torch.full((gb.shape[0],), float(N), dtype=torch.float64)
        stats.append((gb.shape[0], float(nodes.sum()), float(gb.sum()), float((nodes * (nodes - 1) / 2).sum())))
        batches.append({k: v.to(dev) for k, v in mine.items()})
    return batches, stats


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--config", default="headline", choices=sorted(CONFIGS))
    ap.add_argument("--global-batch", type=int, default=0, help="override the configuration's batch")
    ap.add_argument("--per-gpu-batch", type=int, default=0, help="weak scaling: fixed batch per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prefetch", action="store_true", help="build each schedule at the start of its own step")
    ap.add_argument("--no-profile", action="store_true", help="do not bracket GEMM launches with HIP events")
    ap.add_argument("--stochastic", action="store_true", help="c5: stochastic decode (device uniforms)")
    ap.add_argument("--split-optimizer", action="store_true",
                    help="lab: the decoder's Adam as an instance of its own, stepped on the side stream as soon as its "
                         "gradients are final (under the encoder's backward); measured 0.07-0.1 ms SLOWER than one "
                         "fused Adam after the step at B = 512 ... 4096, so it is not the default")
    ap.add_argument("--optimizer", default="flat", choices=["flat", "torch"],
                    help="Adam(lr=1e-4) of the train step: dvae_amd.optim.FlatAdam (one launch per module) or torch's fused Adam")
    ap.add_argument("--no-core-ab", action="store_true", help="N = 1: skip the few steps on the fp32-MFMA GEMM core after the timed region")
    ap.add_argument("--no-weak-probe", action="store_true", help="N > 1: skip the fixed-per-GPU-batch probe after the timed region")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher rehearsal without a GPU: ranks rendezvous over gloo, rank 0 prints the line")
    args = ap.parse_args()
    cfg = CONFIGS[args.config]
    if args.steps is None:
        # (headline: 20 timed steps after 3 warm-up steps read 1.5 % slower than 100 after 10 on the same box -- the first steps of a
        # process are not yet at the steady rate; profiles/r03_level_gemm.txt)
        args.steps = {"headline": 60, "c1": 100, "c2": 40, "c3": 5, "c4": 8, "c5": 3, "c5gen": 3}[args.config]

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, "WORLD_SIZE=%d but --gpus %d" % (world, args.gpus)
    if args.dry_run:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        t = torch.ones(1)
        dist.all_reduce(t)
        if rank == 0:
            print(json.dumps({"metric": cfg["metric"], "dry_run": True, "n_gpus": world, "backend": "gloo",
                              "rccl_ranks": dist.get_world_size(), "allreduce_sum": float(t)}), flush=True)
        dist.barrier()
        dist.destroy_process_group()
        return

    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    force_dist = bool(os.environ.get("DVAE_FORCE_DIST"))    # exercise the RCCL path on one GPU (rehearsal)
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import dvae_amd
    from dvae_amd import _lib, dp

    weak = args.per_gpu_batch > 0
    B_cfg = args.global_batch or cfg["B"]
    B_global = args.per_gpu_batch * world if weak else B_cfg
    lo, hi = dp.shard_range(B_global, rank, world)
    B_local = hi - lo
    N, C = cfg["N"], cfg["C"]

    torch.manual_seed(0)
    model = dvae_amd.Dvae(Namespace(dataset_file=None, mod=cfg["mod"], bidirectional=cfg["bidir"],
                                    num_classes=C, max_seq=N)).to(dev)
    if dist is not None:
        for p in model.parameters():
            dist.broadcast(p.data, 0)
    lib = _lib.load()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if cfg["kind"] in ("decode", "generate"):
        # embarrassingly parallel: every rank decodes its contiguous shard of the latents, no collective
        z = torch.randn(B_global, Z, generator=torch.Generator().manual_seed(0))[lo:hi].to(dev)
        mean_n, mean_e, mean_pairs = float(N), 0.0, N * (N - 1) / 2
        out = None
        report = {}
        if cfg["kind"] == "generate":
            from dvae_amd import metrics
            tg, tx = dvae_amd.synth.enas_like(4096, 4242)
            tg, tx = tg.to(dev), tx.to(dev)

        def step(i):
            e, t = model.decoder.predict(z, is_stochastic=args.stochastic)
            if cfg["kind"] == "generate":     # validity / uniqueness / novelty of this rank's decodes, on the device
                report.update(metrics.generation_report(e, t, tg, tx, 'enas'))
            return e, t
    else:
        # Adam(lr=1e-4) on every parameter each step (dvae_model.py:471-475): one fused instance
        if args.optimizer == "flat" and not args.split_optimizer:
            # the same Adam update as ONE launch per module over flat buffers (dvae_adam_step, optim.FlatAdam)
            opt, opt_dec = dvae_amd.optim.FlatAdam(model, lr=1e-4), None
        elif not args.split_optimizer:
            opt, opt_dec = torch.optim.Adam(model.parameters(), lr=1e-4, fused=True), None
        else:
            opt = torch.optim.Adam(model.encoder.parameters(), lr=1e-4, fused=True)
            opt_dec = torch.optim.Adam(model.decoder.parameters(), lr=1e-4, fused=True)
        dec_step = opt_dec.step if opt_dec is not None else None
        nb = 4 if cfg["N"] <= 16 else 2      # a few distinct batches per rank, generated once, resident in HBM
        batches, stats = make_batches(cfg, B_global, rank, world, dev, nb)
        cnt = sum(s[0] for s in stats)
        mean_n, mean_e, mean_pairs = (sum(s[k] for s in stats) / cnt for k in (1, 2, 3))
        sync_grads = dp.make_grad_sync(dist, force=force_dist)   # SUM all-reduce: the losses are sums (dvae_model.py:405)

        # the level schedule of batch i+1 is queued (own stream) as soon as step i has been
        # launched, as a prefetching data loader would: every step still builds one schedule
        # inside the timed region, but nobody waits for its host round trip
        # (multi-GPU runs queue it on the library's side stream at the start of the step: with RCCL's
        # streams a prefetch stream would be the fifth stream of the process and the box gives a process
        # 4 hardware queues)
        prepared = {}
        prefetch = not args.no_prefetch
        on_side = dist is not None     # with RCCL in the process: no third stream, the side stream builds it

        def step(i):
            nb = len(batches)
            sched = prepared.pop(i, None)
            if prefetch and on_side:   # queued ahead of this step's own side-stream work (idle until the decoder starts)
                prepared[i + 1] = model.prepare(batches[(i + 1) % nb], resident=True, on_side_stream=True)
            out4 = model.loss_and_grad(batches[i % nb], grad_sync=sync_grads, sched=sched, on_decoder_grads=dec_step)
            if prefetch and not on_side:
                prepared[i + 1] = model.prepare(batches[(i + 1) % nb], resident=True)
            opt.step()
            return out4

    for i in range(args.warmup):
        step(i)
    barrier()
    # per-launch HIP events (roofline figures, an N=1 report) on every PROF_EVERY-th timed step: two
    # event records around each of ~110 launches cost 0.44 ms on a 12.4 ms step when every step has them
    prof = not args.no_profile and world == 1
    prof_steps = 0
    prof_every = PROF_EVERY if args.steps >= 16 else 4
    t0 = time.perf_counter()
    for i in range(args.steps):
        sample = prof and i % prof_every == 0
        if sample:
            lib.dvae_gemm_profile(2 if prof_steps else 1)
            prof_steps += 1
        out = step(args.warmup + i)
        if sample:
            lib.dvae_gemm_profile(0)
    barrier()
    dt = time.perf_counter() - t0
    gemm = hbm = None
    if prof:
        buf = (ctypes.c_double * 12)()
        _lib.check(lib.dvae_hbm_profile_read(ctypes.cast(buf, ctypes.c_void_p)))    # before the GEMM read, which clears the log
        hbm = list(buf)
        _lib.check(lib.dvae_gemm_profile_read(ctypes.cast(buf, ctypes.c_void_p)))
        gemm = list(buf)
    # N = 1: the same step on the OTHER arithmetic core of the GEMM (v_mfma_f32_32x32x2_f32 instead of the default
    # six bf16 MFMAs on exactly split fp32 operands), a few steps right after the timed region, reported beside `value`
    fp32_core = None
    if world == 1 and lib.dvae_gemm_core() and not args.no_core_ab:
        lib.dvae_gemm_x6(0)
        for i in range(2):
            step(5000 + i)
        barrier()
        n2 = max(3, min(10, args.steps))
        t2 = time.perf_counter()
        for i in range(n2):
            step(5002 + i)
        barrier()
        t2 = time.perf_counter() - t2
        fp32_core = {"core": "v_mfma_f32_32x32x2_f32", "steps": n2, "ms_per_step": t2 / n2 * 1e3,
                     "value": B_global * n2 / t2, "unit": "DAGs/s"}
        if cfg["kind"] == "train":
            # the same batch, the same eps, both cores: how far the default core's losses and gradients are from the
            # fp32 MFMA's (fp32 rounding noise; the parity tests hold both against the reference's golden vectors)
            eps0 = torch.randn(batches[0]['graph'].shape[0], Z, device=dev)
            ref4 = model.loss_and_grad(batches[0], eps=eps0).double().clone()
            refg = torch.cat([p.grad.flatten() for p in model.parameters()]).double().clone()
            lib.dvae_gemm_x6(1)
            got4 = model.loss_and_grad(batches[0], eps=eps0).double()
            gotg = torch.cat([p.grad.flatten() for p in model.parameters()]).double()
            fp32_core["default_core_vs_this_core"] = {
                "losses_max_rel_diff": float(((got4 - ref4).abs() / ref4.abs().clamp_min(1e-30)).max()),
                "gradient_rel_l2_diff": float((gotg - refg).norm() / refg.norm().clamp_min(1e-30)),
                "gradient_max_abs_diff_over_max_abs": float((gotg - refg).abs().max() / refg.abs().max().clamp_min(1e-30)),
                "note": "fp32 rounding noise: a pre-activation that moves by 1e-7 across add_vertex's / add_edge's ReLU flips one "
                        "unit's gradient term, so two summation orders of the SAME core differ by as much "
                        "(tests/test_gpu_variants.py); both cores meet the reference's golden vectors at rtol 1e-4"}
        lib.dvae_gemm_x6(1)
    rccl_ranks = None
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        rccl_ranks = dist.get_world_size()
    # N > 1: the same step at a FIXED per-GPU batch (= the configuration's batch on every GPU) right after the
    # timed region -- the weak-scaling point beside the strong-scaling `value` (the global batch of the metric
    # leaves each of 8 GPUs a 512-graph shard, which is launch-bound; this shows what the exchange itself costs)
    weak_probe = None
    if cfg["kind"] == "train" and dist is not None and not weak and cfg["N"] <= 16 and not args.no_weak_probe:
        wb, _ = make_batches(cfg, B_cfg * world, rank, world, dev, 2)
        prepared.clear()
        batches[:] = wb
        nb = len(wb)
        for i in range(2):
            step(1000 + i)
        barrier()
        wsteps = max(4, min(10, args.steps))
        tw = time.perf_counter()
        for i in range(wsteps):
            step(1002 + i)
        barrier()
        tw = torch.tensor([time.perf_counter() - tw], device=dev, dtype=torch.float64)
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
        weak_probe = {"per_gpu_batch": B_cfg, "global_batch": B_cfg * world, "steps": wsteps,
                      "ms_per_step": float(tw) / wsteps * 1e3, "value": B_cfg * world * wsteps / float(tw),
                      "unit": "DAGs/s", "scaling": "weak"}
    if cfg["kind"] == "train":
        final = float(out[3])
        assert math.isfinite(final), "loss is not finite"
    else:
        final = float(out[0].sum()) / B_local        # mean edges per decoded DAG
        mean_e = final

    if rank == 0:
        ms = dt / args.steps * 1e3
        value = B_global * args.steps / dt
        res = {
            "metric": cfg["metric"] if B_global == cfg["B"] else cfg["metric"] + " [batch %d]" % B_global,
            "value": value, "unit": "DAGs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms, "higher_is_better": True, "scaling": "weak" if weak else "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": cfg["workload"], "name": args.config,
                       "global_batch": B_global, "per_gpu_batch": B_local, "mean_nodes": round(mean_n, 2),
                       "mean_edges": round(mean_e, 2), "parallelism": "dp%d" % world},
        }
        if cfg["kind"] != "train":
            res["config"]["decode"] = "stochastic (device uniforms: inverse-CDF type draw, u < sigmoid edge draw)" if args.stochastic else "greedy"
        if cfg["kind"] == "generate":
            res["config"]["report"] = {k: (round(float(v), 6) if isinstance(v, (int, float)) else v) for k, v in report.items()}
        if cfg["kind"] == "train":
            res["config"]["final_loss"] = final
            res["config"]["optimizer"] = ("Adam(lr=1e-4), dvae_amd.optim.FlatAdam (torch.optim.Adam's update, one launch per module)"
                                          if args.optimizer == "flat" and not args.split_optimizer else "torch.optim.Adam(lr=1e-4, fused=True)")
            last = getattr(model, "_step_keep", None)
            if last is not None:
                # the encoder's forward sweep runs on one row per distinct node VALUE (index, type, predecessor values) of the
                # batch -- exact common-subexpression sharing, every graph's outputs unchanged; the algorithmic
                # FLOP count above still charges one row per node, as the reference computes
                sr = last[0].shared_rows
                res["config"]["encoder_rows"] = {"nodes": last[0].R, "value_rows": sr[0] if sr else last[0].R,
                                                 "shared": sr is not None}
            if cfg["N"] <= 16:
                g0 = batches[0]['graph'][:256].cpu()
                res["config"]["depth"] = round(dvae_amd.synth.graph_stats(g0)[1], 2)
        if rccl_ranks is not None:
            res["rccl_ranks"] = rccl_ranks
        if weak_probe is not None:
            res["weak_scaling_probe"] = weak_probe
        res["gemm_core"] = ("bf16 x6: fp32 operands split exactly into three bf16 pieces, six v_mfma_f32_32x32x16_bf16 per "
                            "product, fp32 accumulation; error against fp64 no larger than the fp32 MFMA's "
                            "(tests/test_gpu_kernels.py::test_gemm_x6_is_fp32_accurate)" if lib.dvae_gemm_core()
                            else "fp32 MFMA (v_mfma_f32_32x32x2_f32)")
        if fp32_core is not None:
            res["same_step_on_fp32_mfma_core"] = fp32_core
        if gemm is not None:
            # dominant kernel: gemm_f32_kernel (all three layouts are one template)
            launches = gemm[0] + gemm[3] + gemm[9]
            kms = gemm[1] + gemm[4] + gemm[10]
            executed = gemm[2] + gemm[5] + gemm[11]
            if cfg["kind"] == "train":
                per_dag = algorithmic_flops(mean_n, mean_e, mean_pairs, cfg["mod"], cfg["bidir"], C)
            else:
                per_dag = decode_flops(N)      # the metrics of the evaluation path are integer work on 0/1 tensors
            algo = per_dag * B_local * prof_steps
            busy = gemm[6] if gemm[6] > 0 else kms     # wall time with >= 1 GEMM launch running
            ach = algo / (busy * 1e-3) / 1e12 if busy > 0 else 0.0
            x6 = bool(lib.dvae_gemm_core())
            peak = PEAK_X6_TFLOPS if x6 else PEAK_F32_MFMA_TFLOPS
            res["roofline"] = {
                "bound": "mfma",
                "kernel": "gemm_f32_kernel, x6 core (fp32 operands split exactly into 3 bf16 pieces, 6 x v_mfma_f32_32x32x16_bf16 "
                          "per product, fp32 accumulate)" if x6 else "gemm_f32_kernel (v_mfma_f32_32x32x2_f32)",
                "achieved": ach, "peak": peak, "unit": "TFLOP/s",
                "frac": ach / peak, "traffic": None,
                "peak_is": "dense bf16 MFMA peak 2500 TFLOP/s / 6 bf16 products per fp32 product" if x6
                           else "dense fp32 MFMA peak",
                "frac_of_fp32_mfma_peak": ach / PEAK_F32_MFMA_TFLOPS,
                "launches_per_step": launches / prof_steps, "kernel_busy_ms_per_step": busy / prof_steps,
                "sum_launch_ms_per_step": kms / prof_steps, "timed_steps_with_events": prof_steps,
                "avg_launch_us": kms / max(launches, 1) * 1e3,
                "executed_tflops": executed / (busy * 1e-3) / 1e12 if busy > 0 else 0.0,
                "whole_step_tflops": per_dag * B_local / (ms * 1e-3) / 1e12,
                "note": "HIP events around every launch of the kernel on every %d-th timed step; launches run on "
                        "two streams and overlap: achieved = algorithmic FLOPs / wall time with at least one "
                        "launch of the kernel running; avg_launch_us is the plain per-launch mean; "
                        "executed_tflops counts the FLOPs the launches really did" % prof_every,
                "algorithmic_gflop_per_dag": per_dag / 1e9,
            }
            # HBM bytes of the dominant kernel from the kept counter pass (rocprofv3 --pmc, separate passes, FETCH_SIZE
            # doubled per MI355X_MICROARCH.md's gfx950 note + WRITE_SIZE), per launch like `achieved`
            traffic = kept_pmc_traffic(args.config)
            if traffic is not None:
                res["roofline"]["traffic"] = traffic["gemm_bytes_per_launch"]
                res["roofline"]["traffic_source"] = traffic["source"]
            if hbm is not None and cfg["kind"] == "train":
                # the HBM-bound kernels north_star names, bracketed by the same sampled HIP events on their own stream:
                # achieved = ALGORITHMIC bytes (aggregation 4H(E + 2N) per sweep, pair scoring 16H(E + 2N)
                # + 4 per pair, backward 2 x) / measured launch time, against the 8 TB/s HBM peak.  They share the CUs with
                # the side stream's GEMMs while they run.
                names = ("pair scorer forward (pair_fwd*)", "pair scorer backward (pair_bwd*)",
                         "gated-sum aggregation forward (agg_fwd, prefix_gru_fwd)",
                         "aggregation backward: transpose gather (dgm_bwd, dgm_gather)")
                rows = []
                for k, name in enumerate(names):
                    n_l, ms_k, by = hbm[3 * k:3 * k + 3]
                    if n_l <= 0 or ms_k <= 0:
                        continue
                    tbps = by / (ms_k * 1e-3) / 1e12
                    row = {"kernel": name, "launches_per_step": n_l / prof_steps, "algorithmic_GB_per_step": by / prof_steps / 1e9,
                           "ms_per_step": ms_k / prof_steps, "TBps": tbps, "frac_of_8": tbps / PEAK_HBM_TBS}
                    if traffic is not None and traffic.get("hbm", {}).get(str(k)):
                        row["traffic_GB_per_step"] = traffic["hbm"][str(k)]
                    rows.append(row)
                res["roofline"]["hbm"] = rows
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(args.config, stochastic=args.stochastic)
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
