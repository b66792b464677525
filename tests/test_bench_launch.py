"""bench.py's own launcher (no GPU): `python bench.py --gpus 2` with WORLD_SIZE unset must start two
ranks itself (children of a process that never touched the GPU), rendezvous them, and print exactly
ONE JSON line from rank 0.  --dry-run swaps RCCL for gloo and skips the GPU work, so the launch path
is what runs here; the torchrun-wrapped form (WORLD_SIZE set by the caller) must keep working."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["OMP_NUM_THREADS"] = "1"
    return env


def _json_lines(out):
    return [json.loads(ln) for ln in out.splitlines() if ln.startswith("{")]


def test_bench_spawns_its_own_ranks():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"],
                       env=_env(), capture_output=True, text=True, timeout=240)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = _json_lines(p.stdout)
    assert len(lines) == 1, p.stdout
    assert lines[0]["n_gpus"] == 2 and lines[0]["rccl_ranks"] == 2 and lines[0]["allreduce_sum"] == 2.0


def test_bench_under_torchrun_still_works():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29533", os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--dry-run"]
    p = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=240)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = _json_lines(p.stdout)
    assert len(lines) == 1 and lines[0]["rccl_ranks"] == 2


def test_bench_flop_model_matches_survey():
    """0.353 GFLOP per ENAS-8 DAG (E=14) and 3.78 GFLOP per 64-node DAG (E=200)."""
    sys.path.insert(0, ROOT)
    import bench
    assert abs(bench.algorithmic_flops(8, 14, 28) / 1e9 - 0.353) < 2e-3
    assert abs(bench.algorithmic_flops(64, 200, 2016) / 1e9 - 3.78) < 3e-2
    assert abs(bench.decode_flops(8) / 1e9 - 0.14) < 5e-3
