"""`python bench.py --gpus N` with no launcher environment starts its own ranks (a child torch.distributed.run, before the
parent has touched the GPU), relays rank 0's JSON line and the ranks' exit status -- the way the reference's drivers are
self-contained under mpirun (apps/3d/steady.cpp:74-78)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(args, env_extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                          timeout=900)


def test_self_launch_relays_failure_without_gpu():
    """no GPU here: both child ranks refuse to run (no CPU fallback) and the parent must exit non-zero, printing nothing on
    stdout"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("CPU-only check")
    p = run_bench(["--gpus", "2", "--steps", "1", "--warmup", "1", "--size", "64"], {})
    assert p.returncode != 0
    assert p.stdout.strip() == ""
    assert "torch.distributed.run" in p.stderr and "needs a GPU" in p.stderr


def test_mismatched_world_size_is_refused():
    p = run_bench(["--gpus", "2"], {"WORLD_SIZE": "4", "RANK": "0"})
    assert p.returncode != 0 and "WORLD_SIZE=4" in p.stderr


@pytest.mark.gpu
def test_self_launch_two_ranks_gloo_rehearsal():
    """two ranks on the one GPU of the test box, gloo for the exchanges: one JSON line, n_gpus 2"""
    p = run_bench(["--gpus", "2", "--steps", "3", "--warmup", "2", "--size", "128", "--no-cpu-baseline"],
                  {"TE_BENCH_BACKEND": "gloo"})
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["value"] > 0
    assert out["config"]["parallelism"].startswith("patch-sharded x2")
    assert out["config"]["residual_reduction_per_cycle"] < 0.5
