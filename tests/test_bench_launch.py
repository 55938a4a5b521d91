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


def json_lines(stdout):
    return [json.loads(ln) for ln in stdout.splitlines() if ln.strip().startswith("{")]


def test_self_launch_relays_failure_without_gpu():
    """no GPU here: both child ranks refuse to run (no CPU fallback); the parent must exit non-zero and stdout must carry ONE
    JSON line whose `error` says why and whose value is null (every failure path ends in a line a driver can record)"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("CPU-only check")
    p = run_bench(["--gpus", "2", "--steps", "1", "--warmup", "1", "--size", "64"], {})
    assert p.returncode != 0
    lines = json_lines(p.stdout)
    assert len(lines) == 1 and lines[0]["value"] is None and lines[0]["n_gpus"] == 2, p.stdout
    assert "needs a GPU" in lines[0]["error"]
    assert "torch.distributed.run" in p.stderr and "needs a GPU" in p.stderr


def test_mismatched_world_size_is_refused():
    p = run_bench(["--gpus", "2"], {"WORLD_SIZE": "4", "RANK": "0"})
    assert p.returncode != 0 and "WORLD_SIZE=4" in p.stderr


@pytest.mark.parametrize("die", ["at-once", "after-provisional"])
def test_supervisor_prints_a_line_when_the_worker_dies(die):
    """N > 1: each rank's process supervises the real rank (its child). A worker that ends like the exchange watchdog does
    (exit 86, nothing on stdout) leaves rank 0's supervisor to print the line: value null and the reason; or, when the worker had
    finished its headline (PROVISIONAL) and died in an optional block behind it, that line with `error` naming the block."""
    env = {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0", "TE_BENCH_TEST_DIE": die}
    p = run_bench(["--gpus", "2", "--size", "64"], env)
    assert p.returncode == 86
    lines = json_lines(p.stdout)
    assert len(lines) == 1, p.stdout
    assert "exit 86" in lines[0]["error"] and lines[0]["n_gpus"] == 2
    if die == "after-provisional":
        assert lines[0]["value"] == 1.0 and "a test block" in lines[0]["error"] and "next_block" not in lines[0]
    else:
        assert lines[0]["value"] is None
    q = run_bench(["--gpus", "2", "--size", "64"], {**env, "RANK": "1", "LOCAL_RANK": "1"})  # (other ranks: the status, no line)
    assert q.returncode == 86 and json_lines(q.stdout) == []


def test_worker_killed_inside_the_direct_store_block_keeps_the_headline():
    """the one block of the N > 1 line that has never met two GPUs is secondary.direct_store: a worker that is KILLED in it (SIGKILL: no
    handler runs, nothing more is printed) leaves rank 0's supervisor with the provisional line -- the measured headline -- and the
    error names the block and the signal"""
    env = {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0", "TE_BENCH_TEST_DIE": "killed-in-direct-store"}
    p = run_bench(["--gpus", "2", "--size", "64"], env)
    assert p.returncode != 0
    lines = json_lines(p.stdout)
    assert len(lines) == 1, p.stdout
    assert lines[0]["value"] == 1.0 and "secondary.direct_store" in lines[0]["error"] and "signal 9" in lines[0]["error"], lines[0]


def test_supervisor_prefers_the_provisional_line_to_a_value_null_last_line():
    """(round 5 advisor) a worker that has handed over its measured headline (PROVISIONAL) and then ends in an exception that is not the
    library's prints a value-null error line on its way out: rank 0's supervisor must keep the measured headline and put the
    exception beside it, not the other way round"""
    env = {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0", "TE_BENCH_TEST_DIE": "raise-after-provisional"}
    p = run_bench(["--gpus", "2", "--size", "64"], env)
    assert p.returncode != 0
    lines = json_lines(p.stdout)
    assert len(lines) == 1, p.stdout
    assert lines[0]["value"] == 1.0 and "ValueError" in lines[0]["error"] and "a test block" in lines[0]["error"], lines[0]


def test_in_process_headline_survives_any_exception_behind_it():
    """N = 1 (no supervisor, no provisional line on stdout): once the headline is built, an exception of ANY type in what follows
    ends in that line with `error`, not in a value-null line"""
    p = run_bench(["--gpus", "1", "--size", "64"], {"TE_BENCH_TEST_DIE": "raise-in-optional-block", "TE_BENCH_WORKER": "1"})
    assert p.returncode != 0
    lines = json_lines(p.stdout)
    assert len(lines) == 1 and lines[0]["value"] == 1.0 and "ValueError" in lines[0]["error"], p.stdout


@pytest.mark.gpu
def test_optional_block_that_raises_leaves_its_error_under_its_own_name():
    """on the GPU box, the real thing at N = 1: secondary.reference_smoother raises a ValueError; the line keeps its value, carries
    secondary.reference_smoother.error, and the blocks behind it (secondary.solve) still ran"""
    p = run_bench(["--gpus", "1", "--size", "128", "--steps", "3", "--warmup", "2", "--no-cpu-baseline"], {"TE_BENCH_TEST_RAISE": "reference_smoother"})
    assert p.returncode == 0, p.stderr[-3000:]
    out = json_lines(p.stdout)[-1]
    assert out["value"] > 0 and "error" not in out
    assert "ValueError" in out["secondary"]["reference_smoother"]["error"]
    assert out["secondary"]["solve"]["rbgs"]["iterations"] > 0


@pytest.mark.gpu
def test_self_launch_two_ranks_gloo_rehearsal():
    """two ranks on the one GPU of the test box, gloo for the exchanges: one JSON line, n_gpus 2 -- and the line proves what it
    computed: u_checksum_after_timed_region equals the N = 1 line's on the same workload (sharded == single rank, bit for bit),
    the other transport (direct stores between the two processes, hipIpc) reproduced the headline transport's result bit for
    bit (config.sharded.verified_bit_identical) and is reported as a SECOND figure, config.exchange names what carried `value`"""
    args = ["--steps", "3", "--warmup", "2", "--size", "128", "--no-cpu-baseline"]
    p = run_bench(["--gpus", "2"] + args, {"TE_BENCH_BACKEND": "gloo"})
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["value"] > 0 and "error" not in out
    assert out["config"]["parallelism"].startswith("patch-sharded x2")
    assert out["config"]["residual_reduction_per_cycle"] < 0.5
    assert out["config"]["exchange"] == "torch.distributed" and out["config"]["exchange_timeout_s"] <= 60
    one = run_bench(["--gpus", "1", "--no-secondary"] + args, {})
    assert one.returncode == 0, one.stderr[-3000:]
    single = json_lines(one.stdout)[-1]
    assert out["u_checksum_after_timed_region"] == single["u_checksum_after_timed_region"]
    ds = out["secondary"]["direct_store"]
    assert ds.get("verified_bit_identical") is True and ds["value"] > 0, ds
    assert ds["u_checksum"] == ds["u_checksum_headline_transport"] == out["u_checksum_after_timed_region"]
    assert out["config"]["sharded"]["verified_bit_identical"]["identical"] is True
    assert out["secondary"]["solve"]["rbgs"]["x_checksum"]
