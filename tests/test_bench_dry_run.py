"""bench.py's multi-GPU path without devices: `--gpus N --dry-run` builds the N
ranks' launch environments, checks both scaling modes' window partitions and assembles + validates the
record an N-rank run prints (synthetic timings, marked as such).  No GPU, no RCCL."""
import json
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(*extra):
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--dry-run"] + list(extra),
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, "ONE JSON line expected, got %d" % len(lines)
    return json.loads(lines[0])


@pytest.mark.parametrize("n", [1, 2, 4, 8])
def test_dry_run_record_schema(n):
    r = run("--gpus", str(n), "--steps", "5", "--warmup", "1")
    assert r["n_gpus"] == n and r["ranks"] == n and r["steps"] == 5 and r["warmup"] == 1
    assert r["data"].startswith("DRY RUN")
    assert r["scaling"] == "weak" and r["config"]["sequences"] == n
    assert r["vs_baseline"] is None and r["higher_is_better"] is True and r["dtype"] == "f64"
    rf = r["roofline"]
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic", "frac_events", "whole_step_frac",
                "traffic_ratio")) <= set(rf)
    assert 0.0 < rf["whole_step_frac"] < 1.0
    if n > 1:
        assert len(r["per_rank_ms_per_step"]) == n and r["ranks_source"] == "ncclCommCount"
        assert r["allreduce"]["bytes"] == (64 * 64 + 64 * 32 + 64 + 64 * 32 * 32 + 1) * 8
        for leg in ("epoch", "minibatch_s64"):
            assert r["strong_scaling"][leg]["scaling"] == "strong"
    else:
        assert "per_rank_ms_per_step" not in r and r["ranks_source"] == "no communicator"
    # the headline region spans >= 5 s of GPU time by default (a coarse utilisation sampler sees it)
    assert r["reps"] * r["steps"] * 3.1e-3 >= 4.8


def test_dry_run_strong_scaling_line():
    r = run("--gpus", "8", "--scaling", "strong")
    assert r["scaling"] == "strong" and r["config"]["sequences"] == 1
    assert "windows[rank::N]" in r["config"]["parallelism"]


def test_rank_environments():
    sys.path.insert(0, REPO)
    import bench
    envs = bench.rank_envs(8, 29511, base={"PATH": "/usr/bin"})
    assert [e["RANK"] for e in envs] == [str(i) for i in range(8)]
    assert all(e["LOCAL_RANK"] == e["RANK"] and e["WORLD_SIZE"] == "8" for e in envs)
    assert all(e["MASTER_ADDR"] == "127.0.0.1" and e["MASTER_PORT"] == "29511" for e in envs)
    assert all(e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and e["SVIHMM_BENCH_CHILD"] == "1" for e in envs)
    # a caller's own setting of the IPC mode is respected
    assert bench.rank_envs(1, 1, base={"HSA_ENABLE_IPC_MODE_LEGACY": "1"})[0]["HSA_ENABLE_IPC_MODE_LEGACY"] == "1"


def test_failed_rank_zero_still_prints_one_json_line():
    """VERDICT r4 next #5b: a multi-rank run whose rank 0 cannot come up (here: no HIP device in the CPU
    container, under a torchrun-style environment) leaves the error record, not silence."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999",
               SVIHMM_RENDEZVOUS_TIMEOUT="2")
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode != 0
    lines = [l for l in p.stdout.splitlines() if l.lstrip().startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["value"] is None and rec["n_gpus"] == 2 and "error" in rec


def test_self_launch_refusal_prints_the_error_record():
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2"], env=env, capture_output=True,
                       text=True, timeout=120)
    rec = json.loads([l for l in p.stdout.splitlines() if l.lstrip().startswith("{")][-1])
    assert rec["value"] is None and "HIP device" in rec["error"]


def test_self_launch_without_the_built_library_prints_the_error_record():
    """first contact with a box whose libsvihmm_hip.so is missing (or does not load) -- the
    launcher's preamble raises before any rank exists; the ONE JSON line is still there."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["SVIHMM_HIP_LIB"] = "/nonexistent/libsvihmm_hip.so"
    for gpus in ("2", "1"):
        p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", gpus, "--steps", "1", "--warmup", "0"],
                           env=env, capture_output=True, text=True, timeout=120)
        assert p.returncode != 0
        lines = [l for l in p.stdout.splitlines() if l.lstrip().startswith("{")]
        assert len(lines) == 1, p.stdout + p.stderr
        rec = json.loads(lines[0])
        assert rec["value"] is None and rec["n_gpus"] == int(gpus) and "error" in rec
