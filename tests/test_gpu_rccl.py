"""The N>1 path on real GPUs: one process per GPU, RCCL all-reduce of the packed statistics
through the C ABI (svihmm_comm_init / svihmm_allreduce_packed), windows of each minibatch
dealt round-robin to the ranks.  World size 2 needs two visible GPUs (skipped otherwise, so an
8-GPU box exercises it); the same job at world size 1 always runs (the communicator, the
all-reduce and the host protocol are then exercised with one rank)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(REPO, "tests", "golden")


def _ndev():
    from pysvihmm_amd.engine import device_count
    return device_count()


def _run_job(tmp_path, world):
    fixture = os.path.join(GOLDEN, "metaobs_K4_D2_L10_mask.npz")
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world),
                   SVIHMM_TEST_TAG="t%d_%d" % (os.getpid(), world), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(REPO, "tests", "_rccl_worker.py"),
                                       str(tmp_path), fixture], cwd=REPO, env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-4000:]
    g = np.load(fixture)
    res = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % i)) for i in range(world)]
    for o in res:
        assert int(o["nranks"]) == world
        np.testing.assert_allclose(o["var_tran"], g["it_var_tran_new"][-1], rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(o["mu"], g["it_new_mu"][-1], rtol=1e-8, atol=1e-8)
        np.testing.assert_allclose(o["sigma"], g["it_new_sigma"][-1], rtol=1e-8, atol=1e-7)
        np.testing.assert_allclose(o["elbo"], g["elbo_vec"], rtol=1e-8)
    for o in res[1:]:                       # replicas stay in lock-step bit for bit
        np.testing.assert_array_equal(o["var_tran"], res[0]["var_tran"])
        np.testing.assert_array_equal(o["sigma"], res[0]["sigma"])
        np.testing.assert_array_equal(o["elbo"], res[0]["elbo"])


def test_rccl_world_size_one_job(tmp_path):
    _run_job(tmp_path, 1)


def test_rccl_world_size_two_sharded_minibatch(tmp_path):
    if _ndev() < 2:
        pytest.skip("needs 2 visible GPUs (svihmm_device_count() = %d)" % _ndev())
    _run_job(tmp_path, 2)


def _bench(extra_env, *argv):
    env = dict(os.environ, **extra_env)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + list(argv), cwd=REPO, env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_forced_comm_dry_run():
    """SVIHMM_FORCE_COMM=1: bench.py's multi-GPU code path (communicator, all-reduce in every
    step, RCCL barriers, max-over-ranks timing) at world size 1."""
    j = _bench({"SVIHMM_FORCE_COMM": "1"}, "--steps", "3", "--warmup", "1", "--reps", "3", "--no-side",
               "--no-cpu-baseline")
    assert j["n_gpus"] == 1 and j["ranks"] == 1 and j["ranks_source"] == "ncclCommCount"
    assert j["steps"] == 3 and len(j["ms_per_step_reps"]) == 3 and j["value"] > 1e9
    assert "allreduce" in j["kernels"] and len(j["per_rank_ms_per_step"]) == 1
    assert j["scaling"] == "weak" and j["allreduce"]["ms_per_step"] > 0
    # the weak run carries the strong-scaling record of the same job (one sequence, windows dealt)
    ss = j["strong_scaling"]
    for name, nwin in (("epoch", 3891), ("minibatch_s64", 64)):
        assert ss[name]["scaling"] == "strong" and ss[name]["windows"] == nwin == ss[name]["windows_per_rank"]
        assert ss[name]["value"] > 1e9 and ss[name]["allreduce_ms_per_step"] > 0


def test_bench_strong_scaling_dry_run():
    """`--scaling strong` through the forced communicator at world size 1: the headline is then
    the one-sequence epoch with the windows dealt over the ranks."""
    j = _bench({"SVIHMM_FORCE_COMM": "1"}, "--scaling", "strong", "--steps", "3", "--warmup", "1", "--reps", "3",
               "--no-side", "--no-cpu-baseline")
    assert j["scaling"] == "strong" and j["config"]["sequences"] == 1 and j["ranks"] == 1
    assert j["value"] > 1e9 and "strong_scaling" not in j


def test_bench_self_launch_two_gpus():
    """`python bench.py --gpus 2` with no launcher starts both ranks itself and reports what
    RCCL saw."""
    if _ndev() < 2:
        pytest.skip("needs 2 visible GPUs")
    j = _bench({}, "--gpus", "2", "--steps", "3", "--warmup", "1", "--reps", "3")
    assert j["n_gpus"] == 2 and j["ranks"] == 2 and len(j["per_rank_ms_per_step"]) == 2
    assert j["config"]["sequences"] == 2


def test_bench_refuses_more_gpus_than_visible():
    n = _ndev()
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", str(n + 1)], cwd=REPO,
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "{" not in r.stdout
