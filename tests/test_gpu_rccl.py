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
    # (round 5: a refused or failed multi-rank run still leaves ONE JSON line, the error record)
    import json
    lines = [l for l in r.stdout.splitlines() if l.lstrip().startswith("{")]
    assert r.returncode != 0 and len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["value"] is None and "HIP device" in rec["error"] and rec["n_gpus"] == n + 1


def test_allreduce_in_caller_coordinates_rehearsal(tmp_path):
    """Every rank's handle has its own centre, so the multi-rank all-reduce forms the sum in the
    ranks' common caller coordinates (k_packed_shift +1 -> ncclAllReduce -> -1).  That round trip
    only runs at nranks > 1; variant 11 switches it on at one rank: statistics after the
    all-reduce == statistics before it (NIW and diagonal family, data far from the origin), and
    the device-resident SVI loop (which all-reduces inside svihmm_svi_iteration) is unchanged."""
    from pysvihmm_amd.comm import RcclComm, file_uid_exchange
    from pysvihmm_amd.engine import HipEngine
    from pysvihmm_amd import hmmsgd_metaobs
    from pysvihmm_amd.distributions import Gaussian
    from pysvihmm_amd import _lib as L
    from tests.helpers import make_problem
    K, D, T = 5, 3, 1500
    pb = make_problem(K, D, T, seed=4, sep=3.0)
    off = np.array([2e3, -40.0, 0.5])
    obs, mu = pb["obs"] + off, pb["mu"] + off
    starts = np.arange(40, dtype=np.int64) * 33
    e = HipEngine(0)
    try:
        comm = RcclComm(e, 0, 1, file_uid_exchange(0, tag="rehearsal_%d" % os.getpid(), directory=str(tmp_path)))
        e.set_obs(obs, None)
        assert np.all(np.abs(e.get_shift() - off) < 10.0)
        e.set_globals(pb["mod_init"], pb["ltran"])
        rng = np.random.default_rng(0)
        diag = (mu, 0.5 + rng.random((K, D)), 2.0 + rng.random((K, D)), 1.0 + rng.random((K, D)))
        for upload in (lambda: e.set_emission_niw(mu, pb["sigma"], pb["kappa"], pb["nu"]),
                       lambda: e.set_emission_diag(*diag)):
            upload()
            ref = e.estep(starts, 33, flags=L.TRANS_WRAP).buf.copy()
            for v in (0, 1):
                e.set_variant(11, v)
                e.estep(starts, 33, flags=L.TRANS_WRAP, read=False)
                e.allreduce_packed()
                got = e.read_packed().buf
                np.testing.assert_allclose(got, ref, rtol=1e-12, atol=1e-9 * np.abs(ref).max())
            e.set_variant(11, 0)
        # the loop with the all-reduce inside the iteration
        res = []
        for v in (0, 1):
            e.set_variant(11, v)
            np.random.seed(2)
            prior = np.array([Gaussian(mu_0=obs.mean(0), sigma_0=0.75 * np.cov(obs.T), kappa_0=0.01, nu_0=D + 2)
                              for _ in range(K)])
            m = hmmsgd_metaobs.VBHMM(obs, np.ones(K), np.ones((K, K)), prior, tau=1.0, kappa=0.7, metaobs_half=8,
                                     mb_sz=6, maxit=5, seed=4, engine=e, comm=comm)
            m.infer()
            res.append(m)
        np.testing.assert_allclose(res[1].var_tran, res[0].var_tran, rtol=1e-9)
        np.testing.assert_allclose(res[1].elbo_vec, res[0].elbo_vec, rtol=1e-9)
        for k in range(K):
            np.testing.assert_allclose(res[1].var_emit[k].mu_mf, res[0].var_emit[k].mu_mf, rtol=1e-9, atol=1e-9)
            np.testing.assert_allclose(res[1].var_emit[k].sigma_mf, res[0].var_emit[k].sigma_mf, rtol=1e-8, atol=1e-9)
    finally:
        e.set_variant(11, 0)
        e.close()


@pytest.mark.parametrize("family", ["niw", "diag"])
def test_two_handles_with_different_centres_host_sum_stands_in_for_the_allreduce(family):
    """The arithmetic of svihmm_allreduce_packed on ranks whose resident copies have DIFFERENT
    centres (each rank its own sequence: bench.py's weak mode, BASELINE configs[3]) has never run at
    world size 2 on hardware.  Its two halves -- statistics into the callers' common coordinates
    (+c_r), the reduced vector back into the handle's centred `packed` (-c_r) -- are what
    svihmm_export_packed / svihmm_import_packed run; here two handles on ONE device hold two sequences
    whose centres lie 150 apart, a host-side sum stands in for ncclAllReduce, and the result is
    checked against the C oracle on the union of the windows and for equality across the handles."""
    from pysvihmm_amd.engine import HipEngine
    from pysvihmm_amd import _lib as L
    from oracle import ref_c
    from tests.helpers import make_problem, unpack
    K, D, T, Lm = 6, 5, 4000, 41
    offs = [np.full(D, 100.0), np.linspace(-60.0, -40.0, D)]
    half = [make_problem(K // 2, D, T, seed=31 + r, miss=0.03) for r in range(2)]
    pm = make_problem(K, D, 100, seed=33)                      # globals / factor shapes of the 6-state model
    xs = [half[r]["obs"] + offs[r] for r in range(2)]          # rank r's sequence lives around offs[r]
    mu = np.concatenate([half[r]["mu"] + offs[r] for r in range(2)])   # one model: three states at each place
    starts = [np.arange(30, dtype=np.int64) * 97 + 11 * r for r in range(2)]
    nus = 0.5 + pm["kappa"][:, None] * np.ones((K, D))
    al = 3.0 + 0.1 * np.arange(K)[:, None] + np.zeros((K, D))
    be = 2.0 + 0.05 * np.arange(D)[None, :] + np.zeros((K, D))
    engs = [HipEngine(0), HipEngine(0)]
    try:
        own, refs = [], []
        for r, e in enumerate(engs):
            e.set_obs(xs[r], half[r]["mask"])
            e.set_globals(pm["mod_init"], pm["ltran"])
            if family == "niw":
                e.set_emission_niw(mu, pm["sigma"], pm["kappa"], pm["nu"])
                refs.append(ref_c.estep_minibatch(xs[r], half[r]["mask"], starts[r], Lm, pm["mod_init"],
                                                  pm["ltran"], mu, pm["sigma"], pm["kappa"], pm["nu"], flags=2))
            else:
                e.set_emission_diag(mu, nus, al, be)
            e.estep(starts[r], Lm, flags=L.TRANS_WRAP, read=False)
            own.append(e.export_packed())
            np.testing.assert_array_equal(own[r], e.read_packed().buf)   # = what read_packed hands out
        c0, c1 = engs[0].get_shift(), engs[1].get_shift()
        assert np.abs(c0 - c1).min() > 100.0                             # the handles' centres differ
        total = own[0] + own[1]                                          # stands in for ncclAllReduce(sum)
        back = []
        for e in engs:
            e.import_packed(total)
            back.append(e.read_packed().buf)
        # each handle's -c_r / +c_r round trip returns the sum (second moments are ~ n |x|^2)
        n = sum(len(s) for s in starts) * Lm
        scale = np.maximum(np.abs(total), 1e-12 * n * 150.0 ** 2)
        for b in back:
            assert np.max(np.abs(b - total) / scale) < 1e-11, np.max(np.abs(b - total) / scale)
        if family == "niw":
            A, xbar, neff, S, lb = unpack(refs[0] + refs[1], K, D)
            g = engs[1].read_packed()
            np.testing.assert_allclose(g.A_raw, A, rtol=1e-6, atol=1e-9 * n)
            np.testing.assert_allclose(g.neff, neff, rtol=1e-6, atol=1e-9 * n)
            np.testing.assert_allclose(g.xbar, xbar, rtol=1e-6, atol=1e-7 * n)
            np.testing.assert_allclose(g.S, S, rtol=1e-6, atol=1e-5 * n)
            np.testing.assert_allclose(g.lb[0], lb, rtol=1e-9)
            # every state of the model took mass on exactly one of the two ranks
            assert (neff > 50).all()
    finally:
        for e in engs:
            e.close()
