"""The fused E-step launch (kernels_fused.h, round 6: sweeps + statistics, and the emission tiles where the batch's
emission kernel is the 16-row minibatch one) against the C oracle and against the separate launches it replaces: every
shape class it takes -- one window to a quarter of the device, windows of 3 to 300 rows (bands thinner than a stage,
stages mostly padding, emission rounds of one tile), masks, no wrap-around, the fp32 storage format -- with the fused
path forced (variant "pipeline" = 3), forced without the emission tiles (= 4 keeps the emission kernel; with < 16 windows
that is the separate path) and switched off (= 1)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _problem(K, D, T, seed, miss=0.05):
    from tests.helpers import make_problem
    return make_problem(K, D, T, seed=seed, miss=miss)


@pytest.mark.parametrize("B,Lm,D,wrap", [(1, 257, 32, True), (9, 33, 32, True), (64, 257, 32, True), (23, 3, 32, False),
                                         (100, 65, 40, True), (130, 9, 48, True), (5, 300, 32, False)])
def test_fused_launch_vs_oracle_and_separate_launches(B, Lm, D, wrap):
    from oracle import ref_c
    from pysvihmm_amd.engine import HipEngine
    from pysvihmm_amd import _lib as L
    K, T = 64, 6000
    pb = _problem(K, D, T, seed=B + Lm + D)
    starts = np.random.default_rng(B).integers(0, T - Lm + 1, size=B)
    flags = L.TRANS_WRAP if wrap else 0
    eng = HipEngine(0)
    try:
        eng.set_obs(pb["obs"], pb["mask"])
        eng.set_globals(pb["mod_init"], pb["ltran"])
        eng.set_emission_niw(pb["mu"], pb["sigma"], pb["kappa"], pb["nu"])
        res = {}
        for mode in (1, 3, 4, 3):
            eng.set_variant("pipeline", mode)
            res[mode] = eng.estep(starts, Lm, flags=flags).buf.copy()
            res[mode, "kernel"] = eng.last_kernel("forward_backward")
        eng.set_variant("pipeline", 0)
    finally:
        eng.close()
    assert "k_sweep_stats" in res[3, "kernel"], res[3, "kernel"]
    assert "k_sweep_stats" not in res[1, "kernel"]
    # the emission tiles ride in the launch where the batch's emission kernel is the 16-row one and its theta operands
    # fit a wave's registers (D % 8 == 0, D <= 32)
    assert (", true" in res[3, "kernel"]) == (D <= 32), res[3, "kernel"]
    if B >= 16:
        # same sweeps and statistics with the emission kernel as a launch of its own: the tile arithmetic is the same
        # code, so the statistics agree bit for bit -- a row read before its tile was complete would show here
        assert "k_sweep_stats" in res[4, "kernel"] and ", false" in res[4, "kernel"], res[4, "kernel"]
        assert np.array_equal(res[3], res[4]), float(np.max(np.abs(res[3] - res[4])))
    ref = ref_c.estep_minibatch(pb["obs"], pb["mask"], starts, Lm, pb["mod_init"], pb["ltran"], pb["mu"], pb["sigma"],
                                pb["kappa"], pb["nu"], flags=2 if wrap else 0)
    scale = np.maximum(np.abs(ref), 1e-9 * B * Lm)
    assert np.max(np.abs(res[3] - ref) / scale) < 1e-6, "fused vs oracle"
    assert np.max(np.abs(res[3] - res[1]) / scale) < 1e-9, "fused vs separate launches"


def test_fused_launch_in_the_fp32_format():
    """fp32 mode with the fused launch forced: float messages, fp64 statistics stages -- within the mode's 1e-3."""
    from pysvihmm_amd.engine import HipEngine
    from pysvihmm_amd import _lib as L
    K, D, T, B, Lm = 64, 32, 6000, 40, 129
    pb = _problem(K, D, T, seed=3, miss=0.0)
    starts = np.random.default_rng(1).integers(0, T - Lm + 1, size=B)
    eng = HipEngine(0)
    try:
        eng.set_obs(pb["obs"], None)
        eng.set_globals(pb["mod_init"], pb["ltran"])
        eng.set_emission_niw(pb["mu"], pb["sigma"], pb["kappa"], pb["nu"])
        ref = eng.estep(starts, Lm, flags=L.TRANS_WRAP).buf.copy()
        eng.set_precision("f32")
        eng.set_variant("pipeline", 3)
        out = eng.estep(starts, Lm, flags=L.TRANS_WRAP).buf.copy()
        used, kern = eng.precision()[1], eng.last_kernel("forward_backward")
        eng.set_variant("pipeline", 0)
        eng.set_precision("f64")
    finally:
        eng.close()
    assert used and "k_sweep_stats" in kern and "float" in kern, (used, kern)
    scale = np.maximum(np.abs(ref), 1e-6 * B * Lm)
    assert np.max(np.abs(out - ref) / scale) < 1e-3


def test_fused_launch_inside_the_resident_loop_matches_the_separate_launches():
    """The device-resident SVI loop at the S = 64 shape class (K = 64, 24 windows) with the fused launch and without:
    same state and ELBO trace to rounding (the fused statistics normalise every row by its own sum)."""
    from pysvihmm_amd.engine import HipEngine
    from pysvihmm_amd.distributions import niw_prior_logpart
    from pysvihmm_amd import _lib as L
    K, D, T, B, Lm, nit = 64, 32, 6000, 24, 65, 6
    pb = _problem(K, D, T, seed=11, miss=0.0)
    rng = np.random.default_rng(K)
    prior_tran = 1.0 + rng.random((K, K))
    mu0 = np.tile(pb["obs"].mean(0), (K, 1)) + 0.1 * rng.normal(size=(K, D))
    sg0 = np.tile(0.75 * np.cov(pb["obs"].T).reshape(D, D), (K, 1, 1))
    ka0, nu0 = np.full(K, 0.01), np.full(K, D + 2.0)
    res = []
    for mode in (1, 3, 4):
        eng = HipEngine(0)
        try:
            eng.set_variant("pipeline", mode)
            eng.set_obs(pb["obs"], None)
            eng.svi_begin(prior_tran, pb["var_tran"], (mu0, sg0, ka0, nu0), (pb["mu"], pb["sigma"], pb["kappa"], pb["nu"]),
                          niw_prior_logpart(sg0, nu0), nit, 1.0)
            r2 = np.random.default_rng(5)
            for it in range(nit):
                eng.svi_iteration(it, r2.integers(0, T - Lm, size=B), B, Lm, L.TRANS_WRAP, (it + 1.0) ** -0.7, 3.0, 2.5)
            res.append((eng.svi_read_state(), eng.svi_read_elbo(nit)[0]))
        finally:
            eng.close()
    for n, a, b in zip(("var_tran", "var_init", "mu", "sigma", "kappa", "nu"), res[0][0], res[1][0]):
        np.testing.assert_allclose(a, b, rtol=1e-7, atol=1e-9, err_msg=n)
    np.testing.assert_allclose(res[0][1], res[1][1], rtol=1e-9)
    # emission tiles inside the launch (3) or as the launch of their own (4): the same arithmetic, bit for bit -- in the
    # loop the emission buffers are REUSED every iteration, so a stale row (last iteration's, from a cache) would show
    for n, a, b in zip(("var_tran", "var_init", "mu", "sigma", "kappa", "nu"), res[1][0], res[2][0]):
        assert np.array_equal(a, b), n
    assert np.array_equal(res[1][1], res[2][1])


def test_fused_launch_on_random_shapes_vs_oracle():
    """Fixed-seed sweep over what the fused launch's plan depends on: feature count (five-tile shapes of K = 64 and their
    neighbours that fall back), windows from 1 to 150, window lengths from 3 rows (bands thinner than a stage) to 300,
    masks, wrap-around on / off -- forced (variant 3: emission tiles inside where the shape allows) and automatic."""
    from oracle import ref_c
    from pysvihmm_amd.engine import HipEngine
    from pysvihmm_amd import _lib as L
    rng = np.random.default_rng(20260930)
    K, T = 64, 4000
    taken = 0
    for case in range(22):
        D = int(rng.choice([25, 28, 31, 32, 33, 40, 48, 56, 64]))
        Lm = int(rng.choice([3, 4, 9, 33, 64, 129, 257, 300]))
        B = int(rng.choice([1, 2, 16, 17, 40, 64, 100, 150]))
        B = max(1, min(B, 8000 // Lm))
        wrap = bool(rng.integers(2))
        miss = float(rng.choice([0.0, 0.1]))
        mode = int(rng.choice([3, 3, 0]))
        pb = _problem(K, D, T, seed=1000 + case, miss=miss)
        starts = rng.integers(0, T - Lm + 1, size=B)
        flags = L.TRANS_WRAP if wrap else 0
        eng = HipEngine(0)
        try:
            eng.set_obs(pb["obs"], pb["mask"])
            eng.set_globals(pb["mod_init"], pb["ltran"])
            eng.set_emission_niw(pb["mu"], pb["sigma"], pb["kappa"], pb["nu"])
            eng.set_variant("pipeline", mode)
            out = eng.estep(starts, Lm, flags=flags).buf.copy()
            kern = eng.last_kernel("forward_backward")
            eng.set_variant("pipeline", 0)
        finally:
            eng.close()
        taken += "k_sweep_stats" in kern
        ref = ref_c.estep_minibatch(pb["obs"], pb["mask"], starts, Lm, pb["mod_init"], pb["ltran"], pb["mu"], pb["sigma"],
                                    pb["kappa"], pb["nu"], flags=2 if wrap else 0)
        scale = np.maximum(np.abs(ref), 1e-9 * B * Lm)
        err = float(np.max(np.abs(out - ref) / scale))
        assert err < 1e-6, (case, D, Lm, B, wrap, miss, mode, kern, err)
    assert taken >= 8, taken       # (the sweep is about the fused launch: most shapes must have taken it)


@pytest.mark.parametrize("B,Lm,pipeline", [(9, 257, 0), (40, 129, 0), (64, 257, 3)])
def test_renormalising_every_fourth_step_changes_nothing(B, Lm, pipeline):
    """The register-resident sweep (stand-alone below 16 windows, inside the fused launch above) re-normalises its
    vector every fourth step where the transition expectations lie inside a float's range (variant 16 = 1: every
    step).  Scaling by powers of two is exact, so statistics and local bound must agree BIT FOR BIT -- also with
    transition expectations at the edge of that range (-59 nats: a step may shrink the vector by 2^-85)."""
    from pysvihmm_amd.engine import HipEngine
    from pysvihmm_amd import _lib as L
    K, D, T = 64, 32, 6000
    pb = _problem(K, D, T, seed=B + Lm, miss=0.05)
    rng = np.random.default_rng(B)
    starts = rng.integers(0, T - Lm + 1, size=B)
    for edge in (False, True):
        ltran = pb["ltran"].copy()
        if edge:      # a third of the transitions at the edge of the range, the diagonal kept
            m = (rng.random((K, K)) < 0.33) & ~np.eye(K, dtype=bool)
            ltran[m] = -59.0
        out = {}
        for every in (0, 1):
            eng = HipEngine(0)
            try:
                eng.set_variant(16, every)
                eng.set_variant("pipeline", pipeline)
                eng.set_obs(pb["obs"], pb["mask"])
                eng.set_globals(pb["mod_init"], ltran)
                eng.set_emission_niw(pb["mu"], pb["sigma"], pb["kappa"], pb["nu"])
                out[every] = eng.estep(starts, Lm, flags=L.TRANS_WRAP).buf.copy()
                kern = eng.last_kernel("forward_backward")
            finally:
                eng.close()
        assert np.all(np.isfinite(out[0])), (edge, kern)
        # (statistics: bit for bit; the local bound is a sum of logarithms of differently scaled -- equal -- sums: rounding)
        assert np.array_equal(out[0][:-1], out[1][:-1]), (edge, kern, float(np.max(np.abs(out[0] - out[1]))))
        assert abs(out[0][-1] - out[1][-1]) <= 1e-13 * abs(out[1][-1]), (edge, kern, out[0][-1], out[1][-1])


def test_f32_mode_minibatch_takes_the_fused_launch_with_fp64_messages():
    """fp32 mode, minibatch-sized batch: the mode's bf16 emission kernel writes float emission rows, sweeps and
    statistics run as the fused launch on fp64 messages (k_sweep_stats<..., double, ..., float>) -- reported as the fp32
    format, within the mode's tolerance of the fp64 statistics (closer than the float-message path: variant 4 = 6)."""
    from pysvihmm_amd.engine import HipEngine
    from pysvihmm_amd import _lib as L
    K, D, T, B, Lm = 64, 32, 40000, 64, 257
    pb = _problem(K, D, T, seed=64, miss=0.0)
    starts = (np.arange(B, dtype=np.int64) * (T // B)) % (T - Lm)
    out = {}
    for mode, prec in ((0, "f64"), (0, "f32"), (6, "f32")):
        eng = HipEngine(0)
        try:
            eng.set_precision(prec)
            eng.set_variant("pipeline", mode)
            eng.set_obs(pb["obs"], None)
            eng.set_globals(pb["mod_init"], pb["ltran"])
            eng.set_emission_niw(pb["mu"], pb["sigma"], pb["kappa"], pb["nu"])
            out[mode, prec] = (eng.estep(starts, Lm, flags=L.TRANS_WRAP).buf.copy(), eng.last_kernel("forward_backward"),
                               eng.precision())
        finally:
            eng.close()
    ref = out[0, "f64"][0]
    scale = np.maximum(np.abs(ref), 1e-6 * B * Lm)
    kern = out[0, "f32"][1]
    assert "k_sweep_stats" in kern and kern.endswith("float>") and ", double," in kern, kern
    assert out[0, "f32"][2] == ("f32", True)
    assert "k_sweep_stats" not in out[6, "f32"][1] and out[6, "f32"][2] == ("f32", True)
    e_mixed = float(np.max(np.abs(out[0, "f32"][0] - ref) / scale))
    e_float = float(np.max(np.abs(out[6, "f32"][0] - ref) / scale))
    assert e_mixed < 1e-3 and e_float < 1e-3, (e_mixed, e_float)
    assert e_mixed <= e_float * 1.5, (e_mixed, e_float)
