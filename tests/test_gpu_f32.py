"""fp32 mode (north_star: "posteriors and natural gradients within 1e-3 fp32"): svihmm_set_precision
(SVIHMM_F32) stores the scaled emission likelihoods and messages as float and runs the
expected-sufficient-statistics GEMM on v_mfma_f32_16x16x4_f32.  Tolerance, written here:
statistics within rtol 1e-3 of the fp64 C oracle (atol = 1e-6 x the batch's row count, i.e.
1e-6 per row of accumulated mass), posteriors within 1e-3 absolute, local bound within 1e-6
relative; the measured errors are ~1e-6 and are asserted to stay below 1e-4 so that a
regression to "barely 1e-3" is noticed."""
import glob
import os

import numpy as np
import pytest

from tests.helpers import make_problem, unpack, effective_cores
from tests.test_host_logic import emit_from_fixture

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _close(got, ref, scale, what):
    err = np.abs(got - ref)
    tol = 1e-3 * np.abs(ref) + 1e-6 * scale
    assert np.all(err <= tol), (what, float((err / (np.abs(ref) + 1e-6 * scale)).max()))
    return float((err / (np.abs(ref) + 1e-6 * scale)).max())


@pytest.mark.parametrize("K,D,B,Lm", [(5, 3, 7, 40), (16, 8, 203, 33), (33, 16, 60, 65), (64, 32, 24, 257),
                                      (64, 32, 1100, 17), (50, 8, 1500, 9)])
def test_f32_estep_vs_fp64_oracle(K, D, B, Lm):
    """wave-per-window (B < 1025, K <= 16), four-wave (K > 16, B <= 256) and MFMA sweeps (B >= 1025)
    in the fp32 format, ragged K, masked rows, wrap statistic."""
    from pysvihmm_amd.engine import HipEngine
    from pysvihmm_amd import _lib as L
    from oracle import ref_c
    T = max(4000, B * 3 + Lm)
    pb = make_problem(K, D, T, seed=K * 7 + D, miss=0.05, sep=4.0)
    starts = np.random.default_rng(B).integers(0, T - Lm + 1, size=B)
    e = HipEngine(0, dtype="f32")
    e.set_obs(pb["obs"], pb["mask"])
    e.set_globals(pb["mod_init"], pb["ltran"])
    e.set_emission_niw(pb["mu"], pb["sigma"], pb["kappa"], pb["nu"])
    st = e.estep(starts, Lm, flags=L.TRANS_WRAP)
    assert e.precision() == ("f32", True)
    ref = ref_c.estep_minibatch(pb["obs"], pb["mask"], starts, Lm, pb["mod_init"], pb["ltran"], pb["mu"],
                                pb["sigma"], pb["kappa"], pb["nu"], flags=2, threads=effective_cores())
    A, xbar, neff, S, lb = unpack(ref, K, D)
    sc = B * Lm
    worst = max(_close(st.A_raw, A, sc, "A"), _close(st.neff, neff, sc, "neff"),
                _close(st.xbar, xbar, sc * np.abs(pb["obs"]).max(), "xbar"),
                _close(st.S, S, sc * np.abs(pb["obs"]).max() ** 2, "S"))
    assert worst < 1e-4, worst
    np.testing.assert_allclose(st.lb[0], lb, rtol=1e-6)
    # posteriors of the first and last window: |dq| <= 1e-3 (measured ~1e-7), rows sum to one
    for b in (0, B - 1):
        x = pb["obs"][starts[b]:starts[b] + Lm]
        ll = ref_c.lliks_niw(x, pb["mu"], pb["sigma"], pb["kappa"], pb["nu"])
        q, _ = ref_c.posterior(ref_c.forward(ll, pb["mod_init"], pb["ltran"]), ref_c.backward(ll, pb["ltran"]))
        got = e.read_rows("var_x", b * Lm, Lm)
        assert np.abs(got - q).max() < 1e-5
        np.testing.assert_allclose(got.sum(1), 1.0, atol=1e-6)
    # log-domain rows are rebuilt in fp64 on demand, whatever the mode
    la = e.read_rows("lalpha", 0, Lm)
    x = pb["obs"][starts[0]:starts[0] + Lm]
    ll = ref_c.lliks_niw(x, pb["mu"], pb["sigma"], pb["kappa"], pb["nu"])
    np.testing.assert_allclose(la, ref_c.forward(ll, pb["mod_init"], pb["ltran"]), rtol=1e-9, atol=1e-7)
    # switching back restores bit-identical fp64 results
    e.set_precision("f64")
    a = e.estep(starts, Lm, flags=L.TRANS_WRAP).buf.copy()
    assert e.precision() == ("f64", False)
    e2 = HipEngine(0)
    e2.set_obs(pb["obs"], pb["mask"]); e2.set_globals(pb["mod_init"], pb["ltran"])
    e2.set_emission_niw(pb["mu"], pb["sigma"], pb["kappa"], pb["nu"])
    np.testing.assert_array_equal(a, e2.estep(starts, Lm, flags=L.TRANS_WRAP).buf)
    e.close(); e2.close()


def test_f32_outside_the_fast_path_runs_fp64():
    """K > 64 and the whole-chain scan are not covered by the mode: they run as fp64 and say so."""
    from pysvihmm_amd.engine import HipEngine
    from pysvihmm_amd import _lib as L
    pb = make_problem(100, 3, 6000, seed=3, miss=0.0)
    res = []
    for dt in ("f32", "f64"):
        e = HipEngine(0, dtype=dt)
        e.set_obs(pb["obs"], None); e.set_globals(pb["mod_init"], pb["ltran"])
        e.set_emission_niw(pb["mu"], pb["sigma"], pb["kappa"], pb["nu"])
        a = e.estep(np.arange(200) * 20, 15, flags=L.TRANS_WRAP).buf.copy()
        used = e.precision()[1]
        b = e.estep([0], 6000, flags=0).buf.copy()
        res.append((a, b, used, e.precision()[1]))
        e.close()
    assert res[0][2] is False and res[0][3] is False
    np.testing.assert_array_equal(res[0][0], res[1][0])
    np.testing.assert_array_equal(res[0][1], res[1][1])


@pytest.mark.parametrize("name", ["metaobs_K4_D2_L10_mask", "metaobs_K16_D8_L16", "metaobs_K64_D32_L8"])
def test_f32_class_infer_vs_reference_trace(name):
    """hmmsgd_metaobs.VBHMM(dtype='f32').infer() against the executed reference's trace at the
    fp32 tolerance (1e-3 relative on the natural-gradient results = the updated parameters)."""
    from pysvihmm_amd import hmmsgd_metaobs
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    K = int(g["K"])
    hmm = hmmsgd_metaobs.VBHMM(
        g["obs"].copy(), np.ones(K), g["prior_tran"], emit_from_fixture(g, K),
        tau=float(g["tau"]), kappa=float(g["kappa"]), metaobs_half=int(g["L"]), mb_sz=int(g["S"]),
        mask=g["mask"], init_tran=g["init_tran"], maxit=int(g["maxit"]), seed=int(g["seed"]), dtype="f32")
    hmm.infer()
    assert hmm.engine.precision()[0] == "f32"
    np.testing.assert_allclose(hmm.var_tran, g["it_var_tran_new"][-1], rtol=1e-3, atol=1e-6)
    for k in range(K):
        np.testing.assert_allclose(hmm.var_emit[k].mu_mf, g["it_new_mu"][-1][k], rtol=1e-3, atol=1e-5)
        np.testing.assert_allclose(hmm.var_emit[k].sigma_mf, g["it_new_sigma"][-1][k], rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(hmm.elbo_vec, g["elbo_vec"], rtol=1e-5)
    assert np.abs(hmm.var_x - g["w_var_x"][-1]).max() < 1e-3


@pytest.mark.parametrize("K,D,B,Lm,off,sep", [(64, 32, 160, 257, 0.0, 4.0), (33, 16, 600, 65, 0.0, 4.0),
                                              (5, 3, 1200, 40, 0.0, 1.0), (64, 32, 130, 257, 1e4, 4.0),
                                              (17, 32, 2100, 16, -30.0, 0.5), (64, 7, 1040, 33, 0.0, 1.0),
                                              (64, 32, 140, 257, 0.0, 0.25), (48, 24, 300, 129, 7.0, 0.4)])
def test_f32_bf16x3_emission_large_batch(K, D, B, Lm, off, sep):
    """Batches of >= 256 x 128 rows (NIW, K <= 64, D <= 32) in the fp32 mode take k_emission_bf16x3:
    the centred quadratic form on the bf16 matrix pipe, x and U as three bf16 terms each.  Tolerance,
    written here: statistics as the mode's other kernels (1e-3, asserted below 1e-4), posteriors
    within 1e-4 absolute, local bound within 1e-5 relative (the log-likelihoods are fp32 values of
    size |ll|), and the same batch through the fp64 feature GEMM of the mode (variant 5 = 3) agrees
    to the same bounds.  Ragged K (state groups and pairs with padding), ragged D, masked rows, data
    far from the origin (the handle's centre keeps the operands small), overlapping states (sep < 1:
    soft posteriors, where an error in the log-likelihoods shows)."""
    from pysvihmm_amd.engine import HipEngine
    from pysvihmm_amd import _lib as L
    from oracle import ref_c
    T = max(6000, B * 3 + Lm)
    pb = make_problem(K, D, T, seed=K * 5 + D + 1, miss=0.03, sep=sep)
    obs = pb["obs"] + off
    mu = pb["mu"] + off
    starts = np.random.default_rng(B + 1).integers(0, T - Lm + 1, size=B)
    assert B * Lm >= 256 * 128
    e = HipEngine(0, dtype="f32")
    e.set_obs(obs, pb["mask"])
    e.set_globals(pb["mod_init"], pb["ltran"])
    e.set_emission_niw(mu, pb["sigma"], pb["kappa"], pb["nu"])
    st = e.estep(starts, Lm, flags=L.TRANS_WRAP)
    assert e.precision() == ("f32", True)
    buf = st.buf.copy()
    ref = ref_c.estep_minibatch(obs, pb["mask"], starts, Lm, pb["mod_init"], pb["ltran"], mu,
                                pb["sigma"], pb["kappa"], pb["nu"], flags=2, threads=effective_cores())
    A, xbar, neff, S, lb = unpack(ref, K, D)
    sc = B * Lm
    xs = max(np.abs(obs).max(), 1.0)
    worst = max(_close(st.A_raw, A, sc, "A"), _close(st.neff, neff, sc, "neff"),
                _close(st.xbar, xbar, sc * xs, "xbar"), _close(st.S, S, sc * xs ** 2, "S"))
    # measured: 1e-5 .. 3e-5 on separated states; overlapping states (soft posteriors) 4.4e-4 on entries of
    # S at the absolute floor (fp64 emission in front of the same fp32 sweeps: 5e-5), posteriors 4e-6
    assert worst < (1e-4 if sep >= 1.0 else 1e-3), worst
    np.testing.assert_allclose(st.lb[0], lb, rtol=1e-5)
    for b in (0, B // 2, B - 1):
        x = obs[starts[b]:starts[b] + Lm]
        ll = ref_c.lliks_niw(x, mu, pb["sigma"], pb["kappa"], pb["nu"])
        q, _ = ref_c.posterior(ref_c.forward(ll, pb["mod_init"], pb["ltran"]), ref_c.backward(ll, pb["ltran"]))
        assert np.abs(e.read_rows("var_x", b * Lm, Lm) - q).max() < 2e-5
    # the same batch with the fp64 feature GEMM in front of the fp32 sweeps / statistics
    e.set_variant(5, 3)
    st2 = e.estep(starts, Lm, flags=L.TRANS_WRAP)
    assert e.precision() == ("f32", True)
    np.testing.assert_allclose(st2.lb[0], lb, rtol=1e-5)
    d = np.abs(st2.buf - buf)
    assert np.all(d <= 1e-3 * np.abs(buf) + 1e-6 * sc * xs ** 2)
    # the mode switched on AFTER the parameter upload: factors built from the resident NIW block
    e3 = HipEngine(0)
    e3.set_obs(obs, pb["mask"]); e3.set_globals(pb["mod_init"], pb["ltran"])
    e3.set_emission_niw(mu, pb["sigma"], pb["kappa"], pb["nu"])
    e3.set_precision("f32")
    np.testing.assert_array_equal(e3.estep(starts, Lm, flags=L.TRANS_WRAP).buf, buf)
    e.close(); e3.close()


@pytest.mark.parametrize("K,D", [(40, 20), (48, 48), (64, 40), (130, 24), (256, 64)])
@pytest.mark.parametrize("flags_name", ["TRANS_WRAP", "MASK_AS_NAN"])
def test_f32_bf16x3_emission_nan_rows_and_outliers(flags_name, K, D):
    """Rows with NaN entries (log-likelihood 0 for every state, hmmbase.py:220), the missing-data flag and
    a row 1e6 standard deviations away from every state, in a batch large enough for k_emission_bf16x3:
    the fp32 mode's statistics stay within its tolerance of the fp64 oracle and nothing non-finite
    leaves the kernel."""
    from pysvihmm_amd.engine import HipEngine
    from pysvihmm_amd import _lib as L
    from oracle import ref_c
    # (40, 20): k_emission_bf16x3; (48, 48), (64, 40): k_emission_bf16x3d (round 5; K = 64: k_stats_bf16x3w);
    # (130, 24), (256, 64): the wide path (k_emission_bf16x3d<WIDE>, k_scale_ll_f32, k_sweeps_lin2<float>, k_stats_bf16x3w)
    B, Lm = 520, 64
    T = 6000
    pb = make_problem(K, D, T, seed=11, miss=0.05, sep=2.0)
    obs = pb["obs"].copy()
    rng = np.random.default_rng(5)
    obs[rng.integers(0, T, size=40)] = np.nan                  # whole rows
    obs[rng.integers(0, T, size=40), rng.integers(0, D, size=40)] = np.nan
    obs[1234] = 1e6                                            # far from every state
    starts = rng.integers(0, T - Lm + 1, size=B)
    flags = getattr(L, flags_name)
    e = HipEngine(0, dtype="f32")
    e.set_obs(obs, pb["mask"]); e.set_globals(pb["mod_init"], pb["ltran"])
    e.set_emission_niw(pb["mu"], pb["sigma"], pb["kappa"], pb["nu"])
    st = e.estep(starts, Lm, flags=flags)
    assert e.precision() == ("f32", True)
    ref = ref_c.estep_minibatch(obs, pb["mask"], starts, Lm, pb["mod_init"], pb["ltran"], pb["mu"],
                                pb["sigma"], pb["kappa"], pb["nu"], flags=int(flags), threads=effective_cores())
    A, xbar, neff, S, lb = unpack(ref, K, D)
    sc = B * Lm
    # (the moments of a state that a NaN row contributes to are NaN in the reference as well: posteriors,
    #  transition counts and the bound are what a NaN row must leave intact)
    assert np.all(np.isfinite(st.A_raw)) and np.all(np.isfinite(st.neff)) and np.isfinite(st.lb[0])
    _close(st.A_raw, A, sc, "A"); _close(st.neff, neff, sc, "neff")
    np.testing.assert_array_equal(np.isnan(st.xbar), np.isnan(xbar))
    ok = ~np.isnan(xbar)
    if ok.any():
        _close(st.xbar[ok], xbar[ok], sc * 1e6, "xbar")
    np.testing.assert_allclose(st.lb[0], lb, rtol=1e-5)
    # the same through the fp64 feature GEMM of the mode
    e.set_variant(5, 3)
    st2 = e.estep(starts, Lm, flags=flags)
    np.testing.assert_allclose(st2.neff, st.neff, rtol=1e-3, atol=1e-6 * sc)
    e.close()


@pytest.mark.parametrize("D,B,Lm,flags_wrap,inner", [
    (32, 24, 257, True, None),          # minibatch-sized: one chunk per four stages
    (32, 1100, 17, True, None),         # windows much shorter than a stage: many window starts per stage
    (32, 300, 65, False, None),         # no wrap statistic: window starts have no predecessor
    (16, 700, 33, True, None),          # D = 16: Fp = 160, five feature tiles + two transition tiles
    (32, 64, 129, True, (20, 89)),      # buffered meta-observations: inner segment
    (32, 4200, 9, True, None),          # > 256 chunks' worth of rows, ragged last chunk
])
def test_f32_statistics_on_the_bf16_pipe(D, B, Lm, flags_wrap, inner):
    """Round 4: the fp32 mode's statistics GEMM as three-term bf16 products (k_stats_bf16x3; K = 64,
    D <= 32, whole 32-feature tiles).  Against the fp64 C oracle at the mode's tolerance (measured
    errors asserted below 1e-4) and against the fp32-input MFMA kernel it replaces (variant 10 = 2)."""
    from pysvihmm_amd.engine import HipEngine
    from pysvihmm_amd import _lib as L
    from oracle import ref_c
    K = 64
    T = max(6000, B * 3 + Lm)
    pb = make_problem(K, D, T, seed=D * 11 + B, miss=0.06, sep=4.0)
    starts = np.random.default_rng(B + 1).integers(0, T - Lm + 1, size=B)
    flags = L.TRANS_WRAP if flags_wrap else 0
    e = HipEngine(0, dtype="f32")
    try:
        e.set_obs(pb["obs"], pb["mask"])
        e.set_globals(pb["mod_init"], pb["ltran"])
        e.set_emission_niw(pb["mu"], pb["sigma"], pb["kappa"], pb["nu"])
        e.set_variant(10, 3)              # the bf16 kernel also below its batch-size floor of 32 768 rows
        st = e.estep(starts, Lm, flags=flags, inner=inner)
        assert e.precision() == ("f32", True)
        got = st.buf.copy()
        e.set_variant(10, 2)
        old = e.estep(starts, Lm, flags=flags, inner=inner).buf.copy()
        assert not np.array_equal(got, old)       # (two different kernels did run)
        e.set_variant(10, 0)
        if B * Lm >= 32768:               # ... and by default above it
            np.testing.assert_array_equal(e.estep(starts, Lm, flags=flags, inner=inner).buf, got)
        if inner is None:
            ref = ref_c.estep_minibatch(pb["obs"], pb["mask"], starts, Lm, pb["mod_init"], pb["ltran"], pb["mu"],
                                        pb["sigma"], pb["kappa"], pb["nu"], flags=2 if flags_wrap else 0,
                                        threads=effective_cores())
            A, xbar, neff, S, lb = unpack(ref, K, D)
            sc = B * Lm
            xs = np.abs(pb["obs"]).max()
            worst = max(_close(st.A_raw, A, sc, "A"), _close(st.neff, neff, sc, "neff"),
                        _close(st.xbar, xbar, sc * xs, "xbar"), _close(st.S, S, sc * xs ** 2, "S"))
            assert worst < 1e-4, worst
            np.testing.assert_allclose(st.lb[0], lb, rtol=1e-6)
        # the two fp32 kernels agree far inside the mode's tolerance (same inputs, fp32 arithmetic both)
        nin = (inner[1] if inner else Lm) * B
        scale = np.maximum(np.abs(old), 1e-6 * nin * np.abs(pb["obs"]).max() ** 2)
        assert np.max(np.abs(got - old) / scale) < 1e-4, float(np.max(np.abs(got - old) / scale))
        assert np.all(np.isfinite(got))
        np.testing.assert_allclose(st.A_raw.sum(), nin if flags_wrap else nin - B, rtol=1e-5)
    finally:
        e.close()


@pytest.mark.parametrize("K,D,B,Lm", [(256, 64, 208, 257), (200, 48, 260, 129), (130, 40, 300, 129), (128, 32, 200, 257),
                                      (64, 64, 160, 257), (64, 48, 300, 129), (33, 40, 280, 129)])
def test_f32_wide_models_and_d_above_32_vs_fp64_oracle(K, D, B, Lm):
    """Round 5: the fp32 mode at the shapes it did not reach -- wide models
    (64 < K <= 256: k_emission_bf16x3d<WIDE> + k_scale_ll_f32 + k_sweeps_lin2<float> + k_stats_bf16x3w,
    ragged K included) and D in {40, 48, 64} at K <= 64 (k_emission_bf16x3d; K = 64: k_stats_bf16x3w) --
    against the fp64 C oracle: the mode's 1e-3 (canary 1e-4), the batch really ran in the fp32 format."""
    from pysvihmm_amd.engine import HipEngine
    from pysvihmm_amd import _lib as L
    from oracle import ref_c
    T = max(6000, B * 3 + Lm)
    pb = make_problem(K, D, T, seed=K + D, miss=0.05, sep=4.0)
    starts = np.random.default_rng(B).integers(0, T - Lm + 1, size=B)
    e = HipEngine(0, dtype="f32")
    try:
        e.set_obs(pb["obs"], pb["mask"])
        e.set_globals(pb["mod_init"], pb["ltran"])
        e.set_emission_niw(pb["mu"], pb["sigma"], pb["kappa"], pb["nu"])
        st = e.estep(starts, Lm, flags=L.TRANS_WRAP)
        assert e.precision() == ("f32", True)
        ref = ref_c.estep_minibatch(pb["obs"], pb["mask"], starts, Lm, pb["mod_init"], pb["ltran"], pb["mu"],
                                    pb["sigma"], pb["kappa"], pb["nu"], flags=2, threads=effective_cores())
        A, xbar, neff, S, lb = unpack(ref, K, D)
        sc = B * Lm
        xs = np.abs(pb["obs"]).max()
        worst = max(_close(st.A_raw, A, sc, "A"), _close(st.neff, neff, sc, "neff"),
                    _close(st.xbar, xbar, sc * xs, "xbar"), _close(st.S, S, sc * xs ** 2, "S"))
        assert worst < 1e-4, worst
        np.testing.assert_allclose(st.lb[0], lb, rtol=1e-6)
        for b in (0, B - 1):
            x = pb["obs"][starts[b]:starts[b] + Lm].copy()
            ll = ref_c.lliks_niw(x, pb["mu"], pb["sigma"], pb["kappa"], pb["nu"])
            q, _ = ref_c.posterior(ref_c.forward(ll, pb["mod_init"], pb["ltran"]), ref_c.backward(ll, pb["ltran"]))
            got = e.read_rows("var_x", b * Lm, Lm)
            assert np.abs(got - q).max() < 1e-4
            assert np.abs(got.sum(1) - 1.0).max() < 1e-5
        # the inner-segment statistic of buffered windows on the same kernels, against the fp64 device path
        inner = (3, Lm - 7)
        st2 = e.estep(starts, Lm, flags=L.TRANS_WRAP, inner=inner)
        assert e.precision()[1]
        e.set_precision("f64")
        ref2 = e.estep(starts, Lm, flags=L.TRANS_WRAP, inner=inner)
        sc2 = B * inner[1]
        for a, b_, s_, w in ((st2.A_raw, ref2.A_raw, sc2, "A"), (st2.neff, ref2.neff, sc2, "neff"),
                             (st2.xbar, ref2.xbar, sc2 * xs, "xbar"), (st2.S, ref2.S, sc2 * xs ** 2, "S")):
            _close(a, b_, s_, "inner " + w)
    finally:
        e.close()


@pytest.mark.parametrize("B,form", [(64, 0), (64, 8), (100, 0), (31, 0)],
                         ids=["64w_32row_groups", "64w_128row", "100w_64row_groups", "31w_below_floor"])
def test_f32_s64_minibatch_vs_fp64_oracle(B, form):
    """The literal configs[2] minibatch (64 windows of 257 rows, K = 64, D = 32) in the fp32 mode against the
    fp64 C oracle: the minibatch forms of the centred bf16 emission kernel -- k_emission_bf16x3h<4, 1> (32-row
    workgroups, the state pairs split over four one-wave groups; up to ~2 tiles per CU), <2, 2> (64-row
    workgroups, two two-wave groups: 100 windows) and the 128-row k_emission_bf16x3<1> (variant 5 = 8) --, the
    one-wave register-resident sweep with fp64 arithmetic on float messages (k_wave_linr<float, double>) and the
    bf16 statistics kernel with six-tile feature groups.  31 windows lie below the bf16 kernels' 8192-row floor: the
    fp32 format with the fp64 feature GEMM writing float Eh (k_emission_orbit_ks) and the fp32-input MFMA statistics."""
    from pysvihmm_amd.engine import HipEngine
    from pysvihmm_amd import _lib as L
    from oracle import ref_c
    K, D, Lm, T = 64, 32, 257, 40000
    pb = make_problem(K, D, T, seed=64, miss=0.0, sep=4.0)
    starts = (np.arange(B, dtype=np.int64) * (T // B)) % (T - Lm)
    e = HipEngine(0, dtype="f32")
    try:
        if form:
            e.set_variant("emission_orbit", form)
        e.set_obs(pb["obs"], None)
        e.set_globals(pb["mod_init"], pb["ltran"])
        e.set_emission_niw(pb["mu"], pb["sigma"], pb["kappa"], pb["nu"])
        st = e.estep(starts, Lm, flags=L.TRANS_WRAP)
        assert e.precision() == ("f32", True)
        ref = ref_c.estep_minibatch(pb["obs"], None, starts, Lm, pb["mod_init"], pb["ltran"], pb["mu"], pb["sigma"],
                                    pb["kappa"], pb["nu"], flags=2, threads=effective_cores())
        A, xbar, neff, S, lb = unpack(ref, K, D)
        sc = B * Lm
        xs = np.abs(pb["obs"]).max()
        worst = max(_close(st.A_raw, A, sc, "A"), _close(st.neff, neff, sc, "neff"),
                    _close(st.xbar, xbar, sc * xs, "xbar"), _close(st.S, S, sc * xs ** 2, "S"))
        assert worst < 1e-4, worst
        np.testing.assert_allclose(st.lb[0], lb, rtol=1e-6)
    finally:
        e.close()


def test_f32_device_loop_enters_the_fp32_format_and_tracks_the_fp64_loop():
    """Round 5: a loop whose initial var_tran (the class's 1/K) lies below the mode's range for E[log A] used to
    run fp64 to its end; the host-side lower bound of var_tran lets it enter the fp32 format after its first
    steps.  Final state against the fp64 loop at the mode's tolerance."""
    from pysvihmm_amd import hmmsgd_metaobs
    from pysvihmm_amd.distributions import Gaussian
    K, D, T = 8, 4, 6000
    pb = make_problem(K, D, T, seed=5, miss=0.0, sep=5.0)
    obs = pb["obs"]
    np.random.seed(0)
    prior = np.array([Gaussian(mu_0=obs.mean(0), sigma_0=0.75 * np.cov(obs.T), kappa_0=0.01, nu_0=D + 2) for _ in range(K)])
    res = {}
    for dt in ("f64", "f32"):
        np.random.seed(1)
        m = hmmsgd_metaobs.VBHMM(obs.copy(), np.ones(K), np.ones((K, K)), prior, tau=1.0, kappa=0.7, metaobs_half=32,
                                 mb_sz=16, maxit=12, seed=2, dtype=dt)
        assert m._svi_device_ok()
        m.infer()
        res[dt] = (m.var_tran.copy(), np.array([g.mu_mf for g in m.var_emit]), m.elbo_vec.copy(), m.engine.precision())
    assert res["f64"][3] == ("f64", False)
    assert res["f32"][3] == ("f32", True)
    a, b = res["f32"], res["f64"]
    np.testing.assert_allclose(a[0], b[0], rtol=1e-3, atol=1e-3 * np.abs(b[0]).max())
    np.testing.assert_allclose(a[1], b[1], rtol=1e-3, atol=1e-3 * np.abs(b[1]).max())
    np.testing.assert_allclose(a[2], b[2], rtol=1e-3)
