"""Scaled linear-domain sweeps (fb variant 3, the E-step fast path for K <= 64 and large
batches) against the C oracle and the log-domain kernels: ragged K / B, edge lengths,
masked rows, far outliers (emission underflow), host-supplied lliks, on-demand logs."""
import numpy as np
import pytest

from helpers import make_problem, unpack

pytestmark = pytest.mark.gpu
RTOL = 1e-6


@pytest.fixture(scope="module")
def eng():
    from pysvihmm_amd.engine import HipEngine
    e = HipEngine(0)
    yield e
    e.close()


def _push(eng, pb, mask=True):
    eng.set_obs(pb["obs"], pb["mask"] if mask else None)
    eng.set_globals(pb["mod_init"], pb["ltran"])
    eng.set_emission_niw(pb["mu"], pb["sigma"], pb["kappa"], pb["nu"])


@pytest.mark.parametrize("K", [3, 16, 17, 33, 48, 50, 64, 65, 100, 128, 200, 256])
def test_auto_path_vs_oracle(eng, K):
    """B >= 192 -> the automatic choice is the scaled sweep (K > 64: transition tile streamed
    from L2, emission scaled in a second pass); statistics vs the C oracle."""
    from pysvihmm_amd import _lib as L
    from oracle import ref_c
    D, T, Lm, B = 3, 6000, 11, 203            # B not a multiple of 16
    pb = make_problem(K, D, T, seed=300 + K, miss=0.05)
    rng = np.random.default_rng(K)
    starts = rng.integers(0, T - Lm + 1, size=B)
    _push(eng, pb)
    for flags in (L.TRANS_WRAP, L.TRANS_WRAP | L.MASK_AS_NAN):
        st = eng.estep(starts, Lm, flags=flags)
        ref = ref_c.estep_minibatch(pb["obs"], pb["mask"], starts, Lm, pb["mod_init"], pb["ltran"],
                                    pb["mu"], pb["sigma"], pb["kappa"], pb["nu"], flags=flags)
        A, xbar, neff, S, lb = unpack(ref, K, D)
        sc = B * Lm
        np.testing.assert_allclose(st.A_raw, A, rtol=RTOL, atol=1e-9 * sc)
        np.testing.assert_allclose(st.neff, neff, rtol=RTOL, atol=1e-9 * sc)
        np.testing.assert_allclose(st.xbar, xbar, rtol=RTOL, atol=1e-8 * sc)
        np.testing.assert_allclose(st.S, S, rtol=RTOL, atol=1e-7 * sc)
        np.testing.assert_allclose(st.lb[0], lb, rtol=1e-10)


@pytest.mark.parametrize("Lm", [1, 2, 3, 4, 40])
def test_edge_lengths_lin_equals_log(eng, Lm):
    K, D, T, B = 20, 2, 3000, 200
    pb = make_problem(K, D, T, seed=41)
    starts = (np.arange(B) * 13) % (T - Lm)
    _push(eng, pb, mask=False)
    out = {}
    for fv in (2, 3):
        eng.set_variant("fb", fv)
        r = eng.forward_backward(starts, Lm, want=("var_x", "local_lb"))
        out[fv] = r
    eng.set_variant("fb", 0)
    np.testing.assert_allclose(out[3]["var_x"], out[2]["var_x"], rtol=1e-9, atol=1e-13)
    np.testing.assert_allclose(out[3]["local_lb"], out[2]["local_lb"], rtol=1e-12)
    np.testing.assert_allclose(out[3]["var_x"].sum(-1), 1.0, rtol=1e-12)


def test_far_outliers_and_flat_rows(eng):
    """Observations hundreds of sigma away (ll ~ -1e5, every state but the nearest underflows
    in Eh) and masked rows (ll = 0): the integer exponents carry the scale exactly."""
    from pysvihmm_amd import _lib as L
    from oracle import ref_c
    K, D, T, Lm, B = 12, 4, 5000, 30, 160
    pb = make_problem(K, D, T, seed=5, miss=0.1, sep=6.0)
    rng = np.random.default_rng(0)
    idx = rng.integers(0, T, size=60)
    pb["obs"][idx] += rng.normal(size=(60, D)) * 300.0
    starts = (np.arange(B) * 31) % (T - Lm)
    eng.set_variant("fb", 3)
    _push(eng, pb)
    flags = L.TRANS_WRAP | L.MASK_AS_NAN
    st = eng.estep(starts, Lm, flags=flags)
    eng.set_variant("fb", 0)
    ref = ref_c.estep_minibatch(pb["obs"], pb["mask"], starts, Lm, pb["mod_init"], pb["ltran"],
                                pb["mu"], pb["sigma"], pb["kappa"], pb["nu"], flags=flags)
    A, xbar, neff, S, lb = unpack(ref, K, D)
    assert np.all(np.isfinite(st.buf))
    sc = B * Lm
    np.testing.assert_allclose(st.A_raw, A, rtol=RTOL, atol=1e-9 * sc)
    np.testing.assert_allclose(st.neff, neff, rtol=RTOL, atol=1e-9 * sc)
    np.testing.assert_allclose(st.xbar, xbar, rtol=RTOL, atol=1e-7 * sc)
    np.testing.assert_allclose(st.S, S, rtol=RTOL, atol=1e-4 * sc)
    np.testing.assert_allclose(st.lb[0], lb, rtol=1e-10)


def test_host_lliks_through_the_scaled_sweep(eng):
    """Generic emission plugins: lliks uploaded by the host, B >= 192."""
    from pysvihmm_amd import _lib as L
    from oracle import ref_numpy as R
    K, Lm, B = 7, 9, 200
    rng = np.random.default_rng(12)
    ll = rng.normal(size=(B, Lm, K)) * 30.0 - 50.0
    pb = make_problem(K, 2, 500, seed=3)
    eng.set_obs(pb["obs"], None)
    eng.set_globals(pb["mod_init"], pb["ltran"])
    eng.set_lliks(ll)
    r = eng.forward_backward(None, Lm, flags=L.USE_HOST_LLIKS, want=("var_x", "local_lb"), B=B)
    for b in (0, 57, B - 1):
        la = R.forward_msgs(ll[b], pb["mod_init"], pb["ltran"])
        lb = R.backward_msgs(ll[b], pb["ltran"])
        np.testing.assert_allclose(r["var_x"][b], R.posterior(la, lb), rtol=RTOL, atol=1e-13)
        np.testing.assert_allclose(r["local_lb"][b], R.local_lower_bound(la), rtol=1e-12)
        # logs on demand; the uploaded lliks come back untouched
        np.testing.assert_allclose(eng.read_rows("lalpha", b * Lm, Lm), la, rtol=1e-10, atol=1e-9)
        np.testing.assert_allclose(eng.read_rows("lbeta", b * Lm, Lm), lb, rtol=1e-10, atol=1e-9)
        np.testing.assert_array_equal(eng.read_rows("lliks", b * Lm, Lm), ll[b])


def test_logs_on_demand_and_staleness(eng):
    from pysvihmm_amd import _lib as L
    from oracle import ref_numpy as R
    K, D, T, Lm, B = 10, 3, 4000, 15, 220
    pb = make_problem(K, D, T, seed=8)
    starts = (np.arange(B) * 17) % (T - Lm)
    _push(eng, pb, mask=False)
    eng.estep(starts, Lm, flags=L.TRANS_WRAP, read=False)
    # a row range that straddles two windows
    got = eng.read_rows("lalpha", 5 * Lm - 4, 10)
    ref = []
    for b in (4, 5):
        s0 = int(starts[b])
        ll = R.lliks_niw(pb["obs"][s0:s0 + Lm], pb["mu"], pb["sigma"], pb["kappa"], pb["nu"])
        ref.append(R.forward_msgs(ll, pb["mod_init"], pb["ltran"]))
    ref = np.concatenate(ref)[Lm - 4:Lm + 6]
    np.testing.assert_allclose(got, ref, rtol=1e-10, atol=1e-9)
    full = eng.read_intermediate("lbeta", B, Lm)
    assert full.shape == (B, Lm, K) and np.all(full[:, -1] == 0.0)
    # new parameters: the cached range is still served, anything else is refused loudly
    eng.set_globals(pb["mod_init"], pb["ltran"] - 0.1)
    eng.read_rows("lbeta", 0, Lm)
    eng.estep(starts, Lm, flags=L.TRANS_WRAP, read=False)
    eng.set_globals(pb["mod_init"], pb["ltran"])
    with pytest.raises(RuntimeError, match="rebuilt on demand"):
        eng.read_rows("lalpha", 0, Lm)
    eng.read_rows("var_x", 0, Lm)            # posteriors are always resident


@pytest.mark.parametrize("K,B,inner", [(64, 203, None), (20, 75, (3, 5)), (33, 1000, None)])
def test_two_stream_pipeline_vs_oracle(eng, K, B, inner):
    """Forced two-stream E-step (emission/statistics of one half overlap the sweeps of the
    other): same statistics as the oracle, bit-identical when repeated (deterministic
    partials), and identical posteriors to the single-stream path."""
    from pysvihmm_amd import _lib as L
    from oracle import ref_c
    D, T, Lm = 3, 9000, 11
    pb = make_problem(K, D, T, seed=500 + K, miss=0.05)
    rng = np.random.default_rng(K + B)
    starts = rng.integers(0, T - Lm + 1, size=B)
    _push(eng, pb)
    flags = L.TRANS_WRAP | L.MASK_AS_NAN
    eng.set_variant("fb", 3)
    eng.set_variant("pipeline", 1)
    st0 = eng.estep(starts, Lm, flags=flags, inner=inner)
    q0 = eng.read_intermediate("var_x", B, Lm)
    eng.set_variant("pipeline", 2)
    st1 = eng.estep(starts, Lm, flags=flags, inner=inner)
    q1 = eng.read_intermediate("var_x", B, Lm)
    st2 = eng.estep(starts, Lm, flags=flags, inner=inner)
    la = eng.read_rows("lalpha", (B - 1) * Lm, Lm)     # logs on demand after the pipeline
    eng.set_variant("pipeline", 0); eng.set_variant("fb", 0)
    np.testing.assert_array_equal(st1.buf, st2.buf)
    np.testing.assert_array_equal(q0, q1)
    np.testing.assert_allclose(st1.buf, st0.buf, rtol=1e-12, atol=1e-12 * B * Lm)
    if inner is None:
        ref = ref_c.estep_minibatch(pb["obs"], pb["mask"], starts, Lm, pb["mod_init"], pb["ltran"],
                                    pb["mu"], pb["sigma"], pb["kappa"], pb["nu"], flags=flags)
        A, xbar, neff, S, lb = unpack(ref, K, D)
        sc = B * Lm
        np.testing.assert_allclose(st1.A_raw, A, rtol=RTOL, atol=1e-9 * sc)
        np.testing.assert_allclose(st1.neff, neff, rtol=RTOL, atol=1e-9 * sc)
        np.testing.assert_allclose(st1.xbar, xbar, rtol=RTOL, atol=1e-8 * sc)
        np.testing.assert_allclose(st1.S, S, rtol=RTOL, atol=1e-7 * sc)
        np.testing.assert_allclose(st1.lb[0], lb, rtol=1e-10)
    s0 = int(starts[B - 1])
    x = pb["obs"][s0:s0 + Lm].copy()
    x[pb["mask"][s0:s0 + Lm]] = np.nan
    ll = ref_c.lliks_niw(x, pb["mu"], pb["sigma"], pb["kappa"], pb["nu"])
    np.testing.assert_allclose(la, ref_c.forward(ll, pb["mod_init"], pb["ltran"]), rtol=1e-9, atol=1e-8)


@pytest.mark.parametrize("K,Lm,B", [(1, 7, 200), (2, 3000, 192), (64, 1500, 193)])
def test_extreme_shapes(eng, K, Lm, B):
    """Degenerate state counts and long windows on the scaled sweeps: exponents accumulate over
    thousands of steps (|log-likelihood| ~ 1e4 per window) without losing the statistics."""
    from pysvihmm_amd import _lib as L
    from oracle import ref_c
    D = 2
    T = Lm * 4 + 50
    pb = make_problem(K, D, T, seed=900 + K, miss=0.05, sep=4.0)
    rng = np.random.default_rng(K + Lm)
    starts = rng.integers(0, T - Lm + 1, size=B)
    _push(eng, pb)
    st = eng.estep(starts, Lm, flags=L.TRANS_WRAP)
    sub = np.unique(rng.integers(0, B, size=6))
    st2 = eng.estep(starts[sub], Lm, flags=L.TRANS_WRAP)      # same windows on the per-window path
    ref = ref_c.estep_minibatch(pb["obs"], pb["mask"], starts[sub], Lm, pb["mod_init"], pb["ltran"],
                                pb["mu"], pb["sigma"], pb["kappa"], pb["nu"], flags=2)
    np.testing.assert_allclose(st2.buf, ref, rtol=RTOL, atol=1e-7 * len(sub) * Lm)
    assert np.all(np.isfinite(st.buf))
    np.testing.assert_allclose(st.A_raw.sum(), B * Lm, rtol=1e-10)
    # the scaled path on those windows alone (forced) equals the per-window path
    eng.set_variant("fb", 3)
    st3 = eng.estep(starts[sub], Lm, flags=L.TRANS_WRAP)
    eng.set_variant("fb", 0)
    np.testing.assert_allclose(st3.buf, st2.buf, rtol=1e-8, atol=1e-9 * len(sub) * Lm)


@pytest.mark.parametrize("K,B", [(200, 2061), (256, 2100), (130, 2049)])
def test_wide_sweeps_32_windows_per_workgroup(eng, K, B):
    """More than 2048 windows on a wide model (K > 128): `k_sweeps_lin2` runs 32 windows per
    workgroup (two window tiles per wave sharing the streamed transition tile) and the
    transition statistic 128 x 64 blocks.  Ragged K and B (B % 32 != 0: clamped tail windows);
    bit-identical to the 16-window / 64 x 64 variants (variant 13 / 14) and equal to the C oracle."""
    from pysvihmm_amd import _lib as L
    from oracle import ref_c
    D, T, Lm = 2, 9000, 5
    pb = make_problem(K, D, T, seed=900 + K, miss=0.03)
    rng = np.random.default_rng(K + B)
    starts = rng.integers(0, T - Lm + 1, size=B)
    _push(eng, pb)
    flags = L.TRANS_WRAP
    st = eng.estep(starts, Lm, flags=flags)
    post = eng.forward_backward(starts, Lm, want=("var_x", "local_lb"))
    try:
        eng.set_variant(13, 1); eng.set_variant(14, 1)
        st16 = eng.estep(starts, Lm, flags=flags)
        post16 = eng.forward_backward(starts, Lm, want=("var_x", "local_lb"))
    finally:
        eng.set_variant(13, 0); eng.set_variant(14, 0)
    assert np.array_equal(st.buf, st16.buf)
    assert np.array_equal(post["var_x"], post16["var_x"]) and np.array_equal(post["local_lb"], post16["local_lb"])
    ref = ref_c.estep_minibatch(pb["obs"], pb["mask"], starts, Lm, pb["mod_init"], pb["ltran"],
                                pb["mu"], pb["sigma"], pb["kappa"], pb["nu"], flags=flags)
    A, xbar, neff, S, lb = unpack(ref, K, D)
    sc = B * Lm
    np.testing.assert_allclose(st.A_raw, A, rtol=RTOL, atol=1e-9 * sc)
    np.testing.assert_allclose(st.neff, neff, rtol=RTOL, atol=1e-9 * sc)
    np.testing.assert_allclose(st.xbar, xbar, rtol=RTOL, atol=1e-8 * sc)
    np.testing.assert_allclose(st.S, S, rtol=RTOL, atol=1e-7 * sc)
    np.testing.assert_allclose(st.lb[0], lb, rtol=1e-10)
