"""Parity at BASELINE.json's full sizes.

* configs[2] literal shape: K=64, D=32, T=1e6, L=128 (Lm=257) -- (a) the epoch sweep over all
  3891 tiled windows (the bench workload) against the C oracle on EVERY window (the oracle's
  windows are dealt to the host cores the container may use; on a small host: a sample of 96 windows spread over the
  sequence) plus size-independent properties, per-window posteriors of 32 windows spread over
  the sequence; (b) ``hmmsgd_metaobs.VBHMM(metaobs_half=128, mb_sz=64).infer(maxit=3)`` on the
  HIP engine against the same class on the oracle engine.
* configs[4]: K=256, D=64 full covariance, Lm=257, B=208 windows -- the large-batch wide path
  (two-pass emission, k_sweeps_lin2, 64x64-block transition statistics) against the C oracle.
"""
import os

import numpy as np
import pytest

from tests.helpers import make_problem, unpack, effective_cores

pytestmark = pytest.mark.gpu
NCORE = effective_cores()


def _bench_problem(eng, T, K=64, D=32, seed=8675309):
    """The bench's own workload: sequence generated in HBM, variational state of SURVEY 8d."""
    import bench
    rs = np.random.RandomState(seed)
    tran = 0.9 * np.eye(K) + 0.1 / (K - 1) * (1.0 - np.eye(K))
    means = rs.normal(0.0, 5.0, size=(K, D))
    chols = np.broadcast_to(np.eye(D), (K, D, D)).copy()
    eng.generate(tran, means, chols, T, seed=seed)
    obs, sts = eng.read_generated()
    pb = bench.variational_state(rs, means, obs[:20000], K, D, T)
    pb["obs"], pb["sts"] = obs, sts
    return pb


def test_config3_epoch_sweep_t1e6_vs_oracle():
    from pysvihmm_amd.engine import HipEngine
    from pysvihmm_amd import _lib as L
    from oracle import ref_c
    K, D, T, Lm = 64, 32, 1000000, 257
    e = HipEngine(0)
    pb = _bench_problem(e, T)
    B = T // Lm
    starts = np.arange(B, dtype=np.int64) * Lm
    par = (pb["mod_init"], pb["ltran"], pb["mu"], pb["sigma"], pb["kappa"], pb["nu"])
    e.set_globals(pb["mod_init"], pb["ltran"])
    e.set_emission_niw(pb["mu"], pb["sigma"], pb["kappa"], pb["nu"])
    st = e.estep(starts, Lm, flags=L.TRANS_WRAP)
    n = B * Lm
    obs = pb["obs"][:n]
    # size-independent properties
    assert np.all(np.isfinite(st.buf))
    assert abs(st.A_raw.sum() / n - 1.0) < 1e-11          # posteriors sum to one, Lm products per window
    assert abs(st.neff.sum() / n - 1.0) < 1e-11
    np.testing.assert_allclose(st.xbar.sum(0), obs.sum(0), rtol=1e-9, atol=1e-6)
    np.testing.assert_allclose(st.S.sum(0), obs.T.dot(obs), rtol=1e-9)
    assert np.all(np.linalg.eigvalsh(st.S) > -1e-6 * n)
    # per-window posteriors of 32 windows spread over the sequence (oracle window by window)
    for b in np.linspace(0, B - 1, 32).astype(int):
        x = pb["obs"][starts[b]:starts[b] + Lm]
        ll = ref_c.lliks_niw(x, pb["mu"], pb["sigma"], pb["kappa"], pb["nu"])
        la = ref_c.forward(ll, pb["mod_init"], pb["ltran"])
        q, lz = ref_c.posterior(la, ref_c.backward(ll, pb["ltran"]))
        got = e.read_rows("var_x", int(b) * Lm, Lm)
        np.testing.assert_allclose(got, q, rtol=1e-6, atol=1e-12)
    # the whole step against the oracle
    if NCORE >= 12:
        sel = starts
        got = st.buf
    else:                                                    # small host: 96 windows, own E-step
        sel = starts[np.linspace(0, B - 1, 96).astype(int)]
        got = e.estep(sel, Lm, flags=L.TRANS_WRAP).buf
    ref = ref_c.estep_minibatch(pb["obs"], None, sel, Lm, *par, flags=2, threads=NCORE)
    A, xbar, neff, S, lb = unpack(ref, K, D)
    g = unpack(got, K, D)
    sc = len(sel) * Lm
    np.testing.assert_allclose(g[0], A, rtol=1e-6, atol=1e-10 * sc)
    np.testing.assert_allclose(g[1], xbar, rtol=1e-6, atol=1e-9 * sc)
    np.testing.assert_allclose(g[2], neff, rtol=1e-6, atol=1e-10 * sc)
    np.testing.assert_allclose(g[3], S, rtol=1e-6, atol=1e-8 * sc)
    np.testing.assert_allclose(g[4], lb, rtol=1e-11)
    e.close()


def test_config3_class_infer_l128_s64_vs_oracle_engine():
    """configs[2] as the reference words it: hmmsgd_metaobs, metaobs half-length 128, minibatch 64,
    T=1e6 -- three SVI iterations on the HIP engine == the same class on the oracle engine."""
    from pysvihmm_amd import hmmsgd_metaobs
    from pysvihmm_amd.distributions import Gaussian
    from pysvihmm_amd.engine import HipEngine
    from oracle.engine import OracleEngine
    K, D, T = 64, 32, 1000000
    e = HipEngine(0)
    pb = _bench_problem(e, T)
    obs = pb["obs"]
    head = obs[:20000]

    def model(engine):
        np.random.seed(3)
        prior = np.array([Gaussian(mu_0=head.mean(0), sigma_0=0.75 * np.cov(head.T), kappa_0=0.01, nu_0=D + 2)
                          for _ in range(K)])
        return hmmsgd_metaobs.VBHMM(obs, np.ones(K), np.ones((K, K)), prior, tau=1.0, kappa=0.7,
                                    metaobs_half=128, mb_sz=64, maxit=3, seed=7, engine=engine)
    a = model(e)
    a.infer()
    b = model(OracleEngine())
    b.infer()
    np.testing.assert_allclose(a.var_tran, b.var_tran, rtol=1e-6, atol=1e-9)
    for k in range(K):
        np.testing.assert_allclose(a.var_emit[k].mu_mf, b.var_emit[k].mu_mf, rtol=1e-6, atol=1e-8)
        np.testing.assert_allclose(a.var_emit[k].sigma_mf, b.var_emit[k].sigma_mf, rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(a.var_emit[k].kappa_mf, b.var_emit[k].kappa_mf, rtol=1e-9)
    np.testing.assert_allclose(a.elbo_vec, b.elbo_vec, rtol=1e-9)
    np.testing.assert_allclose(a.var_x, b.var_x, rtol=1e-6, atol=1e-12)
    np.testing.assert_allclose(a.lalpha, b.lalpha, rtol=1e-9, atol=1e-7)
    assert a.cur_mo.i1 == b.cur_mo.i1
    e.close()


@pytest.mark.skipif(NCORE < 8, reason="the K=256 oracle needs ~1 s per window: multi-core host only")
def test_config5_k256_d64_large_batch_vs_oracle():
    """configs[4] shape at the batch size that selects the wide large-batch kernels."""
    from pysvihmm_amd.engine import HipEngine
    from pysvihmm_amd import _lib as L
    from oracle import ref_c
    K, D, Lm, B = 256, 64, 257, 208
    T = B * Lm + 500
    pb = make_problem(K, D, T, seed=2560, miss=0.03, sep=4.0)
    starts = (np.arange(B, dtype=np.int64) * Lm + 137) % (T - Lm)
    e = HipEngine(0)
    e.set_obs(pb["obs"], pb["mask"])
    e.set_globals(pb["mod_init"], pb["ltran"])
    e.set_emission_niw(pb["mu"], pb["sigma"], pb["kappa"], pb["nu"])
    par = (pb["mod_init"], pb["ltran"], pb["mu"], pb["sigma"], pb["kappa"], pb["nu"])
    for flags in (L.TRANS_WRAP, L.TRANS_WRAP | L.MASK_AS_NAN):
        e.profile(True); e.profile_reset()
        st = e.estep(starts, Lm, flags=flags)
        prof = e.profile_read(); e.profile(False)
        # scaled sweeps; the statistics GEMM forms the posteriors itself (no k_lin_posterior pass)
        assert "forward_backward" in prof and "posterior" not in prof
        ref = ref_c.estep_minibatch(pb["obs"], pb["mask"], starts, Lm, *par, flags=flags, threads=NCORE)
        A, xbar, neff, S, lb = unpack(ref, K, D)
        sc = B * Lm
        np.testing.assert_allclose(st.A_raw, A, rtol=1e-6, atol=1e-10 * sc)
        np.testing.assert_allclose(st.xbar, xbar, rtol=1e-6, atol=1e-9 * sc)
        np.testing.assert_allclose(st.neff, neff, rtol=1e-6, atol=1e-10 * sc)
        np.testing.assert_allclose(st.S, S, rtol=1e-6, atol=1e-8 * sc)
        np.testing.assert_allclose(st.lb[0], lb, rtol=1e-10)
    # posteriors of a few windows, row by row
    for b in (0, 77, B - 1):
        x = pb["obs"][starts[b]:starts[b] + Lm].copy()
        x[pb["mask"][starts[b]:starts[b] + Lm]] = np.nan
        ll = ref_c.lliks_niw(x, pb["mu"], pb["sigma"], pb["kappa"], pb["nu"])
        q, _ = ref_c.posterior(ref_c.forward(ll, pb["mod_init"], pb["ltran"]), ref_c.backward(ll, pb["ltran"]))
        np.testing.assert_allclose(e.read_rows("var_x", b * Lm, Lm), q, rtol=1e-6, atol=1e-12)
    e.close()


@pytest.mark.skipif(NCORE < 8, reason="the K=256 oracle needs ~1 s per window: multi-core host only")
def test_config5_k256_d64_t1e6_epoch_sweep():
    """configs[4] at its FULL size (K=256, D=64 full covariance, T=1e6, the 3891-window epoch sweep
    bench.py's `c5_k256_d64` record times): size-independent properties of the whole step, a spread
    sample of 48 windows' statistics against the C oracle (own E-step on those windows: identical
    kernels, the batch is above the wide-kernel threshold... at 48 the wave kernels run, so the
    sample is ALSO checked window by window against the epoch's own posteriors), and the posteriors
    of 6 windows of the epoch step row by row."""
    from pysvihmm_amd.engine import HipEngine
    from pysvihmm_amd.gen_synthetic import generate_data_fast
    from pysvihmm_amd import _lib as L
    import bench
    from oracle import ref_c
    K, D, T, Lm = 256, 64, 1000000, 257
    rs = np.random.RandomState(bench.SEED + 4)
    rng = np.random.default_rng(bench.SEED + 4)
    tran = 0.9 * np.eye(K) + 0.1 / (K - 1) * (1.0 - np.eye(K))
    means = rs.normal(0.0, 5.0, size=(K, D))
    obs, _ = generate_data_fast(tran, means, None, T, rng)
    pw = bench.variational_state(rs, means, obs[:20000], K, D, T)
    par = (pw["mod_init"], pw["ltran"], pw["mu"], pw["sigma"], pw["kappa"], pw["nu"])
    B = T // Lm
    starts = np.arange(B, dtype=np.int64) * Lm
    e = HipEngine(0)
    try:
        e.set_obs(obs, None)
        e.set_globals(pw["mod_init"], pw["ltran"])
        e.set_emission_niw(pw["mu"], pw["sigma"], pw["kappa"], pw["nu"])
        st = e.estep(starts, Lm, flags=L.TRANS_WRAP)
        n = B * Lm
        x = obs[:n]
        assert np.all(np.isfinite(st.buf))
        assert abs(st.A_raw.sum() / n - 1.0) < 1e-11 and abs(st.neff.sum() / n - 1.0) < 1e-11
        np.testing.assert_allclose(st.xbar.sum(0), x.sum(0), rtol=1e-9, atol=1e-6)
        np.testing.assert_allclose(st.S.sum(0), x.T.dot(x), rtol=1e-9)
        np.testing.assert_allclose(st.S, np.swapaxes(st.S, 1, 2), rtol=0, atol=0)
        # posteriors of the epoch step, 6 windows spread over the sequence, row by row
        qs = {}
        for b in np.linspace(0, B - 1, 6).astype(int):
            xw = obs[starts[b]:starts[b] + Lm]
            ll = ref_c.lliks_niw(xw, pw["mu"], pw["sigma"], pw["kappa"], pw["nu"])
            q, _ = ref_c.posterior(ref_c.forward(ll, pw["mod_init"], pw["ltran"]), ref_c.backward(ll, pw["ltran"]))
            got = e.read_rows("var_x", int(b) * Lm, Lm)
            np.testing.assert_allclose(got, q, rtol=1e-6, atol=1e-12)
            qs[int(b)] = got
        # marginal consistency of the whole step: row / column sums of the transition statistic
        qsum = np.zeros(K)
        for r0 in range(0, n, 65536):
            qsum += e.read_rows("var_x", r0, min(65536, n - r0)).sum(0)
        np.testing.assert_allclose(st.A_raw.sum(1), qsum, rtol=1e-9)
        np.testing.assert_allclose(st.A_raw.sum(0), qsum, rtol=1e-9)
        np.testing.assert_allclose(st.neff, qsum, rtol=1e-9)
        # 208 windows spread over the sequence (the wide large-batch kernels) against the C oracle
        sel = starts[np.linspace(0, B - 1, 208).astype(int)]
        got = e.estep(sel, Lm, flags=L.TRANS_WRAP)
        ref = ref_c.estep_minibatch(obs, None, sel, Lm, *par, flags=2, threads=NCORE)
        A, xbar, neff, S, lb = unpack(ref, K, D)
        sc = len(sel) * Lm
        np.testing.assert_allclose(got.A_raw, A, rtol=1e-6, atol=1e-10 * sc)
        np.testing.assert_allclose(got.xbar, xbar, rtol=1e-6, atol=1e-9 * sc)
        np.testing.assert_allclose(got.neff, neff, rtol=1e-6, atol=1e-10 * sc)
        np.testing.assert_allclose(got.S, S, rtol=1e-6, atol=1e-8 * sc)
        np.testing.assert_allclose(got.lb[0], lb, rtol=1e-10)
        # ---- the same configuration in the fp32 mode (round 5: k_emission_bf16x3d<WIDE>, k_scale_ll_f32,
        #      k_sweeps_lin2<float>, k_stats_bf16x3w): north_star's 1e-3 against the fp64 C oracle on the 208
        #      spread windows, against the fp64 epoch step on all 3891, posteriors of the 6 windows
        def close32(a, b, scale, what):
            err = np.abs(a - b) / (np.abs(b) + 1e-6 * scale)
            assert err.max() < 1e-3, (what, float(err.max()))
        xs = float(np.abs(obs[:n]).max())
        e.set_precision("f32")
        g32 = e.estep(sel, Lm, flags=L.TRANS_WRAP)
        assert e.precision() == ("f32", True)
        for a, b, sc_, w in ((g32.A_raw, A, sc, "A"), (g32.neff, neff, sc, "neff"), (g32.xbar, xbar, sc * xs, "xbar"),
                             (g32.S, S, sc * xs * xs, "S")):
            close32(a, b, sc_, "sample " + w)
        np.testing.assert_allclose(g32.lb[0], lb, rtol=1e-6)
        s32 = e.estep(starts, Lm, flags=L.TRANS_WRAP)
        assert e.precision() == ("f32", True)
        assert np.all(np.isfinite(s32.buf)) and abs(s32.A_raw.sum() / n - 1.0) < 1e-5
        for a, b, sc_, w in ((s32.A_raw, st.A_raw, n, "A"), (s32.neff, st.neff, n, "neff"), (s32.xbar, st.xbar, n * xs, "xbar"),
                             (s32.S, st.S, n * xs * xs, "S")):
            close32(a, b, sc_, "epoch " + w)
        np.testing.assert_allclose(s32.lb[0], st.lb[0], rtol=1e-6)
        for b, q in qs.items():
            assert np.abs(e.read_rows("var_x", b * Lm, Lm) - q).max() < 1e-4
    finally:
        e.close()
