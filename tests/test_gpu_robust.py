"""Dynamic-range robustness of the scaled linear-domain paths (found by tests/fuzz_gpu.py).

* A window that STARTS in a state the initial distribution makes rare: psi(var_init) of a state
  with stationary mass 1e-4 is -1e4, the other states' emission terms are thousands of nats below
  the row maximum when the states are well separated -- the reference adds the two in the log
  domain (hmmbase.py:292, hmmsgd_metaobs.py:800); a product of two separately shifted
  exponentials is 0 for every state and the window's posterior turns into NaN.
* Transition expectations below the range of exp() (Dirichlet pseudo-counts of ~1e-3 and less:
  psi(1e-3) = -1000): the scaled recursions cannot represent them; the engine must take the
  exact log-domain recursion there.
"""
import numpy as np
import pytest
from scipy.special import digamma

from tests.helpers import make_problem, unpack

pytestmark = pytest.mark.gpu


def _rare_init_problem(K, D, T, seed):
    pb = make_problem(K, D, T, seed=seed, miss=0.05, sep=20.0)
    rng = np.random.default_rng(seed + 1)
    var_init = np.where(rng.random(K) < 0.5, 1e-4, 0.3) * (0.5 + rng.random(K))
    var_init[0] = 0.4
    var_init[K - 1] = 3e-5
    pb["mod_init"] = digamma(var_init + 1e-9) - digamma(var_init.sum() + 1e-9)
    return pb


def _check(e, L, ref_c, pb, starts, Lm, rtol=1e-6):
    K, D = pb["K"], pb["D"]
    par = (pb["mod_init"], pb["ltran"], pb["mu"], pb["sigma"], pb["kappa"], pb["nu"])
    e.set_obs(pb["obs"], pb["mask"])
    e.set_globals(pb["mod_init"], pb["ltran"])
    e.set_emission_niw(pb["mu"], pb["sigma"], pb["kappa"], pb["nu"])
    B = len(starts)
    sc = B * Lm
    xs = max(1.0, float(np.nanmax(np.abs(pb["obs"]))))
    for flags in (L.TRANS_WRAP, L.MASK_AS_NAN):
        st = e.estep(starts, Lm, flags=flags)
        assert np.all(np.isfinite(st.buf)), "non-finite statistics"
        ref = ref_c.estep_minibatch(pb["obs"], pb["mask"], starts, Lm, *par, flags=flags)
        g, r = unpack(st.buf, K, D), unpack(ref, K, D)
        np.testing.assert_allclose(g[0], r[0], rtol=rtol, atol=1e-9 * sc)
        np.testing.assert_allclose(g[1], r[1], rtol=rtol, atol=1e-9 * sc * xs)
        np.testing.assert_allclose(g[2], r[2], rtol=rtol, atol=1e-9 * sc)
        np.testing.assert_allclose(g[3], r[3], rtol=rtol, atol=1e-9 * sc * xs * xs)
        np.testing.assert_allclose(g[4], r[4], rtol=1e-9, atol=1e-6)
    # posteriors of a few windows against the oracle's log-domain recursion
    for b in np.unique(np.linspace(0, B - 1, 4).astype(int)):
        x = pb["obs"][starts[b]:starts[b] + Lm].copy()
        x[pb["mask"][starts[b]:starts[b] + Lm]] = np.nan
        ll = ref_c.lliks_niw(x, pb["mu"], pb["sigma"], pb["kappa"], pb["nu"])
        q, _ = ref_c.posterior(ref_c.forward(ll, pb["mod_init"], pb["ltran"]), ref_c.backward(ll, pb["ltran"]))
        np.testing.assert_allclose(e.read_rows("var_x", int(b) * Lm, Lm), q, rtol=1e-6, atol=1e-12)


@pytest.mark.parametrize("K,D,Lm,B", [(8, 4, 33, 20), (8, 4, 33, 300), (8, 4, 17, 1100), (64, 8, 9, 1100),
                                      (33, 8, 65, 64), (100, 4, 9, 200), (200, 4, 9, 200)])
def test_window_starts_in_a_rare_state(K, D, Lm, B):
    """Every sweep kernel of the scaled path (k_wave_lin4, k_wave_lin, k_sweeps_lin, the streamed
    wide-model variants): mod_init spans 1e4 nats and many windows start in a rare state."""
    from pysvihmm_amd.engine import HipEngine
    from pysvihmm_amd import _lib as L
    from oracle import ref_c
    T = max(4 * Lm, 600)
    pb = _rare_init_problem(K, D, T, seed=500 + K + B)
    starts = np.random.default_rng(B).integers(0, T - Lm + 1, size=B)
    e = HipEngine(0)
    try:
        _check(e, L, ref_c, pb, starts, Lm)
    finally:
        e.close()


@pytest.mark.parametrize("K,T", [(8, 3000), (64, 2500), (100, 2200)])
def test_chain_starts_in_a_rare_state(K, T):
    """The blocked scan's first boundary vector (k_chunk_scan / k_chunk_scan_wide)."""
    from pysvihmm_amd.engine import HipEngine
    from pysvihmm_amd import _lib as L
    from oracle import ref_c
    pb = _rare_init_problem(K, 4, T, seed=900 + K)
    # make sure the chain itself starts in a rare state
    first = int(pb["sts"][0])
    vi = np.exp(pb["mod_init"])
    var_init = np.full(K, 0.3); var_init[first] = 2e-5
    pb["mod_init"] = digamma(var_init + 1e-9) - digamma(var_init.sum() + 1e-9)
    pb["mask"][0] = False
    e = HipEngine(0)
    try:
        _check(e, L, ref_c, pb, np.zeros(1, dtype=np.int64), T)
    finally:
        e.close()


def test_f32_mode_rare_initial_state():
    from pysvihmm_amd.engine import HipEngine
    from pysvihmm_amd import _lib as L
    K, D, Lm, B = 16, 8, 33, 400
    T = 4000
    pb = _rare_init_problem(K, D, T, seed=77)
    starts = np.random.default_rng(5).integers(0, T - Lm + 1, size=B)
    res = {}
    for dt in ("f64", "f32"):
        e = HipEngine(0, dtype=dt)
        e.set_obs(pb["obs"], pb["mask"])
        e.set_globals(pb["mod_init"], pb["ltran"])
        e.set_emission_niw(pb["mu"], pb["sigma"], pb["kappa"], pb["nu"])
        res[dt] = e.estep(starts, Lm, flags=L.TRANS_WRAP)
        e.close()
    assert np.all(np.isfinite(res["f32"].buf))
    sc = B * Lm
    np.testing.assert_allclose(res["f32"].A_raw, res["f64"].A_raw, rtol=2e-3, atol=2e-4 * sc)
    np.testing.assert_allclose(res["f32"].neff, res["f64"].neff, rtol=2e-3, atol=2e-4 * sc)


def _sparse_ltran(K, T, seed):
    """E[log A] of a sticky model with tiny Dirichlet pseudo-counts: psi(1e-3) = -1000."""
    rng = np.random.default_rng(seed)
    vt = 1e-3 + rng.random((K, K)) * (rng.random((K, K)) < 0.2) * T
    vt[np.arange(K), np.arange(K)] += T
    return digamma(vt + 1e-9) - digamma(vt.sum(1)[:, None] + 1e-9)


@pytest.mark.parametrize("K,D,Lm,B,sep", [(3, 3, 2, 15, 60.0), (16, 15, 17, 31, 20.0), (48, 32, 33, 257, 20.0),
                                          (127, 8, 65, 40, 60.0), (8, 8, 257, 3, 0.0), (31, 1, 64, 191, 60.0),
                                          (65, 32, 9, 2, 20.0), (200, 4, 9, 200, 3.0)])
def test_transition_expectations_below_exp_range(K, D, Lm, B, sep):
    """Shapes the randomised campaign failed on before the log-domain route existed (NaN
    statistics, finite-but-wrong statistics, -inf in lalpha / lbeta)."""
    from pysvihmm_amd.engine import HipEngine
    from pysvihmm_amd import _lib as L
    from oracle import ref_c
    T = max(4 * Lm, 400)
    pb = make_problem(K, D, T, seed=3000 + K + B, miss=0.1, sep=sep)
    pb["ltran"] = _sparse_ltran(K, T, seed=K + B)
    assert pb["ltran"].min() < L.LTRAN_LINEAR_MIN
    starts = np.random.default_rng(B).integers(0, T - Lm + 1, size=B)
    e = HipEngine(0)
    try:
        _check(e, L, ref_c, pb, starts, Lm)
        for b in np.unique(np.linspace(0, B - 1, 3).astype(int)):
            fb = e.forward_backward(starts[b:b + 1], Lm, flags=L.MASK_AS_NAN)
            x = pb["obs"][starts[b]:starts[b] + Lm].copy()
            x[pb["mask"][starts[b]:starts[b] + Lm]] = np.nan
            ll = ref_c.lliks_niw(x, pb["mu"], pb["sigma"], pb["kappa"], pb["nu"])
            np.testing.assert_allclose(fb["lalpha"][0], ref_c.forward(ll, pb["mod_init"], pb["ltran"]), rtol=1e-9, atol=1e-7)
            np.testing.assert_allclose(fb["lbeta"][0], ref_c.backward(ll, pb["ltran"]), rtol=1e-9, atol=1e-7)
        # fp32 mode: outside a float's range the engine computes in fp64
        ref64 = e.estep(starts, Lm, flags=L.TRANS_WRAP).buf.copy()
        e.set_precision("f32")
        got = e.estep(starts, Lm, flags=L.TRANS_WRAP).buf
        np.testing.assert_allclose(got, ref64, rtol=1e-12, atol=0)
    finally:
        e.close()


def test_chain_with_sparse_transitions():
    from pysvihmm_amd.engine import HipEngine
    from pysvihmm_amd import _lib as L
    from oracle import ref_c
    K, D, T = 8, 4, 2500
    pb = make_problem(K, D, T, seed=31, miss=0.05, sep=20.0)
    pb["ltran"] = _sparse_ltran(K, T, seed=5)
    e = HipEngine(0)
    try:
        _check(e, L, ref_c, pb, np.zeros(1, dtype=np.int64), T)
    finally:
        e.close()


def test_f32_mode_moderately_sparse_transitions():
    """Pseudo-counts of 0.01 (psi = -100): inside a double's range (fast path), outside a float's
    -- the fp32 mode must not flush the messages."""
    from pysvihmm_amd.engine import HipEngine
    from pysvihmm_amd import _lib as L
    K, D, Lm, B, T = 16, 8, 33, 400, 4000
    pb = make_problem(K, D, T, seed=41, sep=3.0)
    rng = np.random.default_rng(3)
    vt = 1e-2 + rng.random((K, K)) * (rng.random((K, K)) < 0.3) * T
    vt[np.arange(K), np.arange(K)] += T
    pb["ltran"] = digamma(vt + 1e-9) - digamma(vt.sum(1)[:, None] + 1e-9)
    assert L.LTRAN_LINEAR_MIN < pb["ltran"].min() < L.LTRAN_F32_MIN
    starts = rng.integers(0, T - Lm + 1, size=B)
    res = {}
    for dt in ("f64", "f32"):
        e = HipEngine(0, dtype=dt)
        e.set_obs(pb["obs"], pb["mask"])
        e.set_globals(pb["mod_init"], pb["ltran"])
        e.set_emission_niw(pb["mu"], pb["sigma"], pb["kappa"], pb["nu"])
        res[dt] = e.estep(starts, Lm, flags=L.TRANS_WRAP).buf.copy()
        e.close()
    np.testing.assert_allclose(res["f32"], res["f64"], rtol=1e-12, atol=0)


def test_svi_loop_with_tiny_dirichlet_prior():
    """prior_tran = 1e-3: the device-resident loop refuses (svihmm_svi_begin), the class takes the
    host loop by itself, and the run equals the oracle engine's."""
    from pysvihmm_amd import hmmsgd_metaobs
    from pysvihmm_amd.distributions import Gaussian
    from pysvihmm_amd.engine import HipEngine
    from oracle.engine import OracleEngine
    K, D, T = 6, 3, 3000
    pb = make_problem(K, D, T, seed=12, sep=20.0)
    obs = pb["obs"]

    def model(engine):
        np.random.seed(4)
        prior = np.array([Gaussian(mu_0=obs.mean(0), sigma_0=0.75 * np.cov(obs.T), kappa_0=0.01, nu_0=D + 2)
                          for _ in range(K)])
        return hmmsgd_metaobs.VBHMM(obs, np.ones(K), 1e-3 * np.ones((K, K)), prior, tau=1.0, kappa=0.7,
                                    metaobs_half=8, mb_sz=6, maxit=6, seed=11, engine=engine)
    e = HipEngine(0)
    a = model(e)
    assert not a._svi_device_ok()
    a.infer()
    b = model(OracleEngine())
    b.infer()
    assert np.all(np.isfinite(a.var_tran)) and np.all(np.isfinite(a.elbo_vec))
    np.testing.assert_allclose(a.var_tran, b.var_tran, rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(a.elbo_vec, b.elbo_vec, rtol=1e-8)
    for k in range(K):
        np.testing.assert_allclose(a.var_emit[k].mu_mf, b.var_emit[k].mu_mf, rtol=1e-6, atol=1e-8)
    # the C ABI itself refuses the resident loop for such a model
    prior = a._prior_arrays()
    from pysvihmm_amd.distributions import niw_prior_logpart
    with pytest.raises(RuntimeError, match="PSEUDOCOUNT"):
        e.svi_begin(a.prior_tran, a.var_tran, prior, a._emission_arrays(), niw_prior_logpart(prior[1], prior[3]), 3)
    e.close()


@pytest.mark.parametrize("kind", ["metaobs_device_loop", "metaobs_host_loop", "batchcd"])
def test_classes_on_data_far_from_the_origin(kind):
    """The handle keeps the resident observations centred (include/svihmm.h, svihmm_set_obs; the
    C-ABI level tests are tests/test_gpu_coords.py): a sequence offset by 1e5 (|x| / sigma ~ 1e5:
    the form expanded around the origin would be off by 2e-5 in the log-likelihoods) gives what
    the same model gives on the un-shifted
    sequence (the model is shift-equivariant).  Tolerances: with the variational state on the
    device the whole loop runs in centred coordinates (1e-6); the host-side global steps are the
    reference's own arithmetic on raw second moments, which cancel 1e10 : 10 at this offset
    (1e-4, the same for the oracle engine)."""
    from pysvihmm_amd import hmmsgd_metaobs, hmmbatchcd
    from pysvihmm_amd.distributions import Gaussian
    from pysvihmm_amd.engine import HipEngine
    from oracle.engine import OracleEngine
    K, D, T = 4, 3, 1500
    OFF = 1e5
    pb = make_problem(K, D, T, seed=21, sep=3.0)
    mask = np.random.default_rng(2).random(T) < 0.05

    def model(engine, off):
        obs = pb["obs"] + off
        np.random.seed(6)
        prior = np.array([Gaussian(mu_0=pb["obs"].mean(0) + off, sigma_0=0.75 * np.cov(pb["obs"].T), kappa_0=0.01,
                                   nu_0=D + 2) for _ in range(K)])
        if kind == "batchcd":
            return hmmbatchcd.VBHMM(obs, np.ones(K), np.ones((K, K)), prior, mask=mask.copy(), maxit=4, engine=engine)
        return hmmsgd_metaobs.VBHMM(obs, np.ones(K), np.ones((K, K)), prior, tau=1.0, kappa=0.7,
                                    metaobs_half=10, mb_sz=8, mask=mask.copy(), maxit=6, seed=3, engine=engine)
    kw = {"device_loop": False} if kind == "metaobs_host_loop" else {}
    e = HipEngine(0)
    a, a0, b = model(e, OFF), model(e, 0.0), model(OracleEngine(), OFF)
    for m in (a, a0, b):
        m.infer(**kw)
    tol = 1e-6 if kind == "metaobs_device_loop" else 1e-4
    for ref, t in ((a0, tol), (b, 1e-4)):
        np.testing.assert_allclose(a.var_tran, ref.var_tran, rtol=t, atol=1e-8)
        np.testing.assert_allclose(a.var_x, ref.var_x, rtol=10 * t, atol=1e-7)
        for k in range(K):
            np.testing.assert_allclose(a.var_emit[k].mu_mf - OFF, ref.var_emit[k].mu_mf - (OFF if ref is b else 0.0),
                                       rtol=t, atol=10 * t)
            np.testing.assert_allclose(a.var_emit[k].sigma_mf, ref.var_emit[k].sigma_mf, rtol=10 * t, atol=10 * t)
    np.testing.assert_allclose(a.elbo_vec, a0.elbo_vec, rtol=tol)
    e.close()
