"""Diagonal-covariance Gaussian family on the device (svihmm_set_emission_diag):
HIP vs the oracle engine (NumPy lliks + the C port's recursions) on log-likelihoods, posteriors
and the packed statistics [A_raw | xbar | neff | xsq | lb]; BASELINE configs[0] (K=4, D=2,
T=1000, hmmbatchcd) and the headline shape K=64, D=32; the classes end to end."""
import numpy as np
import pytest

from tests.helpers import make_problem

pytestmark = pytest.mark.gpu


def _diag_params(pb, seed):
    rng = np.random.default_rng(seed)
    K, D = pb["K"], pb["D"]
    return (pb["mu"], 0.5 + 4 * rng.random((K, D)), 2.0 + 5 * rng.random((K, D)), 1.0 + 6 * rng.random((K, D)))


def _close(a, b, K, D, rows, xs, rtol=1e-6):
    np.testing.assert_allclose(a.A_raw, b.A_raw, rtol=rtol, atol=1e-9 * rows)
    np.testing.assert_allclose(a.xbar, b.xbar, rtol=rtol, atol=1e-9 * rows * xs)
    np.testing.assert_allclose(a.neff, b.neff, rtol=rtol, atol=1e-9 * rows)
    np.testing.assert_allclose(a.xsq, b.xsq, rtol=rtol, atol=1e-9 * rows * xs * xs)
    np.testing.assert_allclose(a.lb, b.lb, rtol=1e-9, atol=1e-6)


@pytest.mark.parametrize("K,D,T,B,Lm,off", [
    (4, 2, 1000, 1, 1000, 0.0),        # configs[0]: one window = the whole chain
    (4, 2, 1000, 30, 33, 0.0),
    (5, 3, 2000, 250, 17, 1e4),        # large batch (scaled MFMA sweeps), data far from the origin
    (16, 8, 4000, 40, 65, 0.0),
    (64, 32, 20000, 300, 65, -300.0),  # headline shape: 65 features + 64 transition rows
    (33, 7, 3000, 64, 40, 0.0),        # ragged against every tile
    (100, 5, 3000, 200, 9, 0.0),       # K > 64: wide kernels
])
def test_diag_estep_vs_oracle(K, D, T, B, Lm, off):
    from pysvihmm_amd.engine import HipEngine
    from pysvihmm_amd import _lib as L
    from oracle.engine import OracleEngine
    pb = make_problem(K, D, T, seed=K + D, miss=0.05, sep=3.0)
    obs = pb["obs"] + off
    mu, nus, al, be = _diag_params(pb, 1)
    mu = mu + off
    starts = np.random.default_rng(2).integers(0, T - Lm + 1, size=B)
    e, o = HipEngine(0), OracleEngine()
    try:
        for eng in (e, o):
            eng.set_obs(obs, pb["mask"])
            eng.set_globals(pb["mod_init"], pb["ltran"])
            eng.set_emission_diag(mu, nus, al, be)
        xs = max(1.0, abs(off))
        for flags in (L.TRANS_WRAP, L.MASK_AS_NAN):
            a = e.estep(starts, Lm, flags=flags)
            b = o.estep(starts, Lm, flags=flags)
            assert a.buf.shape == b.buf.shape == (K * K + 2 * K * D + K + 1,)
            _close(a, b, K, D, B * Lm, xs)
            np.testing.assert_allclose(e.read_packed().buf, a.buf, rtol=0, atol=0)
        nb = min(B, 3)
        np.testing.assert_allclose(e.loglik(starts[:nb], Lm), o.loglik(starts[:nb], Lm), rtol=1e-9, atol=1e-7)
        ra = e.forward_backward(starts[:nb], Lm)
        rb = o.forward_backward(starts[:nb], Lm)
        np.testing.assert_allclose(ra["var_x"], rb["var_x"], rtol=1e-6, atol=1e-10)
        np.testing.assert_allclose(ra["lalpha"], rb["lalpha"], rtol=1e-9, atol=1e-7)
        np.testing.assert_allclose(ra["lbeta"], rb["lbeta"], rtol=1e-9, atol=1e-7)
        np.testing.assert_allclose(ra["local_lb"], rb["local_lb"], rtol=1e-10)
        # switching family on the same handle: NIW factors, then the diagonal ones again
        if D <= 8:
            e.set_emission_niw(mu, np.broadcast_to(np.eye(D) * 2.0, (K, D, D)).copy(), np.ones(K), np.full(K, D + 3.0))
            assert e.estep(starts, Lm).buf.shape == (K * K + K * D + K + K * D * D + 1,)
            e.set_emission_diag(mu, nus, al, be)
            _close(e.estep(starts, Lm, flags=L.TRANS_WRAP), o.estep(starts, Lm, flags=L.TRANS_WRAP), K, D, B * Lm, xs)
    finally:
        e.close()


def test_diag_parameters_follow_the_centre():
    """The diagonal factors on the device live in centred coordinates like the NIW ones: an explicit
    shift of the centre and a re-upload of the observations under live factors change nothing."""
    from pysvihmm_amd.engine import HipEngine
    from pysvihmm_amd import _lib as L
    from oracle.engine import OracleEngine
    K, D, T, Lm = 6, 4, 1500, 33
    pb = make_problem(K, D, T, seed=9, miss=0.05)
    obs = pb["obs"] + np.array([300.0, -2e4, 0.0, 5.0])
    mu, nus, al, be = _diag_params(pb, 2)
    mu = mu + np.array([300.0, -2e4, 0.0, 5.0])
    starts = np.arange(40, dtype=np.int64) * 33
    e, o = HipEngine(0), OracleEngine()
    try:
        for eng in (e, o):
            eng.set_obs(obs, pb["mask"])
            eng.set_globals(pb["mod_init"], pb["ltran"])
            eng.set_emission_diag(mu, nus, al, be)
        ref = o.estep(starts, Lm, flags=L.TRANS_WRAP)
        _close(e.estep(starts, Lm, flags=L.TRANS_WRAP), ref, K, D, 40 * Lm, 2e4)
        e.shift_obs(np.array([1.0, -7.0, 0.25, 100.0]))
        _close(e.estep(starts, Lm, flags=L.TRANS_WRAP), ref, K, D, 40 * Lm, 2e4)
        e.set_obs(obs, pb["mask"])                      # new centre chosen, factors stay
        _close(e.estep(starts, Lm, flags=L.TRANS_WRAP), ref, K, D, 40 * Lm, 2e4)
        e.set_variant(9, 1)                             # ... also when the new centre is the origin
        e.set_obs(obs[:, :], pb["mask"])
        e.set_variant(9, 0)
        e.shift_obs(obs[~pb["mask"]].mean(0))
        _close(e.estep(starts, Lm, flags=L.TRANS_WRAP), ref, K, D, 40 * Lm, 2e4)
    finally:
        e.close()


def test_diag_refuses_bad_parameters():
    from pysvihmm_amd.engine import HipEngine
    pb = make_problem(4, 3, 300, seed=1)
    mu, nus, al, be = _diag_params(pb, 1)
    e = HipEngine(0)
    try:
        e.set_obs(pb["obs"], None)
        e.set_globals(pb["mod_init"], pb["ltran"])
        be2 = be.copy(); be2[2, 1] = -1.0
        with pytest.raises(RuntimeError):
            e.set_emission_diag(mu, nus, al, be2)
        with pytest.raises(RuntimeError, match="DIAG_MAX_D"):
            e.set_emission_diag(np.zeros((2, 200)), np.ones((2, 200)), np.ones((2, 200)), np.ones((2, 200)))
        e.set_emission_diag(mu, nus, al, be)
        assert np.all(np.isfinite(e.estep([0, 50], 33).buf))
    finally:
        e.close()


def test_diag_f32_mode_within_1e3():
    from pysvihmm_amd.engine import HipEngine
    from pysvihmm_amd import _lib as L
    K, D, T, B, Lm = 16, 8, 8000, 400, 17
    pb = make_problem(K, D, T, seed=3, sep=3.0)
    mu, nus, al, be = _diag_params(pb, 4)
    starts = np.random.default_rng(2).integers(0, T - Lm, size=B)
    e = HipEngine(0)
    try:
        e.set_obs(pb["obs"], None)
        e.set_globals(pb["mod_init"], pb["ltran"])
        e.set_emission_diag(mu, nus, al, be)
        ref = e.estep(starts, Lm, flags=L.TRANS_WRAP).buf.copy()
        e.set_precision("f32")
        got = e.estep(starts, Lm, flags=L.TRANS_WRAP).buf
        scale = np.maximum(np.abs(ref), 1e-6 * B * Lm)
        assert np.max(np.abs(got - ref) / scale) < 1e-3
    finally:
        e.close()


def _configs0(seed=0, T=1000, K=4, D=2):
    from pysvihmm_amd.distributions import DiagonalGaussian
    from pysvihmm_amd import gen_synthetic
    np.random.seed(seed)
    tran = 0.9 * np.eye(K) + 0.1 / (K - 1) * (1 - np.eye(K))
    means = np.array([[-6., -6.], [6., 6.], [-6., 6.], [6., -6.]])
    emit = [DiagonalGaussian(mu=means[k], sigmas=np.ones(D)) for k in range(K)]
    obs, sts, _ = gen_synthetic.generate_data(tran, emit, T)
    prior = np.array([DiagonalGaussian(mu=means[k] + np.random.randn(D), mu_0=obs.mean(0), nus_0=0.01, alphas_0=2.0,
                                       betas_0=obs.var(0)) for k in range(K)])
    return obs, sts, prior


@pytest.mark.parametrize("cls", ["batchcd", "batchsgd", "metaobs"])
def test_configs0_classes_hip_vs_oracle_engine(cls):
    """BASELINE configs[0] through the class surface: HIP engine == oracle engine."""
    from pysvihmm_amd import hmmbatchcd, hmmbatchsgd, hmmsgd_metaobs
    from pysvihmm_amd.engine import HipEngine
    from oracle.engine import OracleEngine
    obs, sts, prior = _configs0()
    K = 4
    mask = np.random.default_rng(1).random(len(obs)) < 0.05
    e = HipEngine(0)
    runs = []
    try:
        for eng in (e, OracleEngine()):
            np.random.seed(3)
            if cls == "batchcd":
                m = hmmbatchcd.VBHMM(obs.copy(), np.ones(K), np.ones((K, K)), prior, mask=mask.copy(), maxit=8, sts=sts,
                                     engine=eng)
            elif cls == "batchsgd":
                m = hmmbatchsgd.VBHMM(obs.copy(), np.ones(K), np.ones((K, K)), prior, tau=1.0, kappa=0.7,
                                      mask=mask.copy(), maxit=5, engine=eng)
            else:
                m = hmmsgd_metaobs.VBHMM(obs.copy(), np.ones(K), np.ones((K, K)), prior, tau=1.0, kappa=0.7,
                                         metaobs_half=8, mb_sz=6, mask=mask.copy(), maxit=6, seed=4, engine=eng)
            m.infer()
            runs.append(m)
    finally:
        e.close()
    a, b = runs
    np.testing.assert_allclose(a.elbo_vec, b.elbo_vec, rtol=1e-7)
    np.testing.assert_allclose(a.var_tran, b.var_tran, rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(a.var_x, b.var_x, rtol=1e-5, atol=1e-9)
    for k in range(K):
        for name in ("mf_mu", "mf_nus", "mf_alphas", "mf_betas"):
            np.testing.assert_allclose(getattr(a.var_emit[k], name), getattr(b.var_emit[k], name), rtol=1e-6, atol=1e-8)
    if cls == "batchcd":
        assert a.hamming < 0.02


@pytest.mark.parametrize("adagrad", [False, True])
def test_diag_device_loop_on_gpu(adagrad):
    """Round 4: the device-resident SVI loop for the diagonal family (svihmm_svi_begin_diag):
    HIP device loop == HIP host loop == oracle engine, data far from the origin included (the
    loop's means live in the handle's centred coordinates)."""
    from pysvihmm_amd import hmmsgd_metaobs
    from oracle.engine import OracleEngine
    obs, sts, prior0 = _configs0(seed=7, T=3000)
    K = 4
    for off in (0.0, 250.0):
        from pysvihmm_amd.distributions import DiagonalGaussian
        prior = np.array([DiagonalGaussian(mu=g.mu + off, sigmas=g.sigmas, mu_0=g.mu_0 + off, nus_0=g.nus_0,
                                           alphas_0=g.alphas_0, betas_0=g.betas_0) for g in prior0])
        runs = []
        for eng, dl in ((None, None), (None, False), (OracleEngine(), False)):
            np.random.seed(3)
            m = hmmsgd_metaobs.VBHMM(obs.copy() + off, np.ones(K), np.ones((K, K)), prior, tau=1.0, kappa=0.7,
                                     metaobs_half=16, mb_sz=8, maxit=6, seed=4, adagrad=adagrad, engine=eng)
            assert m._svi_family() == "diag" and m._svi_device_ok()
            m.infer(device_loop=dl)
            runs.append(m)
        a, b, c = runs
        assert a.engine.name == "hip"
        for o, rt in ((b, 1e-7), (c, 1e-6)):
            np.testing.assert_allclose(a.elbo_vec, o.elbo_vec, rtol=1e-7)
            np.testing.assert_allclose(a.var_tran, o.var_tran, rtol=rt, atol=1e-9)
            for k in range(K):
                for name in ("mf_mu", "mf_nus", "mf_alphas", "mf_betas"):
                    np.testing.assert_allclose(getattr(a.var_emit[k], name), getattr(o.var_emit[k], name), rtol=rt,
                                               atol=1e-8 * (1.0 + off))
            np.testing.assert_allclose(a.var_x, o.var_x, rtol=1e-5, atol=1e-9)


@pytest.mark.parametrize("hooks", ["full_predprob", "adaptive", "growBuffer"])
def test_diag_device_loop_survives_validation_hooks(hooks):
    """the validation hooks of infer() -- full_predprob (default schedule fires at
    it = 0), adaptive select_L, growBuffer -- re-push the emission mid-loop through
    svihmm_set_emission_diag, which used to end the running diagonal-family loop unconditionally:
    the next svihmm_svi_iteration / svihmm_svi_read_elbo then failed with 'call svihmm_svi_begin
    first'.  A re-push of the loop's own family and shape now keeps the loop (as the NIW upload always
    did); HIP device loop == HIP host loop == oracle engine."""
    from pysvihmm_amd import hmmsgd_metaobs
    from oracle.engine import OracleEngine
    obs, sts, prior = _configs0(seed=11, T=2500)
    K = 4
    mask = np.random.default_rng(2).random(len(obs)) < 0.08
    kw = dict(tau=1.0, kappa=0.7, metaobs_half=10, mb_sz=6, maxit=7, seed=4, mask=mask.copy())
    ikw = {}
    if hooks == "full_predprob":
        kw["full_predprob"] = True
    elif hooks == "adaptive":
        ikw = dict(adaptive=True, perIter=3, epsilon=1e-3)
    else:
        kw["growBuffer"] = True
        ikw = dict(perIter=3, epsilon=1e-3)
    runs = []
    for eng, dl in ((None, None), (None, False), (OracleEngine(), None)):
        np.random.seed(3)
        m = hmmsgd_metaobs.VBHMM(obs.copy(), np.ones(K), np.ones((K, K)), prior, engine=eng, **kw)
        assert m._svi_family() == "diag" and m._svi_device_ok()
        m.infer(device_loop=dl, **ikw)
        runs.append(m)
    a, b, c = runs
    assert a.engine.name == "hip"
    assert np.all(np.isfinite(a.elbo_vec))
    for o, rt in ((b, 1e-7), (c, 1e-6)):
        np.testing.assert_allclose(a.elbo_vec, o.elbo_vec, rtol=1e-7)
        np.testing.assert_allclose(a.var_tran, o.var_tran, rtol=rt, atol=1e-9)
        for k in range(K):
            for name in ("mf_mu", "mf_nus", "mf_alphas", "mf_betas"):
                np.testing.assert_allclose(getattr(a.var_emit[k], name), getattr(o.var_emit[k], name), rtol=rt, atol=1e-8)
    if hooks == "full_predprob":
        np.testing.assert_allclose(a.pred_logprob_full_mean, c.pred_logprob_full_mean, rtol=1e-6)
