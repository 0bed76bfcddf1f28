"""DiagonalGaussian (BASELINE configs[0] "K=4 D=2 diagonal-Gaussian HMM, T=1000,
hmmbatchcd.py on CPU via gen_synthetic"): the emission class (third-party arithmetic absent from
/root/reference -> pinned to the mathematics, like tests/test_emission_formula.py), the oracle's
restatement, and the host logic of the three classes on the oracle engine.  CPU only."""
import numpy as np
import pytest
from scipy.special import gammaln

from pysvihmm_amd.distributions import DiagonalGaussian
from pysvihmm_amd import gen_synthetic, hmmbatchcd, hmmbatchsgd, hmmsgd_metaobs
from oracle import ref_numpy as R
from oracle.engine import OracleEngine


def _factor(D, seed):
    rng = np.random.default_rng(seed)
    np.random.seed(seed)
    g = DiagonalGaussian(mu_0=rng.normal(size=D), nus_0=0.3, alphas_0=2.5, betas_0=1.0 + rng.random(D))
    g.mf_mu = rng.normal(size=D) * 2
    g.mf_nus = 1.0 + 5 * rng.random(D)
    g.mf_alphas = 3.0 + 4 * rng.random(D)
    g.mf_betas = 0.5 + 3 * rng.random(D)
    return g, rng


def _draw(g, rng, n):
    s2 = 1.0 / rng.gamma(g.mf_alphas, 1.0 / g.mf_betas, size=(n, len(g.mf_mu)))
    mu = g.mf_mu + np.sqrt(s2 / g.mf_nus) * rng.normal(size=s2.shape)
    return mu, s2


def test_expected_log_likelihood_matches_monte_carlo():
    D, n = 3, 200000
    g, rng = _factor(D, 1)
    x = rng.normal(size=(4, D)) * 2.0
    mu, s2 = _draw(g, rng, n)
    cf = g.expected_log_likelihood(x)
    for i, xi in enumerate(x):
        v = (-0.5 * np.log(2 * np.pi * s2) - 0.5 * (xi - mu) ** 2 / s2).sum(1)
        assert abs(v.mean() - cf[i]) < 5 * v.std() / np.sqrt(n) + 1e-3
    # linear form in the 2 D + 1 features (what the device GEMM evaluates) == centred form
    a, b, c = g.mf_expectations()
    np.testing.assert_allclose((x ** 2).dot(a) + x.dot(b) + c, cf, rtol=1e-12, atol=1e-12)
    # the oracle's restatement
    ll = R.lliks_diag(x, g.mf_mu[None], g.mf_nus[None], g.mf_alphas[None], g.mf_betas[None])
    np.testing.assert_allclose(ll[:, 0], cf, rtol=1e-13, atol=1e-13)
    xn = x.copy(); xn[2, 1] = np.nan
    out = g.expected_log_likelihood(xn)
    assert np.isnan(out[2]) and np.all(np.isfinite(np.delete(out, 2)))


def test_vlb_is_minus_kl_and_zero_at_the_prior():
    D, n = 2, 200000
    g, rng = _factor(D, 3)
    mu, s2 = _draw(g, rng, n)

    def logp(m, nu, al, be):
        return (0.5 * np.log(nu / (2 * np.pi * s2)) - 0.5 * nu * (mu - m) ** 2 / s2
                + al * np.log(be) - gammaln(al) - (al + 1) * np.log(s2) - be / s2).sum(1)
    v = logp(g.mu_0, g.nus_0, g.alphas_0, g.betas_0) - logp(g.mf_mu, g.mf_nus, g.mf_alphas, g.mf_betas)
    assert abs(v.mean() - g.get_vlb()) < 5 * v.std() / np.sqrt(n) + 1e-3
    g.mf_mu, g.mf_nus, g.mf_alphas, g.mf_betas = g.mu_0, g.nus_0, g.alphas_0, g.betas_0
    assert abs(g.get_vlb()) < 1e-12


def test_conjugate_update_and_natural_parameters():
    D = 4
    g, rng = _factor(D, 5)
    x = rng.normal(size=(300, D)) * np.array([1, 2, 0.5, 3]) + np.array([1, -1, 3, 0])
    w = rng.random(300)
    g.meanfieldupdate(x, w)
    n = w.sum()
    xb = w.dot(x) / n
    # textbook normal-inverse-gamma posterior (per dimension)
    np.testing.assert_allclose(g.mf_nus, g.nus_0 + n)
    np.testing.assert_allclose(g.mf_mu, (g.nus_0 * g.mu_0 + n * xb) / (g.nus_0 + n))
    np.testing.assert_allclose(g.mf_alphas, g.alphas_0 + n / 2)
    ss = w.dot((x - xb) ** 2)
    np.testing.assert_allclose(g.mf_betas, g.betas_0 + 0.5 * ss + 0.5 * g.nus_0 * n / (g.nus_0 + n) * (xb - g.mu_0) ** 2,
                               rtol=1e-11)
    # additive natural parameters: prior + statistics == posterior
    sx, nn, sxx = R.diag_suffstats(x, w)
    eta = g.to_natural(g.mu_0, g.nus_0, g.alphas_0, g.betas_0) + np.stack([sx, np.full(D, nn), sxx, np.full(D, nn)])
    for a, b in zip(g.from_natural(eta), (g.mf_mu, g.mf_nus, g.mf_alphas, g.mf_betas)):
        np.testing.assert_allclose(a, b, rtol=1e-11)


def _configs0(seed=0, T=1000, K=4, D=2):
    """BASELINE configs[0]: K=4, D=2 diagonal-Gaussian HMM, T=1000, via gen_synthetic."""
    np.random.seed(seed)
    tran = 0.9 * np.eye(K) + 0.1 / (K - 1) * (1 - np.eye(K))
    means = np.array([[-6., -6.], [6., 6.], [-6., 6.], [6., -6.]])[:K, :D]
    emit = [DiagonalGaussian(mu=means[k], sigmas=np.ones(D)) for k in range(K)]
    obs, sts, _ = gen_synthetic.generate_data(tran, emit, T)
    # (initial means near the blobs, like the reference's demos initialise `mu`)
    prior = np.array([DiagonalGaussian(mu=means[k] + np.random.randn(D), mu_0=obs.mean(0), nus_0=0.01, alphas_0=2.0,
                                       betas_0=obs.var(0)) for k in range(K)])
    return obs, sts, prior


def test_configs0_hmmbatchcd_diag_fused_equals_literal():
    obs, sts, prior = _configs0()
    K = 4
    mask = np.random.default_rng(1).random(len(obs)) < 0.05
    runs = []
    for fused in (True, False):
        np.random.seed(3)
        m = hmmbatchcd.VBHMM(obs.copy(), np.ones(K), np.ones((K, K)), prior, mask=mask.copy(), maxit=8, sts=sts,
                             engine=OracleEngine())
        m.infer(fused=fused)
        runs.append(m)
    a, b = runs
    np.testing.assert_allclose(a.elbo_vec, b.elbo_vec, rtol=1e-9)
    np.testing.assert_allclose(a.var_tran, b.var_tran, rtol=1e-9)
    for k in range(K):
        np.testing.assert_allclose(a.var_emit[k].mf_mu, b.var_emit[k].mf_mu, rtol=1e-9, atol=1e-10)
        np.testing.assert_allclose(a.var_emit[k].mf_betas, b.var_emit[k].mf_betas, rtol=1e-9)
    assert np.all(np.diff(a.elbo_vec) > -1e-6)            # coordinate ascent
    assert a.hamming < 0.02                               # the four blobs are found


def test_metaobs_and_batchsgd_diag_fused_equals_literal():
    obs, sts, prior = _configs0(seed=2)
    K = 4
    for make in (lambda e: hmmsgd_metaobs.VBHMM(obs.copy(), np.ones(K), np.ones((K, K)), prior, tau=1.0, kappa=0.7,
                                                metaobs_half=8, mb_sz=6, maxit=5, seed=4, engine=e),
                 lambda e: hmmbatchsgd.VBHMM(obs.copy(), np.ones(K), np.ones((K, K)), prior, tau=1.0, kappa=0.7,
                                             maxit=4, engine=e)):
        runs = []
        for fused in (True, False):
            np.random.seed(3)
            m = make(OracleEngine())
            m.infer(fused=fused)
            runs.append(m)
        a, b = runs
        assert np.all(np.isfinite(a.elbo_vec))
        np.testing.assert_allclose(a.elbo_vec, b.elbo_vec, rtol=1e-8)
        np.testing.assert_allclose(a.var_tran, b.var_tran, rtol=1e-8)
        for k in range(K):
            np.testing.assert_allclose(a.var_emit[k].mf_mu, b.var_emit[k].mf_mu, rtol=1e-8, atol=1e-9)
            np.testing.assert_allclose(a.var_emit[k].mf_betas, b.var_emit[k].mf_betas, rtol=1e-8)


def test_metaobs_diag_engine_resident_loop_equals_host_loop():
    """Round 4: the SVI loop keeps the diagonal family's state inside the engine too
    (svi_begin_diag: natural-parameter blend, theta rebuild, -KL ELBO term); same trajectory as
    the host loop, with and without AdaGrad."""
    obs, sts, prior = _configs0(seed=5)
    K = 4
    for ada in (False, True):
        runs = []
        for dl in (None, False):
            np.random.seed(3)
            m = hmmsgd_metaobs.VBHMM(obs.copy(), np.ones(K), np.ones((K, K)), prior, tau=1.0, kappa=0.7,
                                     metaobs_half=8, mb_sz=6, maxit=5, seed=4, adagrad=ada, engine=OracleEngine())
            assert m._svi_family() == "diag" and m._svi_device_ok()
            m.infer(device_loop=dl)
            runs.append(m)
        a, b = runs
        np.testing.assert_allclose(a.elbo_vec, b.elbo_vec, rtol=1e-9)
        np.testing.assert_allclose(a.var_tran, b.var_tran, rtol=1e-9)
        for k in range(K):
            for name in ("mf_mu", "mf_nus", "mf_alphas", "mf_betas"):
                np.testing.assert_allclose(getattr(a.var_emit[k], name), getattr(b.var_emit[k], name), rtol=1e-9, atol=1e-10)
        np.testing.assert_allclose(a.var_x, b.var_x, rtol=1e-8, atol=1e-12)
        if ada:
            np.testing.assert_allclose(a.ada_G, b.ada_G, rtol=1e-10)


@pytest.mark.parametrize("hooks", ["full_predprob", "adaptive", "growBuffer"])
def test_device_loop_protocol_survives_validation_hooks_on_the_oracle_engine(hooks):
    """The host logic of the device-resident loop with mid-loop hooks that re-push the emission
    (GPU twin: tests/test_gpu_diag.py): OracleEngine models what an upload does to a
    running loop on libsvihmm_hip.so -- the loop's own family and shape: factors replaced, loop
    kept; anything else: the loop ends."""
    from tests.test_gpu_diag import _configs0
    from pysvihmm_amd import hmmsgd_metaobs
    from oracle.engine import OracleEngine
    obs, sts, prior = _configs0(seed=11, T=1500)
    K = 4
    mask = np.random.default_rng(2).random(len(obs)) < 0.08
    kw = dict(tau=1.0, kappa=0.7, metaobs_half=8, mb_sz=4, maxit=5, seed=4, mask=mask.copy())
    ikw = {}
    if hooks == "full_predprob":
        kw["full_predprob"] = True
    elif hooks == "adaptive":
        ikw = dict(adaptive=True, perIter=2, epsilon=1e-3)
    else:
        kw["growBuffer"] = True
        ikw = dict(perIter=2, epsilon=1e-3)
    res = []
    for dl in (None, False):
        np.random.seed(3)
        m = hmmsgd_metaobs.VBHMM(obs.copy(), np.ones(K), np.ones((K, K)), prior, engine=OracleEngine(), **kw)
        m.infer(device_loop=dl, **ikw)
        res.append(m)
    np.testing.assert_allclose(res[0].elbo_vec, res[1].elbo_vec, rtol=1e-9)
    np.testing.assert_allclose(res[0].var_tran, res[1].var_tran, rtol=1e-9)


def test_oracle_engine_upload_of_another_family_ends_the_loop():
    from oracle.engine import OracleEngine
    rng = np.random.default_rng(0)
    K, D, T = 3, 2, 200
    e = OracleEngine()
    e.set_obs(rng.normal(size=(T, D)))
    blk = tuple(np.abs(rng.normal(size=(K, D))) + 0.5 for _ in range(4))
    e.svi_begin_diag(np.ones((K, K)), np.ones((K, K)) * 2, blk, blk, 3)
    e.set_emission_diag(*blk)                      # own family and shape: the loop lives on
    e.svi_iteration(0, np.array([0, 50]), 2, 21, 2, 0.5, 1.0, 1.0)
    e.set_emission_niw(rng.normal(size=(K, D)), np.tile(np.eye(D), (K, 1, 1)), np.ones(K), np.full(K, D + 2.0))
    with pytest.raises(RuntimeError):
        e.svi_iteration(1, np.array([0, 50]), 2, 21, 2, 0.5, 1.0, 1.0)
