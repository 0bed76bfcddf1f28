"""GPU parity at the shapes of BASELINE.json's configs (the non-bench configs are
parity-test cases): C1 K=4 D=2 T=1000 batch CD; C2 K=16 D=8 T=100k full-chain E-step and
FFBS forward filter vs the C oracle; C5 K=256 D=64 full-covariance."""
import numpy as np
import pytest

from tests.helpers import ffbs_draws_exact, make_problem, unpack

pytestmark = pytest.mark.gpu
RTOL = 1e-6


def test_config1_batchcd_k4_d2_t1000():
    """configs[0]: hmmbatchcd on a K=4, D=2, T=1000 synthetic chain: the class on the HIP
    engine and on the injected oracle engine walk the same ELBO / parameter trajectory."""
    from pysvihmm_amd import hmmbatchcd
    from pysvihmm_amd.distributions import Gaussian
    from oracle.engine import OracleEngine
    pb = make_problem(4, 2, 1000, seed=4, miss=0.05)
    res = []
    for eng in (None, OracleEngine()):
        np.random.seed(12)
        pe = np.array([Gaussian(mu=pb["mu"][k], sigma=np.eye(2), mu_0=np.zeros(2),
                                sigma_0=0.75 * np.cov(pb["obs"].T), kappa_0=0.01, nu_0=4)
                       for k in range(4)])
        for k, g in enumerate(pe):
            g.sigma_mf = pb["sigma"][k]; g.kappa_mf = pb["kappa"][k]; g.nu_mf = pb["nu"][k]
        h = hmmbatchcd.VBHMM(pb["obs"].copy(), np.ones(4), np.ones((4, 4)), pe, mask=pb["mask"],
                             init_tran=pb["var_tran"], maxit=6, sts=pb["sts"], engine=eng)
        h.infer()
        res.append(h)
    a, b = res
    assert a.engine.name == "hip" and b.engine.name == "oracle"
    np.testing.assert_allclose(a.elbo_vec, b.elbo_vec, rtol=1e-9)
    np.testing.assert_allclose(a.var_tran, b.var_tran, rtol=RTOL, atol=1e-9)
    np.testing.assert_allclose(a.var_x, b.var_x, rtol=RTOL, atol=1e-10)
    assert a.hamming == b.hamming


def test_config2_full_chain_k16_d8_t100k():
    """configs[1]: single-GPU full-chain E-step, K=16 D=8 T=100k, against the C port of the
    reference recursions, and the hmm_fast forward filter variant."""
    from pysvihmm_amd.engine import HipEngine
    from pysvihmm_amd import _lib as L
    from oracle import ref_c
    K, D, T = 16, 8, 100000
    pb = make_problem(K, D, T, seed=16, miss=0.02)
    e = HipEngine(0)
    e.set_obs(pb["obs"], pb["mask"])
    e.set_globals(pb["mod_init"], pb["ltran"])
    e.set_emission_niw(pb["mu"], pb["sigma"], pb["kappa"], pb["nu"])
    r = e.forward_backward([0], T, flags=L.MASK_AS_NAN)
    x = pb["obs"].copy(); x[pb["mask"]] = np.nan
    ll = ref_c.lliks_niw(x, pb["mu"], pb["sigma"], pb["kappa"], pb["nu"])
    la = ref_c.forward(ll, pb["mod_init"], pb["ltran"])
    lb = ref_c.backward(ll, pb["ltran"])
    q, lz = ref_c.posterior(la, lb)
    np.testing.assert_allclose(r["lalpha"][0], la, rtol=1e-9, atol=1e-6)
    np.testing.assert_allclose(r["lbeta"][0], lb, rtol=1e-9, atol=1e-6)
    np.testing.assert_allclose(r["var_x"][0], q, rtol=RTOL, atol=1e-10)
    np.testing.assert_allclose(r["local_lb"][0], lz, rtol=1e-10)
    # whole-chain statistics, batch transition form
    st = e.estep([0], T, flags=0)
    ref = ref_c.estep_minibatch(pb["obs"], pb["mask"], [0], T, pb["mod_init"], pb["ltran"],
                                pb["mu"], pb["sigma"], pb["kappa"], pb["nu"], flags=0)
    A, xbar, neff, S, lbt = unpack(ref, K, D)
    np.testing.assert_allclose(st.A_raw, A, rtol=RTOL, atol=1e-6)
    np.testing.assert_allclose(st.xbar, xbar, rtol=RTOL, atol=1e-6)
    np.testing.assert_allclose(st.S, S, rtol=RTOL, atol=1e-5)
    np.testing.assert_allclose(st.neff, neff, rtol=RTOL, atol=1e-7)
    np.testing.assert_allclose(st.lb[0], lbt, rtol=1e-10)
    # FFBS: forward filter of the Cython variant + samples are valid states
    DE = np.finfo(np.float64).eps
    logA = np.log(pb["var_tran"] + DE)
    e.set_globals(pb["mod_init"], logA)
    u = np.random.default_rng(0).random(T)
    z, laf = e.ffbs(logA, u)
    ll0 = ref_c.lliks_niw(pb["obs"], pb["mu"], pb["sigma"], pb["kappa"], pb["nu"])
    np.testing.assert_allclose(laf, ref_c.forward(ll0, pb["mod_init"], logA), rtol=1e-9, atol=1e-6)
    assert z.min() >= 0 and z.max() < K
    # every draw of the path, row by row: z[t] is the inverse-CDF draw of softmax(lalpha[t] +
    # logA[:, z[t+1]]) at u[t] (hmm_fast.pyx:97-122) -- exact outside 1e-12 of a CDF step
    bad, near = ffbs_draws_exact(z, laf, logA, u)
    assert bad == 0, "%d draws differ from the sequential sampler (%d within rounding of a CDF step)" % (bad, near)
    assert near < 10
    # ... and the sampled path follows the true segmentation on this well-separated chain: after the
    # best relabelling (the variational means sit next to the true ones) it agrees on > 90 % of the rows
    # (the sequential sampler on the C port's filter: 94.8 %)
    from pysvihmm_amd.util import munkres_match
    perm = munkres_match(pb["sts"], z, K)
    assert np.mean(perm[z] == pb["sts"]) > 0.90
    e.close()


@pytest.mark.parametrize("var", [2, 3])
def test_config5_k256_d64_full_cov(var):
    """configs[4] shape (K=256, D=64 full covariance) at a small T."""
    from pysvihmm_amd.engine import HipEngine
    from pysvihmm_amd import _lib as L
    from oracle import ref_c
    K, D, T, Lm, B = 256, 64, 3000, 33, 12
    pb = make_problem(K, D, T, seed=256, miss=0.05)
    starts = np.random.default_rng(5).integers(0, T - Lm, size=B)
    e = HipEngine(0)
    e.set_variant("stats", var)                 # 2: the double-buffered statistics GEMM, 3: the pipelined kernels
    e.set_obs(pb["obs"], pb["mask"])
    e.set_globals(pb["mod_init"], pb["ltran"])
    e.set_emission_niw(pb["mu"], pb["sigma"], pb["kappa"], pb["nu"])
    st = e.estep(starts, Lm, flags=L.TRANS_WRAP)
    ref = ref_c.estep_minibatch(pb["obs"], pb["mask"], starts, Lm, pb["mod_init"], pb["ltran"],
                                pb["mu"], pb["sigma"], pb["kappa"], pb["nu"], flags=2)
    A, xbar, neff, S, lbt = unpack(ref, K, D)
    sc = B * Lm
    np.testing.assert_allclose(st.A_raw, A, rtol=RTOL, atol=1e-9 * sc)
    np.testing.assert_allclose(st.xbar, xbar, rtol=RTOL, atol=1e-8 * sc)
    np.testing.assert_allclose(st.neff, neff, rtol=RTOL, atol=1e-9 * sc)
    np.testing.assert_allclose(st.S, S, rtol=RTOL, atol=1e-7 * sc)
    np.testing.assert_allclose(st.lb[0], lbt, rtol=1e-9)
    e.close()
