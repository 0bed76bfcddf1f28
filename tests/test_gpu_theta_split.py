"""NIW -> theta with both halves of the wave at work (k_niw_to_theta_wave32s, round 6) against the builder of rounds
2-5 (variant 13 = 2): every element goes through the same operations in the same order, so the expected
log-likelihoods, the fp32 mode's centred factors and a resident loop's whole trajectory (with the global step still a
launch of its own: variant 13 = 3) must agree BIT FOR BIT; the default, which runs the global step inside the builder's
launch, agrees with them to rounding."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("K,D", [(64, 32), (5, 17), (64, 24), (80, 31)])
def test_both_builders_give_identical_lliks(K, D):
    from pysvihmm_amd.engine import HipEngine
    from tests.helpers import make_problem
    T, Lm, B = 3000, 40, 12
    pb = make_problem(K, D, T, seed=K + D, miss=0.05)
    starts = np.random.default_rng(1).integers(0, T - Lm + 1, size=B)
    out = {}
    for prec in ("f64", "f32"):
        for old in (0, 2):
            eng = HipEngine(0)
            try:
                eng.set_variant(13, old)
                eng.set_precision(prec)
                eng.set_obs(pb["obs"], pb["mask"])
                eng.set_globals(pb["mod_init"], pb["ltran"])
                eng.set_emission_niw(pb["mu"], pb["sigma"], pb["kappa"], pb["nu"])
                out[prec, old] = (eng.loglik(starts, Lm), eng.estep(starts, Lm).buf.copy())
            finally:
                eng.close()
        np.testing.assert_array_equal(out[prec, 0][0], out[prec, 2][0])
        np.testing.assert_array_equal(out[prec, 0][1], out[prec, 2][1])


def test_resident_loop_trajectory_is_identical():
    from pysvihmm_amd.engine import HipEngine
    from pysvihmm_amd.distributions import niw_prior_logpart
    from pysvihmm_amd import _lib as L
    from tests.helpers import make_problem
    K, D, T, B, Lm, nit = 64, 32, 5000, 20, 33, 5
    pb = make_problem(K, D, T, seed=2)
    rng = np.random.default_rng(K)
    prior_tran = 1.0 + rng.random((K, K))
    mu0 = np.tile(pb["obs"].mean(0), (K, 1)) + 0.1 * rng.normal(size=(K, D))
    sg0 = np.tile(0.75 * np.cov(pb["obs"].T).reshape(D, D), (K, 1, 1))
    ka0, nu0 = np.full(K, 0.01), np.full(K, D + 2.0)
    res = []
    for old in (3, 2, 0):
        eng = HipEngine(0)
        try:
            eng.set_variant(13, old)
            eng.set_obs(pb["obs"], pb["mask"])
            eng.svi_begin(prior_tran, pb["var_tran"], (mu0, sg0, ka0, nu0), (pb["mu"], pb["sigma"], pb["kappa"], pb["nu"]),
                          niw_prior_logpart(sg0, nu0), nit, 1.0)
            r2 = np.random.default_rng(5)
            for it in range(nit):
                eng.svi_iteration(it, r2.integers(0, T - Lm, size=B), B, Lm, L.TRANS_WRAP, (it + 1.0) ** -0.7, 3.0, 2.5)
            res.append((eng.svi_read_state(), eng.svi_read_elbo(nit)[0]))
        finally:
            eng.close()
    # the split builder (variant 13 = 3: global step still a launch of its own) against the one-half-wave builder (= 2)
    for a, b in zip(res[0][0], res[1][0]):
        np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(res[0][1], res[1][1])
    # the default (= 0): global step and builder in ONE launch (k_svi_step_theta32s).  Same formulas; the compiler
    # contracts the blend's multiply-adds differently in the merged kernel, so the state agrees to rounding (1e-12
    # relative after five iterations), not bit for bit
    for n, a, b in zip(("var_tran", "var_init", "mu", "sigma", "kappa", "nu"), res[2][0], res[1][0]):
        np.testing.assert_allclose(a, b, rtol=1e-9, atol=1e-12, err_msg=n)
    np.testing.assert_allclose(res[2][1], res[1][1], rtol=1e-9)
