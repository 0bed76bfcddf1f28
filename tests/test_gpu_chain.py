"""Full-chain E-step as an exact blocked scan: chunk transfer matrices ->
boundary vectors -> all chunks as concurrent windows.  Must equal the sequential recursion
(C oracle) and the sequential device path, for every tail length and ragged K."""
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


def _oracle(pb, T, masked):
    from oracle import ref_c
    x = pb["obs"][:T].copy()
    if masked:
        x[pb["mask"][:T]] = np.nan
    ll = ref_c.lliks_niw(x, pb["mu"], pb["sigma"], pb["kappa"], pb["nu"])
    la = ref_c.forward(ll, pb["mod_init"], pb["ltran"])
    lb = ref_c.backward(ll, pb["ltran"])
    q, lz = ref_c.posterior(la, lb)
    return la, lb, q, lz


# T - 1 = 9 * 256 + tail steps: tail of 1, 6 and 256 (the maximum) steps
# ... and one chain long enough for the 1024-step chunks
@pytest.mark.parametrize("K,T", [(5, 2306), (16, 2311), (40, 2561), (64, 2311), (64, 4100), (4, 530000)])
def test_chain_posteriors_vs_oracle(eng, K, T):
    from pysvihmm_amd import _lib as L
    D = 3
    pb = make_problem(K, D, T, seed=700 + K, miss=0.1)
    eng.set_obs(pb["obs"], pb["mask"])
    eng.set_globals(pb["mod_init"], pb["ltran"])
    eng.set_emission_niw(pb["mu"], pb["sigma"], pb["kappa"], pb["nu"])
    la, lb, q, lz = _oracle(pb, T, True)
    r = eng.forward_backward([0], T, flags=L.MASK_AS_NAN, want=("var_x", "local_lb"))
    np.testing.assert_allclose(r["var_x"][0], q, rtol=RTOL, atol=1e-12)
    np.testing.assert_allclose(r["var_x"][0].sum(-1), 1.0, rtol=1e-11)
    np.testing.assert_allclose(r["local_lb"][0], lz, rtol=1e-11)
    # the sequential device path gives the same
    eng.set_variant("chain", 1)
    r2 = eng.forward_backward([0], T, flags=L.MASK_AS_NAN, want=("var_x", "local_lb"))
    eng.set_variant("chain", 0)
    np.testing.assert_allclose(r["var_x"][0], r2["var_x"][0], rtol=1e-8, atol=1e-13)
    np.testing.assert_allclose(r["local_lb"], r2["local_lb"], rtol=1e-12)
    # log-domain messages on demand after the scan
    got = eng.read_rows("lalpha", T - 5, 5)
    np.testing.assert_allclose(got, la[T - 5:], rtol=1e-9, atol=1e-8)


def test_chain_batch_statistics_vs_oracle(eng):
    """hmmbase batch E-step on one long chain (no wrap): statistics + lower bound."""
    from oracle import ref_c
    K, D, T = 12, 4, 3000
    pb = make_problem(K, D, T, seed=21, miss=0.05)
    eng.set_obs(pb["obs"], pb["mask"])
    eng.set_globals(pb["mod_init"], pb["ltran"])
    eng.set_emission_niw(pb["mu"], pb["sigma"], pb["kappa"], pb["nu"])
    for flags in (0, 1, 2):
        st = eng.estep([0], T, flags=flags)
        ref = ref_c.estep_minibatch(pb["obs"], pb["mask"], np.array([0]), T, pb["mod_init"], pb["ltran"],
                                    pb["mu"], pb["sigma"], pb["kappa"], pb["nu"], flags=flags)
        A, xbar, neff, S, lb = unpack(ref, K, D)
        np.testing.assert_allclose(st.A_raw, A, rtol=RTOL, atol=1e-9 * T)
        np.testing.assert_allclose(st.neff, neff, rtol=RTOL, atol=1e-9 * T)
        np.testing.assert_allclose(st.xbar, xbar, rtol=RTOL, atol=1e-8 * T)
        np.testing.assert_allclose(st.S, S, rtol=RTOL, atol=1e-7 * T)
        np.testing.assert_allclose(st.lb[0], lb, rtol=1e-10)
