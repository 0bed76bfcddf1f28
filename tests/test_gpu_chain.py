"""Full-chain E-step as an exact blocked scan: chunk transfer matrices ->
boundary vectors -> all chunks as concurrent windows.  Must equal the sequential recursion
(C oracle) and the sequential device path, for every tail length and ragged K."""
import numpy as np
import pytest

from helpers import make_problem, unpack, ffbs_draws_exact

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
# ... and wide models: transition tile streamed, 8 / 16 state tiles, thread-per-state boundary scan
@pytest.mark.parametrize("K,T", [(5, 2306), (16, 2311), (40, 2561), (64, 2311), (64, 4100), (4, 530000),
                                 (80, 2311), (128, 2561), (150, 2306), (256, 2400)])
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


@pytest.mark.parametrize("K,D,T", [(12, 4, 3000), (100, 4, 2500), (200, 3, 2100)])
def test_chain_batch_statistics_vs_oracle(eng, K, D, T):
    """hmmbase batch E-step on one long chain (no wrap): statistics + lower bound."""
    from oracle import ref_c
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


@pytest.mark.parametrize("K,T,D,sep", [(5, 5000, 3, 1.5), (16, 3000, 3, 1.5), (40, 70001, 3, 1.5),
                                       (64, 9000, 3, 1.5), (16, 6000, 8, 14.0), (33, 4000, 16, 9.0),
                                       # wide models (round 3): k_ffbs_paths_wide, k_lalpha_fix_wide
                                       (80, 3000, 3, 1.5), (130, 2600, 3, 3.0), (256, 2500, 4, 12.0),
                                       (200, 30001, 3, 4.0)])
def test_ffbs_long_chain_blocked(K, T, D, sep):
    """FFBS of a long chain (hmm_fast.pyx:43-124): forward filter through the blocked scan
    (lalpha from the scaled messages) and backward sampling by composition of the per-row draw
    maps, against the C oracle's forward pass and a row-by-row check of every draw:
    z[t] must be the inverse-CDF draw of softmax(lalpha[t] + logA[:, z[t+1]]) at u[t]."""
    from pysvihmm_amd.engine import HipEngine
    from oracle import ref_c
    # sep 1.5: weakly separated, slow coupling of the sampled paths; sep >= 9 with D >= 8: states
    # hundreds of nats apart, the scaled forward messages underflow and lalpha needs the
    # log-domain fix-up to stay finite like the reference's
    pb = make_problem(K, D, T, seed=K + 11, sep=sep, miss=0.0)
    DE = np.finfo(np.float64).eps
    logA = np.log(pb["var_tran"] + DE)
    u = np.random.default_rng(K).random(T)
    e = HipEngine(0)
    e.set_obs(pb["obs"], None)
    e.set_globals(pb["mod_init"], logA)
    e.set_emission_niw(pb["mu"], pb["sigma"], pb["kappa"], pb["nu"])
    z, la = e.ffbs(logA, u)
    ll = ref_c.lliks_niw(pb["obs"], pb["mu"], pb["sigma"], pb["kappa"], pb["nu"])
    ref = ref_c.forward(ll, pb["mod_init"], logA)
    assert np.isfinite(ref).all() and np.isfinite(la).all()
    if sep > 5:
        assert (ref.max(axis=1) - ref.min(axis=1)).max() > 800      # beyond what exp() can span
    np.testing.assert_allclose(la, ref, rtol=1e-9, atol=1e-6)
    assert z.min() >= 0 and z.max() < K
    # every draw, row by row, EXACT (index output): z[t] is the inverse-CDF draw of
    # softmax(lalpha[t] + logA[:, z[t+1]]) at u[t] for the lalpha the call returned, on every row
    # whose uniform is not within 1e-12 of a CDF step
    bad, risky = ffbs_draws_exact(z, la, logA, u)
    assert bad == 0 and risky <= 2, (bad, risky)
    # the sequential sampler (chain off): same exactness against its own lalpha, and the same path
    # as the composed maps wherever neither run had a uniform on a CDF step
    e.set_variant("chain", 1)
    z2, la2 = e.ffbs(logA, u)
    np.testing.assert_allclose(la2, ref, rtol=1e-9, atol=1e-6)
    bad2, risky2 = ffbs_draws_exact(z2, la2, logA, u)
    assert bad2 == 0 and risky2 <= 2, (bad2, risky2)
    if risky == 0 and risky2 == 0 and ffbs_draws_exact(z, la2, logA, u) == (0, 0):
        np.testing.assert_array_equal(z2, z)
    e.close()


# ... and wide models (64 < K <= 256: k_lalpha_fix_wide / k_lbeta_fix_wide, round 3)
@pytest.mark.parametrize("K,T,D,sep", [(7, 4100, 3, 2.0), (16, 6000, 8, 14.0), (64, 5000, 16, 9.0),
                                       (80, 2600, 3, 2.0), (100, 3000, 4, 12.0), (200, 2400, 3, 20.0),
                                       (256, 2500, 2, 15.0)])
def test_chain_logs_from_scaled_messages(K, T, D, sep):
    """lalpha / lbeta of one long chain (hmmbase.local_update's attributes) come from the blocked
    scan's scaled messages plus a row-parallel log-domain fix-up of underflowed entries -- no
    sequential pass -- and match the log-domain C oracle also where states are > 800 nats apart."""
    from pysvihmm_amd.engine import HipEngine
    from oracle import ref_c
    pb = make_problem(K, D, T, seed=K + 3, sep=sep, miss=0.05)
    e = HipEngine(0)
    e.set_obs(pb["obs"], pb["mask"])
    e.set_globals(pb["mod_init"], pb["ltran"])
    e.set_emission_niw(pb["mu"], pb["sigma"], pb["kappa"], pb["nu"])
    r = e.forward_backward([0], T)
    ll = ref_c.lliks_niw(pb["obs"], pb["mu"], pb["sigma"], pb["kappa"], pb["nu"])
    la = ref_c.forward(ll, pb["mod_init"], pb["ltran"])
    lb = ref_c.backward(ll, pb["ltran"])
    q, lz = ref_c.posterior(la, lb)
    assert np.isfinite(r["lalpha"]).all() and np.isfinite(r["lbeta"]).all()
    np.testing.assert_allclose(r["lalpha"][0], la, rtol=1e-9, atol=1e-6)
    np.testing.assert_allclose(r["lbeta"][0], lb, rtol=1e-9, atol=1e-6)
    np.testing.assert_allclose(r["var_x"][0], q, rtol=1e-6, atol=1e-12)
    np.testing.assert_allclose(r["local_lb"][0], lz, rtol=1e-10)
    np.testing.assert_allclose(e.read_intermediate("lliks", 1, T)[0], ll, rtol=1e-9, atol=1e-8)
    # the sequential log-domain kernels give the same thing
    e.set_variant("chain", 1)
    r2 = e.forward_backward([0], T)
    np.testing.assert_allclose(r2["lalpha"][0], r["lalpha"][0], rtol=1e-9, atol=1e-6)
    np.testing.assert_allclose(r2["lbeta"][0], r["lbeta"][0], rtol=1e-9, atol=1e-6)
    e.close()


def test_ffbs_long_chain_host_lliks():
    """The same path fed with host-evaluated lliks (generic emission plugins,
    SVIHMM_USE_HOST_LLIKS): the log-domain fix-up reads the uploaded lliks, not an NIW kernel."""
    from pysvihmm_amd.engine import HipEngine
    from pysvihmm_amd import _lib as L
    from oracle import ref_c
    K, T = 12, 4500
    rng = np.random.default_rng(4)
    ll = rng.normal(size=(T, K)) * 3.0
    ll[np.arange(T), rng.integers(0, K, size=T)] += 900.0        # one state dominates by 900 nats
    vt = 1.0 + rng.random((K, K)) * 5
    logA = np.log(vt + np.finfo(np.float64).eps)
    mod_init = np.log(rng.dirichlet(np.ones(K)))
    u = rng.random(T)
    e = HipEngine(0)
    e.set_obs(np.zeros((T, 1)), None)
    e.set_globals(mod_init, logA)
    e.set_lliks(ll[None])
    z, la = e.ffbs(logA, u, flags=L.USE_HOST_LLIKS)
    ref = ref_c.forward(ll, mod_init, logA)
    assert np.isfinite(la).all()
    np.testing.assert_allclose(la, ref, rtol=1e-9, atol=1e-6)
    assert z.min() >= 0 and z.max() < K
    e.close()
