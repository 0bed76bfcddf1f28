"""Randomised shape / conditioning sweep of the HIP E-step against the C oracle:
ragged K and D (not multiples of the 16-wide tiles), Lm from 1 up, batches on both sides
of the fused-sweep threshold, masks, NaN rows, weakly and extremely separated states
(log-likelihood gaps of ~1e4 exercise the shift logic of the linear-domain recursions)."""
import numpy as np
import pytest

from tests.helpers import make_problem, unpack

pytestmark = pytest.mark.gpu


def _cases():
    rng = np.random.default_rng(20260928)
    out = []
    for i in range(24):
        K = int(rng.choice([1, 2, 3, 7, 15, 16, 17, 31, 33, 48, 63, 64, 65, 70]))
        D = int(rng.choice([1, 2, 3, 5, 8, 13, 16, 31, 32, 33, 40]))
        Lm = int(rng.choice([1, 2, 3, 9, 33, 70]))
        B = int(rng.choice([1, 5, 17, 200, 333]))
        sep = float(rng.choice([0.5, 3.0, 20.0]))
        miss = float(rng.choice([0.0, 0.2]))
        out.append((K, D, Lm, B, sep, miss, i))
    return out


CASES = _cases()


@pytest.mark.parametrize("case", CASES, ids=["K%d_D%d_Lm%d_B%d_sep%g_m%g_%d" % c for c in CASES])
def test_sweep(case):
    from pysvihmm_amd.engine import HipEngine
    from pysvihmm_amd import _lib as L
    from oracle import ref_c
    K, D, Lm, B, sep, miss, i = case
    T = max(4 * Lm, 400)
    pb = make_problem(K, D, T, seed=1000 + i, miss=miss, sep=sep)
    obs = pb["obs"].copy()
    if i % 3 == 0:
        obs[T // 3] = np.nan            # a NaN row (unmasked rows with NaN poison the
        pb["mask"][T // 3] = True       # reference's statistics too, so mask it)
    rng = np.random.default_rng(i)
    starts = rng.integers(0, T - Lm + 1, size=B)
    e = HipEngine(0)
    try:
        e.set_obs(obs, pb["mask"])
        e.set_globals(pb["mod_init"], pb["ltran"])
        e.set_emission_niw(pb["mu"], pb["sigma"], pb["kappa"], pb["nu"])
        for flags in (L.TRANS_WRAP, L.MASK_AS_NAN):
            st = e.estep(starts, Lm, flags=flags)
            ref = ref_c.estep_minibatch(obs, pb["mask"], starts, Lm, pb["mod_init"], pb["ltran"],
                                        pb["mu"], pb["sigma"], pb["kappa"], pb["nu"], flags=flags)
            A, xbar, neff, S, lb = unpack(ref, K, D)
            sc = B * Lm
            xs = max(1.0, np.nanmax(np.abs(obs)))
            np.testing.assert_allclose(st.A_raw, A, rtol=1e-6, atol=1e-9 * sc)
            np.testing.assert_allclose(st.neff, neff, rtol=1e-6, atol=1e-9 * sc)
            np.testing.assert_allclose(st.xbar, xbar, rtol=1e-6, atol=1e-9 * sc * xs)
            np.testing.assert_allclose(st.S, S, rtol=1e-6, atol=1e-9 * sc * xs * xs)
            np.testing.assert_allclose(st.lb[0], lb, rtol=1e-9, atol=1e-6)
            q = e.read_intermediate("var_x", B, Lm)
            assert np.all(np.isfinite(q)) and np.all(q >= 0)
            np.testing.assert_allclose(q.sum(-1), 1.0, rtol=1e-11)
    finally:
        e.close()
