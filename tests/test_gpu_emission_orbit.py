"""Orbit-schedule emission GEMM (the scaled E-step emission for D % 8 == 0, K <= 64) against the C oracle and
against the table-driven kernel it replaces, over every state-tile count (NT = 1..4, ragged K), every
supported D, masked (missing) rows and row counts that are not a multiple of any tile -- in all three forms:
k_emission_orbit_ks (16-row workgroups, k-steps split over the waves: what batches below 32 768 rows take),
k_emission_orbit<.., 1> (64-row workgroups) and k_emission_orbit<.., 2> (128-row workgroups: large batches)."""
import numpy as np
import pytest

from helpers import make_problem

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from pysvihmm_amd.engine import HipEngine
    e = HipEngine(0)
    yield e
    e.set_variant("emission_orbit", 0)
    e.close()


@pytest.mark.parametrize("K,D", [(5, 8), (16, 8), (20, 16), (33, 24), (48, 16), (64, 24), (50, 32), (64, 40),
                                 (7, 40)])
def test_orbit_vs_oracle_and_table(eng, K, D):
    from pysvihmm_amd import _lib as L
    from oracle import ref_c
    T, Lm, B = 9000, 37, 211                    # 7807 rows: the last tile is ragged
    pb = make_problem(K, D, T, seed=K * 100 + D, miss=0.07)
    obs = pb["obs"]
    rng = np.random.default_rng(K)
    starts = rng.integers(0, T - Lm + 1, size=B)
    eng.set_obs(obs, pb["mask"])
    eng.set_globals(pb["mod_init"], pb["ltran"])
    eng.set_emission_niw(pb["mu"], pb["sigma"], pb["kappa"], pb["nu"])
    for flags in (L.TRANS_WRAP, L.TRANS_WRAP | L.MASK_AS_NAN):
        eng.set_variant("emission_orbit", 0)
        a = eng.estep(starts, Lm, flags=flags).buf.copy()
        eng.set_variant("emission_orbit", 1)
        b = eng.estep(starts, Lm, flags=flags).buf.copy()
        ref = ref_c.estep_minibatch(obs, pb["mask"], starts, Lm, pb["mod_init"], pb["ltran"],
                                    pb["mu"], pb["sigma"], pb["kappa"], pb["nu"], flags)
        scale = np.abs(ref).max()
        np.testing.assert_allclose(a, b, rtol=1e-9, atol=1e-11 * scale)
        np.testing.assert_allclose(a, ref, rtol=1e-6, atol=1e-9 * scale)
        for form in (5, 2):                  # the row-tile forms on the same batch
            eng.set_variant("emission_orbit", form)
            c = eng.estep(starts, Lm, flags=flags).buf.copy()
            np.testing.assert_allclose(c, a, rtol=1e-9, atol=1e-11 * scale, err_msg="form %d" % form)
    eng.set_variant("emission_orbit", 0)


def test_orbit_parameters_change_between_calls(eng):
    """theta is written in the orbit layout by the NIW kernel on every parameter upload: a second
    parameter set on the same handle must not see the first one's theta."""
    from pysvihmm_amd import _lib as L
    from oracle import ref_c
    K, D, T, Lm, B = 24, 16, 5000, 21, 200
    pa = make_problem(K, D, T, seed=1, miss=0.0)
    pb = make_problem(K, D, T, seed=2, miss=0.0)
    starts = np.arange(B) * 23
    eng.set_obs(pa["obs"], None)
    eng.set_globals(pa["mod_init"], pa["ltran"])
    for p in (pa, pb, pa):
        eng.set_emission_niw(p["mu"], p["sigma"], p["kappa"], p["nu"])
        got = eng.estep(starts, Lm, flags=L.TRANS_WRAP).buf
        ref = ref_c.estep_minibatch(pa["obs"], None, starts, Lm, pa["mod_init"], pa["ltran"],
                                    p["mu"], p["sigma"], p["kappa"], p["nu"], L.TRANS_WRAP)
        np.testing.assert_allclose(got, ref, rtol=1e-6, atol=1e-9 * np.abs(ref).max())
