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


def _svi_cases():
    rng = np.random.default_rng(20260929)
    out = []
    for i in range(16):
        K = int(rng.choice([1, 2, 5, 16, 17, 40, 64, 65, 96, 130]))
        D = int(rng.choice([1, 3, 8, 16, 24, 33]))
        Lm = int(rng.choice([1, 4, 17, 65]))
        S = int(rng.choice([1, 3, 20, 260]))
        f32 = bool(rng.integers(0, 2)) and K <= 64
        out.append((K, D, Lm, S, f32, i))
    return out


SVI_CASES = _svi_cases()


@pytest.mark.parametrize("case", SVI_CASES, ids=["K%d_D%d_Lm%d_S%d_f32%d_%d" % c for c in SVI_CASES])
def test_svi_loop_sweep(case):
    """Randomised shapes through svihmm_svi_begin / svihmm_svi_iteration (three iterations, state
    resident in HBM, side-stream globals) against the reference arithmetic of the oracle engine:
    K on both sides of 64 and of the LDS / global-scratch switch of k_svi_globals, buffered
    windows (inner segment), masks, fp64 and the fp32 mode (tolerance 1e-3 there)."""
    from oracle.engine import OracleEngine
    from pysvihmm_amd.engine import HipEngine
    from pysvihmm_amd.distributions import niw_prior_logpart
    from pysvihmm_amd import _lib as L
    K, D, Lm, S, f32, i = case
    T = max(6 * Lm, 500)
    pb = make_problem(K, D, T, seed=2000 + i, miss=0.1 if i % 2 else 0.0, sep=3.0)
    rng = np.random.default_rng(100 + i)
    prior_tran = 1.0 + rng.random((K, K))
    mu0 = np.tile(pb["obs"].mean(0), (K, 1))
    sg0 = np.tile((0.75 * np.cov(pb["obs"].T)).reshape(D, D) + 0.05 * np.eye(D), (K, 1, 1))
    prior = (mu0, sg0, np.full(K, 0.01), np.full(K, D + 2.0))
    factors = (pb["mu"], pb["sigma"], pb["kappa"], pb["nu"])
    inner = (Lm // 4, Lm - 2 * (Lm // 4)) if (i % 3 == 0 and Lm >= 4) else None
    res = []
    for eng in (HipEngine(0, dtype="f32" if f32 else "f64"), OracleEngine()):
        eng.set_obs(pb["obs"], pb["mask"])
        eng.svi_begin(prior_tran, pb["var_tran"], prior, factors, niw_prior_logpart(sg0, prior[3]), 3, 1.0)
        r2 = np.random.default_rng(7)
        for it in range(3):
            starts = r2.integers(0, T - Lm + 1, size=S)
            eng.svi_iteration(it, starts, S, Lm, L.TRANS_WRAP, (it + 2.0) ** -0.7, 1.7, 1.3, inner=inner)
        res.append((eng.svi_read_state(), eng.svi_read_elbo(3)[0]))
        eng.close()
    rt, at = (1e-3, 1e-5) if f32 else (1e-6, 1e-8)
    for name, a, b in zip(("var_tran", "var_init", "mu", "sigma", "kappa", "nu"), res[0][0], res[1][0]):
        np.testing.assert_allclose(a, b, rtol=rt, atol=at * max(1.0, float(np.abs(b).max())), err_msg=name)
    np.testing.assert_allclose(res[0][1], res[1][1], rtol=1e-5 if f32 else 1e-9)


@pytest.mark.parametrize("case", [(300, 4, 33, 20, 3.0, 0.1, False, False), (512, 8, 17, 96, 20.0, 0.0, False, True),
                                  (1024, 2, 9, 3, 3.0, 0.1, False, False), (300, 3, 257, 4, 3.0, 0.0, True, False)],
                         ids=lambda c: "K%d_D%d_Lm%d_B%d" % c[:4])
def test_models_beyond_256_states(case):
    """K in (256, 1024]: generic per-window recursions (thread = state), emission / statistics
    tiles in state groups; the full check list of the randomised campaign (statistics in both
    transition conventions, messages, fp32-mode switch, inner segment), incl. a rare initial state
    and transition expectations below exp()'s range."""
    from tests.fuzz_gpu import run_case
    from pysvihmm_amd.engine import HipEngine
    from pysvihmm_amd import _lib as L
    from oracle import ref_c
    e = HipEngine(0)
    try:
        run_case(e, L, ref_c, case, 4242)
    finally:
        e.close()


@pytest.mark.parametrize("case", [(5, 65, 33, 20, 3.0, 0.1, False, False), (16, 72, 17, 300, 3.0, 0.0, False, True),
                                  (64, 79, 65, 250, 3.0, 0.0, False, False), (8, 80, 33, 40, 3.0, 0.1, False, False),
                                  (32, 96, 17, 260, 3.0, 0.0, False, False)],
                         ids=lambda c: "K%d_D%d_Lm%d_B%d" % c[:4])
def test_observations_up_to_the_niw_kernels_width(case):
    """D in (64, SVIHMM_NIW_MAX_D = 96]: generic NIW -> theta kernel (two D x (D+1) matrices in LDS since
    round 3; 79 before), emission / statistics fallbacks."""
    from tests.fuzz_gpu import run_case
    from pysvihmm_amd.engine import HipEngine
    from pysvihmm_amd import _lib as L
    from oracle import ref_c
    e = HipEngine(0)
    try:
        run_case(e, L, ref_c, case, 777)
    finally:
        e.close()


def test_observations_wider_than_the_niw_kernels():
    """D > SVIHMM_NIW_MAX_D: the C ABI refuses the NIW upload and points at svihmm_set_lliks; with
    host-evaluated lliks the recursions and the statistics run on the device for any D; the
    classes take that route by themselves (hmmbase._niw_fastpath)."""
    from pysvihmm_amd.engine import HipEngine
    from pysvihmm_amd import _lib as L, hmmsgd_metaobs, hmmbatchcd
    from pysvihmm_amd.distributions import Gaussian
    from oracle.engine import OracleEngine
    from tests.fuzz_gpu import check_stats
    e = HipEngine(0)
    try:
        for (K, D, Lm, B) in [(5, 100, 33, 20), (16, 200, 9, 300)]:
            T = 600
            pb = make_problem(K, D, T, seed=D, miss=0.1)
            st = np.random.default_rng(1).integers(0, T - Lm + 1, size=B)
            o = OracleEngine()
            for eng in (e, o):
                eng.set_obs(pb["obs"], pb["mask"])
                eng.set_globals(pb["mod_init"], pb["ltran"])
            with pytest.raises(RuntimeError, match="SVIHMM_NIW_MAX_D"):
                e.set_emission_niw(pb["mu"], pb["sigma"], pb["kappa"], pb["nu"])
            o.set_emission_niw(pb["mu"], pb["sigma"], pb["kappa"], pb["nu"])
            ll = o.loglik(st, Lm, flags=L.MASK_AS_NAN)
            for eng in (e, o):
                eng.set_lliks(ll)
            fl = L.TRANS_WRAP | L.MASK_AS_NAN | L.USE_HOST_LLIKS
            a, b = e.estep(st, Lm, flags=fl), o.estep(st, Lm, flags=fl)
            check_stats(a.buf, b.buf, K, D, B * Lm, max(1.0, np.abs(pb["obs"]).max()), 1e-6, 1e-9, "host lliks D=%d" % D)
        K, D, T = 3, 100, 400
        pb = make_problem(K, D, T, seed=3)
        obs = pb["obs"]

        def model(engine, cd):
            np.random.seed(6)
            prior = np.array([Gaussian(mu_0=obs.mean(0), sigma_0=0.75 * np.cov(obs.T) + np.eye(D), kappa_0=0.01,
                                       nu_0=D + 2) for _ in range(K)])
            if cd:
                return hmmbatchcd.VBHMM(obs.copy(), np.ones(K), np.ones((K, K)), prior, maxit=2, engine=engine)
            return hmmsgd_metaobs.VBHMM(obs.copy(), np.ones(K), np.ones((K, K)), prior, tau=1.0, kappa=0.7,
                                        metaobs_half=5, mb_sz=4, maxit=3, seed=3, engine=engine)
        for cd in (False, True):
            a, b = model(e, cd), model(OracleEngine(), cd)
            a.infer(); b.infer()
            np.testing.assert_allclose(a.var_tran, b.var_tran, rtol=1e-6, atol=1e-8)
            np.testing.assert_allclose(a.elbo_vec, b.elbo_vec, rtol=1e-8)
    finally:
        e.close()


@pytest.mark.parametrize("case", [(16, 4, 5000, 8, 3.0, 0.1, False, False), (8, 2, 100000, 2, 3.0, 0.0, False, False),
                                  (64, 8, 20000, 3, 20.0, 0.0, False, True)],
                         ids=lambda c: "K%d_D%d_Lm%d_B%d" % c[:4])
def test_batches_of_long_windows(case):
    """Several windows of thousands of rows each (adaptive / buffered meta-observations grow them;
    one window alone takes the blocked scan, covered elsewhere): same check list."""
    from tests.fuzz_gpu import run_case
    from pysvihmm_amd.engine import HipEngine
    from pysvihmm_amd import _lib as L
    from oracle import ref_c
    e = HipEngine(0)
    try:
        run_case(e, L, ref_c, case, 999)
    finally:
        e.close()


@pytest.mark.parametrize("case", [(16, 8, 257, 1100, 3.0, 0.1, False, False), (8, 3, 300, 1100, 20.0, 0.0, False, True)],
                         ids=lambda c: "K%d_D%d_Lm%d_B%d" % c[:4])
def test_one_state_tile_large_batch(case):
    """K <= 16 with more than 2^18 rows: the statistics plan that puts two workgroups on a CU."""
    from tests.fuzz_gpu import run_case
    from pysvihmm_amd.engine import HipEngine
    from pysvihmm_amd import _lib as L
    from oracle import ref_c
    e = HipEngine(0)
    try:
        run_case(e, L, ref_c, case, 1234)
    finally:
        e.close()
