"""Long randomised campaign of the HIP E-step against the C oracle (run by hand on a GPU box:
``python tests/fuzz_gpu.py --seed 1 --cases 300``).  Not collected by pytest -- the fixed-seed
subset that runs every round is tests/test_gpu_random_sweep.py.  Every case draws a model shape
(K up to 256, D up to 64, ragged against the 16-wide tiles), a window batch on either side of
every kernel-selection threshold, masks / NaN rows, state separation from overlapping to
1e4-nat gaps, near-absorbing transition rows, and checks

  * the packed statistics of ``svihmm_estep_minibatch_ex`` (both transition conventions, random
    inner segment) against ``orc_estep_minibatch``,
  * ``lalpha / lbeta / var_x / local_lb`` of ``svihmm_forward_backward`` on sampled windows,
  * the fp32 mode against the fp64 statistics (north_star's 1e-3),
  * now and then one long chain (B = 1, Lm = T: the blocked scan) against the oracle's
    sequential recursion.

Prints one line per failing case (the case tuple reproduces it) and exits non-zero if any."""
import argparse
import os
import sys
import time
import traceback

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.helpers import make_problem, unpack, effective_cores  # noqa: E402

NCORE = effective_cores()


def draw_case(rng):
    K = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 15, 16, 17, 24, 31, 32, 33, 47, 48, 49, 63, 64, 65, 80, 96,
                        127, 128, 129, 192, 200, 256]))
    D = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 13, 15, 16, 17, 24, 31, 32, 33, 40, 48, 63, 64]))
    Lm = int(rng.choice([1, 2, 3, 4, 5, 8, 9, 16, 17, 31, 33, 64, 70, 129, 257, 300]))
    B = int(rng.choice([1, 2, 3, 5, 15, 16, 17, 31, 33, 64, 65, 191, 192, 200, 255, 256, 257, 333, 600, 1023,
                        1024, 1100]))
    # keep the oracle's work per case bounded (~ a few seconds on one core)
    cost = lambda: B * Lm * K * (2.0 * K + 3.0 * D * D)
    while cost() > 6e9 and B > 1:
        B = max(1, B // 2)
    while cost() > 6e9 and Lm > 1:
        Lm = max(1, Lm // 2)
    sep = float(rng.choice([0.0, 0.5, 3.0, 20.0, 60.0]))
    miss = float(rng.choice([0.0, 0.0, 0.1, 0.5]))
    sparse = bool(rng.random() < 0.25)        # near-absorbing / sparse transition expectations
    rare = bool(rng.random() < 0.3)           # initial distribution with states of mass 1e-4 and less
    return K, D, Lm, B, sep, miss, sparse, rare


def problem(case, seed):
    K, D, Lm, B, sep, miss, sparse, rare = case
    T = max(4 * Lm, 400)
    pb = make_problem(K, D, T, seed=seed, miss=miss, sep=sep)
    rng = np.random.default_rng(seed + 7)
    from scipy.special import digamma
    if rare:
        vi = np.where(rng.random(K) < 0.5, 10.0 ** -rng.uniform(3, 7, K), 0.3) * (0.5 + rng.random(K))
        pb["mod_init"] = digamma(vi + 1e-9) - digamma(vi.sum() + 1e-9)
    if sparse:
        vt = 1e-3 + rng.random((K, K)) * (rng.random((K, K)) < 0.2) * T
        vt[np.arange(K), np.arange(K)] += T
        pb["ltran"] = digamma(vt + 1e-9) - digamma(vt.sum(1)[:, None] + 1e-9)
    if rng.random() < 0.3:
        pb["obs"][T // 3] = np.nan
        pb["mask"][T // 3] = True
    starts = rng.integers(0, T - Lm + 1, size=B)
    return pb, starts, T


def check_stats(got, ref, K, D, sc, xs, rtol, atol_scale, what):
    names = ("A_raw", "xbar", "neff", "S", "lb")
    g = unpack(got, K, D)
    r = unpack(ref, K, D)
    atol = (atol_scale * sc, atol_scale * sc * xs, atol_scale * sc, atol_scale * sc * xs * xs, 1e-6)
    for n, a, b, at in zip(names, g, r, atol):
        rt = 1e-9 if (n == "lb" and rtol < 1e-5) else rtol
        np.testing.assert_allclose(a, b, rtol=rt, atol=at, err_msg="%s: %s" % (what, n))


def run_case(e, L, ref_c, case, seed):
    K, D, Lm, B, sep, miss, sparse, rare = case
    pb, starts, T = problem(case, seed)
    obs, mask = pb["obs"], pb["mask"]
    par = (pb["mod_init"], pb["ltran"], pb["mu"], pb["sigma"], pb["kappa"], pb["nu"])
    e.set_precision("f64")
    e.set_obs(obs, mask)
    e.set_globals(pb["mod_init"], pb["ltran"])
    e.set_emission_niw(pb["mu"], pb["sigma"], pb["kappa"], pb["nu"])
    sc = B * Lm
    xs = max(1.0, float(np.nanmax(np.abs(obs))))
    rng = np.random.default_rng(seed + 11)
    f64 = {}
    for flags in (L.TRANS_WRAP, L.MASK_AS_NAN):
        st = e.estep(starts, Lm, flags=flags)
        ref = ref_c.estep_minibatch(obs, mask, starts, Lm, *par, flags=flags, threads=NCORE)
        check_stats(st.buf, ref, K, D, sc, xs, 1e-6, 1e-9, "estep flags=%d" % flags)
        f64[flags] = st.buf.copy()
    # messages of up to three windows
    for b in rng.choice(B, size=min(B, 3), replace=False):
        fb = e.forward_backward(starts[b:b + 1], Lm, flags=L.MASK_AS_NAN)
        x = obs[starts[b]:starts[b] + Lm].copy()
        x[mask[starts[b]:starts[b] + Lm]] = np.nan
        ll = ref_c.lliks_niw(x, pb["mu"], pb["sigma"], pb["kappa"], pb["nu"])
        la = ref_c.forward(ll, pb["mod_init"], pb["ltran"])
        lbm = ref_c.backward(ll, pb["ltran"])
        q, _ = ref_c.posterior(la, lbm)
        np.testing.assert_allclose(fb["lalpha"][0], la, rtol=1e-9, atol=1e-7, err_msg="lalpha")
        np.testing.assert_allclose(fb["lbeta"][0], lbm, rtol=1e-9, atol=1e-7, err_msg="lbeta")
        np.testing.assert_allclose(fb["var_x"][0], q, rtol=1e-6, atol=1e-12, err_msg="var_x")
    # batch of windows through forward_backward: posteriors sum to one, equal the single-window ones
    fbB = e.forward_backward(starts, Lm, flags=L.MASK_AS_NAN, want=("var_x", "local_lb"))
    assert np.all(np.isfinite(fbB["var_x"])), "var_x not finite"
    np.testing.assert_allclose(fbB["var_x"].sum(-1), 1.0, rtol=0, atol=1e-9, err_msg="var_x rows")
    # fp32 mode (K <= 64 takes the float kernels; other shapes must still give fp64-grade results)
    e.set_precision("f32")
    st = e.estep(starts, Lm, flags=L.TRANS_WRAP)
    e.set_precision("f64")
    assert np.all(np.isfinite(st.buf)), "f32: non-finite statistics"
    g = unpack(st.buf, K, D)
    r = unpack(f64[L.TRANS_WRAP], K, D)
    np.testing.assert_allclose(g[0], r[0], rtol=2e-3, atol=2e-4 * sc, err_msg="f32 A_raw")
    np.testing.assert_allclose(g[2], r[2], rtol=2e-3, atol=2e-4 * sc, err_msg="f32 neff")
    np.testing.assert_allclose(g[1], r[1], rtol=2e-3, atol=2e-4 * sc * xs, err_msg="f32 xbar")
    np.testing.assert_allclose(g[4], r[4], rtol=1e-4, atol=1e-2, err_msg="f32 lb")
    # the same with the bf16 kernels' batch-size floors lifted (round 5): the centred emission kernels
    # (k_emission_bf16x3 / k_emission_bf16x3d), the bf16 statistics kernels and -- from 192 windows -- the whole
    # wide-model path (k_scale_ll_f32, k_sweeps_lin2<float>, k_stats_bf16x3w) on this case's ragged shape
    e.set_precision("f32"); e.set_variant(10, 3); e.set_variant(5, 4)
    try:
        st = e.estep(starts, Lm, flags=L.TRANS_WRAP)
    finally:
        e.set_variant(10, 0); e.set_variant(5, 0); e.set_precision("f64")
    assert np.all(np.isfinite(st.buf)), "f32 (bf16 kernels): non-finite statistics"
    g = unpack(st.buf, K, D)
    np.testing.assert_allclose(g[0], r[0], rtol=2e-3, atol=2e-4 * sc, err_msg="f32/bf16 A_raw")
    np.testing.assert_allclose(g[2], r[2], rtol=2e-3, atol=2e-4 * sc, err_msg="f32/bf16 neff")
    np.testing.assert_allclose(g[1], r[1], rtol=2e-3, atol=2e-4 * sc * xs, err_msg="f32/bf16 xbar")
    np.testing.assert_allclose(g[4], r[4], rtol=1e-4, atol=1e-2, err_msg="f32/bf16 lb")
    # inner segment (buffered meta-observations)
    if Lm >= 3:
        off = int(rng.integers(0, Lm // 2))
        ln = int(rng.integers(1, Lm - off + 1))
        st = e.estep(starts, Lm, flags=L.MASK_AS_NAN, inner=(off, ln))
        acc = None
        from oracle.engine import OracleEngine
        oe = OracleEngine()
        oe.set_obs(obs, mask)
        oe.set_globals(pb["mod_init"], pb["ltran"])
        oe.set_emission_niw(pb["mu"], pb["sigma"], pb["kappa"], pb["nu"])
        acc = oe.estep(starts, Lm, flags=L.MASK_AS_NAN, inner=(off, ln))
        check_stats(st.buf, acc.buf, K, D, sc, xs, 1e-6, 1e-9, "inner (%d,%d)" % (off, ln))


def run_chain(e, L, ref_c, seed):
    rng = np.random.default_rng(seed)
    K = int(rng.choice([2, 5, 16, 33, 64, 100, 128]))
    D = int(rng.choice([1, 3, 8, 17, 32]))
    T = int(rng.choice([2048, 3000, 5000, 12345, 40000]))
    if K > 64:
        T = min(T, 12345)
    pb = make_problem(K, D, T, seed=seed, miss=float(rng.choice([0.0, 0.1])), sep=float(rng.choice([0.5, 3.0, 20.0])))
    par = (pb["mod_init"], pb["ltran"], pb["mu"], pb["sigma"], pb["kappa"], pb["nu"])
    e.set_precision("f64")
    e.set_obs(pb["obs"], pb["mask"])
    e.set_globals(pb["mod_init"], pb["ltran"])
    e.set_emission_niw(pb["mu"], pb["sigma"], pb["kappa"], pb["nu"])
    st = e.estep(np.zeros(1, dtype=np.int64), T, flags=L.MASK_AS_NAN)
    ref = ref_c.estep_minibatch(pb["obs"], pb["mask"], np.zeros(1, dtype=np.int64), T, *par, flags=L.MASK_AS_NAN)
    xs = max(1.0, float(np.nanmax(np.abs(pb["obs"]))))
    check_stats(st.buf, ref, K, D, T, xs, 1e-6, 1e-9, "chain K=%d D=%d T=%d" % (K, D, T))
    q = e.read_rows("var_x", 0, T)
    x = pb["obs"].copy()
    x[pb["mask"]] = np.nan
    ll = ref_c.lliks_niw(x, pb["mu"], pb["sigma"], pb["kappa"], pb["nu"])
    qr, _ = ref_c.posterior(ref_c.forward(ll, pb["mod_init"], pb["ltran"]), ref_c.backward(ll, pb["ltran"]))
    np.testing.assert_allclose(q, qr, rtol=1e-6, atol=1e-12, err_msg="chain var_x")
    return (K, D, T)


def run_api_case(e, L, seed):
    """The callers around the E-step, HipEngine vs OracleEngine on one random problem: FFBS
    (lalpha + exact draws), held-out predictive log-probability, state decoding + count matrix,
    host-supplied lliks, Categorical emissions, three iterations of the device-resident SVI loop."""
    from oracle.engine import OracleEngine
    from pysvihmm_amd.distributions import niw_prior_logpart
    from tests.helpers import ffbs_draws_exact
    from scipy.special import digamma
    rng = np.random.default_rng(seed)
    K = int(rng.choice([1, 2, 3, 5, 8, 16, 17, 33, 64, 65, 100]))
    D = int(rng.choice([1, 2, 3, 8, 9, 16]))
    T = int(rng.choice([300, 1000, 2500, 5000]))
    sep = float(rng.choice([0.5, 3.0, 20.0]))
    pb = make_problem(K, D, T, seed=seed, miss=float(rng.choice([0.0, 0.1, 0.3])), sep=sep)
    if rng.random() < 0.3:
        vi = np.where(rng.random(K) < 0.5, 10.0 ** -rng.uniform(3, 6, K), 0.3) * (0.5 + rng.random(K))
        pb["mod_init"] = digamma(vi + 1e-9) - digamma(vi.sum() + 1e-9)
    what = "K=%d D=%d T=%d sep=%g" % (K, D, T, sep)
    o = OracleEngine()
    e.set_precision("f64")
    for eng in (e, o):
        eng.set_obs(pb["obs"], pb["mask"])
        eng.set_globals(pb["mod_init"], pb["ltran"])
        eng.set_emission_niw(pb["mu"], pb["sigma"], pb["kappa"], pb["nu"])
    # --- FFBS (the reference passes log(var_tran + eps) as transition weights, hmm_fast.pyx:91-93)
    DE = np.finfo(np.float64).eps
    vt = pb["var_tran"] / pb["var_tran"].sum(1)[:, None]
    logA = np.log(vt + DE)
    for eng in (e, o):
        eng.set_globals(pb["mod_init"], logA)
    u = rng.random(T)
    z, la = e.ffbs(logA, u)
    zo, lao = o.ffbs(logA, u)
    np.testing.assert_allclose(la, lao, rtol=1e-9, atol=1e-7, err_msg=what + " ffbs lalpha")
    bad, risky = ffbs_draws_exact(z, la, logA, u)
    assert bad == 0, (what, "ffbs draws", bad, risky)
    for eng in (e, o):
        eng.set_globals(pb["mod_init"], pb["ltran"])
    # --- windows
    Lm = int(rng.choice([1, 3, 17, 65, 257]))
    Lm = min(Lm, T // 2)
    B = int(rng.choice([1, 4, 40, 260]))
    while B * Lm * K * K > 3e7 and B > 1:
        B //= 2
    starts = rng.integers(0, T - Lm + 1, size=B)
    # --- predictive log-probability of the held-out rows
    v, n = e.pred_logprob(starts, Lm)
    vo, no = o.pred_logprob(starts, Lm)
    assert n == no, (what, "pred_logprob count", n, no)
    if n:
        np.testing.assert_allclose(v, vo, rtol=1e-8, err_msg=what + " pred_logprob")
    # --- decoding + count matrix
    true = rng.integers(-1, K + 1, size=B * Lm)
    ro = o.forward_backward(starts, Lm, want=("var_x",))
    e.forward_backward(starts, Lm, want=())
    zz, dm = e.state_argmax(true)
    q = ro["var_x"].reshape(-1, K)
    top2 = np.sort(np.concatenate([q, np.full((len(q), 1), -1.0)], axis=1), axis=1)[:, -2:]
    clear = (top2[:, 1] - top2[:, 0]) > 1e-9
    assert np.array_equal(zz[clear], np.argmax(q, axis=1)[clear]), what + " argmax"
    ok = (true >= 0) & (true < K)
    ref = np.zeros((K, K), dtype=np.int64)
    np.add.at(ref, (zz[ok], true[ok]), 1)
    assert np.array_equal(dm, ref), what + " count matrix"
    # --- host-supplied lliks == the NIW kernel's own
    st1 = e.estep(starts, Lm, flags=L.TRANS_WRAP | L.MASK_AS_NAN)
    ll = o.loglik(starts, Lm, flags=L.MASK_AS_NAN)
    e.set_lliks(ll)
    st2 = e.estep(starts, Lm, flags=L.TRANS_WRAP | L.MASK_AS_NAN | L.USE_HOST_LLIKS)
    sc = B * Lm
    xs = max(1.0, float(np.nanmax(np.abs(pb["obs"]))))
    check_stats(st2.buf, st1.buf, K, D, sc, xs, 1e-6, 1e-9, what + " host lliks")
    # --- device-resident SVI loop, three iterations
    # >= 1: quirk Q2 adds nwin * (prior_tran - 1); below 1 var_tran goes negative and both
    # implementations turn chaotic (psi of negative arguments)
    prior_tran = 1.0 + rng.random((K, K)) if rng.random() < 0.5 else np.ones((K, K))
    var_tran0 = np.maximum(pb["var_tran"], 1.0)
    mu0 = np.tile(pb["obs"].mean(0), (K, 1))
    sg0 = np.tile(0.75 * np.atleast_2d(np.cov(pb["obs"].T)).reshape(D, D), (K, 1, 1))
    ka0, nu0 = np.full(K, 0.01), np.full(K, D + 2.0)
    prior = (mu0, sg0, ka0, nu0)
    factors = (pb["mu"], pb["sigma"], pb["kappa"], pb["nu"])
    Lh = max(Lm // 2, 1)
    bA, bE = (T - 2 * Lh - 1) / (2. * Lh * B), (T - 2 * Lh - 1) / ((2. * Lh + 1) * B)
    res = []
    for eng in (e, o):
        eng.svi_begin(prior_tran, var_tran0, prior, factors, niw_prior_logpart(sg0, nu0), 3, 1.0)
        r2 = np.random.default_rng(seed + 1)
        for it in range(3):
            st = r2.integers(0, T - Lm + 1, size=B)
            eng.svi_iteration(it, st, B, Lm, L.TRANS_WRAP, (it + 1.0) ** -0.7, bA, bE)
        res.append((eng.svi_read_state(), eng.svi_read_elbo(3)[0]))
    (sa, ea), (sb, eb) = res
    for nme, a, b in zip(("var_tran", "var_init", "mu", "sigma", "kappa", "nu"), sa, sb):
        np.testing.assert_allclose(a, b, rtol=1e-5 if nme == "sigma" else 1e-6, atol=1e-8, err_msg=what + " svi " + nme)
    np.testing.assert_allclose(ea, eb, rtol=1e-8, err_msg=what + " svi elbo")
    # --- Categorical emissions on the same state path
    V = int(rng.choice([2, 5, 9, 30]))
    theta = rng.dirichlet(np.ones(V) * 0.3, size=K)
    cobs = np.array([rng.choice(V, p=theta[s_]) for s_ in pb["sts"][:T]], dtype=float)
    alpha = rng.random((K, V)) * 5 + 0.2
    table = digamma(alpha) - digamma(alpha.sum(1))[:, None]
    for eng in (e, o):
        eng.set_obs(cobs, pb["mask"])
        eng.set_globals(pb["mod_init"], pb["ltran"])
        eng.set_emission_cat(table)
    for flags in (L.TRANS_WRAP, L.TRANS_WRAP | L.MASK_AS_NAN):
        a, b = e.estep(starts, Lm, flags=flags), o.estep(starts, Lm, flags=flags)
        np.testing.assert_allclose(a.A_raw, b.A_raw, rtol=1e-6, atol=1e-9 * sc, err_msg=what + " cat A_raw")
        np.testing.assert_allclose(a.counts, b.counts, rtol=1e-6, atol=1e-9 * sc, err_msg=what + " cat counts")
        np.testing.assert_allclose(a.lb[0], b.lb[0], rtol=1e-9, atol=1e-6, err_msg=what + " cat lb")
    # --- the device-resident loop of the Categorical family (round 4), AdaGrad on or off
    alpha0 = np.full((K, V), float(rng.choice([1.0, 1.5, 3.0])))
    ada = rng.random() < 0.5
    sd = int(rng.integers(1 << 30))
    res = []
    for eng in (e, o):
        eng.svi_begin_cat(np.ones((K, K)), np.maximum(pb["var_tran"], 1.0), alpha0, alpha0 + alpha, 3)
        if ada:
            eng.svi_set_adagrad(np.ones((K, K)))
        r2 = np.random.default_rng(sd)
        for it in range(3):
            st = r2.integers(0, T - Lm + 1, size=B)
            eng.svi_iteration(it, st, B, Lm, L.TRANS_WRAP, (it + 1.0) ** -0.7, bA, bE)
        res.append((eng.svi_read_factors(), eng.svi_read_elbo(3)[0], eng.svi_read_adagrad() if ada else None))
        if ada:
            eng.svi_set_adagrad(None)
    (fa, ea, ga), (fb, eb, gb) = res
    for nme, a, b in zip(("var_tran", "var_init", "alpha"), fa, fb):
        np.testing.assert_allclose(a, b, rtol=1e-6, atol=1e-8, err_msg=what + " cat loop " + nme + (" adagrad" if ada else ""))
    np.testing.assert_allclose(ea, eb, rtol=1e-8, err_msg=what + " cat loop elbo")
    if ada:
        np.testing.assert_allclose(ga, gb, rtol=1e-6, atol=1e-12, err_msg=what + " cat loop ada_G")
    return what


def class_case(seed):
    """The class-level case of `seed`: (kind, K, D, T, obs, mask, opts, infer_kw, what, model) with model(engine) ->
    a fresh instance of the drawn class on that engine (run_class_case; tests/test_gpu_tiny_windows.py replays the
    recorded deviating seeds through it)."""
    from pysvihmm_amd import hmmsgd_metaobs, hmmbatchcd, hmmbatchsgd
    from pysvihmm_amd.distributions import Gaussian
    rng = np.random.default_rng(seed)
    K = int(rng.choice([2, 3, 5, 8, 17]))
    D = int(rng.choice([1, 2, 3, 8]))
    T = int(rng.choice([300, 700, 1500]))
    pb = make_problem(K, D, T, seed=seed, miss=0.0, sep=float(rng.choice([1.0, 3.0, 20.0])))
    obs = pb["obs"]
    mask = (rng.random(T) < 0.1) if rng.random() < 0.5 else None
    kind = str(rng.choice(["metaobs", "metaobs", "metaobs", "batchcd", "batchsgd"]))
    half = int(rng.choice([1, 3, 8, 20]))
    opts = dict(metaobs_half=half, mb_sz=int(rng.choice([1, 2, 7, 20])), maxit=int(rng.choice([2, 5])),
                metaobs_fun=str(rng.choice(["unif", "noverlap"])), full_predprob=bool(rng.random() < 0.3) and mask is not None,
                growBuffer=bool(rng.random() < 0.25))
    if opts["metaobs_fun"] == "noverlap" and 3 * (opts["mb_sz"] + 1) * (half + 1) > T - 2 * half:
        opts["metaobs_fun"] = "unif"      # the reference's rejection loop (:246-252) never ends when the windows cannot fit
    infer_kw = {}
    if kind == "metaobs" and not opts["growBuffer"] and rng.random() < 0.25:
        infer_kw = dict(adaptive=True, perIter=2, epsilon=1e-2, minHalfL=1, Lincrement=2, Lcutoff=10)
    if kind == "metaobs" and rng.random() < 0.3:
        infer_kw["device_loop"] = False
    what = "%s K=%d D=%d T=%d %r %r" % (kind, K, D, T, opts if kind == "metaobs" else "", infer_kw)

    def model(engine):
        np.random.seed(seed % 100000)
        sg0 = 0.75 * np.atleast_2d(np.cov(obs.T)).reshape(D, D)
        prior = np.array([Gaussian(mu_0=obs.mean(0), sigma_0=sg0, kappa_0=0.01, nu_0=D + 2) for _ in range(K)])
        m = None if mask is None else mask.copy()
        if kind == "metaobs":
            return hmmsgd_metaobs.VBHMM(obs.copy(), np.ones(K), np.ones((K, K)), prior, tau=1.0, kappa=0.7, mask=m,
                                        seed=seed % 1000, engine=engine, **opts)
        if kind == "batchcd":
            return hmmbatchcd.VBHMM(obs.copy(), np.ones(K), np.ones((K, K)), prior, mask=m, maxit=3, engine=engine)
        return hmmbatchsgd.VBHMM(obs.copy(), np.ones(K), np.ones((K, K)), prior, tau=1.0, kappa=0.7, mask=m, maxit=3,
                                 engine=engine)
    return dict(kind=kind, K=K, D=D, T=T, obs=obs, mask=mask, opts=opts, infer_kw=infer_kw, what=what, model=model)


def run_class_case(e, seed):
    """The reference's three classes end to end (small problems, a few iterations), HIP engine vs
    oracle engine: hmmsgd_metaobs with random options (mask, full_predprob, growBuffer,
    adaptive, noverlap sampler, host / device loop), hmmbatchcd, hmmbatchsgd."""
    from oracle.engine import OracleEngine
    e.set_precision("f64")     # (a call sequence before may have left the shared handle in the fp32 mode)
    c = class_case(seed)
    kind, K, opts, infer_kw, what, model = c["kind"], c["K"], c["opts"], c["infer_kw"], c["what"], c["model"]
    a, b = model(e), model(OracleEngine())
    a.infer(**infer_kw)
    b.infer(**infer_kw)
    np.testing.assert_allclose(a.var_tran, b.var_tran, rtol=1e-6, atol=1e-9, err_msg=what + " var_tran")
    np.testing.assert_allclose(a.var_init, b.var_init, rtol=1e-6, atol=1e-9, err_msg=what + " var_init")
    for k in range(K):
        np.testing.assert_allclose(a.var_emit[k].mu_mf, b.var_emit[k].mu_mf, rtol=1e-6, atol=1e-6, err_msg=what + " mu")
        np.testing.assert_allclose(a.var_emit[k].sigma_mf, b.var_emit[k].sigma_mf, rtol=2e-5, atol=1e-6, err_msg=what + " sigma")
    np.testing.assert_allclose(a.elbo_vec, b.elbo_vec, rtol=2e-6, err_msg=what + " elbo")
    np.testing.assert_allclose(a.var_x, b.var_x, rtol=1e-6, atol=1e-11, err_msg=what + " var_x")
    np.testing.assert_allclose(a.lalpha, b.lalpha, rtol=1e-7, atol=1e-7, err_msg=what + " lalpha")
    if kind == "metaobs" and opts["full_predprob"]:
        np.testing.assert_allclose(a.pred_logprob_full_mean, b.pred_logprob_full_mean, rtol=1e-8, err_msg=what + " predprob")
    return what


def _run_sequence(e, L, seed, nops=30):
    """One random sequence of API calls on ONE handle (and the same sequence on an OracleEngine):
    parameter / data uploads, precision switches, E-steps of changing shapes, message reads of the
    lazily rebuilt intermediates, the callers around the E-step, short device-resident SVI runs --
    what is compared after every call is that call's own result.  Aims at the state the handle
    carries between calls (cached shapes, lazily materialised logs, side streams, pending sums)."""
    from oracle.engine import OracleEngine
    from pysvihmm_amd.distributions import niw_prior_logpart
    from scipy.special import digamma
    rng = np.random.default_rng(seed)
    K = int(rng.choice([2, 3, 7, 16, 33, 64, 70]))
    D = int(rng.choice([1, 3, 8, 16]))
    T = int(rng.choice([400, 1200, 3000]))
    o = OracleEngine()
    hist = []
    state = {"prec": "f64", "fresh": None, "pb": None}

    def new_problem(keep_obs=False):
        pb = make_problem(K, D, T, seed=int(rng.integers(1 << 30)), miss=float(rng.choice([0.0, 0.1])),
                          sep=float(rng.choice([0.5, 3.0, 20.0])))
        if rng.random() < 0.3:
            vi = np.where(rng.random(K) < 0.5, 10.0 ** -rng.uniform(3, 6, K), 0.3) * (0.5 + rng.random(K))
            pb["mod_init"] = digamma(vi + 1e-9) - digamma(vi.sum() + 1e-9)
        # data anywhere on the axis: the handle keeps its resident copy centred, the C ABI is
        # coordinate-invariant (means / statistics in the caller's coordinates)
        offs = float(rng.choice([0.0, 0.0, 40.0, -3e3, 1e5]))
        pb["obs"] = pb["obs"] + offs
        pb["mu"] = pb["mu"] + offs
        pb["offset"] = offs
        if keep_obs and state["pb"] is not None:
            pb["mu"] = pb["mu"] - offs + state["pb"]["offset"]
            for k in ("obs", "mask", "sts", "offset"):
                pb[k] = state["pb"][k]
        state["pb"] = pb
        return pb

    # replay aid (FUZZ_ORACLE_CENTRED=1): the oracle sees the data at offset 0 -- sigma_mf, var_tran and
    # the ELBO are translation invariant, so an "svi" mismatch at a large offset can be attributed:
    # the side that disagrees with the centred oracle is the one whose rounding it is
    octr = bool(os.environ.get("FUZZ_ORACLE_CENTRED"))
    real_allclose = np.testing.assert_allclose
    if octr:
        # (the other ops compare offset-dependent quantities: under the aid they only run, to keep the
        #  handle's state and the random stream of the sequence)
        def gated(*a, **k):
            if state.get("op") in ("svi", "svi_diag"):
                return real_allclose(*a, **k)
        np.testing.assert_allclose = gated

    def upload(what):
        pb = state["pb"]
        for eng in (e, o):
            sh = pb["offset"] if (octr and eng is o) else 0.0
            if what in ("all", "obs"):
                eng.set_obs(pb["obs"] - sh, pb["mask"] if pb["mask"].any() else None)
            if what in ("all", "params"):
                eng.set_globals(pb["mod_init"], pb["ltran"])
                eng.set_emission_niw(pb["mu"] - sh, pb["sigma"], pb["kappa"], pb["nu"])
        state["fresh"] = None

    def windows():
        Lm = int(rng.choice([1, 2, 9, 33, 65, 257]))
        Lm = min(Lm, T // 2)
        B = int(rng.choice([1, 3, 20, 64, 200, 300]))
        while B * Lm * K * (K + 3 * D * D) > 4e8 and B > 1:
            B //= 2
        return rng.integers(0, T - Lm + 1, size=B), Lm

    new_problem()
    e.set_precision("f64")
    upload("all")
    for step in range(nops):
        pb = state["pb"]
        f32 = state["prec"] == "f32"
        op = str(rng.choice(["estep", "estep", "fb", "read", "read", "params", "obs", "prec", "predlp", "argmax",
                             "ffbs", "hostll", "svi", "loglik", "inner", "shift", "class", "reobs", "diag",
                             "svi_diag"]))
        hist.append(op)
        state["op"] = op
        what = "seq seed=%d K=%d D=%d T=%d offset=%g step %d %s (history %s)" % (seed, K, D, T, pb["offset"], step, op,
                                                                                " ".join(hist[-8:]))
        xs = max(1.0, float(np.nanmax(np.abs(pb["obs"]))))
        if op == "params":
            new_problem(keep_obs=True)
            upload("params")
        elif op == "shift":
            # the centre moves, nothing else: whatever follows must still agree with the oracle
            e.shift_obs(rng.normal(size=D) * float(rng.choice([0.1, 5.0, 300.0])))
            state["fresh"] = None
        elif op == "reobs":
            # the SAME data uploaded again under the parameters already on the device (the new
            # centre is chosen afresh; device-side means and theta follow it)
            for eng in (e, o):
                eng.set_obs(pb["obs"], pb["mask"] if pb["mask"].any() else None)
            state["fresh"] = None
        elif op == "class":
            # a class borrows the handle for its own model and data (the state
            # that broke the bench's cross-check); afterwards the sequence's data come back and the
            # raw calls go on with the caller's parameters
            from pysvihmm_amd import hmmsgd_metaobs
            from pysvihmm_amd.distributions import Gaussian
            Kc, Dc, Tc = 3, D, 500
            cp = make_problem(Kc, Dc, Tc, seed=int(rng.integers(1 << 30)), sep=3.0)
            coff = float(rng.choice([0.0, 77.0, -2e4]))
            cobs = cp["obs"] + coff
            out = []
            for eng in (e, o):
                np.random.seed(11)
                prior = np.array([Gaussian(mu_0=cobs.mean(0), sigma_0=0.75 * np.atleast_2d(np.cov(cobs.T)),
                                           kappa_0=0.01, nu_0=Dc + 2) for _ in range(Kc)])
                m = hmmsgd_metaobs.VBHMM(cobs, np.ones(Kc), np.ones((Kc, Kc)), prior, tau=1.0, kappa=0.7, metaobs_half=6,
                                         mb_sz=5, maxit=3, seed=2, engine=eng)
                m.infer()
                out.append(m)
            if not f32:     # (tolerance: the oracle's raw-moment arithmetic at the class's offset, see "svi")
                ctol = max(1e-6, 1e-12 * coff ** 2)
                np.testing.assert_allclose(out[0].var_tran, out[1].var_tran, rtol=ctol, atol=1e-9, err_msg=what)
                np.testing.assert_allclose(out[0].elbo_vec, out[1].elbo_vec, rtol=10 * ctol, err_msg=what)
            upload("all")
        elif op == "diag":
            st, Lm = windows()
            dp = (pb["mu"], 0.5 + 4 * rng.random((K, D)), 2.0 + 5 * rng.random((K, D)), 1.0 + 6 * rng.random((K, D)))
            for eng in (e, o):
                eng.set_emission_diag(*dp)
            a, b = e.estep(st, Lm, flags=L.TRANS_WRAP), o.estep(st, Lm, flags=L.TRANS_WRAP)
            sc = len(st) * Lm
            tol, at = (2e-3, 2e-4) if f32 else (1e-6, 1e-9)
            np.testing.assert_allclose(a.A_raw, b.A_raw, rtol=tol, atol=at * sc, err_msg=what)
            np.testing.assert_allclose(a.neff, b.neff, rtol=tol, atol=at * sc, err_msg=what)
            if not f32:
                np.testing.assert_allclose(a.xbar, b.xbar, rtol=tol, atol=at * sc * xs, err_msg=what)
                np.testing.assert_allclose(a.xsq, b.xsq, rtol=tol, atol=at * sc * xs * xs, err_msg=what)
                np.testing.assert_allclose(a.lb, b.lb, rtol=1e-9, atol=1e-6, err_msg=what)
            upload("params")          # back to the NIW family
            state["fresh"] = None
        elif op == "svi_diag":
            # the diagonal family's device-resident loop (round 4), AdaGrad on or off
            st, Lm = windows()
            B = len(st)
            ok = ~np.isnan(pb["obs"]).any(1)
            m0 = np.tile(np.nanmean(pb["obs"], 0), (K, 1))
            v0 = np.tile(np.var(pb["obs"][ok], 0) + 1e-3, (K, 1))
            prior = (m0, np.full((K, D), 0.05), np.full((K, D), 2.0), 2.0 * v0)
            factors = (pb["mu"], 0.5 + 4 * rng.random((K, D)), 2.0 + 5 * rng.random((K, D)), (1.0 + 6 * rng.random((K, D))) * v0)
            Lh = max(Lm // 2, 1)
            bA, bE = (T - 2 * Lh - 1) / (2. * Lh * B), (T - 2 * Lh - 1) / ((2. * Lh + 1) * B)
            ada = rng.random() < 0.5
            sd = int(rng.integers(1 << 30))
            res = []
            oc = None
            if pb["offset"] != 0.0 and not octr:
                oc = OracleEngine()
                oc.set_obs(pb["obs"] - pb["offset"], pb["mask"] if pb["mask"].any() else None)
            for eng in (e, oc if oc is not None else o):
                sh = pb["offset"] if (eng is oc or (octr and eng is o)) else 0.0
                eng.svi_begin_diag(np.ones((K, K)), np.maximum(pb["var_tran"], 1.0), (prior[0] - sh,) + prior[1:],
                                   (factors[0] - sh,) + factors[1:], 2)
                if ada:
                    eng.svi_set_adagrad(np.ones((K, K)))
                r2 = np.random.default_rng(sd)
                for it in range(2):
                    eng.svi_iteration(it, r2.integers(0, T - Lm + 1, size=B), B, Lm, L.TRANS_WRAP, (it + 1.0) ** -0.7, bA, bE)
                res.append((eng.svi_read_factors(), eng.svi_read_elbo(2)[0]))
                if ada:
                    eng.svi_set_adagrad(None)
            (fa, ea), (fb, eb) = res
            if os.environ.get("FUZZ_VERBOSE"):
                print(what, "B", B, "Lm", Lm, "bE", bE, "adagrad", ada, "prec", state["prec"], "elbo", ea, eb)
            if oc is not None:
                oc.close()
            if octr or oc is not None:
                fb = (fb[0], fb[1], (fb[2][0] + pb["offset"],) + tuple(fb[2][1:]))
            offs_eff = 0.0 if oc is not None else pb["offset"]
            tol = 5e-3 if f32 else 1e-6
            tol = max(tol, 1e-12 * offs_eff ** 2)
            small = f32 and (B * Lm < 2000 or bE > 20.0)     # (limits of the fp32 mode, see "svi")
            if small:
                assert all(np.all(np.isfinite(a)) for a in (fa[0], fa[1]) + tuple(fa[2])), what
            else:
                np.testing.assert_allclose(fa[0], fb[0], rtol=tol, atol=tol * 1e-2 * (1 + np.abs(fb[0]).max()), err_msg=what + " var_tran")
                np.testing.assert_allclose(fa[1], fb[1], rtol=tol, atol=tol * 1e-2, err_msg=what + " var_init")
                for nme, a, b in zip(("mu", "nus", "alphas", "betas"), fa[2], fb[2]):
                    if f32 and nme == "betas":
                        continue      # (raw second moments of the fp32 statistics: the scale cancels, as sigma in "svi")
                    # (fp32 mode behind a `shift` op -- the resident copy up to 300 spreads off its centre:
                    #  seed 61380147, one mean 2.8e-3 off on 16 448 rows)
                    at = tol * (1e-1 if f32 else 1e-2) * (1 + np.abs(b).max())
                    if nme == "betas":
                        at += 4e-16 * float(np.max(fb[2][1])) * offs_eff ** 2 * 10
                    np.testing.assert_allclose(a, b, rtol=tol, atol=at, err_msg=what + " " + nme)
                if not f32:
                    np.testing.assert_allclose(ea, eb, rtol=max(1e-8, 1e-11 * offs_eff ** 2), err_msg=what + " elbo")
            new_problem(keep_obs=True)
            upload("params")
        elif op == "obs":
            # new data under the parameters already on the device (their means move with the data's
            # offset on the host side of the comparison; on the device they must follow the new centre)
            old = state["pb"]
            keep = {k: old[k] for k in ("mod_init", "ltran", "sigma", "kappa", "nu")}
            mu_rel = old["mu"] - old["offset"]
            new_problem()
            state["pb"].update(keep)
            state["pb"]["mu"] = mu_rel + state["pb"]["offset"]
            upload("obs")
            for eng in (e, o):     # (the means moved with the data: an upload in the caller's coordinates)
                eng.set_emission_niw(state["pb"]["mu"], keep["sigma"], keep["kappa"], keep["nu"])
        elif op == "prec":
            state["prec"] = "f32" if state["prec"] == "f64" else "f64"
            if os.environ.get("FUZZ_F64"):          # replay aid: the same sequence without the fp32 mode
                state["prec"] = "f64"
            e.set_precision(state["prec"])
            state["fresh"] = None
        elif op in ("estep", "inner"):
            st, Lm = windows()
            flags = int(rng.choice([L.TRANS_WRAP, L.MASK_AS_NAN, L.TRANS_WRAP | L.MASK_AS_NAN, 0]))
            inner = None
            if op == "inner" and Lm >= 3:
                off = int(rng.integers(0, Lm // 2))
                inner = (off, int(rng.integers(1, Lm - off + 1)))
            a, b = e.estep(st, Lm, flags=flags, inner=inner), o.estep(st, Lm, flags=flags, inner=inner)
            sc = len(st) * Lm
            if f32:
                assert np.all(np.isfinite(a.buf)), what
                np.testing.assert_allclose(a.A_raw, b.A_raw, rtol=2e-3, atol=2e-4 * sc, err_msg=what)
                np.testing.assert_allclose(a.neff, b.neff, rtol=2e-3, atol=2e-4 * sc, err_msg=what)
            else:
                check_stats(a.buf, b.buf, K, D, sc, xs, 1e-6, 1e-9, what)
            state["fresh"] = (len(st), Lm)
        elif op == "fb":
            st, Lm = windows()
            want = tuple(w for w in ("lalpha", "lbeta", "var_x", "local_lb") if rng.random() < 0.6)
            flags = int(rng.choice([0, L.MASK_AS_NAN]))
            a, b = e.forward_backward(st, Lm, flags=flags, want=want), o.forward_backward(st, Lm, flags=flags, want=want)
            for w in want:
                if f32 and w not in ("lalpha", "lbeta"):
                    np.testing.assert_allclose(a[w], b[w], rtol=1e-3, atol=1e-4, err_msg=what + " " + w)
                else:
                    np.testing.assert_allclose(a[w], b[w], rtol=1e-6 if w == "var_x" else 1e-9,
                                               atol=1e-11 if w == "var_x" else 1e-7, err_msg=what + " " + w)
            state["fresh"] = (len(st), Lm)
        elif op == "read":
            if state["fresh"] is None:
                continue
            B, Lm = state["fresh"]
            w = str(rng.choice(["lliks", "lalpha", "lbeta", "var_x"]))
            r0 = int(rng.integers(0, B * Lm))
            n = int(rng.integers(1, min(B * Lm - r0, 300) + 1))
            a, b = e.read_rows(w, r0, n), o.read_rows(w, r0, n)
            if f32 and w == "var_x":
                np.testing.assert_allclose(a, b, rtol=1e-3, atol=1e-4, err_msg=what + " " + w)
            else:
                np.testing.assert_allclose(a, b, rtol=1e-6 if w == "var_x" else 1e-9,
                                           atol=1e-11 if w == "var_x" else 1e-7, err_msg=what + " " + w)
        elif op == "predlp":
            st, Lm = windows()
            a, b = e.pred_logprob(st, Lm), o.pred_logprob(st, Lm)
            assert a[1] == b[1], what
            if a[1]:
                np.testing.assert_allclose(a[0], b[0], rtol=1e-3 if f32 else 1e-8, err_msg=what)
            state["fresh"] = None
        elif op == "argmax":
            st, Lm = windows()
            rb = o.forward_backward(st, Lm, want=("var_x",))
            e.forward_backward(st, Lm, want=())
            z, _ = e.state_argmax()
            q = rb["var_x"].reshape(-1, K)
            top2 = np.sort(np.concatenate([q, np.full((len(q), 1), -1.0)], axis=1), axis=1)[:, -2:]
            clear = (top2[:, 1] - top2[:, 0]) > (1e-3 if f32 else 1e-9)
            assert np.array_equal(z[clear], np.argmax(q, axis=1)[clear]), what
            state["fresh"] = (len(st), Lm)
        elif op == "ffbs":
            if T > 1500:
                continue
            DE = np.finfo(np.float64).eps
            vt = pb["var_tran"] / pb["var_tran"].sum(1)[:, None]
            logA = np.log(vt + DE)
            for eng in (e, o):
                eng.set_globals(pb["mod_init"], logA)
            u = rng.random(T)
            (z, la), (zo, lao) = e.ffbs(logA, u), o.ffbs(logA, u)
            np.testing.assert_allclose(la, lao, rtol=1e-9, atol=1e-7, err_msg=what)
            from tests.helpers import ffbs_draws_exact
            bad, risky = ffbs_draws_exact(z, la, logA, u)
            assert bad == 0, (what, bad, risky)
            for eng in (e, o):
                eng.set_globals(pb["mod_init"], pb["ltran"])
            state["fresh"] = None
        elif op == "hostll":
            st, Lm = windows()
            ll = o.loglik(st, Lm, flags=L.MASK_AS_NAN)
            for eng in (e, o):
                eng.set_lliks(ll)
            fl = L.TRANS_WRAP | L.MASK_AS_NAN | L.USE_HOST_LLIKS
            a, b = e.estep(st, Lm, flags=fl), o.estep(st, Lm, flags=fl)
            check_stats(a.buf, b.buf, K, D, len(st) * Lm, xs, 2e-3 if f32 else 1e-6, 2e-4 if f32 else 1e-9, what)
            state["fresh"] = (len(st), Lm)
        elif op == "loglik":
            st, Lm = windows()
            st = st[:5]
            a, b = e.loglik(st, Lm, flags=L.MASK_AS_NAN), o.loglik(st, Lm, flags=L.MASK_AS_NAN)
            np.testing.assert_allclose(a, b, rtol=1e-9, atol=1e-7, err_msg=what)
            state["fresh"] = None
        elif op == "svi":
            st, Lm = windows()
            B = len(st)
            prior_tran = np.ones((K, K))
            mu0 = np.tile(np.nanmean(pb["obs"], 0), (K, 1))
            sg0 = np.tile(0.75 * np.atleast_2d(np.cov(pb["obs"][~np.isnan(pb["obs"]).any(1)].T)).reshape(D, D), (K, 1, 1))
            prior = (mu0, sg0, np.full(K, 0.01), np.full(K, D + 2.0))
            factors = (pb["mu"], pb["sigma"], pb["kappa"], pb["nu"])
            Lh = max(Lm // 2, 1)
            bA, bE = (T - 2 * Lh - 1) / (2. * Lh * B), (T - 2 * Lh - 1) / ((2. * Lh + 1) * B)
            res = []
            sd = int(rng.integers(1 << 30))
            ada = rng.random() < 0.3      # AdaGrad accumulator beside var_tran (round 4)
            # The oracle side of this op runs on the data CENTRED (its own instance): in the caller's
            # coordinates the oracle's raw-moment arithmetic loses ~eps kappa offset^2 / (nu - D - 1)
            # (seed 51280046: 1.7e-4 on sigma at offset -3000, reproduced oracle-against-oracle on the CPU),
            # and it is the device, not the oracle's rounding, that the campaign tests
            oc = None
            if pb["offset"] != 0.0 and not octr:
                oc = OracleEngine()
                oc.set_obs(pb["obs"] - pb["offset"], pb["mask"] if pb["mask"].any() else None)
            for eng in (e, oc if oc is not None else o):
                sh = pb["offset"] if (eng is oc or (octr and eng is o)) else 0.0
                eng.svi_begin(prior_tran, np.maximum(pb["var_tran"], 1.0), (mu0 - sh,) + prior[1:],
                              (factors[0] - sh,) + factors[1:], niw_prior_logpart(sg0, prior[3]), 2, 1.0)
                if ada:
                    eng.svi_set_adagrad(np.ones((K, K)))
                r2 = np.random.default_rng(sd)
                for it in range(2):
                    eng.svi_iteration(it, r2.integers(0, T - Lm + 1, size=B), B, Lm, L.TRANS_WRAP, (it + 1.0) ** -0.7, bA, bE)
                try:
                    res.append((eng.svi_read_state(), eng.svi_read_elbo(2)[0]))
                    if ada:
                        eng.svi_set_adagrad(None)
                except RuntimeError as ex:
                    if f32 and B * Lm < 2000 and eng is e and "positive definite" in str(ex):
                        # fp32 raw moments of a few rows times a batch factor of hundreds: the scale
                        # matrix can cancel to an indefinite one (limits of the mode); the
                        # engine says so instead of computing on
                        res = None
                        break
                    raise RuntimeError("%s B=%d Lm=%d prec=%s engine=%s: %s" % (what, B, Lm, state["prec"], eng.name, ex))
            if oc is not None:
                oc.close()
            if res is None:
                new_problem(keep_obs=True)
                upload("params")
                continue
            (sa, ea), (sb, eb) = res
            if octr or oc is not None:
                sb = list(sb); sb[2] = sb[2] + pb["offset"]
            offs_eff = 0.0 if oc is not None else pb["offset"]
            if os.environ.get("FUZZ_VERBOSE"):
                print(what, "B", B, "Lm", Lm, "elbo", ea, eb)
            tol = 5e-3 if f32 else 1e-6
            # The ORACLE runs the reference's natural-gradient arithmetic on raw second moments in the
            # caller's coordinates (kappa mu mu^T cancels against sigma: ~1e-16 kappa offset^2, kappa
            # up to ~1e4 here), the device in centred ones: at large offsets the comparison is
            # limited by the oracle's own rounding, not the device's
            tol = max(tol, 1e-12 * offs_eff ** 2)
            if f32 and (B * Lm < 2000 or bE > 20.0):
                # a few dozen rows times a batch factor of ~100: the fp32 raw moments' cancellation
                # (limits of the mode) reaches the second iteration's posteriors
                assert all(np.all(np.isfinite(a)) for a in sa) and np.all(np.isfinite(ea)), what
                new_problem(keep_obs=True)
                upload("params")
                continue
            for nme, a, b in zip(("var_tran", "var_init", "mu", "sigma", "kappa", "nu"), sa, sb):
                if f32 and nme == "sigma":
                    # fp32 statistics are raw second moments: the scale matrix S - n xbar xbar^T of
                    # a tiny, batch-factor-amplified minibatch cancels 1e3 : 1 (the tolerance of the
                    # mode is stated on the natural gradients, which are the raw moments)
                    continue
                # (fp32 mode: single entries of a batch-factor-amplified update can be off by a few times
                #  the mode's tolerance relative to the array's scale)
                at = tol * (4e-2 if f32 else 1e-2) * (1 + np.abs(b).max())
                if nme == "sigma":
                    # the oracle's sigma = (e3 - kappa mu mu^T) / (nu - p - 1) in raw coordinates: absolute
                    # rounding ~eps kappa offset^2 (seed 424280063, offset -3000: 2e-6; against an oracle fed
                    # the same data centred -- FUZZ_ORACLE_CENTRED=1 -- the device agrees)
                    at += 4e-16 * float(np.max(sb[4])) * offs_eff ** 2
                np.testing.assert_allclose(a, b, rtol=tol, atol=at, err_msg=what + " " + nme)
            if not f32:     # (the ELBO's NIW terms inherit the scale matrices' cancellation)
                # (the ELBO is a difference of terms of the size of the minibatch's rows x states: seed 51380002,
                #  offset 1e5 behind two `shift` ops, 4e-6 on 403 against the centred oracle)
                np.testing.assert_allclose(ea, eb, rtol=max(1e-8, 1e-11 * offs_eff ** 2), atol=1e-10 * B * Lm * K,
                                           err_msg=what + " elbo")
            else:
                assert np.all(np.isfinite(ea)), what + " elbo"
            new_problem(keep_obs=True)      # both engines get fresh, identical parameters again
            upload("params")
    o.close()
    return hist


def run_sequence(e, L, seed, nops=30):
    """_run_sequence with numpy.testing.assert_allclose restored on EVERY exit path: the
    FUZZ_ORACLE_CENTRED replay aid gates it, and an assertion raised under the aid must not leave
    the gate installed for the tests that follow in the same process."""
    real = np.testing.assert_allclose
    try:
        return _run_sequence(e, L, seed, nops)
    finally:
        np.testing.assert_allclose = real


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--cases", type=int, default=100)
    ap.add_argument("--chains", type=int, default=10)
    ap.add_argument("--replay", type=int, default=None, help="re-run one call sequence by its seed")
    ap.add_argument("--replay-class", type=int, default=None, help="re-run one class-level case by its seed")
    ap.add_argument("--variant", default="", help="engine switches for A/B replays, e.g. 10:1,8:256")
    ap.add_argument("--sequences", type=int, default=0, help="random call sequences on one handle")
    ap.add_argument("--classes", type=int, default=0, help="cases of the class-level campaign")
    ap.add_argument("--api", type=int, default=0, help="cases of the API-level campaign (callers around the E-step)")
    ap.add_argument("--seconds", type=float, default=1e9, help="stop drawing new cases after this long")
    args = ap.parse_args(argv)
    from pysvihmm_amd.engine import HipEngine
    from pysvihmm_amd import _lib as L
    from oracle import ref_c
    rng = np.random.default_rng(args.seed)
    e = HipEngine(0)
    t0 = time.time()
    nfail = ndone = 0
    for kv in args.variant.split(","):
        if kv:
            e.set_variant(int(kv.split(":")[0]), int(kv.split(":")[1]))
    if args.replay is not None:
        print(run_sequence(e, L, args.replay))
        return 0
    if args.replay_class is not None:
        print(run_class_case(e, args.replay_class))
        return 0
    for i in range(args.cases):
        if time.time() - t0 > args.seconds:
            break
        case = draw_case(rng)
        seed = args.seed * 100000 + i
        try:
            run_case(e, L, ref_c, case, seed)
        except Exception as ex:       # report and go on: a campaign wants all the failures
            nfail += 1
            msg = str(ex).strip().splitlines()
            print("FAIL case=%r seed=%d: %s | %s" % (case, seed, type(ex).__name__, " / ".join(msg[:6])[:600]))
            if not isinstance(ex, AssertionError):
                traceback.print_exc()
                e.close()
                e = HipEngine(0)
        ndone += 1
        if ndone % 20 == 0:
            print("... %d cases, %d failures, %.0f s" % (ndone, nfail, time.time() - t0))
            sys.stdout.flush()
    for i in range(args.chains):
        if time.time() - t0 > args.seconds:
            break
        seed = args.seed * 100000 + 50000 + i
        try:
            run_chain(e, L, ref_c, seed)
        except Exception as ex:
            nfail += 1
            msg = str(ex).strip().splitlines()
            print("FAIL chain seed=%d: %s | %s" % (seed, type(ex).__name__, " / ".join(msg[:6])[:600]))
            if not isinstance(ex, AssertionError):
                traceback.print_exc()
                e.close()
                e = HipEngine(0)
        ndone += 1
    for i in range(args.api):
        if time.time() - t0 > args.seconds:
            break
        seed = args.seed * 100000 + 70000 + i
        try:
            run_api_case(e, L, seed)
        except Exception as ex:
            nfail += 1
            msg = str(ex).strip().splitlines()
            print("FAIL api seed=%d: %s | %s" % (seed, type(ex).__name__, " / ".join(msg[:6])[:600]))
            if not isinstance(ex, AssertionError):
                traceback.print_exc()
            e.close()
            e = HipEngine(0)
        ndone += 1
        if (i + 1) % 10 == 0:
            print("... api %d, %d failures, %.0f s" % (i + 1, nfail, time.time() - t0))
            sys.stdout.flush()
    for i in range(args.sequences):
        if time.time() - t0 > args.seconds:
            break
        seed = args.seed * 100000 + 80000 + i
        try:
            run_sequence(e, L, seed)
        except Exception as ex:
            nfail += 1
            msg = str(ex).strip().splitlines()
            print("FAIL sequence seed=%d: %s | %s" % (seed, type(ex).__name__, " / ".join(msg[:8])[:900]))
            if not isinstance(ex, AssertionError):
                traceback.print_exc()
            e.close()
            e = HipEngine(0)
        ndone += 1
        if (i + 1) % 10 == 0:
            print("... sequences %d, %d failures, %.0f s" % (i + 1, nfail, time.time() - t0))
            sys.stdout.flush()
    for i in range(args.classes):
        if time.time() - t0 > args.seconds:
            break
        seed = args.seed * 100000 + 90000 + i
        try:
            run_class_case(e, seed)
        except Exception as ex:
            nfail += 1
            msg = str(ex).strip().splitlines()
            print("FAIL class seed=%d: %s | %s" % (seed, type(ex).__name__, " / ".join(msg[:6])[:600]))
            if not isinstance(ex, AssertionError):
                traceback.print_exc()
            e.close()
            e = HipEngine(0)
        ndone += 1
        if (i + 1) % 10 == 0:
            print("... classes %d, %d failures, %.0f s" % (i + 1, nfail, time.time() - t0))
            sys.stdout.flush()
    print("fuzz: %d cases, %d failures, %.0f s (seed %d)" % (ndone, nfail, time.time() - t0, args.seed))
    e.close()
    return 1 if nfail else 0


if __name__ == "__main__":
    sys.exit(main())
