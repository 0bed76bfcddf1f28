"""The C ABI is coordinate-invariant.

The library keeps the resident observations centred (``obs_dev = obs - c``, c chosen at upload,
moved by ``svihmm_shift_obs``) because the emission GEMM expands the NIW quadratic form around the
resident copy's origin; nothing a caller passes in or reads back depends on c.  The reference
evaluates everything in the coordinates of ``self.obs`` (hmmbase.py:219-229,
hmmsgd_metaobs.py:857-928), and so does the oracle these tests compare with.

* raw engine calls on data far from the origin, any mix of automatic and explicit shifts;
* a class running ``infer(engine=e)`` followed by raw calls on the same handle with the caller's
  parameters (the state that broke round 2's bench cross-check);
* block uploads, the generator's read-back, the device-resident SVI loop's state, the ELBO terms,
  a Categorical table following NIW factors on the same symbol column.
"""
import numpy as np
import pytest

from tests.helpers import make_problem, unpack

pytestmark = pytest.mark.gpu


def _stats_close(got, ref, K, D, rows, xs, rtol=1e-6):
    g, r = unpack(got, K, D), unpack(ref, K, D)
    np.testing.assert_allclose(g[0], r[0], rtol=rtol, atol=1e-9 * rows)
    np.testing.assert_allclose(g[1], r[1], rtol=rtol, atol=1e-9 * rows * xs)
    np.testing.assert_allclose(g[2], r[2], rtol=rtol, atol=1e-9 * rows)
    np.testing.assert_allclose(g[3], r[3], rtol=rtol, atol=1e-9 * rows * xs * xs)
    np.testing.assert_allclose(g[4], r[4], rtol=1e-9, atol=1e-6)


@pytest.mark.parametrize("off", [0.0, 37.5, 1e5, -1e7])
@pytest.mark.parametrize("B,Lm", [(12, 33), (300, 17)])
def test_raw_calls_in_caller_coordinates(off, B, Lm):
    """Statistics, posteriors and log-likelihoods of data offset by `off` equal the oracle's on
    the same (offset) inputs; explicit shifts in between change nothing."""
    from pysvihmm_amd.engine import HipEngine
    from pysvihmm_amd import _lib as L
    from oracle import ref_c
    K, D, T = 6, 3, 1200
    pb = make_problem(K, D, T, seed=5, miss=0.05, sep=3.0)
    obs, mu = pb["obs"] + off, pb["mu"] + off
    par = (pb["mod_init"], pb["ltran"], mu, pb["sigma"], pb["kappa"], pb["nu"])
    starts = np.random.default_rng(1).integers(0, T - Lm, size=B)
    ref = ref_c.estep_minibatch(obs, pb["mask"], starts, Lm, *par, flags=L.TRANS_WRAP)
    xs = max(1.0, abs(off))
    e = HipEngine(0)
    try:
        e.set_obs(obs, pb["mask"])
        c = e.get_shift()
        assert np.all(np.abs(c - off) < 10.0), c       # a point inside the data
        e.set_globals(pb["mod_init"], pb["ltran"])
        e.set_emission_niw(mu, pb["sigma"], pb["kappa"], pb["nu"])
        st = e.estep(starts, Lm, flags=L.TRANS_WRAP)
        _stats_close(st.buf, ref, K, D, B * Lm, xs)
        np.testing.assert_allclose(e.read_packed().buf, st.buf, rtol=0, atol=0)
        # an explicit shift moves the centre only: same parameters, same answers
        e.shift_obs(np.array([0.5, -2.0, 1.25]))
        np.testing.assert_allclose(e.get_shift(), c + np.array([0.5, -2.0, 1.25]), rtol=1e-15)
        st2 = e.estep(starts, Lm, flags=L.TRANS_WRAP)         # (theta was rebuilt by the shift)
        _stats_close(st2.buf, ref, K, D, B * Lm, xs)
        e.set_emission_niw(mu, pb["sigma"], pb["kappa"], pb["nu"])
        st3 = e.estep(starts, Lm, flags=L.TRANS_WRAP)
        _stats_close(st3.buf, ref, K, D, B * Lm, xs)
        # log-likelihoods and posteriors of a window
        x = obs[starts[0]:starts[0] + Lm]
        ll = ref_c.lliks_niw(x, mu, pb["sigma"], pb["kappa"], pb["nu"])
        np.testing.assert_allclose(e.loglik(starts[:1], Lm)[0], ll, rtol=1e-9, atol=1e-7)
        r = e.forward_backward(starts[:1], Lm, want=("var_x",))
        q, _ = ref_c.posterior(ref_c.forward(ll, pb["mod_init"], pb["ltran"]), ref_c.backward(ll, pb["ltran"]))
        np.testing.assert_allclose(r["var_x"][0], q, rtol=1e-6, atol=1e-10)
    finally:
        e.close()


def test_guard_without_automatic_centring_and_its_cure():
    """variant 9 = 1 switches the automatic centre off: factors 1e7 spreads from the origin are
    refused (the expanded form would lose ~0.2 nats); svihmm_shift_obs cures it."""
    from pysvihmm_amd.engine import HipEngine
    from pysvihmm_amd import _lib as L
    from oracle import ref_c
    K, D, T = 5, 3, 500
    pb = make_problem(K, D, T, seed=8, sep=3.0)
    off = 1e7
    obs, mu = pb["obs"] + off, pb["mu"] + off
    starts = np.array([0, 100])
    ref = ref_c.estep_minibatch(obs, None, starts, 33, pb["mod_init"], pb["ltran"], mu, pb["sigma"], pb["kappa"],
                                pb["nu"], flags=L.TRANS_WRAP)
    e = HipEngine(0)
    try:
        e.set_variant(9, 1)
        e.set_obs(obs, None)
        assert np.all(e.get_shift() == 0.0)
        e.set_globals(pb["mod_init"], pb["ltran"])
        with pytest.raises(RuntimeError, match="centre"):
            e.set_emission_niw(mu, pb["sigma"], pb["kappa"], pb["nu"])
            e.estep(starts, 33)
        e.shift_obs(np.full(D, off))
        e.set_emission_niw(mu, pb["sigma"], pb["kappa"], pb["nu"])
        _stats_close(e.estep(starts, 33, flags=L.TRANS_WRAP).buf, ref, K, D, 66, off)
        e.set_variant(9, 0)
        e.set_obs(obs, None)                     # automatic again
        e.set_emission_niw(mu, pb["sigma"], pb["kappa"], pb["nu"])
        _stats_close(e.estep(starts, 33, flags=L.TRANS_WRAP).buf, ref, K, D, 66, off)
    finally:
        e.close()


@pytest.mark.parametrize("cls", ["metaobs", "batchcd"])
def test_class_then_raw_calls_on_one_handle(cls):
    """A class runs infer(engine=e); afterwards raw E-steps on the same handle with the CALLER's
    parameters equal the oracle on the caller's data (round 2: the class had centred the shared
    resident copy and every later raw call silently ran in other coordinates)."""
    from pysvihmm_amd import hmmsgd_metaobs, hmmbatchcd
    from pysvihmm_amd.distributions import Gaussian
    from pysvihmm_amd.engine import HipEngine
    from pysvihmm_amd import _lib as L
    from oracle import ref_c
    K, D, T = 4, 3, 1500
    pb = make_problem(K, D, T, seed=21, sep=3.0)
    obs = pb["obs"] + 250.0
    mu = pb["mu"] + 250.0
    par = (pb["mod_init"], pb["ltran"], mu, pb["sigma"], pb["kappa"], pb["nu"])
    starts = np.arange(40, dtype=np.int64) * 33
    e = HipEngine(0)
    try:
        # raw use first: the handle holds the caller's sequence
        e.set_obs(obs, None)
        e.set_globals(pb["mod_init"], pb["ltran"])
        e.set_emission_niw(mu, pb["sigma"], pb["kappa"], pb["nu"])
        ref = ref_c.estep_minibatch(obs, None, starts, 33, *par, flags=L.TRANS_WRAP)
        _stats_close(e.estep(starts, 33, flags=L.TRANS_WRAP).buf, ref, K, D, 40 * 33, 250.0)
        # a class on the same handle (its own data: a different offset)
        np.random.seed(6)
        cobs = pb["obs"] - 1e4
        prior = np.array([Gaussian(mu_0=cobs.mean(0), sigma_0=0.75 * np.cov(cobs.T), kappa_0=0.01, nu_0=D + 2)
                          for _ in range(K)])
        if cls == "metaobs":
            m = hmmsgd_metaobs.VBHMM(cobs, np.ones(K), np.ones((K, K)), prior, tau=1.0, kappa=0.7,
                                     metaobs_half=10, mb_sz=8, maxit=5, seed=3, engine=e)
        else:
            m = hmmbatchcd.VBHMM(cobs, np.ones(K), np.ones((K, K)), prior, maxit=3, engine=e)
        m.infer()
        assert np.all(np.isfinite(m.elbo_vec[:3]))
        # the resident copy is now the class's: raw calls see ITS data, in ITS coordinates
        cmu = pb["mu"] - 1e4
        e.set_globals(pb["mod_init"], pb["ltran"])
        e.set_emission_niw(cmu, pb["sigma"], pb["kappa"], pb["nu"])
        refc = ref_c.estep_minibatch(cobs, None, starts, 33, pb["mod_init"], pb["ltran"], cmu, pb["sigma"],
                                     pb["kappa"], pb["nu"], flags=L.TRANS_WRAP)
        _stats_close(e.estep(starts, 33, flags=L.TRANS_WRAP).buf, refc, K, D, 40 * 33, 1e4)
        # and after the caller uploads again, the caller's
        e.set_obs(obs, None)
        e.set_emission_niw(mu, pb["sigma"], pb["kappa"], pb["nu"])
        _stats_close(e.estep(starts, 33, flags=L.TRANS_WRAP).buf, ref, K, D, 40 * 33, 250.0)
        # the class notices that the resident copy is no longer its own
        m.maxit = 2
        m.infer()
        assert np.all(np.isfinite(m.elbo_vec))
    finally:
        e.close()


def test_block_upload_and_generator_readback():
    from pysvihmm_amd.engine import HipEngine
    from pysvihmm_amd import _lib as L
    from oracle import ref_c
    K, D, T = 5, 4, 2000
    pb = make_problem(K, D, T, seed=3, miss=0.1, sep=2.0)
    off = np.array([1e4, -3e3, 0.0, 77.0])
    obs, mu = pb["obs"] + off, pb["mu"] + off
    starts = np.arange(30, dtype=np.int64) * 65
    ref = ref_c.estep_minibatch(obs, pb["mask"], starts, 65, pb["mod_init"], pb["ltran"], mu, pb["sigma"],
                                pb["kappa"], pb["nu"], flags=L.TRANS_WRAP)
    e = HipEngine(0)
    try:
        n = e.set_obs_blocks((obs[i:i + 300] for i in range(0, T, 300)), T, D, mask=pb["mask"])
        assert n == T
        c = e.get_shift()
        assert np.all(np.abs(c - off) < 10.0)          # fixed by the first block
        e.set_globals(pb["mod_init"], pb["ltran"])
        e.set_emission_niw(mu, pb["sigma"], pb["kappa"], pb["nu"])
        _stats_close(e.estep(starts, 65, flags=L.TRANS_WRAP).buf, ref, K, D, 30 * 65, 1e4)
        # generator: the read-back is in the caller's coordinates (means included)
        tran = 0.9 * np.eye(K) + 0.1 / (K - 1) * (1 - np.eye(K))
        means = np.random.default_rng(0).normal(0, 3, size=(K, D)) + off
        chols = np.broadcast_to(np.eye(D), (K, D, D)).copy()
        e.generate(tran, means, chols, 5000, seed=11)
        assert np.all(np.abs(e.get_shift() - off) < 10.0)
        x, z = e.read_generated()
        assert np.max(np.abs(x - means[z])) < 7.0      # unit-variance noise around the state means
        e.set_emission_niw(means, chols * 3.0, np.ones(K), np.full(K, D + 3.0))
        st = e.estep(np.arange(50, dtype=np.int64) * 100, 100, flags=L.TRANS_WRAP)
        refg = ref_c.estep_minibatch(x, None, np.arange(50, dtype=np.int64) * 100, 100, pb["mod_init"], pb["ltran"],
                                     means, chols * 3.0, np.ones(K), np.full(K, D + 3.0), flags=L.TRANS_WRAP)
        _stats_close(st.buf, refg, K, D, 5000, 1e4)
    finally:
        e.close()


def test_svi_state_and_elbo_terms_in_caller_coordinates():
    """svi_begin / svi_read_state and the NIW ELBO terms: offset data + offset means give the
    un-offset run's state plus the offset (1e-9 relative to the spread)."""
    from pysvihmm_amd import hmmsgd_metaobs
    from pysvihmm_amd.distributions import Gaussian
    from pysvihmm_amd.engine import HipEngine
    K, D, T = 4, 2, 1200
    pb = make_problem(K, D, T, seed=9, sep=3.0)
    res = []
    for off in (0.0, 5e4):
        e = HipEngine(0)
        try:
            np.random.seed(2)
            obs = pb["obs"] + off
            prior = np.array([Gaussian(mu_0=pb["obs"].mean(0) + off, sigma_0=0.75 * np.cov(pb["obs"].T),
                                       kappa_0=0.01, nu_0=D + 2) for _ in range(K)])
            m = hmmsgd_metaobs.VBHMM(obs, np.ones(K), np.ones((K, K)), prior, tau=1.0, kappa=0.7, metaobs_half=8,
                                     mb_sz=6, maxit=6, seed=4, engine=e)
            m.infer()
            mu = np.array([g.mu_mf for g in m.var_emit])
            sg = np.array([g.sigma_mf for g in m.var_emit])
            ka = np.array([g.kappa_mf for g in m.var_emit]); nu = np.array([g.nu_mf for g in m.var_emit])
            e.set_emission_prior(np.array([g.mu_0 for g in m.var_emit]), np.array([g.sigma_0 for g in m.var_emit]))
            terms = e.niw_vlb_terms(mu, sg, ka, nu)
            # (move the centre between the prior upload and a second evaluation: still the same)
            e.shift_obs(np.array([3.0, -1.0]))
            terms2 = e.niw_vlb_terms(mu, sg, ka, nu)
            for a, b in zip(terms, terms2):
                np.testing.assert_allclose(a, b, rtol=1e-9, atol=1e-9)
            res.append((mu - off, sg, m.var_tran.copy(), m.elbo_vec.copy(), terms))
        finally:
            e.close()
    (mu0, sg0, vt0, el0, tm0), (mu1, sg1, vt1, el1, tm1) = res
    np.testing.assert_allclose(mu1, mu0, rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(sg1, sg0, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(vt1, vt0, rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(el1, el0, rtol=1e-6)
    for a, b in zip(tm0, tm1):
        np.testing.assert_allclose(b, a, rtol=1e-6, atol=1e-6)


def test_categorical_table_after_niw_factors_on_a_symbol_column():
    """A D = 1 symbol column that a NIW upload has seen centred goes back to exact integers when
    a Categorical table follows (the upload cannot know the emission family)."""
    from pysvihmm_amd.engine import HipEngine
    from pysvihmm_amd import _lib as L
    from oracle.engine import OracleEngine
    K, V, T = 3, 7, 900
    rng = np.random.default_rng(12)
    sym = rng.integers(0, V, size=T).astype(np.float64)[:, None]
    mod_init = np.log(np.full(K, 1.0 / K))
    ltran = np.log(0.8 * np.eye(K) + 0.2 / K)
    logp = np.log(rng.dirichlet(np.ones(V), size=K))
    starts = np.arange(20, dtype=np.int64) * 40
    e, o = HipEngine(0), OracleEngine()
    try:
        for eng in (e, o):
            eng.set_obs(sym, None)
            eng.set_globals(mod_init, ltran)
        assert abs(e.get_shift()[0] - sym.mean()) < 1.0
        e.set_emission_niw(np.arange(K, dtype=float)[:, None], np.ones((K, 1, 1)), np.ones(K), np.full(K, 4.0))
        e.estep(starts, 40, flags=L.TRANS_WRAP)
        for eng in (e, o):
            eng.set_emission_cat(logp)
        assert e.get_shift()[0] == 0.0
        a = e.estep(starts, 40, flags=L.TRANS_WRAP)
        b = o.estep(starts, 40, flags=L.TRANS_WRAP)
        np.testing.assert_allclose(a.buf, b.buf, rtol=1e-9, atol=1e-9)
    finally:
        e.close()


def test_categorical_table_set_before_the_observations():
    """ABI order  set_emission_cat -> set_obs(new symbol column) -> estep  (valid since ABI v1; the
    classes never use it: _push_emission follows _upload_obs).  Round-3 advisor finding: every upload
    centred the resident copy, the lookup / count kernels then truncated x - c with (int)x and the
    lliks / counts were silently wrong.  Under an active table the column stays exactly as uploaded;
    an explicit svihmm_shift_obs afterwards is undone (with rounding) in front of the next launch.
    Block upload and the loglik entry point included."""
    from pysvihmm_amd.engine import HipEngine
    from pysvihmm_amd import _lib as L
    from oracle.engine import OracleEngine
    K, V, T = 3, 7, 900
    rng = np.random.default_rng(13)
    mod_init = np.log(np.full(K, 1.0 / K))
    ltran = np.log(0.8 * np.eye(K) + 0.2 / K)
    logp = np.log(rng.dirichlet(np.ones(V), size=K))
    starts = np.arange(20, dtype=np.int64) * 40
    e, o = HipEngine(0), OracleEngine()
    try:
        for eng in (e, o):
            eng.set_obs(rng.integers(0, V, size=(T, 1)).astype(np.float64), None)   # some earlier column
            eng.set_globals(mod_init, ltran)
            eng.set_emission_cat(logp)
        sym = rng.integers(2, V, size=T).astype(np.float64)[:, None]                # mean well away from 0
        for eng in (e, o):
            eng.set_obs(sym, None)                                                  # table first, data second
        assert e.get_shift()[0] == 0.0
        a = e.estep(starts, 40, flags=L.TRANS_WRAP)
        b = o.estep(starts, 40, flags=L.TRANS_WRAP)
        np.testing.assert_allclose(a.buf, b.buf, rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(e.loglik(starts[:3], 40), o.loglik(starts[:3], 40), rtol=0, atol=1e-12)
        # an explicit shift under the table: put back before the next lookup
        e.shift_obs(np.array([2.5]))
        a2 = e.estep(starts, 40, flags=L.TRANS_WRAP)
        assert e.get_shift()[0] == 0.0
        np.testing.assert_array_equal(a2.buf, a.buf)
        # block upload under the table
        sym2 = rng.integers(1, V, size=T).astype(np.float64)[:, None]
        e.set_obs_blocks([sym2[:500], sym2[500:]], T, 1)
        o.set_obs(sym2, None)
        assert e.get_shift()[0] == 0.0
        np.testing.assert_allclose(e.estep(starts, 40, flags=L.TRANS_WRAP).buf,
                                   o.estep(starts, 40, flags=L.TRANS_WRAP).buf, rtol=1e-9, atol=1e-9)
    finally:
        e.close()


def test_gaussian_family_after_a_categorical_table_gets_the_skipped_centring():
    """Found by the seed-411 campaign of round 4: uploads under an active Categorical table do not centre
    (round-3 advisor fix), so NIW factors that FOLLOW on data far from the origin met an uncentred copy
    and tripped the range guard.  The library now takes the skipped centring when the Gaussian family
    arrives (column means of a row sample, on the device)."""
    from pysvihmm_amd.engine import HipEngine
    from pysvihmm_amd import _lib as L
    from oracle import ref_c
    from tests.helpers import make_problem
    K, D, T, Lm = 4, 3, 3000, 33
    pb = make_problem(K, D, T, seed=77, miss=0.04)
    off = np.array([4.0e4, -2.5e4, 1.0e4])
    starts = np.arange(40, dtype=np.int64) * 70
    e = HipEngine(0)
    try:
        e.set_obs(np.round(np.abs(pb["obs"][:, :1])) % 5, None)          # a symbol column first
        e.set_globals(pb["mod_init"], pb["ltran"])
        e.set_emission_cat(np.log(np.full((K, 5), 0.2)))
        e.set_obs(pb["obs"] + off, pb["mask"])                            # uploaded under the table: not centred
        assert np.all(e.get_shift() == 0.0)
        e.set_emission_niw(pb["mu"] + off, pb["sigma"], pb["kappa"], pb["nu"])
        assert np.abs(e.get_shift() - off).max() < 20.0                   # centred now
        st = e.estep(starts, Lm, flags=L.TRANS_WRAP)
        ref = ref_c.estep_minibatch(pb["obs"], pb["mask"], starts, Lm, pb["mod_init"], pb["ltran"], pb["mu"],
                                    pb["sigma"], pb["kappa"], pb["nu"], flags=2)
        g = e.estep(starts, Lm, flags=L.TRANS_WRAP)
        np.testing.assert_allclose(g.A_raw, ref[:K * K].reshape(K, K), rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(g.neff, ref[K * K + K * D:K * K + K * D + K], rtol=1e-6, atol=1e-7)
        # first moments in the caller's coordinates: oracle ran un-offset, so xbar = xbar_ref + neff * off
        xb = ref[K * K:K * K + K * D].reshape(K, D) + g.neff[:, None] * off[None, :]
        np.testing.assert_allclose(g.xbar, xb, rtol=1e-6, atol=1e-4)
        np.testing.assert_array_equal(g.buf, st.buf)
    finally:
        e.close()
