"""Adaptive window length, buffered meta-observations and the Categorical branch against the
EXECUTED REFERENCE (fixtures written by tests/golden/make_golden.py):

* ``adaptive_K4_D2``        infer(adaptive=True): select_L (hmmsgd_metaobs.py:521-569) re-sizes the
                            windows every perIter iterations
* ``growbuf_K4_D2``         growBuffer=True: select_buffer (:579-661) + intermediate_pars_buffer
                            (:932-1008), statistics over the inner segment only
* ``growbuf_budget_K3_D2``  + bufferBudget=True: buffer_budget (:571-577) re-sizes the minibatch
* ``categorical_K3_V5``     Categorical emissions: local_update on table-lookup lliks, the
                            natural-gradient direction (:907-926) and global_update (:1071-1084)

Each case runs on the oracle engine (CPU: the host logic) and on the HIP engine (``-m gpu``: the
same host logic on the device kernels -- batched candidate windows through
svihmm_forward_backward, buffered statistics through svihmm_estep_minibatch_ex, symbol counts
through k_stats_cat).  Index outputs (chosen L, chosen buffer, window bounds) must be exact.
"""
import os

import numpy as np
import pytest
from scipy.special import digamma

from pysvihmm_amd import hmmsgd_metaobs
from pysvihmm_amd.distributions import Categorical
from tests.test_host_logic import emit_from_fixture

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
ENGINES = [pytest.param("oracle", id="oracle"), pytest.param("hip", id="hip", marks=pytest.mark.gpu)]


def _engine(kind):
    if kind == "oracle":
        from oracle.engine import OracleEngine
        return OracleEngine()
    return None            # default engine of the class = HipEngine (raises without a GPU)


def _tol(kind):
    return dict(rtol=1e-9, atol=1e-9) if kind == "oracle" else dict(rtol=1e-6, atol=1e-8)


class _Trace(object):
    """Records what the fixture recorded, by wrapping instance attributes (the class-level
    dispatch of infer() is unaffected)."""

    def __init__(self, hmm):
        self.L, self.buf, self.its, self.windows = [], [], [], []
        sel_L, sel_b, glob, mbe = hmm.select_L, hmm.select_buffer, hmm.global_update, hmm._minibatch_estep

        def select_L(*a, **k):
            r = sel_L(*a, **k)
            self.L.append((len(self.its), int(r)))
            return r

        def select_buffer(*a, **k):
            r = sel_b(*a, **k)
            self.buf.append((len(self.its), int(r)))
            return r

        def minibatch_estep(minibatch, miniL, buffer=None):
            self.windows += [(len(self.its), mo.i1, mo.i2) for mo in minibatch]
            return mbe(minibatch, miniL, buffer)

        def global_update(A_inter, emit_inter):
            d = dict(A_inter=np.array(A_inter), emit_inter=emit_inter)
            glob(A_inter, emit_inter)
            d["var_tran_new"] = hmm.var_tran.copy()
            self.its.append(d)

        hmm.select_L, hmm.select_buffer = select_L, select_buffer
        hmm._minibatch_estep, hmm.global_update = minibatch_estep, global_update


@pytest.mark.parametrize("kind", ENGINES)
@pytest.mark.parametrize("name", ["adaptive_K4_D2", "growbuf_K4_D2", "growbuf_budget_K3_D2"])
def test_adaptive_and_buffered_infer_match_reference(name, kind):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    K = int(g["K"])
    hmm = hmmsgd_metaobs.VBHMM(
        g["obs"].copy(), np.ones(K), g["prior_tran"], emit_from_fixture(g, K),
        tau=float(g["tau"]), kappa=float(g["kappa"]), metaobs_half=int(g["L"]), mb_sz=int(g["S"]),
        mask=g["mask"], init_tran=g["init_tran"], maxit=int(g["maxit"]), seed=int(g["seed"]),
        growBuffer=bool(g["ctor_growBuffer"]), bufferBudget=bool(g["ctor_bufferBudget"]),
        engine=_engine(kind))
    kw = {k[len("infer_"):]: g[k].item() for k in g.files if k.startswith("infer_")}
    tr = _Trace(hmm)
    hmm.infer(**kw)
    if kind == "hip":
        assert hmm.engine.name == "hip"
    # index outputs: exact
    np.testing.assert_array_equal(np.array(tr.L, dtype=np.int64).reshape(-1, 2), g["chosen_L"])
    np.testing.assert_array_equal(np.array(tr.buf, dtype=np.int64).reshape(-1, 2), g["chosen_buffer"])
    w = np.array(tr.windows, dtype=np.int64)
    np.testing.assert_array_equal(w[:, 0], g["w_iter"])
    np.testing.assert_array_equal(w[:, 1], g["w_i1"])
    np.testing.assert_array_equal(w[:, 2], g["w_i2"])
    np.testing.assert_array_equal(w[:, 2] - w[:, 1] + 1, g["w_len"])
    # accumulated natural-gradient direction and the global step, every iteration
    tol = _tol(kind)
    assert len(tr.its) == int(g["maxit"])
    for it, d in enumerate(tr.its):
        np.testing.assert_allclose(d["A_inter"], g["it_A_inter"][it], **tol)
        E = d["emit_inter"]
        np.testing.assert_allclose(np.array([e[0] for e in E]), g["it_E_xbar"][it], **tol)
        np.testing.assert_allclose(np.array([float(e[1]) for e in E]), g["it_E_neff"][it], **tol)
        np.testing.assert_allclose(np.array([e[2] for e in E]), g["it_E_S"][it], **tol)
        np.testing.assert_allclose(d["var_tran_new"], g["it_var_tran_new"][it], **tol)
    for k in range(K):
        np.testing.assert_allclose(hmm.var_emit[k].mu_mf, g["it_new_mu"][-1][k], **tol)
        np.testing.assert_allclose(hmm.var_emit[k].sigma_mf, g["it_new_sigma"][-1][k],
                                   rtol=tol["rtol"], atol=10 * tol["atol"])
        np.testing.assert_allclose(hmm.var_emit[k].kappa_mf, g["it_new_kappa"][-1][k], rtol=tol["rtol"])
        np.testing.assert_allclose(hmm.var_emit[k].nu_mf, g["it_new_nu"][-1][k], rtol=tol["rtol"])
    np.testing.assert_allclose(hmm.elbo_vec, g["elbo_vec"], rtol=1e-8)
    # get_local_messages (:663-700) called directly after infer
    j = 0
    while "probe%d_ind" % j in g.files:
        q = hmm.get_local_messages(int(g["probe%d_ind" % j]), int(g["probe%d_half" % j]))
        np.testing.assert_allclose(q, g["probe%d_var_x" % j], rtol=tol["rtol"], atol=1e-11)
        j += 1
    assert hmm.buffer_budget(7) == int(np.ceil(400 / 15.))


def _cat_model(g, engine):
    K, V = int(g["K"]), int(g["V"])
    emit = np.array([Categorical(alphav_0=g["alphav_0"], alpha_mf=g["init_alpha_mf"][k], weights=np.ones(V) / V)
                     for k in range(K)])
    return hmmsgd_metaobs.VBHMM(
        g["obs"].copy(), np.ones(K), np.ones((K, K)), emit, tau=float(g["tau"]), kappa=float(g["kappa"]),
        metaobs_half=int(g["L"]), mb_sz=int(g["S"]), mask=g["mask"], init_tran=g["init_tran"], maxit=2,
        seed=int(np.int64(8675309 + 10) % (2 ** 31)), engine=engine)


@pytest.mark.parametrize("kind", ENGINES)
def test_categorical_branch_matches_reference(kind):
    g = np.load(os.path.join(GOLDEN, "categorical_K3_V5.npz"))
    K, V, S, L_ = int(g["K"]), int(g["V"]), int(g["S"]), int(g["L"])
    assert int(g["cat_intermediate_raises"]) == 1     # the reference's own branch cannot execute
    hmm = _cat_model(g, _engine(kind))
    assert hmm._cat_fastpath()
    tr = _Trace(hmm)
    hmm.infer()
    tol = _tol(kind)
    w = np.array(tr.windows, dtype=np.int64)
    np.testing.assert_array_equal(w[:, 1], g["w_i1"])        # same minibatches as the reference drew
    np.testing.assert_array_equal(w[:, 2], g["w_i2"])
    for it, d in enumerate(tr.its):
        np.testing.assert_allclose(d["A_inter"], g["it_A_inter"][it], **tol)
        np.testing.assert_allclose(np.array(d["emit_inter"]), g["it_emit_inter"][it], **tol)
        np.testing.assert_allclose(d["var_tran_new"], g["it_var_tran_new"][it], **tol)
    np.testing.assert_allclose(np.array([e.alpha_mf for e in hmm.var_emit]), g["it_alpha_new"][-1], **tol)
    np.testing.assert_allclose(np.array([e.weights for e in hmm.var_emit]), g["it_weights_new"][-1], **tol)
    # per-window quantities of the first minibatch (parameters = the fixture's initial ones):
    # lliks is the table lookup, everything downstream is reference arithmetic
    eng = hmm.engine
    a0 = g["init_alpha_mf"]
    eng.set_obs(g["obs"], g["mask"])
    vt = g["it_var_tran_old"][0]
    ltran = digamma(vt + 1e-9) - digamma(vt.sum(1)[:, None] + 1e-9)
    vi = g["w_var_init"][0]
    eng.set_globals(digamma(vi + 1e-9) - digamma(vi.sum() + 1e-9), ltran)
    eng.set_emission_cat(digamma(a0) - digamma(a0.sum(1))[:, None])
    Lm = 2 * L_ + 1
    r = eng.forward_backward(g["w_i1"][:S], Lm)
    np.testing.assert_allclose(eng.loglik(g["w_i1"][:S], Lm), g["w_lliks"][:S], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(r["lalpha"], g["w_lalpha"][:S], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(r["lbeta"], g["w_lbeta"][:S], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(r["var_x"], g["w_var_x"][:S], rtol=tol["rtol"], atol=1e-11)
    np.testing.assert_allclose(r["local_lb"], g["w_local_lb"][:S], rtol=1e-10)
    # one window's direction through the engine's packed statistics: A_i = prior - 1 + A_raw,
    # e_i[k] = alphav_0 + counts[k] - 1
    st = eng.estep(g["w_i1"][:1], Lm, flags=2)
    np.testing.assert_allclose(st.A_raw + 0.0, g["w_A_i"][0], **tol)        # prior_tran - 1 = 0 here
    np.testing.assert_allclose(g["alphav_0"][None, :] + st.counts - 1.0, g["w_e_i"][0], **tol)


@pytest.mark.parametrize("kind", ENGINES)
@pytest.mark.parametrize("name", ["adaptive_K4_D2", "growbuf_K4_D2", "growbuf_budget_K3_D2"])
def test_adaptive_and_buffered_infer_device_loop(name, kind):
    """The same three runs with nothing wrapped on the instance: infer() then keeps the
    variational state on the device between iterations (engine.svi_*) and pulls it to the host
    only where select_L / select_buffer need it.  Same ELBO trace and final parameters as the
    executed reference (which implies the same chosen L / buffers and the same minibatches)."""
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    K = int(g["K"])
    hmm = hmmsgd_metaobs.VBHMM(
        g["obs"].copy(), np.ones(K), g["prior_tran"], emit_from_fixture(g, K),
        tau=float(g["tau"]), kappa=float(g["kappa"]), metaobs_half=int(g["L"]), mb_sz=int(g["S"]),
        mask=g["mask"], init_tran=g["init_tran"], maxit=int(g["maxit"]), seed=int(g["seed"]),
        growBuffer=bool(g["ctor_growBuffer"]), bufferBudget=bool(g["ctor_bufferBudget"]),
        engine=_engine(kind))
    assert hmm._svi_device_ok()
    kw = {k[len("infer_"):]: g[k].item() for k in g.files if k.startswith("infer_")}
    hmm.infer(**kw)
    tol = _tol(kind)
    np.testing.assert_allclose(hmm.elbo_vec, g["elbo_vec"], rtol=1e-8)
    np.testing.assert_allclose(hmm.var_tran, g["it_var_tran_new"][-1], **tol)
    for k in range(K):
        np.testing.assert_allclose(hmm.var_emit[k].mu_mf, g["it_new_mu"][-1][k], **tol)
        np.testing.assert_allclose(hmm.var_emit[k].sigma_mf, g["it_new_sigma"][-1][k],
                                   rtol=tol["rtol"], atol=10 * tol["atol"])
        np.testing.assert_allclose(hmm.var_emit[k].kappa_mf, g["it_new_kappa"][-1][k], rtol=tol["rtol"])
        np.testing.assert_allclose(hmm.var_emit[k].nu_mf, g["it_new_nu"][-1][k], rtol=tol["rtol"])
    assert hmm.cur_mo.i1 == int(g["w_i1"][-1]) and hmm.cur_mo.i2 == int(g["w_i2"][-1])
    assert hmm.var_x.shape == (int(g["w_len"][-1]), K)
