"""The oracle (oracle/ref_numpy.py) against golden vectors produced by running
the reference's own code (tests/golden/make_golden.py).  CPU only."""
import glob
import os

import numpy as np
import pytest

from oracle import ref_numpy as R

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
META = sorted(glob.glob(os.path.join(GOLDEN, "metaobs_*.npz")))
TOL = dict(rtol=1e-12, atol=1e-12)


def test_fixtures_present():
    assert len(META) >= 4
    assert os.path.exists(os.path.join(GOLDEN, "batchcd_K4_D2_T300.npz"))


@pytest.mark.parametrize("path", META, ids=[os.path.basename(p)[:-4] for p in META])
def test_window_trace(path):
    g = np.load(path)
    obs, mask, prior_tran = g["obs"], g["mask"], g["prior_tran"]
    nw = g["w_i1"].shape[0]
    for w in range(nw):
        i1, i2 = int(g["w_i1"][w]), int(g["w_i2"][w])
        var_tran = g["w_var_tran"][w]
        # a2: stationary init -- same eig call on the same matrix
        np.testing.assert_allclose(R.stationary_init(var_tran), g["w_var_init"][w], **TOL)
        mi, mt = R.psi_expectations(g["w_var_init"][w], var_tran)
        np.testing.assert_allclose(mi, g["w_mod_init"][w], **TOL)
        np.testing.assert_allclose(mt, g["w_mod_tran"][w], **TOL)
        # a3 (unpinned arithmetic; same formula here and in the emission class)
        ll = R.lliks_niw(obs[i1:i2 + 1], g["w_mu"][w], g["w_sigma"][w],
                         g["w_kappa"][w], g["w_nu"][w])
        np.testing.assert_allclose(ll, g["w_lliks"][w], rtol=1e-10, atol=1e-9)
        # a4..a8 from the recorded lliks: pure reference arithmetic
        r = R.estep_window(g["w_lliks"][w], obs[i1:i2 + 1], mask[i1:i2 + 1],
                           mi, mt, prior_tran)
        np.testing.assert_allclose(r["lalpha"], g["w_lalpha"][w], **TOL)
        np.testing.assert_allclose(r["lbeta"], g["w_lbeta"][w], **TOL)
        np.testing.assert_allclose(r["var_x"], g["w_var_x"][w], **TOL)
        np.testing.assert_allclose(r["lb"], g["w_local_lb"][w], rtol=1e-13)
        np.testing.assert_allclose(r["A_i"], g["w_A_i"][w], **TOL)
        np.testing.assert_allclose(r["xbar"], g["w_xbar"][w], rtol=1e-12, atol=1e-11)
        np.testing.assert_allclose(r["neff"], g["w_neff"][w], **TOL)
        np.testing.assert_allclose(r["S"], g["w_Sk"][w], rtol=1e-12, atol=1e-10)


@pytest.mark.parametrize("path", META, ids=[os.path.basename(p)[:-4] for p in META])
def test_global_update_trace(path):
    g = np.load(path)
    T, L, S = int(g["T"]), int(g["L"]), int(g["S"])
    K = int(g["K"])
    wpi = int(g["windows_per_iter"])
    prior = [(g["prior_mu0"][k], g["prior_sigma0"][k], g["prior_kappa0"][k],
              g["prior_nu0"][k]) for k in range(K)]
    for it in range(int(g["maxit"])):
        w0 = it * wpi
        # a9: accumulation over the minibatch
        A = g["w_A_i"][w0:w0 + wpi].sum(0)
        np.testing.assert_allclose(A, g["it_A_inter"][it], rtol=1e-12, atol=1e-12)
        var_tran = g["w_var_tran"][w0]
        mf = [(g["w_mu"][w0][k], g["w_sigma"][w0][k], g["w_kappa"][w0][k],
               g["w_nu"][w0][k]) for k in range(K)]
        ei = [(g["it_E_xbar"][it][k], g["it_E_neff"][it][k], g["it_E_S"][it][k])
              for k in range(K)]
        vt, out = R.global_update_metaobs(var_tran, g["it_A_inter"][it], mf, prior,
                                          ei, float(g["it_lrate"][it]), T, L, S)
        np.testing.assert_allclose(vt, g["it_var_tran_new"][it], **TOL)
        for k in range(K):
            np.testing.assert_allclose(out[k][0], g["it_new_mu"][it][k], rtol=1e-11, atol=1e-11)
            np.testing.assert_allclose(out[k][1], g["it_new_sigma"][it][k], rtol=1e-11, atol=1e-9)
            np.testing.assert_allclose(out[k][2], g["it_new_kappa"][it][k], rtol=1e-12)
            np.testing.assert_allclose(out[k][3], g["it_new_nu"][it][k], rtol=1e-12)


@pytest.mark.parametrize("path", META, ids=[os.path.basename(p)[:-4] for p in META])
def test_full_local_update(path):
    """hmmsgd_metaobs.py:1147-1205: masked rows are NaN'd -> lliks 0."""
    g = np.load(path)
    K = int(g["K"])
    obs = g["obs"].copy()
    obs[g["mask"], :] = np.nan
    last = -1
    mi, mt = R.psi_expectations(g["full_var_init"], g["it_var_tran_new"][last])
    ll = R.lliks_niw(obs, g["it_new_mu"][last], g["it_new_sigma"][last],
                     g["it_new_kappa"][last], g["it_new_nu"][last])
    la = R.forward_msgs(ll, mi, mt)
    lb = R.backward_msgs(ll, mt)
    np.testing.assert_allclose(R.posterior(la, lb), g["full_var_x"], rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("name", ["batchcd_K4_D2_T300", "batchsgd_K4_D3_T250"])
def test_batch_trace(name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    for it in range(g["it_lliks"].shape[0]):
        ll = g["it_lliks"][it]
        la = R.forward_msgs(ll, g["it_mod_init"][it], g["it_mod_tran"][it])
        lb = R.backward_msgs(ll, g["it_mod_tran"][it])
        np.testing.assert_allclose(la, g["it_lalpha"][it], **TOL)
        np.testing.assert_allclose(lb, g["it_lbeta"][it], **TOL)
        vx = R.posterior(la, lb)
        np.testing.assert_allclose(vx, g["it_var_x"][it], **TOL)
        tran = g["prior_tran"] + R.transition_stat_batch(vx)
        if not int(g["sgd"]):
            np.testing.assert_allclose(tran, g["it_var_tran_new"][it], **TOL)
            np.testing.assert_allclose(g["prior_init"] + vx[0], g["it_var_init_new"][it], **TOL)


def test_ffbs_forward():
    g = np.load(os.path.join(GOLDEN, "ffbs_K5_D3_T120.npz"))
    ll = R.lliks_niw(g["obs"], g["mu"], g["sigma"], g["kappa"], g["nu"])
    la = R.ffbs_forward(ll, g["var_init"], g["var_tran"])
    np.testing.assert_allclose(la, g["lalpha"], rtol=1e-10, atol=1e-9)
    z = g["z"]
    assert z.shape == (int(g["T"]),) and z.min() >= 0 and z.max() < int(g["K"])


# ---- the C port (oracle/ref_c.c) pinned DIRECTLY to the reference-executed fixtures --------------
# Every full-size GPU parity test leans on ref_c; until round 3 it was pinned only through the
# OracleEngine class traces.  CPU only (liboracle.so is built by oracle/Makefile).
@pytest.mark.parametrize("path", META, ids=[os.path.basename(p)[:-4] for p in META])
def test_c_port_window_trace(path):
    from oracle import ref_c
    g = np.load(path)
    obs, mask, prior_tran = g["obs"], g["mask"], g["prior_tran"]
    K, D = int(g["K"]), obs.shape[1]
    for w in range(g["w_i1"].shape[0]):
        i1, i2 = int(g["w_i1"][w]), int(g["w_i2"][w])
        Lm = i2 - i1 + 1
        mi, mt = g["w_mod_init"][w], g["w_mod_tran"][w]
        par = (g["w_mu"][w], g["w_sigma"][w], g["w_kappa"][w], g["w_nu"][w])
        # a3 (same formula as the emission class; the fixture's lliks come from that class
        #     executed inside the reference's loop)
        ll = ref_c.lliks_niw(obs[i1:i2 + 1], *par)
        np.testing.assert_allclose(ll, g["w_lliks"][w], rtol=1e-10, atol=1e-9)
        # a4..a6 from the RECORDED lliks: the reference's own arithmetic, bit for bit on the folds
        la = ref_c.forward(g["w_lliks"][w], mi, mt)
        lb = ref_c.backward(g["w_lliks"][w], mt)
        np.testing.assert_allclose(la, g["w_lalpha"][w], rtol=1e-13, atol=1e-12)
        np.testing.assert_allclose(lb, g["w_lbeta"][w], rtol=1e-13, atol=1e-12)
        q, lbv = ref_c.posterior(la, lb)
        np.testing.assert_allclose(q, g["w_var_x"][w], rtol=1e-12, atol=1e-15)
        np.testing.assert_allclose(lbv, g["w_local_lb"][w], rtol=1e-13)
        # a7..a9: the fused minibatch entry point on this one window (quirks Q1, Q4, Q9)
        buf = ref_c.estep_minibatch(obs, mask, np.array([i1]), Lm, mi, mt, *par, flags=2)
        o = 0
        A = buf[o:o + K * K].reshape(K, K); o += K * K
        xbar = buf[o:o + K * D].reshape(K, D); o += K * D
        neff = buf[o:o + K]; o += K
        S = buf[o:o + K * D * D].reshape(K, D, D); o += K * D * D
        np.testing.assert_allclose(A + prior_tran - 1.0, g["w_A_i"][w], rtol=1e-10, atol=1e-11)
        np.testing.assert_allclose(xbar, g["w_xbar"][w], rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(neff, g["w_neff"][w], rtol=1e-9, atol=1e-10)
        np.testing.assert_allclose(S, g["w_Sk"][w], rtol=1e-9, atol=1e-8)
        np.testing.assert_allclose(buf[o], g["w_local_lb"][w], rtol=1e-10)


@pytest.mark.parametrize("name", ["batchcd_K4_D2_T300", "batchsgd_K4_D3_T250"])
def test_c_port_batch_trace(name):
    from oracle import ref_c
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    for it in range(g["it_lliks"].shape[0]):
        ll = g["it_lliks"][it]
        la = ref_c.forward(ll, g["it_mod_init"][it], g["it_mod_tran"][it])
        lb = ref_c.backward(ll, g["it_mod_tran"][it])
        np.testing.assert_allclose(la, g["it_lalpha"][it], rtol=1e-13, atol=1e-12)
        np.testing.assert_allclose(lb, g["it_lbeta"][it], rtol=1e-13, atol=1e-12)
        np.testing.assert_allclose(ref_c.posterior(la, lb)[0], g["it_var_x"][it], rtol=1e-12, atol=1e-15)
