"""Host-side cost of one svihmm_svi_iteration call: the resident loop on a minibatch so small that the device is never
the limiter (16 windows of 17 rows) -- wall per call is what the host spends submitting an iteration."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pysvihmm_amd.engine import HipEngine
from pysvihmm_amd.distributions import niw_prior_logpart
from pysvihmm_amd import _lib as L
from tests.helpers import make_problem
K, D, T = 64, 32, 20000
pb = make_problem(K, D, T, seed=1)
rng = np.random.default_rng(K)
prior_tran = 1.0 + rng.random((K, K))
mu0 = np.tile(pb["obs"].mean(0), (K, 1)) + 0.1 * rng.normal(size=(K, D))
sg0 = np.tile(0.75 * np.cov(pb["obs"].T).reshape(D, D), (K, 1, 1))
ka0, nu0 = np.full(K, 0.01), np.full(K, D + 2.0)
for B, Lm in ((16, 17), (64, 257)):
    eng = HipEngine(0)
    nit = 3000
    eng.set_obs(pb["obs"], None)
    eng.svi_begin(prior_tran, pb["var_tran"], (mu0, sg0, ka0, nu0), (pb["mu"], pb["sigma"], pb["kappa"], pb["nu"]), niw_prior_logpart(sg0, nu0), nit, 1.0)
    st = [rng.integers(0, T - Lm, size=B) for _ in range(nit)]
    for it in range(200):
        eng.svi_iteration(it, st[it], B, Lm, L.TRANS_WRAP, 0.01, 3.0, 2.5)
    eng.sync()
    t0 = time.perf_counter()
    for it in range(200, nit):
        eng.svi_iteration(it, st[it], B, Lm, L.TRANS_WRAP, 0.01, 3.0, 2.5)
    t1 = time.perf_counter()
    eng.sync()
    t2 = time.perf_counter()
    print("B=%d Lm=%d: submit %.1f us per call, until the device is done %.1f us per iteration" % (B, Lm, (t1 - t0) / (nit - 200) * 1e6, (t2 - t0) / (nit - 200) * 1e6))
    eng.close()
