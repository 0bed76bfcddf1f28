"""FFBS (hmm_fast.pyx:43-124) on a long chain: blocked scan path vs the sequential kernels."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pysvihmm_amd.engine import HipEngine
from tests.helpers import make_problem
for K, D, T, sep in ((16, 8, 100000, 4.0), (64, 32, 1000000, 4.0), (64, 32, 1000000, 0.5)):
    pb = make_problem(K, D, T, seed=3, sep=sep, miss=0.0)
    logA = np.log(pb["var_tran"] + np.finfo(np.float64).eps)
    u = np.random.default_rng(1).random(T)
    e = HipEngine(0)
    e.set_obs(pb["obs"], None); e.set_globals(pb["mod_init"], logA)
    e.set_emission_niw(pb["mu"], pb["sigma"], pb["kappa"], pb["nu"])
    for mode in ("blocked", "sequential"):
        e.set_variant("chain", 0 if mode == "blocked" else 1)
        for rep in range(2):
            e.profile(True); e.profile_reset()
            t0 = time.time(); z, _ = e.ffbs(logA, u, want_lalpha=False); dt = time.time() - t0
            pr = e.profile_read(); e.profile(False)
        print("K=%d D=%d T=%d sep=%.1f FFBS %-10s %8.1f ms wall (z only)  kernels %s" % (
            K, D, T, sep, mode, dt * 1e3, {k: round(v[0], 2) for k, v in pr.items() if v[0] > 0.005}))
        if mode == "blocked":
            zb = z
            t0 = time.time(); z, la = e.ffbs(logA, u); dt = time.time() - t0
            print("      with lalpha[T,K] read back (%d MB): %.1f ms" % (la.nbytes >> 20, dt * 1e3))
        else:
            print("      paths agree on %.4f of the rows" % (zb == z).mean())
    e.close()
