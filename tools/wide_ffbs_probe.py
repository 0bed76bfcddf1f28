"""Whole-chain callers on a wide model (64 < K <= 256): FFBS and the log read-back, blocked
(scan + row-parallel conversion + composed draw maps) against the sequential device path."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pysvihmm_amd.engine import HipEngine
from tests.helpers import make_problem
K = int(sys.argv[1]) if len(sys.argv) > 1 else 256
T = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
D = 8
pb = make_problem(K, D, T, seed=3, sep=3.0)
logA = np.log(pb["var_tran"] + np.finfo(np.float64).eps)
u = np.random.default_rng(1).random(T)
e = HipEngine(0)
e.set_obs(pb["obs"], None)
e.set_emission_niw(pb["mu"], pb["sigma"], pb["kappa"], pb["nu"])
for name, v in (("blocked", 0), ("sequential", 1)):
    e.set_variant("chain", v)
    e.set_globals(pb["mod_init"], logA)
    e.ffbs(logA, u, want_lalpha=False)
    t0 = time.time(); z, _ = e.ffbs(logA, u, want_lalpha=False); t1 = time.time()
    e.set_globals(pb["mod_init"], pb["ltran"])
    t2 = time.time(); r = e.forward_backward([0], T, want=("lalpha", "lbeta", "local_lb")); t3 = time.time()
    print("K=%d T=%d %-10s ffbs %.1f ms (z only)   lalpha+lbeta read-back %.1f ms (incl. %.0f MB over PCIe)"
          % (K, T, name, (t1 - t0) * 1e3, (t3 - t2) * 1e3, 2 * T * K * 8 / 1e6))
e.close()
