"""Workload for the whole-chain profile: full-chain E-step + FFBS, K=64 D=32 T=1e6 (3 repetitions)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pysvihmm_amd.engine import HipEngine
from tests.helpers import make_problem
K, D, T = 64, 32, 1000000
pb = make_problem(K, D, T, seed=3, sep=4.0, miss=0.0)
logA = np.log(pb["var_tran"] + np.finfo(np.float64).eps)
u = np.random.default_rng(1).random(T)
e = HipEngine(0); e.set_obs(pb["obs"], None)
e.set_emission_niw(pb["mu"], pb["sigma"], pb["kappa"], pb["nu"])
for rep in range(3):
    e.set_globals(pb["mod_init"], pb["ltran"]); e.estep([0], T, flags=0)
    e.set_globals(pb["mod_init"], logA); e.ffbs(logA, u, want_lalpha=False)
e.close()
