import sys, time; sys.path.insert(0, '.')
import numpy as np
from pysvihmm_amd.engine import HipEngine
from tests.helpers import make_problem
K, D, T = 16, 8, 6000000
pb = make_problem(K, D, T, seed=5, sep=3.0, miss=0.05)
e = HipEngine(0); e.set_obs(pb['obs'], pb['mask']); e.set_globals(pb['mod_init'], pb['ltran'])
e.set_emission_niw(pb['mu'], pb['sigma'], pb['kappa'], pb['nu'])
out = {}
for mode in ("scan", "sequential"):
    e.set_variant("chain", 0 if mode == "scan" else 1)
    t0 = time.time(); st = e.estep([0], T, flags=1); dt = time.time() - t0
    out[mode] = st.buf.copy()
    print("T=%d K=%d %s: %.1f ms, lb %.12e" % (T, K, mode, dt * 1e3, st.lb[0]))
d = np.abs(out["scan"] - out["sequential"]) / (np.abs(out["sequential"]) + 1e-12)
print("statistics scan vs sequential: max rel diff %.3g" % d.max())
u = np.random.default_rng(0).random(T)
logA = np.log(np.exp(pb["ltran"]) + 2.2e-16)
e.set_variant("chain", 0); e.set_globals(pb["mod_init"], logA)
t0 = time.time(); z, _ = e.ffbs(logA, u, want_lalpha=False); print("FFBS T=%d: %.1f ms; states used %d" % (T, (time.time() - t0) * 1e3, len(np.unique(z))))
