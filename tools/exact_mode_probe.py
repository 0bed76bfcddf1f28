"""Cost of the literal log-domain recursion (k_fb_exact) that models with transition expectations
below exp()'s range take: the bench epoch step with such a transition matrix against the same step
on the scaled sweeps."""
import os
import sys
import time

import numpy as np
from scipy.special import digamma

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _workload import bench_problem  # noqa: E402
import bench  # noqa: E402
from pysvihmm_amd.engine import HipEngine  # noqa: E402
from pysvihmm_amd import _lib as L  # noqa: E402

eng = HipEngine(0)
pb = bench_problem(eng)
K = bench.K
B = bench.T // bench.LM
st = np.arange(B, dtype=np.int64) * bench.LM
rng = np.random.default_rng(1)
vt = 1e-3 + rng.random((K, K)) * (rng.random((K, K)) < 0.2) * bench.T / K
vt[np.arange(K), np.arange(K)] += bench.T / K
sparse = digamma(vt + 1e-9) - digamma(vt.sum(1)[:, None] + 1e-9)
eng.set_emission_niw(pb["mu"], pb["sigma"], pb["kappa"], pb["nu"])
for name, lt, n in (("scaled sweeps (dense expectations)", pb["ltran"], 10), ("k_fb_exact (psi(1e-3) entries)", sparse, 3)):
    eng.set_globals(pb["mod_init"], lt)
    for nwin in (64, B):
        s = st[:nwin]
        eng.estep(s, bench.LM, flags=L.TRANS_WRAP, read=False); eng.sync()
        t0 = time.perf_counter()
        for _ in range(n):
            eng.estep(s, bench.LM, flags=L.TRANS_WRAP, read=False)
        eng.sync()
        print("%-38s %5d windows: %8.3f ms per E-step" % (name, nwin, (time.perf_counter() - t0) / n * 1e3))
