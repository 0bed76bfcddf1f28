import sys, time; sys.path.insert(0, '.')
import numpy as np
from pysvihmm_amd.engine import HipEngine
from tests.helpers import make_problem
"""Single long chain on wide models (64 < K <= 256): blocked scan vs the sequential kernels."""
T0 = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
for K, D, T in ((128, 16, T0), (256, 64, T0)):
    pb = make_problem(K, D, T, seed=3, sep=4.0, miss=0.0)
    e = HipEngine(0); e.set_obs(pb['obs'], None); e.set_globals(pb['mod_init'], pb['ltran'])
    e.set_emission_niw(pb['mu'], pb['sigma'], pb['kappa'], pb['nu'])
    for mode in ("scan", "sequential"):
        if mode == "sequential" and T > 200000:
            continue
        e.set_variant("chain", 0 if mode == "scan" else 1)
        for rep in range(2):
            e.profile(True); e.profile_reset()
            t0 = time.time(); r = e.forward_backward([0], T, want=("local_lb",)); dt = time.time() - t0
            pr = e.profile_read(); e.profile(False)
        print("K=%d D=%d T=%d single chain (%s): %.1f ms = %.2f us/step; lb %.10e; %s" % (K, D, T, mode, dt * 1e3, dt * 1e6 / T, r["local_lb"][0], {k: round(v[0], 1) for k, v in pr.items() if v[0] > 0.05}))
    e.close()
