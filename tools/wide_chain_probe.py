import sys, time; sys.path.insert(0, '.')
import numpy as np
from pysvihmm_amd.engine import HipEngine
from tests.helpers import make_problem
for K, D, T in ((128, 16, 50000), (256, 64, 50000)):
    pb = make_problem(K, D, T, seed=3, sep=4.0, miss=0.0)
    e = HipEngine(0); e.set_obs(pb['obs'], None); e.set_globals(pb['mod_init'], pb['ltran'])
    e.set_emission_niw(pb['mu'], pb['sigma'], pb['kappa'], pb['nu'])
    for rep in range(2):
        e.profile(True); e.profile_reset()
        t0 = time.time(); r = e.forward_backward([0], T, want=("local_lb",)); dt = time.time() - t0
        pr = e.profile_read(); e.profile(False)
    print("K=%d D=%d T=%d single chain: %.1f ms = %.2f us/step; %s" % (K, D, T, dt * 1e3, dt * 1e6 / T, {k: round(v[0], 1) for k, v in pr.items() if v[0] > 0.05}))
    e.close()
