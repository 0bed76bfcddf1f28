import sys, time; sys.path.insert(0, '.')
import numpy as np
from pysvihmm_amd.engine import HipEngine
from tests.helpers import make_problem
K, D, T = 64, 32, 1000000
pb = make_problem(K, D, T, seed=3, sep=4.0, miss=0.1)
e = HipEngine(0); e.set_obs(pb['obs'], pb['mask']); e.set_globals(pb['mod_init'], pb['ltran'])
e.set_emission_niw(pb['mu'], pb['sigma'], pb['kappa'], pb['nu'])
for mode in ("scan", "sequential"):
    e.set_variant("chain", 0 if mode == "scan" else 1)
    for rep in range(2):
        e.profile(True); e.profile_reset()
        t0 = time.time(); r = e.forward_backward([0], T); dt = time.time() - t0
        pr = e.profile_read(); e.profile(False)
    print("local_update-style fetch (lalpha, lbeta, var_x: 3 x 488 MB) %-10s %.1f ms wall; kernels %s" % (
        mode, dt * 1e3, {k: round(v[0], 1) for k, v in pr.items() if v[0] > 0.05}))
    if mode == "scan": keep = r
    else: print("   max |lalpha diff| %.3g  max |lbeta diff| %.3g" % (np.abs(keep["lalpha"] - r["lalpha"]).max(), np.abs(keep["lbeta"] - r["lbeta"]).max()))
