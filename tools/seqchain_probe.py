"""Per-step time of the sequential (one window, chain scan off) sweeps against T."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pysvihmm_amd.engine import HipEngine
from tests.helpers import make_problem
K, D = 64, 32
pb = make_problem(K, D, 1 << 20, seed=3, sep=4.0, miss=0.1)
e = HipEngine(0)
e.set_obs(pb['obs'], pb['mask']); e.set_globals(pb['mod_init'], pb['ltran'])
e.set_emission_niw(pb['mu'], pb['sigma'], pb['kappa'], pb['nu'])
e.set_variant("chain", 1)
for B, T in ((1, 257), (1, 2048), (1, 16384), (1, 131072), (1, 1 << 20), (64, 257), (64, 4096), (512, 2048)):
    starts = [i * T for i in range(B)]
    for rep in range(2):
        e.profile(True); e.profile_reset()
        r = e.forward_backward(starts, T, flags=1, want=("local_lb",))
        pr = e.profile_read(); e.profile(False)
    fb = pr['forward_backward'][0]
    print("B=%d T=%d: sweeps %.3f ms = %.3f us/step; %s" % (B, T, fb, fb * 1e3 / T, {k: round(v[0], 3) for k, v in pr.items()}))
e.close()
