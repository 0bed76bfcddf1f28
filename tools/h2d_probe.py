import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from pysvihmm_amd.engine import HipEngine
e = HipEngine(0)
obs = np.random.default_rng(0).normal(size=(1000000, 32))
for i in range(3):
    t0 = time.perf_counter(); e.set_obs(obs, None); e.sync(); dt = time.perf_counter() - t0
    print("set_obs 256 MB: %.2f ms  (%.1f GB/s)" % (dt * 1e3, obs.nbytes / dt / 1e9))
