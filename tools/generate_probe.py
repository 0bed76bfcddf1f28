"""Device-side sequence generation (svihmm_generate): time and rate."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pysvihmm_amd.engine import HipEngine
K, D = 64, 32
rng = np.random.default_rng(0)
tran = 0.9 * np.eye(K) + 0.1 / (K - 1) * (1 - np.eye(K))
means = rng.normal(0, 5, size=(K, D))
chols = np.tile(np.eye(D), (K, 1, 1))
e = HipEngine(0)
for T in (1000000, 10000000, 100000000):
    for rep in range(2):
        t0 = time.time(); e.generate(tran, means, chols, T, seed=1); dt = time.time() - t0
    t0 = time.time(); _, sts = e.read_generated(want_obs=False); dr = time.time() - t0
    print("T=%d K=%d D=%d: generated in HBM in %.1f ms (%.2f GB of observations, %.0f GB/s); states back in %.1f ms; host vectorised generator: " % (
        T, K, D, dt * 1e3, T * D * 8 / 1e9, T * D * 8 / dt / 1e9, dr * 1e3), end="")
    if T <= 10000000:
        from pysvihmm_amd import gen_synthetic
        t0 = time.time(); gen_synthetic.generate_data_fast(tran, means, chols, T, rng=np.random.default_rng(1)); print("%.0f ms" % ((time.time() - t0) * 1e3))
    else:
        print("(skipped)")
