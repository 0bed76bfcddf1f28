"""Scale test: sequence generated in HBM, one E-step over all tiled windows (K=64, D=32, Lm=257)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from scipy.special import digamma
from pysvihmm_amd.engine import HipEngine
from pysvihmm_amd import _lib as L
K, D, LM = 64, 32, 257
T = int(float(sys.argv[1])) if len(sys.argv) > 1 else 30000000
rng = np.random.default_rng(0)
tran = 0.9 * np.eye(K) + 0.1 / (K - 1) * (1 - np.eye(K))
means = rng.normal(0, 5, size=(K, D))
chols = np.tile(np.eye(D), (K, 1, 1))
e = HipEngine(0)
t0 = time.time(); e.generate(tran, means, chols, T, seed=1); print("generate T=%d: %.2f s" % (T, time.time() - t0))
vt = 1.0 + rng.random((K, K)) * 10 + 50 * np.eye(K)
e.set_globals(np.log(np.full(K, 1.0 / K)), digamma(vt) - digamma(vt.sum(1))[:, None])
e.set_emission_niw(means + 0.1, np.tile(np.eye(D) * 1.2, (K, 1, 1)), np.full(K, 5.0), np.full(K, D + 5.0))
B = T // LM
starts = np.arange(B, dtype=np.int64) * LM
for rep in range(2):
    t0 = time.time(); st = e.estep(starts, LM, flags=L.TRANS_WRAP); dt = time.time() - t0
print("E-step over %d windows (%d rows): %.1f ms -> %.3g updates/s; sum A_raw / rows = %.12f; lb %.6e" % (
    B, B * LM, dt * 1e3, B * LM * K / dt, st.A_raw.sum() / (B * LM), st.lb[0]))
