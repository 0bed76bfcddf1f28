"""BASELINE.json configs[0] (K=4, D=2, T=1e3) and configs[1] (K=16, D=8, T=1e5): wall time of the
windowed epoch E-step (L = 128 where it fits, else L = 16) and of the whole-chain E-step."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from pysvihmm_amd.engine import HipEngine  # noqa: E402
from pysvihmm_amd import _lib as L  # noqa: E402

eng = HipEngine(0)
for (K, D, T, Lm) in ((4, 2, 1000, 33), (16, 8, 100000, 257)):
    rs = np.random.RandomState(2)
    tran = 0.9 * np.eye(K) + 0.1 / (K - 1) * (1.0 - np.eye(K))
    means = rs.normal(0.0, 5.0, size=(K, D))
    chols = np.broadcast_to(np.eye(D), (K, D, D)).copy()
    eng.generate(tran, means, chols, T, seed=5)
    head = eng.read_generated(want_sts=False)[0][:20000]
    pb = bench.variational_state(rs, means, head, K, D, T)
    eng.set_globals(pb["mod_init"], pb["ltran"])
    eng.set_emission_niw(pb["mu"], pb["sigma"], pb["kappa"], pb["nu"])
    B = T // Lm
    st = np.arange(B, dtype=np.int64) * Lm
    for name, s, lm in (("windows %d x %d" % (B, Lm), st, Lm), ("whole chain", np.zeros(1, dtype=np.int64), T)):
        for _ in range(3):
            eng.estep(s, lm, flags=L.TRANS_WRAP)
        t0 = time.perf_counter()
        n = 50
        for _ in range(n):
            eng.estep(s, lm, flags=L.TRANS_WRAP)
        dt = (time.perf_counter() - t0) / n
        print("K=%d D=%d T=%d  %-22s %.3f ms per E-step (statistics read back)  %.2e updates/s" % (
            K, D, T, name, dt * 1e3, len(s) * lm * K / dt))
