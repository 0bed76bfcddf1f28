"""Epoch E-step (T = 1e6 rows, K = 64, Lm = 257) against the observation width: per-kernel
HIP-event times and achieved fp64 TF/s -- looks for cliffs between the emission / statistics
kernel variants (orbit schedule for D % 8 == 0 up to 40, table-driven beyond)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from pysvihmm_amd.engine import HipEngine  # noqa: E402
from pysvihmm_amd import _lib as L  # noqa: E402

eng = HipEngine(0)
T, K, LM = bench.T, bench.K, bench.LM
B = T // LM
st = np.arange(B, dtype=np.int64) * LM
for D in [int(a) for a in sys.argv[1:]] or [8, 16, 24, 31, 32, 40, 48, 64, 79]:
    rs = np.random.RandomState(1)
    tran = 0.9 * np.eye(K) + 0.1 / (K - 1) * (1.0 - np.eye(K))
    means = rs.normal(0.0, 5.0, size=(K, D))
    chols = np.broadcast_to(np.eye(D), (K, D, D)).copy()
    eng.generate(tran, means, chols, T, seed=3)
    head = eng.read_generated(want_sts=False)[0][:20000]
    pb = bench.variational_state(rs, means, head, K, D, T)
    eng.set_globals(pb["mod_init"], pb["ltran"])
    eng.set_emission_niw(pb["mu"], pb["sigma"], pb["kappa"], pb["nu"])
    for _ in range(3):
        eng.estep(st, LM, flags=L.TRANS_WRAP, read=False)
    eng.sync(); eng.profile(True); eng.profile_reset()
    for _ in range(6):
        eng.estep(st, LM, flags=L.TRANS_WRAP, read=False)
    p = eng.profile_read(); eng.profile(False)
    ms = {k: v[0] / v[1] for k, v in p.items() if v[1] and k in ("emission", "forward_backward", "stats")}
    F = (D + 1) * (D + 2) // 2
    rows = B * LM
    print("D=%3d: emission %.3f ms (%.1f TF/s)  sweeps %.3f  stats %.3f ms (%.1f TF/s)" % (
        D, ms["emission"], 2.0 * F * K * rows / ms["emission"] * 1e-9, ms["forward_backward"],
        ms["stats"], 2.0 * (F + K) * K * rows / ms["stats"] * 1e-9))
