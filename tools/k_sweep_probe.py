"""Epoch E-step (T = 1e6, D = 32, Lm = 257) against the number of states: per-kernel HIP-event
times -- looks for cliffs between the tile instantiations (K <= 16 / 32 / 48 / 64)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from pysvihmm_amd.engine import HipEngine  # noqa: E402
from pysvihmm_amd import _lib as L  # noqa: E402

eng = HipEngine(0)
for kv in os.environ.get("SVIHMM_SET", "").split(","):      # engine switches for A/B runs, e.g. 8:256
    if kv:
        eng.set_variant(int(kv.split(":")[0]), int(kv.split(":")[1]))
T, D, LM = bench.T, bench.D, bench.LM
B = T // LM
st = np.arange(B, dtype=np.int64) * LM
for K in [int(a) for a in sys.argv[1:]] or [16, 24, 32, 40, 48, 56, 64]:
    rs = np.random.RandomState(1)
    tran = 0.9 * np.eye(K) + 0.1 / (K - 1) * (1.0 - np.eye(K))
    means = rs.normal(0.0, 5.0, size=(K, D))
    chols = np.broadcast_to(np.eye(D), (K, D, D)).copy()
    if K <= 64:
        eng.generate(tran, means, chols, T, seed=3)
        head = eng.read_generated(want_sts=False)[0][:20000]
    else:                                      # the device generator covers K <= 64
        from pysvihmm_amd.gen_synthetic import generate_data_fast
        obs, _ = generate_data_fast(tran, means, None, T, np.random.default_rng(3))
        eng.set_obs(obs, None)
        head = obs[:20000]
    pb = bench.variational_state(rs, means, head, K, D, T)
    eng.set_globals(pb["mod_init"], pb["ltran"])
    eng.set_emission_niw(pb["mu"], pb["sigma"], pb["kappa"], pb["nu"])
    for _ in range(3):
        eng.estep(st, LM, flags=L.TRANS_WRAP, read=False)
    eng.sync(); eng.profile(True); eng.profile_reset()
    for _ in range(10):
        eng.estep(st, LM, flags=L.TRANS_WRAP, read=False)
    p = eng.profile_read(); eng.profile(False)
    ms = {k: v[0] / v[1] for k, v in p.items() if v[1] and k in ("emission", "forward_backward", "posterior", "stats")}
    tot = sum(ms.values())
    print("K=%3d: emission %.3f  sweeps %.3f  posterior pass %.3f  stats %.3f  sum %.3f ms  -> %.2e upd/s" % (
        K, ms["emission"], ms["forward_backward"], ms.get("posterior", 0.0), ms["stats"], tot, B * LM * K / tot * 1e3))
