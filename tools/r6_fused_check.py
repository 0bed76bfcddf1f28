"""round 6: the fused sweep + statistics launch against the separate launches (same handle, variant 4 = 1 switches
the fused path off; parity against the oracle is the test suite's business); per-call wall time of the 64-window E-step both ways."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from pysvihmm_amd.engine import HipEngine
from pysvihmm_amd import _lib as L
sys.path.insert(0, os.path.join(ROOT, "tools"))
from _workload import bench_problem

e = HipEngine(0)
pb = bench_problem(e)
LM, T, K = bench.LM, bench.T, bench.K
obs = e.read_generated(want_sts=False)[0]
for B in (64, 9, 1, 100, 128):
    st = (np.arange(B, dtype=np.int64) * (T // max(B, 1))) % (T - LM)
    e.set_globals(pb["mod_init"], pb["ltran"]); e.set_emission_niw(pb["mu"], pb["sigma"], pb["kappa"], pb["nu"])
    res = {}
    for mode in (1, 0):
        e.set_variant("pipeline", mode)
        out = e.estep(st, LM, flags=L.TRANS_WRAP)
        for _ in range(5):
            e.estep(st, LM, flags=L.TRANS_WRAP, read=False)
        e.sync()
        t0 = time.perf_counter()
        n = 200
        for _ in range(n):
            e.estep(st, LM, flags=L.TRANS_WRAP, read=False)
        e.sync()
        dt = (time.perf_counter() - t0) / n
        res[mode] = (out.buf.copy(), dt, e.last_kernel("forward_backward") if hasattr(e, "last_kernel") else "")
    a, b = res[1][0], res[0][0]
    sc = np.maximum(np.abs(a), 1e-9 * B * LM)
    print("B=%d  separate %.1f us (%s)  fused %.1f us (%s)  max rel diff %.3g" % (B, res[1][1] * 1e6, res[1][2], res[0][1] * 1e6, res[0][2], float(np.max(np.abs(a - b) / sc))))
e.set_variant("pipeline", 0)
e.close()
