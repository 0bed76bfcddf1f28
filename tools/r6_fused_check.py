"""round 6: the fused E-step launch against the separate launches (same handle; variant 4 = 1 switches the fused path off,
4 = the default's fused sweeps + statistics, 3 forces everything the fused kernel can take -- emission tiles included; parity against the oracle
is the test suite's business); per-call wall time of the E-step each way."""
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
    for mode in (1, 4, 3):
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
    a = res[1][0]
    sc = np.maximum(np.abs(a), 1e-9 * B * LM)
    print("B=%d  separate %.1f us (%s)" % (B, res[1][1] * 1e6, res[1][2]))
    for m, what in ((4, "sweeps + statistics"), (3, "emission + sweeps + statistics")):
        print("      %-32s %.1f us (%s)  max rel diff %.3g" % (what, res[m][1] * 1e6, res[m][2], float(np.max(np.abs(a - res[m][0]) / sc))))
e.set_variant("pipeline", 0)
e.close()
