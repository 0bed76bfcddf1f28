"""A/B of a set_variant switch on the bench shape (same process, same box):
  tools/ab_emission.py                     emission_orbit: 1 keeps K1b (table-driven features), 0 = K1c
  tools/ab_emission.py stats 4 0           statistics GEMM: 4 = single-column staging, 0 = column pairs"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from _workload import bench_problem
from pysvihmm_amd.engine import HipEngine
from pysvihmm_amd import _lib as L
e = HipEngine(0)
pb = bench_problem(e)
B = bench.T // bench.LM
st = np.arange(B, dtype=np.int64) * bench.LM
e.set_globals(pb["mod_init"], pb["ltran"]); e.set_emission_niw(pb["mu"], pb["sigma"], pb["kappa"], pb["nu"])
res = {}
name = sys.argv[1] if len(sys.argv) > 1 else "emission_orbit"
va, vb = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1, 0)
for rnd in range(3):
    for v in (va, vb):
        e.set_variant(name, v)
        for _ in range(2): e.estep(st, bench.LM, flags=L.TRANS_WRAP, read=False)
        e.sync(); e.profile(True); e.profile_reset()
        for _ in range(10): e.estep(st, bench.LM, flags=L.TRANS_WRAP, read=False)
        p = e.profile_read(); e.profile(False)
        res[v] = e.read_packed().buf.copy()
        print(rnd, "%s=%d" % (name, v), {k: round(x[0] / max(x[1], 1), 4) for k, x in p.items() if x[1]})
d = np.abs(res[va] - res[vb]) / (np.abs(res[va]) + 1e-300)
print("packed statistics, %s %d vs %d: max rel diff %.3g" % (name, va, vb, d.max()))
