"""Host-side wall time of each call of one bench step (where do the non-kernel microseconds go?)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from _workload import bench_problem
from pysvihmm_amd.engine import HipEngine
from pysvihmm_amd import _lib as L
e = HipEngine(0)
pb = bench_problem(e)
LM = bench.LM
B = bench.T // LM
st = np.arange(B, dtype=np.int64) * LM
names = ["set_globals", "set_emission_niw", "estep(launch)", "read_packed", "total"]
acc = np.zeros(5)
N = 20
for it in range(N + 3):
    t0 = time.perf_counter()
    e.set_globals(pb["mod_init"], pb["ltran"])
    t1 = time.perf_counter()
    e.set_emission_niw(pb["mu"], pb["sigma"], pb["kappa"], pb["nu"], check=False)
    t2 = time.perf_counter()
    e.estep(st, LM, flags=L.TRANS_WRAP, read=False)
    t3 = time.perf_counter()
    e.read_packed()
    t4 = time.perf_counter()
    if it >= 3:
        acc += [t1 - t0, t2 - t1, t3 - t2, t4 - t3, t4 - t0]
for n, a in zip(names, acc / N * 1e3):
    print("%-18s %.4f ms" % (n, a))
