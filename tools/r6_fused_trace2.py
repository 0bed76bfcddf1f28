import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from pysvihmm_amd.engine import HipEngine
from pysvihmm_amd import _lib as L
sys.path.insert(0, os.path.join(ROOT, "tools"))
from _workload import bench_problem
B = 64
e = HipEngine(0)
pb = bench_problem(e)
LM, T = bench.LM, bench.T
st = (np.arange(B, dtype=np.int64) * (T // B)) % (T - LM)
e.set_globals(pb["mod_init"], pb["ltran"]); e.set_emission_niw(pb["mu"], pb["sigma"], pb["kappa"], pb["nu"])
for _ in range(5):
    e.estep(st, LM, flags=L.TRANS_WRAP, read=False)
e.sync()
os.environ["SVIHMM_PIPE_DBG"] = "/tmp/pipe_dbg.txt"
e.estep(st, LM, flags=L.TRANS_WRAP, read=False); e.sync()
lines = open("/tmp/pipe_dbg.txt").read().splitlines()
nsw = int(lines[0].split()[2])
for i in (0, 1, 2, 3, 30, 31):
    print(lines[1 + i])
e.close()
