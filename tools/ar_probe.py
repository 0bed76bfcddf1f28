"""RCCL all-reduce latency probe at world size 1 (no torch in the process)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pysvihmm_amd.engine import HipEngine
from tests.helpers import make_problem
e = HipEngine(0)
e.comm_init(e.comm_unique_id(), 0, 1)
pb = make_problem(64, 32, 20000, seed=1)
e.set_obs(pb['obs'], None); e.set_globals(pb['mod_init'], pb['ltran']); e.set_emission_niw(pb['mu'], pb['sigma'], pb['kappa'], pb['nu'])
st = np.arange(64) * 257
e.estep(st, 257, read=False); e.sync()
for rep in range(6):
    t0 = time.perf_counter(); e.allreduce_packed(); t1 = time.perf_counter(); e.sync(); t2 = time.perf_counter()
    out = e.read_packed(); t3 = time.perf_counter()
    print('allreduce enqueue %.1f us, sync %.1f us, read_packed %.1f us' % ((t1-t0)*1e6, (t2-t1)*1e6, (t3-t2)*1e6))
t0 = time.perf_counter()
for _ in range(20):
    e.allreduce_host(np.zeros(1))
print('allreduce_host(1 double) %.1f us each' % ((time.perf_counter()-t0)/20*1e6))
