"""Sweep-kernel time vs number of windows (latency- or bandwidth-bound?)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from _workload import bench_problem
from pysvihmm_amd.engine import HipEngine
from pysvihmm_amd import _lib as L
e = HipEngine(0)
pb = bench_problem(e)
for kv in os.environ.get("SVIHMM_SET", "").split(","):
    if kv:
        k, v = kv.split(":"); e.set_variant(k, int(v))
e.set_globals(pb["mod_init"], pb["ltran"])
e.set_emission_niw(pb["mu"], pb["sigma"], pb["kappa"], pb["nu"])
LM = bench.LM
Bs = [int(a) for a in sys.argv[1:]] or [3891, 2048, 1024, 512, 256]
for B in Bs:
    st = np.arange(B, dtype=np.int64) * LM
    for _ in range(2):
        e.estep(st, LM, flags=L.TRANS_WRAP, read=False)
    e.profile(True); e.profile_reset()
    for _ in range(5):
        e.estep(st, LM, flags=L.TRANS_WRAP, read=False)
    e.sync()
    p = e.profile_read(); e.profile(False)
    print(B, {k: round(v[0] / v[1], 4) for k, v in p.items() if k in ("emission", "forward_backward", "posterior", "stats")})
