"""Sweep kernel crossover: wave-per-window against 16-window MFMA tiles, by minibatch size."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from _workload import bench_problem
from pysvihmm_amd.engine import HipEngine
from pysvihmm_amd import _lib as L
e = HipEngine(0)
pb = bench_problem(e)
e.set_globals(pb["mod_init"], pb["ltran"]); e.set_emission_niw(pb["mu"], pb["sigma"], pb["kappa"], pb["nu"])
for B in (64, 128, 256, 384, 512, 768, 1024, 1399):
    st = (np.arange(B, dtype=np.int64) * 257)
    row = []
    for v in (0, 2):
        e.set_variant(7, v)
        for _ in range(3): e.estep(st, 257, flags=L.TRANS_WRAP, read=False)
        e.sync(); e.profile(True); e.profile_reset()
        for _ in range(10): e.estep(st, 257, flags=L.TRANS_WRAP, read=False)
        p = e.profile_read(); e.profile(False)
        row.append(p["forward_backward"][0] / p["forward_backward"][1])
    print("B=%5d sweeps: wave-per-window %.3f ms, MFMA 16-window tiles %.3f ms" % (B, row[0], row[1]))
