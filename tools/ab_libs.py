"""A/B of experiment builds on one box: per-kernel HIP-event times of the bench E-step for
each library given on the command line (run one process per library: SVIHMM_HIP_LIB)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os
sys.path.insert(0, %r)
import numpy as np
import bench
from pysvihmm_amd.engine import HipEngine
from pysvihmm_amd import _lib as L
sys.path.insert(0, os.path.join(%r, "tools"))
from _workload import bench_problem
e = HipEngine(0)
pb = bench_problem(e)
B = bench.T // bench.LM
st = np.arange(B, dtype=np.int64) * bench.LM
e.set_globals(pb["mod_init"], pb["ltran"]); e.set_emission_niw(pb["mu"], pb["sigma"], pb["kappa"], pb["nu"])
for _ in range(3): e.estep(st, bench.LM, flags=L.TRANS_WRAP, read=False)
e.sync(); e.profile(True); e.profile_reset()
for _ in range(10): e.estep(st, bench.LM, flags=L.TRANS_WRAP, read=False)
p = e.profile_read()
print({k: round(v[0] / max(v[1], 1), 4) for k, v in p.items() if v[1]})
''' % (ROOT, ROOT)
for rnd in range(2):
    for lib in sys.argv[1:]:
        env = dict(os.environ)
        if lib != "base":
            env["SVIHMM_HIP_LIB"] = os.path.join(ROOT, lib)
        out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
        print(rnd, lib, out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-400:])
