import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools"))
import numpy as np
import bench
from pysvihmm_amd.engine import HipEngine
from pysvihmm_amd import _lib as L
from _workload import bench_problem
e = HipEngine(0)
pb = bench_problem(e)
B = bench.T // bench.LM
st = np.arange(B, dtype=np.int64) * bench.LM
e.set_globals(pb["mod_init"], pb["ltran"]); e.set_emission_niw(pb["mu"], pb["sigma"], pb["kappa"], pb["nu"])
res = {}
for v in (1, 0, 1, 0):
    e.set_variant(12, v)
    for _ in range(3): out = e.estep(st, bench.LM, flags=L.TRANS_WRAP)
    e.sync(); e.profile(True); e.profile_reset()
    for _ in range(10): e.estep(st, bench.LM, flags=L.TRANS_WRAP, read=False)
    p = e.profile_read(); e.profile(False)
    print("variant12=%d" % v, {k: round(x[0] / max(x[1], 1), 4) for k, x in p.items() if x[1]})
    res[v] = out.buf.copy()
d = np.abs(res[0] - res[1]) / (1e-9 + np.abs(res[1]))
print("max rel diff NB3 vs NB2:", d.max(), "bit-identical:", np.array_equal(res[0], res[1]))
