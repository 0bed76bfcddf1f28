"""Epoch step of the bench workload in the fp32 mode, for rocprofv3 (kernel trace / PMC passes):
the fp64 headline trace cannot tell the emission kernel's float-storing launches apart by name."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from pysvihmm_amd.engine import HipEngine
from pysvihmm_amd import _lib as L
eng = HipEngine(0, dtype=os.environ.get("DTYPE", "f32"))
rs, tran, means, chols = bench.true_process(0)
eng.generate(tran, means, chols, bench.T, seed=bench.SEED)
pb = bench.variational_state(rs, means, eng.read_generated(want_sts=False)[0][:20000])
B = bench.T // bench.LM
st = np.arange(B, dtype=np.int64) * bench.LM
for _ in range(int(os.environ.get("STEPS", 8))):
    eng.set_globals(pb["mod_init"], pb["ltran"])
    eng.set_emission_niw(pb["mu"], pb["sigma"], pb["kappa"], pb["nu"], check=False)
    eng.estep(st, bench.LM, flags=L.TRANS_WRAP, read=False)
    eng.read_packed()
print(eng.precision())
