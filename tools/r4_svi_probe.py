"""Wall time per iteration of hmmsgd_metaobs.VBHMM.infer at the literal configs[2] minibatch (64 windows,
L = 128) on the bench sequence -- bench.py's svi_iteration_s64 leg alone, with engine variants from the
command line (e.g. `5:4` = variant[5] = 4) for A/B runs."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench
from pysvihmm_amd.engine import HipEngine

eng = HipEngine(0)
rs, tran, means, chols = bench.true_process(0)
eng.generate(tran, means, chols, bench.T, seed=bench.SEED)
obs = eng.read_generated(want_sts=False)[0]
for kv in sys.argv[1:]:
    if kv == "f32":
        eng.set_precision("f32")
        continue
    eng.set_variant(int(kv.split(":")[0]), int(kv.split(":")[1]))
for rep in range(3):
    r = bench.svi_iteration(eng, obs)
    print("%s  ms/iteration %.4f  (iter_time median %.4f)" % (" ".join(sys.argv[1:]) or "default", r["ms"], r["iter_time_median_ms"]), flush=True)
