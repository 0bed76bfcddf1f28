"""bench.py's configs[4] record (K=256, D=64, T=1e6: fp64 step + the fp32 mode's) alone."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pysvihmm_amd.engine import HipEngine
from pysvihmm_amd import _lib as L
eng = HipEngine(0)
for kv in sys.argv[1:]:
    eng.set_variant(int(kv.split(":")[0]), int(kv.split(":")[1]))
print(json.dumps(bench.wide_model(eng, L), indent=1))
