"""fp32-mode epoch step of the bench workload: step time and per-kernel HIP-event times, for the
statistics GEMM on the bf16 pipe (default) and the fp32-input MFMA kernel (variant 10 = 2)."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench
from _workload import bench_problem
from pysvihmm_amd.engine import HipEngine
from pysvihmm_amd import _lib as L
e = HipEngine(0)
pb = bench_problem(e)
B = bench.T // bench.LM
st = np.arange(B, dtype=np.int64) * bench.LM


def step():
    e.set_emission_niw(pb["mu"], pb["sigma"], pb["kappa"], pb["nu"], check=False)
    e.set_globals(pb["mod_init"], pb["ltran"])
    e.estep(st, bench.LM, flags=L.TRANS_WRAP, read=False)
    return e.read_packed()


ref = step().buf.copy()
e.set_precision("f32")
cases = (("bf16x3", 0),) if os.environ.get("SVIHMM_HIP_LIB") else (("bf16x3", 0), ("f32_mfma", 2), ("bf16x3", 0))
for name, var in cases:
    e.set_variant(10, var)
    for _ in range(3):
        out = step()
    blk = []
    for _ in range(5):
        e.sync(); t0 = time.perf_counter()
        for _ in range(20):
            out = step()
        e.sync(); blk.append((time.perf_counter() - t0) / 20 * 1e3)
    e.profile(True); e.profile_reset()
    for _ in range(10):
        step()
    p = e.profile_read(); e.profile(False)
    scale = np.maximum(np.abs(ref), 1e-6 * B * bench.LM)
    print(json.dumps({"lib": os.path.basename(os.environ.get("SVIHMM_HIP_LIB", "product")), "stats": name, "ms_per_step": round(float(np.median(blk)), 4),
                      "kernels_ms": {k: round(v[0] / max(v[1], 1), 4) for k, v in p.items() if v[1]},
                      "max_rel_err_vs_f64": float(np.max(np.abs(out.buf - ref) / scale))}), flush=True)
