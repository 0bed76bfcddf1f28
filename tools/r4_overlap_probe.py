"""Round-4 bounds on what overlapping / fusing the three launches of the headline step can give
(VERDICT r3 next #1).  One process, the bench workload, median of blocks of whole steps
(parameter upload -> emission -> sweeps -> statistics -> read-back) under:
  base          the default path, HIP-event profiling off
  prof          the same with the per-kernel event pairs bench.py records
  no_sweeps     (needs the measurement build: make -C pysvihmm_amd/csrc measure and
                SVIHMM_HIP_LIB=build_exp/libsvihmm_measure.so) variant[7] = 9: the sweep launch skipped (statistics on stale messages): the step
                if the sweeps cost NOTHING -- the ceiling of any overlap design
  pipeline      variant[4] = 2: the two-stream split (sweeps of one half co-resident with the
                emission GEMM of the other): co-residency forced at kernel granularity
  q_pass        variant[15] = 2: posteriors by their own pass, statistics GEMM on plain q
                (what the GEMM gains if ah*bh*scale left its staging code / what the pass costs)
Per-kernel HIP-event averages are printed for the configurations that run with events.
`--ko` (one process per library, SVIHMM_HIP_LIB): the kernel-level knock-outs of the sweeps' HBM side."""
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
LM = bench.LM
B = bench.T // LM
st = np.arange(B, dtype=np.int64) * LM


def step():
    e.set_emission_niw(pb["mu"], pb["sigma"], pb["kappa"], pb["nu"], check=False)
    e.set_globals(pb["mod_init"], pb["ltran"])
    e.estep(st, LM, flags=L.TRANS_WRAP, read=False)
    return e.read_packed()


def run(name, prof, variants, steps=20, reps=7):
    for k, v in variants.items():
        e.set_variant(k, v)
    for _ in range(3):
        step()
    e.sync()
    if prof:
        e.profile(True); e.profile_reset()
    blk = []
    for _ in range(reps):
        e.sync(); t0 = time.perf_counter()
        for _ in range(steps):
            step()
        e.sync(); blk.append((time.perf_counter() - t0) / steps * 1e3)
    out = {"config": name, "ms_per_step": round(float(np.median(blk)), 4), "events": bool(prof)}
    if prof:
        p = e.profile_read(); e.profile(False)
        out["kernels_ms"] = {k: round(v[0] / max(v[1], 1), 4) for k, v in p.items() if v[1]}
    for k in variants:
        e.set_variant(k, 0)
    print(json.dumps(out), flush=True)
    return out


if "--ko" in sys.argv:
    # run under SVIHMM_HIP_LIB=build_exp/libsvihmm_koN.so (make -C pysvihmm_amd/csrc ko KO=N: the scaled
    # sweeps without their ah / bh stores (1), without their Eh loads (2), without both (3))
    tag = os.path.basename(os.environ.get("SVIHMM_HIP_LIB", "product"))
    run(tag, False, {})
    run(tag + "+prof", True, {})
    sys.exit(0)
res = []
res.append(run("base", False, {}))
res.append(run("prof", True, {}))
res.append(run("no_sweeps", False, {7: 9}))
res.append(run("no_sweeps+prof", True, {7: 9}))
res.append(run("pipeline", False, {4: 2}))
res.append(run("pipeline+prof", True, {4: 2}))
res.append(run("q_pass", False, {15: 2}))
res.append(run("q_pass+prof", True, {15: 2}))
res.append(run("base_again", False, {}))
