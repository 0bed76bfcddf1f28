"""fp32-mode epoch step: per-kernel HIP-event times and deviation from the fp64 statistics."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bench
from pysvihmm_amd.engine import HipEngine
from pysvihmm_amd import _lib as L
eng = HipEngine(0)
rs, tran, means, chols = bench.true_process(0)
eng.generate(tran, means, chols, bench.T, seed=bench.SEED)
pb = bench.variational_state(rs, means, eng.read_generated(want_sts=False)[0][:20000])
B = bench.T // bench.LM; st = np.arange(B, dtype=np.int64) * bench.LM
eng.set_globals(pb["mod_init"], pb["ltran"]); eng.set_emission_niw(pb["mu"], pb["sigma"], pb["kappa"], pb["nu"])
ref = eng.estep(st, bench.LM, flags=L.TRANS_WRAP).buf.copy()
for mode in ("f64", "f32"):
    eng.set_precision(mode)
    for _ in range(3): out = eng.estep(st, bench.LM, flags=L.TRANS_WRAP)
    eng.profile(True); eng.profile_reset()
    for _ in range(10): eng.estep(st, bench.LM, flags=L.TRANS_WRAP, read=False)
    p = eng.profile_read(); eng.profile(False)
    err = np.max(np.abs(out.buf - ref) / np.maximum(np.abs(ref), 1e-6 * B * bench.LM))
    print(mode, {k: round(v[0] / v[1], 4) for k, v in p.items()}, "err %.2e" % err)
