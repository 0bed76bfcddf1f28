"""64-window minibatch E-step: statistics / finalize time against the number of row chunks."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bench
from pysvihmm_amd.engine import HipEngine
from pysvihmm_amd import _lib as L
eng = HipEngine(0)
rs, tran, means, chols = bench.true_process(0)
eng.generate(tran, means, chols, bench.T, seed=bench.SEED)
pb = bench.variational_state(rs, means, eng.read_generated(want_sts=False)[0][:20000])
for nwin in (64, 256, 1024):
    st = (np.arange(nwin, dtype=np.int64) * (bench.T // nwin)) % (bench.T - bench.LM)
    eng.set_globals(pb["mod_init"], pb["ltran"]); eng.set_emission_niw(pb["mu"], pb["sigma"], pb["kappa"], pb["nu"])
    for tgt in (0, 32, 64, 96, 128, 192, 256):
        eng.set_variant(8, tgt)
        for _ in range(3): eng.estep(st, bench.LM, flags=L.TRANS_WRAP, read=False)
        eng.profile(True); eng.profile_reset()
        for _ in range(20): eng.estep(st, bench.LM, flags=L.TRANS_WRAP, read=False)
        p = eng.profile_read(); eng.profile(False)
        print(nwin, "chunks", tgt, {k: round(v[0] / v[1] * 1e3, 1) for k, v in p.items() if k in ("stats", "finalize", "forward_backward", "emission")})
