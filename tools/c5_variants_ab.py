import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pysvihmm_amd.engine import HipEngine
from tests.helpers import make_problem
K, D, Lm = 256, 64, 257
T = 1000000
pb = make_problem(K, D, T, seed=3, sep=4.0)
e = HipEngine(0); e.set_obs(pb['obs'], None); e.set_globals(pb['mod_init'], pb['ltran'])
e.set_emission_niw(pb['mu'], pb['sigma'], pb['kappa'], pb['nu'])
B = T // Lm; starts = np.arange(B, dtype=np.int64) * Lm
ref = None
for name, var in [("base", {}), ("posterior pass (15:1)", {15: 1}), ("round start (13,14,15:1)", {13: 1, 14: 1, 15: 1})]:
    for i in (3, 12, 13, 14, 15): e.set_variant(i, var.get(i, 0))
    e.estep(starts, Lm, read=False); out = e.read_packed().buf.copy()
    if ref is None: ref = out
    err = np.max(np.abs(out - ref) / (np.abs(ref) + 1e-300 + 1e-12 * np.abs(ref).max()))
    e.profile(True); e.profile_reset()
    t0 = time.time()
    for _ in range(3): e.estep(starts, Lm, read=False)
    e.sync(); dt = (time.time() - t0) / 3
    pr = {k: round(ms / 3, 3) for k, (ms, c) in e.profile_read().items()}
    e.profile(False)
    print("%-20s %.2f ms  relerr vs base %.2e  %s" % (name, dt * 1e3, err, pr), flush=True)
