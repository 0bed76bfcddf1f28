"""configs[4] shape (K=256, D=64 full covariance): kernel time breakdown."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pysvihmm_amd.engine import HipEngine
from tests.helpers import make_problem
K, D, Lm = 256, 64, 257
T = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
pb = make_problem(K, D, T, seed=3, sep=4.0)
e = HipEngine(0); e.set_obs(pb['obs'], None); e.set_globals(pb['mod_init'], pb['ltran'])
e.set_emission_niw(pb['mu'], pb['sigma'], pb['kappa'], pb['nu'])
B = T // Lm; starts = np.arange(B, dtype=np.int64) * Lm
e.estep(starts, Lm, read=False); e.sync()
e.profile(True); e.profile_reset()
t0 = time.time()
for _ in range(3): e.estep(starts, Lm, read=False)
e.sync(); dt = (time.time() - t0) / 3
print("K=%d D=%d T=%d B=%d: %.2f ms/step -> %.3g upd/s" % (K, D, T, B, dt * 1e3, B * Lm * K / dt))
for k, (ms, c) in e.profile_read().items(): print("   %-18s %9.3f ms per step (%d launches)" % (k, ms / 3, c // 3))
F = (D + 1) * (D + 2) // 2
fl = B * Lm * (2.0 * F * K + 2.0 * (F + K) * K + 2 * 2.0 * K * K * 17 / 16)
print("   algorithmic MFMA flops per step %.3g -> %.1f TF/s end to end (fp64 peak 78.6)" % (fl, fl / dt / 1e12))
