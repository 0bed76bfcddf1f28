"""Round 5: the register-resident one-wave sweep (k_wave_linr) against the four-wave kernel
(k_wave_lin4, variant[7] = 4): parity of the minibatch statistics with the C oracle on ragged shapes,
then the sweep launch's HIP-event time at the S = 64 shape (64 windows of 257 rows, K = 64) in both
precisions."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import make_problem, unpack
from pysvihmm_amd.engine import HipEngine
from pysvihmm_amd import _lib as L
from oracle import ref_c


def parity(K, D, Lm, B, dtype, var):
    T = max(4 * Lm, 600)
    pb = make_problem(K, D, T, seed=K + 7 * B, miss=0.1)
    starts = np.random.default_rng(B).integers(0, T - Lm + 1, size=B)
    e = HipEngine(0, dtype=dtype)
    if var: e.set_variant(7, var)
    e.set_obs(pb["obs"], pb["mask"]); e.set_globals(pb["mod_init"], pb["ltran"])
    e.set_emission_niw(pb["mu"], pb["sigma"], pb["kappa"], pb["nu"])
    st = e.estep(starts, Lm, flags=L.TRANS_WRAP)
    ref = ref_c.estep_minibatch(pb["obs"], pb["mask"], starts, Lm, pb["mod_init"], pb["ltran"], pb["mu"],
                                pb["sigma"], pb["kappa"], pb["nu"], flags=2)
    err = np.abs(st.buf - ref) / (1e-9 * B * Lm + np.abs(ref))
    e.close()
    return err.max(), abs(st.buf[-1] - ref[-1]) / abs(ref[-1])


def timing(dtype, var, K=64, D=32, Lm=257, B=64, reps=200):
    T = 20000
    pb = make_problem(K, D, T, seed=11)
    starts = np.random.default_rng(3).integers(0, T - Lm + 1, size=B)
    e = HipEngine(0, dtype=dtype)
    if var: e.set_variant(7, var)
    e.set_obs(pb["obs"], None); e.set_globals(pb["mod_init"], pb["ltran"])
    e.set_emission_niw(pb["mu"], pb["sigma"], pb["kappa"], pb["nu"])
    for _ in range(5): e.estep(starts, Lm, flags=L.TRANS_WRAP)
    e.profile(True); e.profile_reset()
    for _ in range(reps): e.estep(starts, Lm, flags=L.TRANS_WRAP)
    pr = e.profile_read()
    e.profile(False)
    out = {n: 1e3 * m / c for n, (m, c) in pr.items()}
    e.close()
    return out


if __name__ == "__main__":
    for dtype in ("f64", "f32"):
        for (K, D, Lm, B) in () if os.environ.get("NO_PARITY") else ((64, 8, 257, 64), (64, 8, 33, 7), (33, 8, 65, 64), (48, 4, 12, 256), (17, 3, 5, 9), (64, 8, 1, 5), (64, 8, 2, 70)):
            for var in (0, 4):
                er, lb = parity(K, D, Lm, B, dtype, var)
                print("parity %s K=%d D=%d Lm=%d B=%d variant7=%d: max rel err %.3g  lb rel %.3g" % (dtype, K, D, Lm, B, var, er, lb), flush=True)
    for dtype in ("f64", "f32"):
        for var in (0, 4):
            print("timing", dtype, "variant7=%d" % var, {k: round(v, 2) for k, v in timing(dtype, var).items()}, "us per launch", flush=True)
