#!/usr/bin/env python3
"""First-contact GPU diagnostics: layout self-test, fp64 peaks, per-kernel
errors vs the oracle for every variant, and a kernel-time breakdown at the bench
shape.  Prints everything; never stops at the first failure.  (A checker script: it lives
under tests/ because it imports the oracle; run it as `python tests/gpu_diag.py`.)"""
import os
import sys
import time
import traceback

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pysvihmm_amd.engine import HipEngine  # noqa: E402
from pysvihmm_amd import _lib as L  # noqa: E402
from oracle import ref_c  # noqa: E402
from tests.helpers import make_problem, unpack, relerr  # noqa: E402


def section(s):
    print("\n==== " + s, flush=True)


def main():
    eng = HipEngine(0)
    section("mfma layout")
    try:
        rng = np.random.default_rng(1)
        A = rng.normal(size=(16, 4)); B = rng.normal(size=(4, 16))
        C = eng.selftest_mfma(A, B)
        print("max abs err vs A@B:", np.abs(C - A @ B).max(), " vs (A@B).T:", np.abs(C - (A @ B).T).max())
    except Exception:
        traceback.print_exc()
    section("fp64 peaks")
    try:
        sys.path.insert(0, os.path.join(REPO, "tools"))
        from _probe import probe_fp64
        print("v_mfma_f64_16x16x4: %.1f TFLOP/s" % probe_fp64(0))
        print("v_fma_f64:          %.1f TFLOP/s" % probe_fp64(1))
    except Exception:
        traceback.print_exc()

    for (K, D, T, Lm, B, miss) in [(4, 2, 500, 21, 10, 0.1), (16, 8, 4000, 65, 40, 0.1),
                                   (64, 32, 8000, 257, 24, 0.05), (80, 6, 2000, 30, 6, 0.1)]:
        section("parity K=%d D=%d Lm=%d B=%d" % (K, D, Lm, B))
        pb = make_problem(K, D, T, seed=K + D, miss=miss)
        starts = np.random.default_rng(K).integers(0, T - Lm + 1, size=B)
        try:
            eng.set_obs(pb["obs"], pb["mask"])
            eng.set_globals(pb["mod_init"], pb["ltran"])
            eng.set_emission_niw(pb["mu"], pb["sigma"], pb["kappa"], pb["nu"])
            ref_ll = np.stack([ref_c.lliks_niw(pb["obs"][s:s + Lm], pb["mu"], pb["sigma"], pb["kappa"], pb["nu"]) for s in starts])
            ll = eng.loglik(starts, Lm)
            print("  emission: max abs err %.3g (|ll| max %.3g)" % (np.abs(ll - ref_ll).max(), np.abs(ref_ll).max()))
            r = eng.forward_backward(starts, Lm)
            la = np.stack([ref_c.forward(ref_ll[b], pb["mod_init"], pb["ltran"]) for b in range(B)])
            lb = np.stack([ref_c.backward(ref_ll[b], pb["ltran"]) for b in range(B)])
            print("  lalpha err %.3g  lbeta err %.3g" % (np.abs(r["lalpha"] - la).max(), np.abs(r["lbeta"] - lb).max()))
            qs = [ref_c.posterior(la[b], lb[b]) for b in range(B)]
            q = np.stack([x[0] for x in qs]); lbs = np.array([x[1] for x in qs])
            print("  var_x err %.3g  local_lb rel %.3g" % (np.abs(r["var_x"] - q).max(), relerr(r["local_lb"], lbs)))
            ref = ref_c.estep_minibatch(pb["obs"], pb["mask"], starts, Lm, pb["mod_init"], pb["ltran"], pb["mu"], pb["sigma"], pb["kappa"], pb["nu"], flags=2)
            A, xbar, neff, S, lbt = unpack(ref, K, D)
            for sv in (2, 3):
                eng.set_variant("stats", sv)
                st = eng.estep(starts, Lm, flags=L.TRANS_WRAP)
                print("  stats var %d: A %.3g xbar %.3g neff %.3g S %.3g lb %.3g (abs, scale %d)" % (
                    sv, np.abs(st.A_raw - A).max(), np.abs(st.xbar - xbar).max(), np.abs(st.neff - neff).max(),
                    np.abs(st.S - S).max(), abs(st.lb[0] - lbt), B * Lm))
            eng.set_variant("stats", 0)
        except Exception:
            traceback.print_exc()

    section("timing at bench shape K=64 D=32 Lm=257")
    try:
        K, D, Lm = 64, 32, 257
        T = 1000000
        t0 = time.time()
        pb = make_problem(K, D, T, seed=8675309, sep=5.0)
        print("  gen %.1fs" % (time.time() - t0))
        t0 = time.time(); eng.set_obs(pb["obs"], None); print("  H2D obs %.3fs" % (time.time() - t0))
        eng.set_globals(pb["mod_init"], pb["ltran"])
        eng.set_emission_niw(pb["mu"], pb["sigma"], pb["kappa"], pb["nu"])
        for B in (64, 3891):
            starts = (np.arange(B, dtype=np.int64) * Lm) % (T - Lm)
            for ev, sv, fv in ((2, 3, 2), (2, 2, 2), (2, 2, 1)):
                eng.set_variant("stats", sv); eng.set_variant("fb", fv)
                eng.estep(starts, Lm, read=False); eng.sync()
                eng.profile(True); eng.profile_reset()
                t0 = time.time()
                reps = 3
                for _ in range(reps):
                    eng.estep(starts, Lm, read=False)
                eng.sync()
                dt = (time.time() - t0) / reps
                prof = eng.profile_read(); eng.profile(False)
                print("  B=%d variants(em=%d,st=%d,fb=%d): %.3f ms/step -> %.3g upd/s" % (B, ev, sv, fv, dt * 1e3, B * Lm * K / dt))
                for k, (ms, c) in prof.items():
                    print("      %-18s %8.3f ms/launch (%d launches)" % (k, ms / c, c))
    except Exception:
        traceback.print_exc()
    eng.close()


if __name__ == "__main__":
    main()
