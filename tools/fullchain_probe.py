"""Full-chain E-step (one window = the whole sequence) timing."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pysvihmm_amd.engine import HipEngine
from tests.helpers import make_problem
for K, D, T in ((16, 8, 100000), (64, 32, 1000000)):
    if len(sys.argv) > 1 and int(sys.argv[1]) != K:
        continue
    pb = make_problem(K, D, T, seed=3, sep=4.0, miss=0.1)
    e = HipEngine(0)
    e.set_obs(pb['obs'], pb['mask']); e.set_globals(pb['mod_init'], pb['ltran'])
    e.set_emission_niw(pb['mu'], pb['sigma'], pb['kappa'], pb['nu'])
    for mode in ("scan", "sequential"):
        e.set_variant("chain", 0 if mode == "scan" else 1)
        for rep in range(2):
            e.profile(True); e.profile_reset()
            t0 = time.time()
            r = e.forward_backward([0], T, flags=1, want=("local_lb",))
            dt = time.time() - t0
            pr = e.profile_read(); e.profile(False)
        print("K=%d D=%d T=%d full chain (%s): %.1f ms wall without var_x D2H (%.3g upd/s)  lb=%.9e" % (
            K, D, T, mode, dt * 1e3, T * K / dt, r["local_lb"][0]))
        print("    kernels:", {k: round(v[0], 3) for k, v in pr.items()})
        if mode == "scan":
            t0 = time.time(); v = e.pred_logprob([0], T, flags=1); dt = time.time() - t0
            print("    pred_logprob (E-step + held-out term + reduction): %.1f ms -> %r" % (dt * 1e3, v))
            sts = np.asarray(pb.get('sts', np.zeros(T)), dtype=np.int32)
            e.forward_backward([0], T, flags=1, want=())
            t0 = time.time(); _, dm = e.state_argmax(sts, want_z=False); dt = time.time() - t0
            print("    state_argmax + K x K counts (labels in, counts out): %.1f ms, %d rows counted" % (dt * 1e3, dm.sum()))
            t0 = time.time(); q = e.read_intermediate("var_x", 1, T)[0]; zz = np.argmax(q, axis=1); dt = time.time() - t0
            print("    host route (var_x D2H + np.argmax): %.1f ms" % (dt * 1e3))
    e.close()
