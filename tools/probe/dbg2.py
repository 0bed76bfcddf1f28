import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from tests.helpers import make_problem, unpack, effective_cores
from pysvihmm_amd.engine import HipEngine
from pysvihmm_amd import _lib as L
from oracle import ref_c
def rel(got, ref, scale):
    e = np.abs(got - ref) / (np.abs(ref) + 1e-6 * scale)
    i = np.unravel_index(np.argmax(e), e.shape)
    return float(e.max()), i, float(np.asarray(ref)[i]), float(np.asarray(got)[i])
for (K, D, B, Lm, off, sep) in [(64, 32, 140, 257, 0.0, 0.25), (48, 24, 300, 129, 7.0, 0.4), (64, 32, 160, 257, 0.0, 4.0)]:
    T = max(6000, B * 3 + Lm)
    pb = make_problem(K, D, T, seed=K * 5 + D + 1, miss=0.03, sep=sep)
    obs = pb["obs"] + off; mu = pb["mu"] + off
    starts = np.random.default_rng(B + 1).integers(0, T - Lm + 1, size=B)
    ref = ref_c.estep_minibatch(obs, pb["mask"], starts, Lm, pb["mod_init"], pb["ltran"], mu,
                                pb["sigma"], pb["kappa"], pb["nu"], flags=2, threads=effective_cores())
    A, xbar, neff, S, lb = unpack(ref, K, D)
    sc = B * Lm; xs = max(np.abs(obs).max(), 1.0)
    for var in (3, 0):
        e = HipEngine(0, dtype="f32"); e.set_variant(5, var)
        e.set_obs(obs, pb["mask"]); e.set_globals(pb["mod_init"], pb["ltran"])
        e.set_emission_niw(mu, pb["sigma"], pb["kappa"], pb["nu"])
        st = e.estep(starts, Lm, flags=L.TRANS_WRAP)
        print((K, D, B, Lm, sep), "variant", var, "A", rel(st.A_raw, A, sc), "neff", rel(st.neff, neff, sc)[0],
              "xbar", rel(st.xbar, xbar, sc * xs)[0], "S", rel(st.S, S, sc * xs ** 2)[0], "lb", abs(st.lb[0] - lb) / abs(lb))
        qerr = 0
        for b in (0, B // 2):
            x = obs[starts[b]:starts[b] + Lm]
            ll = ref_c.lliks_niw(x, mu, pb["sigma"], pb["kappa"], pb["nu"])
            q, _ = ref_c.posterior(ref_c.forward(ll, pb["mod_init"], pb["ltran"]), ref_c.backward(ll, pb["ltran"]))
            qerr = max(qerr, np.abs(e.read_rows("var_x", b * Lm, Lm) - q).max())
        print("   max|dq|", qerr)
        e.close()
