import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from tests.helpers import make_problem, unpack, effective_cores
from pysvihmm_amd.engine import HipEngine
from pysvihmm_amd import _lib as L
from oracle import ref_c
K, D, B, Lm = 64, 32, 160, 257
T = 6000
pb = make_problem(K, D, T, seed=K * 5 + D + 1, miss=0.03, sep=4.0)
starts = np.random.default_rng(B + 1).integers(0, T - Lm + 1, size=B)
for var in (3, 0):
    for dt in ("f64", "f32"):
        e = HipEngine(0, dtype=dt)
        e.set_variant(5, var)
        e.set_obs(pb["obs"], pb["mask"]); e.set_globals(pb["mod_init"], pb["ltran"])
        e.set_emission_niw(pb["mu"], pb["sigma"], pb["kappa"], pb["nu"])
        st = e.estep(starts, Lm, flags=L.TRANS_WRAP)
        for b in (0, 80, 159):
            x = pb["obs"][starts[b]:starts[b] + Lm]
            ll = ref_c.lliks_niw(x, pb["mu"], pb["sigma"], pb["kappa"], pb["nu"])
            q, _ = ref_c.posterior(ref_c.forward(ll, pb["mod_init"], pb["ltran"]), ref_c.backward(ll, pb["ltran"]))
            got = e.read_rows("var_x", b * Lm, Lm)
            print(var, dt, b, "max|dq|", np.abs(got - q).max(), "rowsum", got.sum(1).min(), got.sum(1).max(), e.precision())
        e.close()
