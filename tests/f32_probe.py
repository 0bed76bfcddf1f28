"""Round 5: fp32-mode E-step against the fp64 C oracle at shapes the mode did not reach before
(D > 32, K > 64): max relative deviation of the packed statistics (tolerance of the mode: 1e-3) and
the kernels' HIP-event times.  usage: python tests/f32_probe.py K D B Lm [variant=value ...]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import make_problem, unpack, effective_cores
from pysvihmm_amd.engine import HipEngine
from pysvihmm_amd import _lib as L
from oracle import ref_c

K, D, B, Lm = [int(a) for a in sys.argv[1:5]]
variants = [tuple(int(x) for x in a.split(":")) for a in sys.argv[5:]]
T = max(B * 3 + Lm, 6000)
pb = make_problem(K, D, T, seed=K + D, miss=0.05, sep=4.0)
starts = np.random.default_rng(B).integers(0, T - Lm + 1, size=B)
res = {}
for dt in ("f64", "f32"):
    e = HipEngine(0, dtype=dt)
    for k, v in variants:
        e.set_variant(k, v)
    e.set_obs(pb["obs"], pb["mask"]); e.set_globals(pb["mod_init"], pb["ltran"])
    e.set_emission_niw(pb["mu"], pb["sigma"], pb["kappa"], pb["nu"])
    st = e.estep(starts, Lm, flags=L.TRANS_WRAP)
    prec = e.precision()
    e.profile(True); e.profile_reset()
    for _ in range(5):
        e.estep(starts, Lm, flags=L.TRANS_WRAP, read=False)
    e.sync()
    pr = e.profile_read(); e.profile(False)
    res[dt] = st.buf.copy()
    print(dt, "precision", prec, {n: round(1e3 * m / c, 1) for n, (m, c) in pr.items()}, "us per launch", flush=True)
    e.close()
sc = B * Lm
def dev(a, b):
    A1, x1, n1, S1, l1 = unpack(a, K, D); A2, x2, n2, S2, l2 = unpack(b, K, D)
    xs = np.abs(pb["obs"]).max()
    out = []
    for u, v, s in ((A1, A2, sc), (n1, n2, sc), (x1, x2, sc * xs), (S1, S2, sc * xs * xs)):
        out.append(float((np.abs(u - v) / (np.abs(v) + 1e-6 * s)).max()))
    return out, abs(l1 - l2) / abs(l2)
print("f32 vs f64 device:", dev(res["f32"], res["f64"]))
if os.environ.get("ORACLE", "1") == "1":
    t0 = time.time()
    ref = ref_c.estep_minibatch(pb["obs"], pb["mask"], starts, Lm, pb["mod_init"], pb["ltran"], pb["mu"], pb["sigma"],
                                pb["kappa"], pb["nu"], flags=2, threads=effective_cores())
    print("oracle %.1f s" % (time.time() - t0))
    print("f64 vs oracle:", dev(res["f64"], ref))
    print("f32 vs oracle:", dev(res["f32"], ref))
