"""Wall time of hmmsgd_metaobs.VBHMM.infer against the sum of the device's per-iteration times, for several
loop lengths: is bench.py's infer(70) - infer(10) difference a fair per-iteration figure?"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pysvihmm_amd.engine import HipEngine
from pysvihmm_amd import hmmsgd_metaobs
from pysvihmm_amd.distributions import Gaussian

eng = HipEngine(0)
rs, tran, means, chols = bench.true_process(0)
eng.generate(tran, means, chols, bench.T, seed=bench.SEED)
obs = eng.read_generated(want_sts=False)[0]
if "f32" in sys.argv[1:]:
    eng.set_precision("f32")
K, D = bench.K, bench.D
head = obs[:20000]
np.random.seed(0)
prior = np.array([Gaussian(mu_0=head.mean(0), sigma_0=0.75 * np.cov(head.T), kappa_0=0.01, nu_0=D + 2) for _ in range(K)])
def run(maxit):
    hmm = hmmsgd_metaobs.VBHMM(obs, np.ones(K), np.ones((K, K)), prior, tau=1.0, kappa=0.7,
                               metaobs_half=bench.LHALF, mb_sz=64, maxit=maxit, seed=1, engine=eng)
    t0 = time.perf_counter()
    hmm.infer()
    return time.perf_counter() - t0, hmm
run(5)
for maxit in (10, 10, 70, 70, 130, 130, 250, 250):
    w, hmm = run(maxit)
    it = hmm.iter_time * 1e3
    print("maxit %3d: wall %.3f ms, sum(iter_time) %.3f ms, iter_time first 3 %s, median %.4f, last %.4f" % (
        maxit, w * 1e3, it.sum(), np.round(it[:3], 3), np.median(it), it[-1]), flush=True)
