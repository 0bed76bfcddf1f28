import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bench
from pysvihmm_amd.engine import HipEngine
from pysvihmm_amd import hmmsgd_metaobs
from pysvihmm_amd.distributions import Gaussian
eng = HipEngine(0)
rs, tran, means, chols = bench.true_process(0)
eng.generate(tran, means, chols, bench.T, seed=bench.SEED)
obs = eng.read_generated(want_sts=False)[0]
eng.set_precision("f32")
K, D = bench.K, bench.D
head = obs[:20000]
np.random.seed(0)
prior = np.array([Gaussian(mu_0=head.mean(0), sigma_0=0.75 * np.cov(head.T), kappa_0=0.01, nu_0=D + 2) for _ in range(K)])
hmm = hmmsgd_metaobs.VBHMM(obs, np.ones(K), np.ones((K, K)), prior, tau=1.0, kappa=0.7, metaobs_half=bench.LHALF, mb_sz=64, maxit=5, seed=1, engine=eng)
print("device ok", hmm._svi_device_ok(), "var_tran min", hmm.var_tran.min())
hmm.infer()
print("precision after infer", eng.precision())
