"""Is the 64-window SVI iteration host- or device-bound?  Per-iteration wall time of
hmmsgd_metaobs.VBHMM.infer with (a) the class's own minibatch sampler, (b) a sampler that costs
nothing (pre-drawn windows), (c) the own sampler plus a busy-wait of 50 / 100 us per iteration."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _workload import bench_problem  # noqa: E402
import bench  # noqa: E402
from pysvihmm_amd import hmmsgd_metaobs  # noqa: E402
from pysvihmm_amd.distributions import Gaussian  # noqa: E402
from pysvihmm_amd.engine import HipEngine  # noqa: E402

K, D = bench.K, bench.D
eng = HipEngine(0)
pb = bench_problem(eng, want_obs=True)
obs = pb["obs"]
head = obs[:20000]
np.random.seed(0)
prior = np.array([Gaussian(mu_0=head.mean(0), sigma_0=0.75 * np.cov(head.T), kappa_0=0.01, nu_0=D + 2)
                  for _ in range(K)])


def per_iteration(patch=None):
    def run(maxit):
        hmm = hmmsgd_metaobs.VBHMM(obs, np.ones(K), np.ones((K, K)), prior, tau=1.0, kappa=0.7,
                                   metaobs_half=bench.LHALF, mb_sz=64, maxit=maxit, seed=1, engine=eng)
        if patch:
            patch(hmm)
        t0 = time.perf_counter()
        hmm.infer()
        return time.perf_counter() - t0
    run(5)
    t1 = min(run(10) for _ in range(2))
    t2 = min(run(110) for _ in range(3))
    return (t2 - t1) / 100 * 1e6


def free_sampler(hmm):
    pool = [hmm.metaobs_unif(hmm.T, bench.LHALF, 64) for _ in range(8)]
    state = {"i": 0}

    def f(N, L_, n):
        state["i"] += 1
        return pool[state["i"] % 8]
    hmm.metaobs_fun = f


def delayed(us):
    def patch(hmm):
        own = hmm.metaobs_fun

        def f(N, L_, n):
            t0 = time.perf_counter()
            while (time.perf_counter() - t0) * 1e6 < us:
                pass
            return own(N, L_, n)
        hmm.metaobs_fun = f
    return patch


print("own sampler            : %.1f us / iteration" % per_iteration())
print("pre-drawn windows      : %.1f us / iteration" % per_iteration(free_sampler))
print("own sampler + 50 us    : %.1f us / iteration" % per_iteration(delayed(50)))
print("own sampler + 100 us   : %.1f us / iteration" % per_iteration(delayed(100)))
