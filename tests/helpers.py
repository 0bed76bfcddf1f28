"""Shared synthetic-problem helpers for the tests (host-side, NumPy only)."""
import numpy as np
from scipy.special import digamma

EPS = 1e-9


def make_problem(K, D, T, seed, miss=0.0, sep=3.0):
    rng = np.random.default_rng(seed)
    tran = 0.9 * np.eye(K) + 0.1 / max(K - 1, 1) * (1 - np.eye(K))
    tran /= tran.sum(1)[:, None]
    means = rng.normal(0, sep, size=(K, D))
    # state path with geometric dwell times (vectorised generator)
    sts = np.empty(T, dtype=np.int64)
    t = 0
    cur = 0
    while t < T:
        dwell = rng.geometric(0.1)
        sts[t:t + dwell] = cur
        t += dwell
        cur = (cur + 1 + rng.integers(0, max(K - 1, 1))) % K
    obs = means[sts] + rng.normal(size=(T, D))
    mask = rng.random(T) < miss
    var_tran = 1.0 + rng.random((K, K)) * T / K
    var_init = rng.random(K) + 0.1
    mod_init = digamma(var_init + EPS) - digamma(var_init.sum() + EPS)
    ltran = digamma(var_tran + EPS) - digamma(var_tran.sum(1)[:, None] + EPS)
    mu = means + rng.normal(size=(K, D))
    A = rng.normal(size=(K, D, D))
    sigma = 0.3 * np.einsum('kij,klj->kil', A, A) + (D + 2.0) * np.eye(D)
    kappa = 0.5 + rng.random(K)
    nu = D + 2 + 3 * rng.random(K)
    return dict(obs=obs, mask=mask, sts=sts, var_tran=var_tran, var_init=var_init,
                mod_init=mod_init, ltran=ltran, mu=mu, sigma=sigma, kappa=kappa, nu=nu,
                K=K, D=D, T=T)


def unpack(buf, K, D):
    o = 0
    A = buf[o:o + K * K].reshape(K, K); o += K * K
    xbar = buf[o:o + K * D].reshape(K, D); o += K * D
    neff = buf[o:o + K]; o += K
    S = buf[o:o + K * D * D].reshape(K, D, D); o += K * D * D
    return A, xbar, neff, S, buf[o]


def relerr(a, b, floor=1e-12):
    a = np.asarray(a, dtype=float); b = np.asarray(b, dtype=float)
    return float(np.max(np.abs(a - b) / (floor + np.abs(b)))) if a.size else 0.0


def ffbs_draws_exact(z, lalpha, logA, u, tol=1e-12):
    """Row-by-row check of a backward-sampled path (hmm_fast.pyx:97-122): given the path's own
    z[t+1], z[t] must be the inverse-CDF draw (rand_discrete, :29-36: first k with u <= cumsum)
    of softmax(lalpha[t] + logA[:, z[t+1]]) at u[t].  Index output: EXACT on every row whose
    uniform is further than ``tol`` from every CDF step (a row closer than that may
    legitimately round either way).  Returns (rows that disagree outside tol, rows within tol)."""
    z = np.asarray(z)
    T, K = lalpha.shape
    lp = np.array(lalpha, dtype=np.float64)
    lp[:-1] += logA[:, z[1:]].T
    p = np.exp(lp - lp.max(axis=1, keepdims=True))
    p /= p.sum(axis=1, keepdims=True)
    c = np.cumsum(p, axis=1)
    want = np.minimum((c < u[:, None]).sum(axis=1), K - 1)
    margin = np.min(np.abs(c[:, :-1] - u[:, None]), axis=1) if K > 1 else np.ones(T)
    risky = margin <= tol
    return int(np.sum((want != z) & ~risky)), int(risky.sum())


def effective_cores():
    """Cores the process may really use (cgroup quota or affinity), for the oracle's thread count."""
    import os
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        pass
    return n
