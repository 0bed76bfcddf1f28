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
