"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported by the product package.

NumPy restatement of the reference's SVI-HMM E-step hot path (dillonalaird/
pysvihmm), as pure functions with explicit arguments.  Every function cites the
reference ``file:line`` it follows and keeps the reference's exact NumPy
expressions (``np.logaddexp.reduce`` folds, no global max-shift) *including the
quirks* of SURVEY.md Appendix D (Q1 wrap-around product-of-marginals transition
statistic, Q2 prior added per window, Q4 "log Z" summed over all t, Q5
un-normalised eigenvector as ``var_init``).

Pinning: checked to <=1e-12 against golden vectors produced by executing the
reference's own modules in this container (``tests/golden/make_golden.py``,
fixtures ``tests/golden/*.npz``; test ``tests/test_oracle_golden.py``).  The one
piece that is NOT pinned is the emission expected log-likelihood
(``niw_expected_log_likelihood``): its arithmetic lives in the third-party
package ``pybasicbayes`` (mattjj/pybasicbayes, pinned version unrecoverable --
``.gitmodules:1-3`` has only a URL and the submodule directory is empty), so it
restates that package's published algorithm (Bishop PRML 10.64-10.71) --
"parity unpinned" for that function only.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.
"""

import numpy as np
from scipy.special import digamma, gammaln
import scipy.linalg as sla

eps = 1e-9  # reference hmmbase.py:30, hmmsgd_metaobs.py:26


# --- a1: psi-expectations -------------------------------------------------- #
def psi_expectations(var_init, var_tran):
    """reference hmmbase.py:214-216 == hmmsgd_metaobs.py:502-504."""
    mod_init = digamma(var_init + eps) - digamma(np.sum(var_init) + eps)
    tran_sum = np.sum(var_tran, axis=1)
    mod_tran = digamma(var_tran + eps) - digamma(tran_sum[:, None] + eps)
    return mod_init, mod_tran


# --- a2: stationary initialisation ------------------------------------------ #
def stationary_init(var_tran):
    """reference hmmsgd_metaobs.py:413-418 (unit-L2 eigenvector, abs; Q5)."""
    A_mean = var_tran / np.sum(var_tran, axis=1)[:, None]
    ew, ev = np.linalg.eig(A_mean.T)
    ew_dec = np.argsort(ew)[::-1]
    return np.abs(ev[:, ew_dec[0]])


# --- a3: emission expected log-likelihood (third-party arithmetic) ---------- #
def niw_expected_log_likelihood(x, mu, sigma, kappa, nu):
    """pybasicbayes ``Gaussian.expected_log_likelihood`` (published algorithm,
    parity unpinned; call sites hmmbase.py:219-220, hmmsgd_metaobs.py:508-509).
    NaN rows give NaN (callers nan_to_num)."""
    D = len(mu)
    xc = np.reshape(x, (-1, D)) - mu
    chol = np.linalg.cholesky(sigma)
    xs = sla.solve_triangular(chol, xc.T, lower=True, check_finite=False)
    llt = (digamma((nu - np.arange(D)) / 2.).sum() + D * np.log(2.)
           - 2. * np.log(chol.diagonal()).sum())
    return (llt / 2. - D / (2. * kappa)
            - nu / 2. * np.einsum('ij,ij->j', xs, xs)
            - D / 2. * np.log(2. * np.pi))


def lliks_niw(obs_rows, mu, sigma, kappa, nu):
    """lliks[:,k] = nan_to_num(E log p) -- hmmbase.py:219-220."""
    K = len(kappa)
    out = np.empty((obs_rows.shape[0], K))
    for k in range(K):
        out[:, k] = np.nan_to_num(niw_expected_log_likelihood(
            obs_rows, mu[k], sigma[k], kappa[k], nu[k]))
    return out


def lliks_diag(obs_rows, mu, nus, alphas, betas):
    """lliks[:,k] = nan_to_num(E_q log N(x | mu_k, diag sigma_k^2)) for diagonal Gaussian factors
    (per dimension a normal-inverse-gamma mean-field factor; pybasicbayes DiagonalGaussian, evaluated
    where the reference evaluates any emitter: hmmbase.py:219-220).  Centred form:
        sum_d [ -1/2 (alpha/beta) (x - m)^2 - 1/(2 nu) - 1/2 (log beta - psi(alpha)) ] - D/2 log 2 pi."""
    from scipy.special import digamma
    K, D = mu.shape
    out = np.empty((obs_rows.shape[0], K))
    for k in range(K):
        d = obs_rows - mu[k]
        v = ((-0.5 * (alphas[k] / betas[k]) * d ** 2).sum(1)
             + (-0.5 / nus[k] - 0.5 * (np.log(betas[k]) - digamma(alphas[k]))).sum() - 0.5 * D * np.log(2. * np.pi))
        out[:, k] = np.nan_to_num(v)
    return out


def diag_suffstats(obs_rows, w):
    """(sum_t w x, sum_t w, sum_t w x^2): the diagonal family's expected sufficient statistics (the
    diagonal of util.NIW_suffstats' second moment, util.py:73-83)."""
    return w.dot(obs_rows), w.sum(), w.dot(obs_rows ** 2)


# --- a4 / a5: messages ------------------------------------------------------ #
def forward_msgs(ll, mod_init, ltran):
    """reference hmmbase.py:292-295 == hmmsgd_metaobs.py:800-803."""
    T, K = ll.shape
    lalpha = np.empty((T, K))
    lalpha[0, :] = mod_init + ll[0, :]
    for t in range(1, T):
        lalpha[t] = np.logaddexp.reduce(lalpha[t - 1] + ltran.T, axis=1) + ll[t]
    return lalpha


def backward_msgs(ll, ltran):
    """reference hmmbase.py:316-320 == hmmsgd_metaobs.py:851-855."""
    T, K = ll.shape
    lbeta = np.empty((T, K))
    lbeta[T - 1, :] = 0.
    for t in range(T - 2, -1, -1):
        np.logaddexp.reduce(ltran + lbeta[t + 1] + ll[t + 1], axis=1,
                            out=lbeta[t])
    return lbeta


# --- a6: posterior marginals ------------------------------------------------ #
def posterior(lalpha, lbeta):
    """reference hmmbase.py:226-229 == hmmsgd_metaobs.py:516-519."""
    var_x = lalpha + lbeta
    var_x -= np.max(var_x, axis=1)[:, None]
    var_x = np.exp(var_x)
    var_x /= np.sum(var_x, axis=1)[:, None]
    return var_x


# --- a7: local ELBO term ---------------------------------------------------- #
def local_lower_bound(lalpha):
    """reference hmmsgd_metaobs.py:271 / hmmbase.py:194 (sum over ALL t; Q4)."""
    return np.sum(np.logaddexp.reduce(lalpha, axis=1))


# --- a8: expected sufficient statistics ------------------------------------- #
def transition_stat_wrap(var_x):
    """sum_{t=0}^{Lm-1} var_x[t-1] (x) var_x[t], index -1 wraps (Q1);
    reference hmmsgd_metaobs.py:876-878 without the prior."""
    K = var_x.shape[1]
    acc = np.zeros((K, K))
    for t in range(var_x.shape[0]):
        acc += np.outer(var_x[t - 1, :], var_x[t, :])
    return acc


def transition_stat_batch(var_x):
    """sum_{t=1}^{T-1} var_x[t-1] (x) var_x[t]; hmmbatchcd.py:183-184,
    hmmbatchsgd.py:224-225 (no wrap)."""
    K = var_x.shape[1]
    acc = np.zeros((K, K))
    for t in range(1, var_x.shape[0]):
        acc += np.outer(var_x[t - 1, :], var_x[t, :])
    return acc


def niw_suffstats(data, weights):
    """reference util.py:73-83 -> (xbar, neff, S)."""
    tmp = weights[:, None] * data
    S = data.T.dot(tmp)
    xbar = np.sum(tmp, axis=0)
    neff = weights.sum()
    return xbar, neff, S


def intermediate_pars(var_x, obs_win, mask_win, prior_tran):
    """reference hmmsgd_metaobs.py:857-904 (Gaussian branch).
    Returns A_i[K,K], xbar[K,D], neff[K], S[K,D,D]."""
    K = var_x.shape[1]
    D = obs_win.shape[1]
    tran_mf = prior_tran.copy()
    for t in range(var_x.shape[0]):
        tran_mf += np.outer(var_x[t - 1, :], var_x[t, :])
    A_i = tran_mf - 1.
    inds = np.logical_not(mask_win)
    xbar = np.empty((K, D)); neff = np.empty(K); S = np.empty((K, D, D))
    for k in range(K):
        xbar[k], neff[k], S[k] = niw_suffstats(obs_win[inds, :], var_x[inds, k])
    return A_i, xbar, neff, S


# --- one window / one minibatch (a3..a9) ------------------------------------ #
def estep_window(ll, obs_win, mask_win, mod_init, ltran, prior_tran):
    """hmmsgd_metaobs.py:487-519 + 857-904 + 271 for one meta-observation,
    starting from its ``lliks``."""
    lalpha = forward_msgs(ll, mod_init, ltran)
    lbeta = backward_msgs(ll, ltran)
    var_x = posterior(lalpha, lbeta)
    A_i, xbar, neff, S = intermediate_pars(var_x, obs_win, mask_win, prior_tran)
    return dict(lalpha=lalpha, lbeta=lbeta, var_x=var_x, A_i=A_i, xbar=xbar,
                neff=neff, S=S, lb=local_lower_bound(lalpha))


def estep_minibatch(obs, mask, starts, Lm, mod_init, ltran, prior_tran,
                    mu, sigma, kappa, nu):
    """The minibatch loop hmmsgd_metaobs.py:405-436 (serial accumulation)."""
    K = ltran.shape[0]; D = obs.shape[1]
    A = np.zeros((K, K)); xbar = np.zeros((K, D)); neff = np.zeros(K)
    S = np.zeros((K, D, D)); lb = 0.
    for s in starts:
        ow = obs[s:s + Lm]
        ll = lliks_niw(ow, mu, sigma, kappa, nu)
        r = estep_window(ll, ow, mask[s:s + Lm], mod_init, ltran, prior_tran)
        A += r['A_i']; xbar += r['xbar']; neff += r['neff']; S += r['S']
        lb += r['lb']
    return dict(A_inter=A, xbar=xbar, neff=neff, S=S, lb=lb)


# --- a10: global natural-gradient step -------------------------------------- #
def niw_nat(mu, sigma, kappa, nu):
    """reference util.py:28-37."""
    p = len(mu)
    return [kappa * mu, kappa, sigma + np.outer(mu, mu) * kappa, nu + 2 + p]


def niw_moment(e1, e2, e3, e4):
    """reference util.py:40-60 -> (mu, sigma, kappa, nu)."""
    p = len(e1)
    mu = e1 / e2
    kappa = e2
    sigma = e3 - np.outer(mu, mu) * kappa
    nu = e4 - 2 - p
    return mu, sigma, kappa, nu


def global_update_metaobs(var_tran, A_inter, mf, prior, emit_inter, lrate,
                          T, L, S):
    """reference hmmsgd_metaobs.py:1010-1069 (no adagrad).  ``mf``/``prior`` are
    lists of (mu, sigma, kappa, nu); ``emit_inter[k] = (xbar, neff, S)``."""
    nats_old = var_tran - 1.
    bfact = (T - 2 * L - 1) / (2. * L * S)
    nats_new = (1. - lrate) * nats_old + lrate * (bfact * A_inter)
    var_tran_new = nats_new + 1.
    bfact = (T - 2 * L - 1) / ((2. * L + 1.) * S)
    out = []
    for k in range(len(mf)):
        n_old = niw_nat(*mf[k])
        n_0 = niw_nat(*prior[k])
        xbar, neff, Sk = emit_inter[k]
        e = [xbar, neff, Sk, neff]
        n_new = [(1. - lrate) * n_old[i] + lrate * (n_0[i] + bfact * e[i])
                 for i in range(4)]
        out.append(niw_moment(*n_new))
    return var_tran_new, out


def dirichlet_lower_bound(prior, var):
    """A_energy + A_entropy of reference hmmsgd_metaobs.py:277-292 (row-wise Dirichlet factors)."""
    eps = 1e-9
    p_sum = np.sum(prior, axis=1)
    q_dg = digamma(var + eps)
    q_sum = np.sum(var, axis=1)
    dg_q_sum = digamma(q_sum + eps)
    energy = (gammaln(p_sum + eps) - np.sum(gammaln(prior + eps), axis=1)
              + np.sum((prior - 1) * (q_dg - dg_q_sum[:, None]), axis=1))
    entropy = -(gammaln(q_sum + eps) - np.sum(gammaln(var + eps), axis=1)
                + np.sum((var - 1) * (q_dg - dg_q_sum[:, None]), axis=1))
    return np.sum(energy) + np.sum(entropy)


# --- a12: FFBS (Cython variant) --------------------------------------------- #
DBL_EPSILON = np.finfo(np.float64).eps


def ffbs_forward(ll, var_init, var_tran):
    """reference hmm_fast.pyx:74-93: mod_init with DBL_EPSILON, transitions
    ``log(var_tran + DBL_EPSILON)`` (un-normalised; quirk Q6)."""
    mod_init = (digamma(var_init + DBL_EPSILON)
                - digamma(np.sum(var_init) + DBL_EPSILON))
    T, K = ll.shape
    lalpha = np.empty((T, K))
    lalpha[0] = mod_init + ll[0]
    lA = np.log(var_tran + DBL_EPSILON)
    for t in range(1, T):
        lalpha[t] = np.logaddexp.reduce(lalpha[t - 1] + lA.T, axis=1) + ll[t]
    return lalpha


def rand_discrete(p, r):
    """reference hmm_fast.pyx:29-36 with the uniform ``r`` supplied."""
    rsum = 0.
    for i in range(len(p)):
        rsum += p[i]
        if r <= rsum:
            return i
    return len(p) - 1  # the C code falls off the end (UB); clamp


def ffbs_backward_sample(lalpha, var_tran, uniforms):
    """reference hmm_fast.pyx:97-122 with a recorded uniform stream
    (``uniforms[t]`` is consumed for ``z[t]``)."""
    T, K = lalpha.shape
    z = np.empty(T, dtype=np.int64)
    lp = lalpha[T - 1]
    p = np.exp(lp - lp.max()); p /= p.sum()
    z[T - 1] = rand_discrete(p, uniforms[T - 1])
    for t in range(T - 2, -1, -1):
        lp = lalpha[t] + np.log(var_tran[:, z[t + 1]] + DBL_EPSILON)
        p = np.exp(lp - lp.max()); p /= p.sum()
        z[t] = rand_discrete(p, uniforms[t])
    return z


# ---- device generator restated (pysvihmm_amd/csrc/kernels_misc.h k_gen_*): Philox4x32-10 ----
def philox4x32_10(seed, rows, stream):
    """Four uint32 words per row for counter (row_lo, row_hi, stream, 0), key = seed."""
    M32 = np.uint64(0xFFFFFFFF)
    rows = np.asarray(rows, dtype=np.uint64)
    c0 = rows & M32
    c1 = rows >> np.uint64(32)
    c2 = np.full(rows.shape, stream, dtype=np.uint64)
    c3 = np.zeros(rows.shape, dtype=np.uint64)
    k0 = np.uint64(seed & 0xFFFFFFFF)
    k1 = np.uint64((seed >> 32) & 0xFFFFFFFF)
    for _ in range(10):
        p0 = np.uint64(0xD2511F53) * c0
        p1 = np.uint64(0xCD9E8D57) * c2
        n0 = ((p1 >> np.uint64(32)) ^ c1 ^ k0) & M32
        n1 = p1 & M32
        n2 = ((p0 >> np.uint64(32)) ^ c3 ^ k1) & M32
        n3 = p0 & M32
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 = (k0 + np.uint64(0x9E3779B9)) & M32
        k1 = (k1 + np.uint64(0xBB67AE85)) & M32
    return c0, c1, c2, c3


def _u53(hi, lo):
    return ((hi >> np.uint64(5)).astype(np.float64) * 67108864.0
            + (lo >> np.uint64(6)).astype(np.float64)) / 9007199254740992.0


def generate_counter_based(tran, means, chols, T, seed):
    """gen_synthetic.py:27-44 semantics (start in state 0, np.random.choice's inverse CDF,
    mean + chol n) on the device generator's random stream: states exactly, obs to rounding."""
    tran = np.asarray(tran, float)
    K = tran.shape[0]
    D = means.shape[1]
    cdf = np.cumsum(tran, axis=1)
    cdf = cdf / cdf[:, -1:]
    rows = np.arange(T, dtype=np.uint64)
    w = philox4x32_10(seed, rows, 0)
    u = _u53(w[0], w[1])
    z = np.zeros(T, dtype=np.int32)
    for t in range(1, T):
        z[t] = min(int(np.searchsorted(cdf[z[t - 1]], u[t], side='right')), K - 1)
    n = np.empty((T, D + 1))
    for p in range((D + 1) // 2):
        w = philox4x32_10(seed, rows, 1 + p)
        u1, u2 = _u53(w[0], w[1]), _u53(w[2], w[3])
        r = np.sqrt(-2.0 * np.log(1.0 - u1))
        n[:, 2 * p] = r * np.cos(2 * np.pi * u2)
        n[:, 2 * p + 1] = r * np.sin(2 * np.pi * u2)
    obs = means[z] + np.einsum('tij,tj->ti', np.tril(chols)[z], n[:, :D])
    return obs, z
