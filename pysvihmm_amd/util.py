"""Host-side helpers with the reference's names and semantics (``util.py``).

Only what sits on / next to the E-step path: NIW natural <-> moment conversions
(reference ``util.py:12-60``), the weighted sufficient statistics
(``util.py:73-83``, the CPU statement of what the device ``suffstats`` kernels
reduce), masks (``util.py:163-206``), state matching (``util.py:236-277``, via
SciPy's Hungarian solver instead of the vendored Munkres).  Plot helpers are out
of scope.
"""

import numpy as np
import numpy.linalg as npl


def _obj(*items):
    """Ragged 4-slot parameter vector; the reference relies on legacy implicit
    object arrays (quirk Q13) so ``(1-rho)*a + rho*b`` works slot-wise."""
    out = np.empty(len(items), dtype=object)
    for i, it in enumerate(items):
        out[i] = it
    return out


def NIW_zero_nat_pars(G):
    p = len(G.mu_mf)
    return _obj(np.zeros(p), 0., np.zeros((p, p)), 0)


def NIW_nat2moment_pars(e1, e2, e3, e4):
    p = len(e1)
    mu = e1 / e2
    kappa = e2
    sigma = e3 - np.outer(mu, mu) / kappa
    nu = e4 - 2 - p
    return _obj(mu, sigma, kappa, nu)


def NIW_mf_natural_pars(mu, sigma, kappa, nu):
    """Moment -> natural parameters, reference ``util.py:28-37``:
    ``[kappa*mu, kappa, sigma + kappa*mu mu', nu + 2 + p]``."""
    p = len(mu)
    eta3 = sigma + np.outer(mu, mu) * kappa
    return _obj(kappa * mu, kappa, eta3, nu + 2 + p)


def NIW_mf_moment_pars(G, e1, e2, e3, e4):
    """Natural -> moment parameters written back into ``G`` (``util.py:40-60``)."""
    p = len(e1)
    mu = e1 / e2
    kappa = e2
    sigma = e3 - np.outer(mu, mu) * kappa
    nu = e4 - 2 - p
    G.mu_mf = mu
    G.sigma_mf = sigma
    G.kappa_mf = kappa
    G.nu_mf = nu
    G.mu = G.mu_mf
    G.sigma = G.sigma_mf / (G.nu_mf - p - 1)


def NIW_meanfield(G, data, weights):
    D = len(G.mu_0)
    mu_mf, sigma_mf, kappa_mf, nu_mf = \
        G._posterior_hypparams(*G._get_weighted_statistics(data, weights, D))
    return _obj(mu_mf, sigma_mf, kappa_mf, nu_mf)


def NIW_suffstats(G, data, weights):
    """``[sum w x, sum w, sum w x x', sum w]`` (``util.py:73-83``)."""
    tmp = weights[:, np.newaxis] * data
    S = data.T.dot(tmp)
    xbar = np.sum(tmp, axis=0)
    neff = weights.sum()
    return _obj(xbar, neff, S, neff)


def KL_gaussian(mu0, sig0, mu1, sig1):
    D = len(mu0)
    if D != len(mu1) or D != sig0.shape[0] or D != sig1.shape[0]:
        raise RuntimeError("Means and covariances my be the same dimension.")
    if sig0.shape[0] != sig0.shape[1] or sig1.shape[0] != sig1.shape[1]:
        raise RuntimeError("Covariance matrices must be square.")
    s1inv = npl.inv(sig1)
    s0_ld = npl.slogdet(sig0)[1]
    s1_ld = npl.slogdet(sig1)[1]
    x = mu1 - mu0
    tmp = np.trace(np.dot(s1inv, sig0)) + np.dot(x.T, np.dot(s1inv, x))
    tmp += -D - s0_ld + s1_ld
    return 0.5 * tmp


def dirichlet_natural_pars(alpha):
    return alpha - 1.


def dirichlet_moment_pars(eta):
    return eta + 1.


def mvnrand(mean, cov, size=1):
    mu = np.squeeze(mean)
    D = mu.shape[0]
    C = npl.cholesky(cov)
    z = np.random.randn(size, D)
    return np.squeeze(mu + np.dot(z, C.T))


def make_mask(sts, miss=0., left=0):
    """Mark a ``miss`` fraction of the observations right of ``left`` as
    missing, evenly over states (``util.py:163-191``)."""
    sts = np.asarray(sts)
    sts_l = sts[left:]
    K = np.unique(sts_l).shape[0]
    mask = np.zeros(len(sts), dtype='bool')
    if miss > 0.:
        for k in range(K):
            obs_k = np.where(sts_l == k)[0]
            if obs_k.shape[0] < 10:
                continue
            nobs_k = np.ceil(miss * np.sum(sts == k))
            if obs_k.shape[0] < nobs_k:
                nobs_k = np.ceil(miss * obs_k.shape[0])
            nobs_k = int(nobs_k)
            inds = np.random.choice(obs_k, size=nobs_k, replace=False)
            mask[left + inds] = True
    return mask


def make_mask_prediction(sts, miss=0.):
    nobs = len(sts)
    mask = np.zeros(nobs, dtype='bool')
    if miss == 0.:
        return mask
    nmiss = int(np.ceil(miss * nobs))
    mask[-nmiss:] = True
    return mask


def munkres_match(sts_true, sts_pred, K):
    """Permutation of predicted labels minimising the Hamming distance
    (``util.py:236-277``); solved with ``scipy.optimize.linear_sum_assignment``."""
    from scipy.optimize import linear_sum_assignment
    sts_true = np.asarray(sts_true).astype('int')
    sts_pred = np.asarray(sts_pred).astype('int')
    DM = np.zeros((K, K))
    np.add.at(DM, (sts_pred, sts_true), 1.)
    return match_from_counts(DM)


def match_from_counts(DM):
    """The assignment step of ``munkres_match`` on a ready count matrix ``DM[pred, true]``
    (``util.py:262-277``; the device builds the counts, ``svihmm_state_argmax``)."""
    from scipy.optimize import linear_sum_assignment
    DM = np.asarray(DM, dtype=np.float64)
    cost_mat = 1 - (DM / np.sum(DM))
    rows, cols = linear_sum_assignment(cost_mat)
    out = np.empty(DM.shape[0], dtype=int)
    out[rows] = cols
    return out


match_state_seq = munkres_match
