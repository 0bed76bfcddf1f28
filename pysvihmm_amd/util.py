"""Host-side helpers with the reference's names and semantics (``util.py``).

Only what sits on / next to the E-step path: NIW natural <-> moment conversions
(reference ``util.py:12-60``), the weighted sufficient statistics
(``util.py:73-83``, the CPU statement of what the device ``suffstats`` kernels
reduce), the hold-out masks ``gen_synthetic`` needs (``util.py:163-206``), state matching
(``util.py:236-277``, via SciPy's Hungarian solver instead of the vendored Munkres).
Plot helpers, ``KL_gaussian``, ``mvnrand`` and the Dirichlet one-liners are out of scope
 and not provided.
"""

import numpy as np


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


def make_mask(sts, miss=0., left=0):
    """Hold out a ``miss`` fraction of every state's observations at or after ``left``
    (the held-out rows of ``gen_synthetic``'s smoothing setups; reference ``util.py:163-191``).
    Per state label ``k = 0 .. n_labels-1`` in order: skipped when fewer than 10 of its rows lie
    right of ``left``; the target count is ``ceil(miss * rows of k in the WHOLE sequence)``,
    falling back to the fraction of the rows right of ``left`` when that is more than there
    are; one ``np.random.choice(..., replace=False)`` draw per state (the global legacy
    stream, so a seeded run reproduces the reference's masks)."""
    sts = np.asarray(sts)
    out = np.zeros(sts.shape[0], dtype=bool)
    if not miss > 0.:
        return out
    tail = sts[left:]
    for k in range(np.unique(tail).shape[0]):
        cand = np.flatnonzero(tail == k)
        if cand.size < 10:
            continue
        want = np.ceil(miss * np.count_nonzero(sts == k))
        if want > cand.size:
            want = np.ceil(miss * cand.size)
        out[left + np.random.choice(cand, size=int(want), replace=False)] = True
    return out


def make_mask_prediction(sts, miss=0.):
    """Hold out the last ``ceil(miss * T)`` rows (prediction setup, reference ``util.py:194-206``)."""
    T = len(sts)
    out = np.zeros(T, dtype=bool)
    if miss != 0.:
        out[T - int(np.ceil(miss * T)):] = True
    return out


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
