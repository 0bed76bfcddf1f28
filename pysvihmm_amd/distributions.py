"""Emission-distribution plugins (the pybasicbayes duck type the reference uses).

The reference imports ``Gaussian`` / ``Categorical`` from the un-vendored
``pybasicbayes`` submodule (reference ``.gitmodules:1-3``; the directory is empty
in the reference tree, no commit pin is recoverable).  These classes restate the
published algorithm of that package (mattjj/pybasicbayes ``distributions.py``,
NIW mean-field per Bishop PRML eqs. 10.59-10.65, 10.71, 10.74, 10.77) and expose
exactly the members the reference touches:

  ``Gaussian(mu=, sigma=, mu_0=, sigma_0=, kappa_0=, nu_0=)``
        reference ``test_hmmsgd_metaobs.py:22-33,45-46``
  ``.expected_log_likelihood(x)``   ``hmmbase.py:220``, ``hmmsgd_metaobs.py:509,686,816,1176``
  ``.meanfieldupdate(data, w)``     ``hmmbatchcd.py:189``
  ``._get_weighted_statistics`` / ``._posterior_hypparams``   ``util.py:69``
  ``.get_vlb()``                    ``hmmbase.py:185``, ``hmmsgd_metaobs.py:294``
  ``.rvs(size)``                    ``gen_synthetic.py:32,40``
  ``.mu_mf .sigma_mf .kappa_mf .nu_mf .mu_0 .sigma_0 .kappa_0 .nu_0 .mu .sigma``
        read and written by ``util.NIW_mf_moment_pars`` (``util.py:40-60``)
  ``DiagonalGaussian(mu=, sigmas=, mu_0=, nus_0=, alphas_0=, betas_0=)``
        the package's diagonal-covariance family (BASELINE configs[0]); same protocol, a
        normal-inverse-gamma factor per dimension

Parity status: *unpinned* -- no reference test or fixture pins this arithmetic
and the upstream source is absent; the device emission kernel is checked against
this class (and the C oracle), everything downstream of ``lliks`` is checked
against the reference's own code (tests/golden).
"""

import numpy as np
import scipy.linalg as sla
from scipy.special import digamma, gammaln

__all__ = ["Gaussian", "DiagonalGaussian", "Categorical", "sample_niw", "sample_invwishart",
           "niw_quadratic_form", "niw_vlb_batch", "niw_prior_logpart", "vlb_logz_sign"]


# --------------------------------------------------------------------------- #
#  samplers (pybasicbayes.util.stats)                                         #
# --------------------------------------------------------------------------- #
def sample_invwishart(lmbda, dof):
    """Draw from an inverse-Wishart IW(lmbda, dof) (Bartlett decomposition).

    Used by the reference harness only (``cluster/exper_run_simple.py:12,121``).
    """
    lmbda = np.asarray(lmbda, dtype=np.float64)
    n = lmbda.shape[0]
    chol = np.linalg.cholesky(lmbda)
    if (dof <= 81 + n) and (dof == np.round(dof)):
        x = np.random.randn(int(dof), n)
    else:
        x = np.diag(np.sqrt(np.atleast_1d(
            np.random.chisquare(dof - np.arange(n)))))
        x[np.triu_indices_from(x, 1)] = np.random.randn(n * (n - 1) // 2)
    R = np.linalg.qr(x, 'r')
    T = sla.solve_triangular(R.T, chol.T, lower=True).T
    return T.dot(T.T)


def sample_niw(mu, lmbda, kappa, nu):
    """Draw (mu, Sigma) from a normal-inverse-Wishart."""
    lmbda = sample_invwishart(lmbda, nu)
    mu = np.random.multivariate_normal(mu, lmbda / kappa)
    return mu, lmbda


# --------------------------------------------------------------------------- #
#  NIW mean-field Gaussian                                                    #
# --------------------------------------------------------------------------- #
def _loglmbdatilde(sigma_mf, nu_mf):
    """E_q[log |Lambda|] for Lambda ~ Wishart, Bishop eq. 10.65."""
    D = sigma_mf.shape[0]
    chol = np.linalg.cholesky(sigma_mf)
    return (digamma((nu_mf - np.arange(D)) / 2.).sum() + D * np.log(2.)
            - 2. * np.log(chol.diagonal()).sum())


def niw_quadratic_form(mu_mf, sigma_mf, kappa_mf, nu_mf):
    """Canonical quadratic form of the NIW expected log-likelihood.

    Returns ``(W[D,D], v[D], c)`` with
    ``E_q[log N(x)] = c + v.x - x' W x``  where ``W = (nu/2) sigma_mf^-1``,
    ``v = 2 W mu``, ``c = const - mu' W mu``.  This is the parameter block the
    device emission kernel consumes ("emission feature GEMM").
    """
    D = len(mu_mf)
    W = 0.5 * nu_mf * np.linalg.inv(sigma_mf)
    W = 0.5 * (W + W.T)
    v = 2. * W.dot(mu_mf)
    const = (0.5 * _loglmbdatilde(sigma_mf, nu_mf) - D / (2. * kappa_mf)
             - 0.5 * D * np.log(2. * np.pi))
    c = const - mu_mf.dot(W).dot(mu_mf)
    return W, v, c


class Gaussian(object):
    """Multivariate Gaussian with a normal-inverse-Wishart prior and an NIW
    mean-field factor ``(mu_mf, sigma_mf, kappa_mf, nu_mf)``.
    """

    def __init__(self, mu=None, sigma=None, mu_0=None, sigma_0=None,
                 kappa_0=None, nu_0=None, kappa_mf=None, nu_mf=None):
        f = lambda a: None if a is None else np.array(a, dtype=np.float64)
        self.mu = f(mu)
        self.sigma = f(sigma)
        self.mu_0 = f(mu_0)
        self.sigma_0 = f(sigma_0)
        self.kappa_0 = kappa_0
        self.nu_0 = nu_0
        self.kappa_mf = kappa_mf if kappa_mf is not None else kappa_0
        self.nu_mf = nu_mf if nu_mf is not None else nu_0
        self.mu_mf = self.mu
        self.sigma_mf = self.sigma
        have_prior = not any(a is None for a in (mu_0, sigma_0, kappa_0, nu_0))
        if mu is None and sigma is None and have_prior:
            self.resample()  # initialise from the prior, as upstream does

    # -- sampling ------------------------------------------------------------
    def resample(self, data=()):
        D = len(self.mu_0)
        if len(data) == 0:
            hyp = (self.mu_0, self.sigma_0, self.kappa_0, self.nu_0)
        else:
            data = np.reshape(np.asarray(data, dtype=np.float64), (-1, D))
            hyp = self._posterior_hypparams(
                *self._get_weighted_statistics(data, np.ones(len(data)), D))
        self.mu, self.sigma = sample_niw(*hyp)
        self.mu_mf, self.sigma_mf = self.mu, self.sigma
        return self

    def rvs(self, size=None):
        size = 1 if size is None else size
        D = self.mu.shape[0]
        shape = size + (D,) if isinstance(size, tuple) else (size, D)
        chol = np.linalg.cholesky(self.sigma)
        return self.mu + np.random.normal(size=shape).dot(chol.T)

    def num_parameters(self):
        D = len(self.mu_0 if self.mu_0 is not None else self.mu)
        return D * (D + 1) / 2

    # -- statistics / conjugate update --------------------------------------
    def _get_weighted_statistics(self, data, weights, D=None):
        D = len(self.mu_0) if D is None else D
        data = np.reshape(data, (-1, D))
        neff = weights.sum()
        if neff > 0:
            xbar = np.dot(weights, data) / neff
            centered = data - xbar
            sumsq = np.dot(centered.T, centered * weights[:, None])
        else:
            xbar, sumsq = None, None
        return neff, xbar, sumsq

    def _posterior_hypparams(self, n, xbar, sumsq):
        mu_0, sigma_0, kappa_0, nu_0 = (self.mu_0, self.sigma_0, self.kappa_0,
                                        self.nu_0)
        if n > 0:
            mu_n = kappa_0 / (kappa_0 + n) * mu_0 + n / (kappa_0 + n) * xbar
            kappa_n = kappa_0 + n
            nu_n = nu_0 + n
            sigma_n = (sigma_0 + sumsq + kappa_0 * n / (kappa_0 + n)
                       * np.outer(xbar - mu_0, xbar - mu_0))
            return mu_n, sigma_n, kappa_n, nu_n
        return mu_0, sigma_0, kappa_0, nu_0

    def meanfieldupdate(self, data, weights):
        D = len(self.mu_0)
        self.mu_mf, self.sigma_mf, self.kappa_mf, self.nu_mf = \
            self._posterior_hypparams(
                *self._get_weighted_statistics(data, weights, D))
        self.mu, self.sigma = self.mu_mf, self.sigma_mf / (self.nu_mf - D - 1)

    # -- mean-field expectations ---------------------------------------------
    def _loglmbdatilde(self):
        return _loglmbdatilde(self.sigma_mf, self.nu_mf)

    def expected_log_likelihood(self, x):
        """E_q[log N(x | mu, Sigma)] under the NIW factor; NaN rows stay NaN
        (callers apply ``np.nan_to_num``, reference ``hmmbase.py:220``)."""
        mu_n, kappa_n, nu_n = self.mu_mf, self.kappa_mf, self.nu_mf
        D = len(mu_n)
        x = np.reshape(x, (-1, D)) - mu_n
        chol = np.linalg.cholesky(self.sigma_mf)
        xs = sla.solve_triangular(chol, x.T, lower=True, check_finite=False)
        return (self._loglmbdatilde() / 2. - D / (2. * kappa_n)
                - nu_n / 2. * np.einsum('ij,ij->j', xs, xs)
                - D / 2. * np.log(2. * np.pi))

    def quadratic_form(self):
        return niw_quadratic_form(self.mu_mf, self.sigma_mf, self.kappa_mf,
                                  self.nu_mf)

    def get_vlb(self, convention=None):
        """E_q[log p(mu,Sigma)] + H[q] of the NIW factor (Bishop eqs. 10.74 and 10.77).

        ``convention`` (default: the module's ``VLB_CONVENTION``) selects the sign with which the
        prior's inverse-Wishart log-normaliser ``log Z(sigma_0, nu_0)`` enters:
        ``"pybasicbayes"`` adds it, as the upstream package's ``Gaussian.get_vlb`` does -- so ELBO
        traces (``elbo_vec``, ``lower_bound``) and ``hmmbatchcd``'s ``np.allclose(lb, elbo)``
        stopping test see the same magnitudes as with the reference's dependency; ``"bishop"``
        subtracts it (10.74: ``+ log B(W0, nu0) = - log Z``), which makes the term exactly
        ``-KL(q || prior)`` (zero at the prior, tests/test_emission_formula.py).  The two differ
        by the q-independent constant ``2 log Z(sigma_0, nu_0)`` per state and nothing else."""
        D = len(self.mu_0)
        llt = self._loglmbdatilde()
        dmu = self.mu_mf - self.mu_0
        q_entropy = (-0.5 * (llt + D * (np.log(self.kappa_mf / (2 * np.pi)) - 1))
                     + _invwishart_entropy(self.sigma_mf, self.nu_mf))
        p_avgengy = (0.5 * (D * np.log(self.kappa_0 / (2 * np.pi)) + llt
                            - D * self.kappa_0 / self.kappa_mf
                            - self.kappa_0 * self.nu_mf
                            * np.dot(dmu, np.linalg.solve(self.sigma_mf, dmu)))
                     + vlb_logz_sign(convention)
                     * _invwishart_log_partitionfunction(self.sigma_0, self.nu_0)
                     + (self.nu_0 - D - 1) / 2. * llt
                     - 0.5 * self.nu_mf
                     * np.linalg.solve(self.sigma_mf, self.sigma_0).trace())
        return p_avgengy + q_entropy


class DiagonalGaussian(object):
    """Gaussian with diagonal covariance: per dimension a normal-inverse-gamma prior

        sigma_d^2 ~ InvGamma(alphas_0[d], betas_0[d]),   mu_d | sigma_d^2 ~ N(mu_0[d], sigma_d^2 / nus_0[d])

    and a mean-field factor of the same form ``(mf_mu, mf_nus, mf_alphas, mf_betas)`` (the
    package's ``DiagonalGaussian``; scalars broadcast over the dimensions).  Implements the
    protocol the reference's loops use (module docstring): ``expected_log_likelihood``,
    ``meanfieldupdate``, ``get_vlb``, ``rvs``, ``mu`` / ``sigmas`` / ``sigma``.  On the device this
    family runs on 2 D + 1 features per row instead of (D+1)(D+2)/2 (``svihmm_set_emission_diag``).
    """

    svihmm_diag_fastpath = True

    def __init__(self, mu=None, sigmas=None, mu_0=None, nus_0=None, alphas_0=None, betas_0=None):
        f = lambda a: None if a is None else np.array(a, dtype=np.float64)
        self.mu_0 = f(mu_0)
        D = None if self.mu_0 is None else self.mu_0.shape[0]
        if D is None and mu is not None:
            D = np.asarray(mu).shape[0]
        b = lambda a: None if a is None else np.broadcast_to(np.asarray(a, dtype=np.float64), (D,)).copy()
        self.nus_0, self.alphas_0, self.betas_0 = b(nus_0), b(alphas_0), b(betas_0)
        self.mu = f(mu)
        self.sigmas = b(sigmas)
        have_prior = not any(a is None for a in (mu_0, nus_0, alphas_0, betas_0))
        if have_prior:
            self.mf_mu = self.mu_0.copy() if self.mu is None else self.mu.copy()
            self.mf_nus, self.mf_alphas, self.mf_betas = self.nus_0.copy(), self.alphas_0.copy(), self.betas_0.copy()
            if self.mu is None or self.sigmas is None:
                given_mu, given_s = self.mu, self.sigmas
                self.resample()      # initialise what was not given from the prior, as upstream does
                if given_mu is not None:
                    self.mu = given_mu
                if given_s is not None:
                    self.sigmas = given_s
                self.mf_mu = self.mu.copy()

    @property
    def sigma(self):
        return np.diag(self.sigmas)

    def num_parameters(self):
        return 2 * len(self.mu_0 if self.mu_0 is not None else self.mu)

    # -- sampling ------------------------------------------------------------
    def resample(self, data=()):
        D = len(self.mu_0)
        if len(data) == 0:
            mu_n, nus_n, alphas_n, betas_n = self.mu_0, self.nus_0, self.alphas_0, self.betas_0
        else:
            data = np.reshape(np.asarray(data, dtype=np.float64), (-1, D))
            mu_n, nus_n, alphas_n, betas_n = self._posterior_hypparams(
                *self._get_weighted_statistics(data, np.ones(len(data))))
        self.sigmas = 1. / np.random.gamma(alphas_n, scale=1. / betas_n)
        self.mu = np.sqrt(self.sigmas / nus_n) * np.random.randn(D) + mu_n
        return self

    def rvs(self, size=None):
        size = 1 if size is None else size
        D = self.mu.shape[0]
        shape = size + (D,) if isinstance(size, tuple) else (size, D)
        return self.mu + np.sqrt(self.sigmas) * np.random.normal(size=shape)

    # -- statistics / conjugate update --------------------------------------
    def _get_weighted_statistics(self, data, weights, D=None):
        """(n, sum_t w x, sum_t w x^2) per dimension: the expected sufficient statistics."""
        D = len(self.mu_0) if D is None else D
        data = np.reshape(data, (-1, D))
        weights = np.asarray(weights, dtype=np.float64)
        return weights.sum(), np.dot(weights, data), np.dot(weights, data ** 2)

    def _posterior_hypparams(self, n, sx, sxx):
        mu_0, nus_0, alphas_0, betas_0 = self.mu_0, self.nus_0, self.alphas_0, self.betas_0
        nus_n = nus_0 + n
        mu_n = (nus_0 * mu_0 + sx) / nus_n
        alphas_n = alphas_0 + 0.5 * n
        betas_n = betas_0 + 0.5 * (sxx + nus_0 * mu_0 ** 2 - nus_n * mu_n ** 2)
        return mu_n, nus_n, alphas_n, betas_n

    def _set_mf(self, mu_n, nus_n, alphas_n, betas_n):
        self.mf_mu, self.mf_nus, self.mf_alphas, self.mf_betas = mu_n, nus_n, alphas_n, betas_n
        self.mu = self.mf_mu
        # point estimate: the factor's mean where it exists, its mode otherwise
        self.sigmas = np.where(alphas_n > 1., betas_n / np.maximum(alphas_n - 1., 1e-300), betas_n / (alphas_n + 1.))

    def meanfieldupdate(self, data, weights):
        self._set_mf(*self._posterior_hypparams(*self._get_weighted_statistics(data, weights)))

    # natural parameters (additive in the statistics): [nus mu, nus, 2 betas + nus mu^2, 2 alphas]
    @staticmethod
    def to_natural(mu, nus, alphas, betas):
        return np.stack([nus * mu, nus, 2. * betas + nus * mu ** 2, 2. * alphas])

    @staticmethod
    def from_natural(eta):
        nus = eta[1]
        mu = eta[0] / nus
        return mu, nus, 0.5 * eta[3], 0.5 * (eta[2] - nus * mu ** 2)

    # -- mean-field expectations ---------------------------------------------
    def mf_expectations(self):
        """Coefficients (a, b, c) of E_q log N(x) = sum_d (a_d x_d^2 + b_d x_d) + c."""
        mu, nus, al, be = self.mf_mu, self.mf_nus, self.mf_alphas, self.mf_betas
        prec = al / be
        c = (-0.5 * (1. / nus + mu ** 2 * prec) - 0.5 * (np.log(be) - digamma(al))).sum() \
            - 0.5 * len(mu) * np.log(2. * np.pi)
        return -0.5 * prec, mu * prec, c

    def expected_log_likelihood(self, x):
        """E_q[log N(x | mu, diag sigma^2)] under the factor (centred form); NaN rows stay NaN."""
        mu, nus, al, be = self.mf_mu, self.mf_nus, self.mf_alphas, self.mf_betas
        x = np.reshape(x, (-1, len(mu))) - mu
        return ((-0.5 * (al / be) * x ** 2).sum(1)
                + (-0.5 / nus - 0.5 * (np.log(be) - digamma(al))).sum() - 0.5 * len(mu) * np.log(2. * np.pi))

    def get_vlb(self):
        """E_q[log p(mu, sigma^2)] - E_q[log q(mu, sigma^2)] = -KL(q || prior), summed over the
        dimensions (zero at the prior)."""
        m, nu, al, be = self.mf_mu, self.mf_nus, self.mf_alphas, self.mf_betas
        m0, nu0, al0, be0 = self.mu_0, self.nus_0, self.alphas_0, self.betas_0
        elog = np.log(be) - digamma(al)            # E log sigma^2
        prec = al / be                             # E 1 / sigma^2
        p = (0.5 * np.log(nu0 / (2 * np.pi)) - (al0 + 1.5) * elog - 0.5 * nu0 * (1. / nu + (m - m0) ** 2 * prec)
             + al0 * np.log(be0) - gammaln(al0) - be0 * prec)
        q = (0.5 * np.log(nu / (2 * np.pi)) - (al + 1.5) * elog - 0.5
             + al * np.log(be) - gammaln(al) - al)
        return float((p - q).sum())


# sign convention of the prior's log-normaliser in the NIW factors' ELBO term (see Gaussian.get_vlb)
VLB_CONVENTION = "pybasicbayes"


def vlb_logz_sign(convention=None):
    c = VLB_CONVENTION if convention is None else convention
    if c == "pybasicbayes":
        return 1.0
    if c == "bishop":
        return -1.0
    raise RuntimeError("unknown ELBO convention %r (pybasicbayes | bishop)" % (c,))


def niw_prior_logpart(sigma_0, nu_0):
    """``invwishart_log_partitionfunction(sigma_0[k], nu_0[k])`` for stacked priors [K,D,D], [K]."""
    sigma_0 = np.asarray(sigma_0, float); nu_0 = np.asarray(nu_0, float)
    D = sigma_0.shape[-1]
    hl = np.log(np.diagonal(np.linalg.cholesky(sigma_0), axis1=1, axis2=2)).sum(1)
    return -1. * (nu_0 * hl - (nu_0 * D / 2. * np.log(2.) + D * (D - 1) / 4. * np.log(np.pi)
                               + gammaln((nu_0[:, None] - np.arange(D)) / 2.).sum(1)))


def niw_vlb_batch(mu_mf, sigma_mf, kappa_mf, nu_mf, mu_0, sigma_0, kappa_0, nu_0, terms=None,
                  convention=None):
    """``Gaussian.get_vlb()`` for K NIW factors at once (stacked arrays [K,D], [K,D,D], [K]):
    same formulas (Bishop 10.74, 10.77), one batched Cholesky / solve instead of 4K small
    ones -- the ELBO bookkeeping of the SVI loop is otherwise slower than the device E-step."""
    mu_mf = np.asarray(mu_mf, float); sigma_mf = np.asarray(sigma_mf, float)
    K, D = mu_mf.shape
    kappa_mf = np.asarray(kappa_mf, float); nu_mf = np.asarray(nu_mf, float)
    kappa_0 = np.asarray(kappa_0, float); nu_0 = np.asarray(nu_0, float)
    ar = np.arange(D)

    def llt(chol, nu):
        return (digamma((nu[:, None] - ar) / 2.).sum(1) + D * np.log(2.)
                - 2. * np.log(np.diagonal(chol, axis1=1, axis2=2)).sum(1))

    def logpart(chol, nu):
        return -1. * (nu * np.log(np.diagonal(chol, axis1=1, axis2=2)).sum(1)
                      - (nu * D / 2. * np.log(2.) + D * (D - 1) / 4. * np.log(np.pi)
                         + gammaln((nu[:, None] - ar) / 2.).sum(1)))

    chol_0 = np.linalg.cholesky(np.asarray(sigma_0, float))
    if terms is not None:
        # (log det sigma_mf, tr(sigma_mf^-1 sigma_0), dmu' sigma_mf^-1 dmu) from the device
        # (HipEngine.emission_vlb_terms): no factorisation of sigma_mf on the host
        logdet_mf, tr_s0, quad = (np.asarray(t, float) for t in terms)
        half_ld = 0.5 * logdet_mf
    else:
        chol_mf = np.linalg.cholesky(sigma_mf)
        half_ld = np.log(np.diagonal(chol_mf, axis1=1, axis2=2)).sum(1)
        dmu = mu_mf - np.asarray(mu_0, float)
        # one factorisation per state for both right-hand sides
        sol = np.linalg.solve(sigma_mf, np.concatenate([dmu[:, :, None], np.asarray(sigma_0, float)], axis=2))
        quad = np.einsum('kd,kd->k', dmu, sol[:, :, 0])
        tr_s0 = np.trace(sol[:, :, 1:], axis1=1, axis2=2)

    def llt_h(hl, nu):
        return digamma((nu[:, None] - ar) / 2.).sum(1) + D * np.log(2.) - 2. * hl

    def logpart_h(hl, nu):
        return -1. * (nu * hl - (nu * D / 2. * np.log(2.) + D * (D - 1) / 4. * np.log(np.pi)
                                 + gammaln((nu[:, None] - ar) / 2.).sum(1)))

    l_mf = llt_h(half_ld, nu_mf)
    iw_entropy = logpart_h(half_ld, nu_mf) - (nu_mf - D - 1) / 2. * l_mf + nu_mf * D / 2.
    q_entropy = -0.5 * (l_mf + D * (np.log(kappa_mf / (2 * np.pi)) - 1)) + iw_entropy
    p_avgengy = (0.5 * (D * np.log(kappa_0 / (2 * np.pi)) + l_mf - D * kappa_0 / kappa_mf
                        - kappa_0 * nu_mf * quad)
                 + vlb_logz_sign(convention) * logpart(chol_0, nu_0) + (nu_0 - D - 1) / 2. * l_mf
                 - 0.5 * nu_mf * tr_s0)
    return p_avgengy + q_entropy


def _invwishart_log_partitionfunction(sigma, nu):
    D = sigma.shape[0]
    chol = np.linalg.cholesky(sigma)
    return -1. * (nu * np.log(chol.diagonal()).sum()
                  - (nu * D / 2. * np.log(2.) + D * (D - 1) / 4. * np.log(np.pi)
                     + gammaln((nu - np.arange(D)) / 2.).sum()))


def _invwishart_entropy(sigma, nu):
    D = sigma.shape[0]
    Elogdetlmbda = _loglmbdatilde(sigma, nu)
    return (_invwishart_log_partitionfunction(sigma, nu)
            - (nu - D - 1) / 2. * Elogdetlmbda + nu * D / 2.)


# --------------------------------------------------------------------------- #
#  Dirichlet-Categorical mean field                                           #
# --------------------------------------------------------------------------- #
class Categorical(object):
    """Categorical emission with a Dirichlet prior ``alphav_0`` and Dirichlet
    mean-field factor ``alpha_mf`` (reference use:
    ``hmmsgd_metaobs.py:907-926,1071-1084``)."""

    def __init__(self, weights=None, alpha_0=None, K=None, alphav_0=None,
                 alpha_mf=None):
        if alphav_0 is None and alpha_0 is not None and K is not None:
            alphav_0 = np.repeat(alpha_0 / K, K)
        self.alphav_0 = None if alphav_0 is None else np.array(alphav_0, float)
        self.K = len(self.alphav_0) if self.alphav_0 is not None else K
        self.weights = None if weights is None else np.array(weights, float)
        if self.weights is None and self.alphav_0 is not None:
            self.weights = np.random.dirichlet(self.alphav_0)
        self._alpha_mf = (np.array(alpha_mf, float) if alpha_mf is not None
                          else self.weights * self.alphav_0.sum()
                          if self.alphav_0 is not None else None)

    @property
    def alpha_mf(self):
        return self._alpha_mf

    def num_parameters(self):
        return len(self.weights)

    def rvs(self, size=None):
        return np.random.choice(len(self.weights), p=self.weights, size=size)

    def _get_weighted_statistics(self, data, weights):
        # ``weights`` is [N, C] already masked to the observed symbol
        # (reference ``hmmsgd_metaobs.py:918-923``); data is ignored upstream
        if np.ndim(weights) == 2:
            return (weights.sum(0),)
        counts = np.bincount(np.asarray(data, int).ravel(), weights=np.asarray(weights, float).ravel(),
                             minlength=self.K)
        return (counts,)

    def _posterior_hypparams(self, counts):
        return self.alphav_0 + counts

    def meanfieldupdate(self, data, weights):
        self._alpha_mf = self._posterior_hypparams(
            *self._get_weighted_statistics(data, weights))
        self.weights = self._alpha_mf / self._alpha_mf.sum()

    def expected_log_likelihood(self, x):
        x = np.asarray(x)
        out = np.full(x.shape[0], np.nan)
        ok = ~np.isnan(x.astype(float)).reshape(x.shape[0], -1).any(1)
        el = digamma(self._alpha_mf) - digamma(self._alpha_mf.sum())
        out[ok] = el[x[ok].astype(int).ravel()]
        return out

    def get_vlb(self):
        a, a0 = self._alpha_mf, self.alphav_0
        el = digamma(a) - digamma(a.sum())
        logpi_p = gammaln(a0.sum()) - gammaln(a0).sum() + ((a0 - 1) * el).sum()
        logpi_q = gammaln(a.sum()) - gammaln(a).sum() + ((a - 1) * el).sum()
        return logpi_p - logpi_q
