"""Batch coordinate-ascent VB for HMMs -- class surface of reference ``hmmbatchcd.py``.

Each iteration is one whole-chain E-step on the MI355X (emission log-lik, forward /
backward, posteriors, expected sufficient statistics; only O(K^2 + K D^2) numbers
cross PCIe) followed by the reference's closed-form M-step on the host
(hmmbatchcd.py:172-189)."""
from __future__ import division

import sys
import time

import numpy as np

from .hmmbase import VariationalHMMBase, is_niw_gaussian, is_diag_gaussian

eps = 1e-9


class VBHMM(VariationalHMMBase):
    """ Batch coordinate-descent variational inference for hidden Markov models
    (same constructor as reference hmmbatchcd.py:45-47)."""

    def __init__(self, obs, prior_init, prior_tran, prior_emit, mask=None,
                 init_init=None, init_tran=None, epsilon=1e-8, maxit=100,
                 verbose=False, sts=None, engine=None, device=0, dtype="f64"):
        super(VBHMM, self).__init__(obs, prior_init, prior_tran, prior_emit,
                                    mask=mask, init_init=init_init,
                                    init_tran=init_tran, verbose=verbose,
                                    sts=sts, engine=engine, device=device, dtype=dtype)
        self.epsilon = epsilon
        self.maxit = maxit

        self.var_x = np.random.rand(self.T, self.K)
        self.var_x /= np.sum(self.var_x, axis=1)[:, np.newaxis]

        self.lalpha = np.empty((self.T, self.K))
        self.lbeta = np.empty((self.T, self.K))
        self.lliks = np.empty((self.T, self.K))

        self.mod_init = np.zeros(self.K)
        self.mod_tran = np.zeros((self.K, self.K))

    def infer(self, fused=True):
        """ Run batch VB with coordinate ascent on the full data set
        (reference hmmbatchcd.py:114-170).

        ``fused=True`` keeps the posteriors on the device and brings back only the
        sufficient statistics each iteration; ``fused=False`` follows the reference
        literally (``local_update()`` then ``global_update()`` on host arrays) so
        subclasses overriding those methods keep working."""
        if type(self).local_update is not VariationalHMMBase.local_update \
                or type(self).global_update is not VBHMM.global_update:
            fused = False
        # the reference's "self.obs[self.mask,:]" is a no-op (quirk Q9): masked rows
        # still enter lliks, but are excluded from the emission update
        self.obs_full = self.obs.copy()

        epsilon = self.epsilon
        maxit = self.maxit

        self.elbo_vec = np.inf * np.ones(maxit)
        self.pred_logprob_mean = np.nan * np.ones(maxit)
        self.pred_logprob_std = np.nan * np.ones(maxit)
        self.iter_time = np.nan * np.ones(maxit)
        self._obs_dirty = True

        for it in range(maxit):
            start_time = time.time()
            if fused:
                st = self._batch_estep_stats()
                self._global_update_from_stats(st)
            else:
                self.local_update()
                self.global_update()
            self.iter_time[it] = time.time() - start_time

            lb = self.lower_bound()
            if self.verbose:
                print("iter: %d, ELBO: %.2f" % (it, lb))
                sys.stdout.flush()

            if np.allclose(lb, self.elbo, atol=epsilon):
                break
            else:
                self.elbo = lb
                self.elbo_vec[it] = lb
                if fused and np.any(self.mask):
                    self.var_x = self.engine.read_intermediate("var_x", 1, self.T)[0]
                tmp = self.pred_logprob()
                if tmp is not None:
                    self.pred_logprob_mean[it] = np.mean(tmp)
                    self.pred_logprob_std[it] = np.std(tmp)

        lbidx = np.where(np.logical_not(np.isinf(self.elbo_vec)))[0]
        self.elbo_vec = self.elbo_vec[lbidx]
        self.pred_logprob_mean = self.pred_logprob_mean[lbidx]
        self.pred_logprob_std = self.pred_logprob_std[lbidx]
        self.iter_time = self.iter_time[lbidx]

        if fused:
            self._fetch_local()

        if self.sts is not None:
            self.hamming, self.perm = self.hamming_dist(self.var_x, self.sts)

        self.obs = self.obs_full
        self._obs_dirty = True

    def _global_update_from_stats(self, st):
        """reference hmmbatchcd.py:172-189 expressed on the device statistics:
        ``sum_t outer(q[t-1],q[t])`` is ``A_raw`` (batch form), and
        ``meanfieldupdate(obs[inds], q[inds,k])`` only needs
        ``(neff, xbar/neff, S - neff xbar xbar')``."""
        self.var_init = self.prior_init + self._q0
        self.var_tran = self.prior_tran + st.A_raw
        if hasattr(st, "counts"):
            # Categorical emitters (the engine counted symbols): meanfieldupdate(obs[inds],
            # q[inds, k]) is alpha_mf = alphav_0 + sum_t q[t, k] [x_t = v] over unmasked rows
            for k in range(self.K):
                G = self.var_emit[k]
                G._alpha_mf = G._posterior_hypparams(st.counts[k])
                G.weights = G._alpha_mf / G._alpha_mf.sum()
            return
        if hasattr(st, "xsq"):
            # diagonal family: meanfieldupdate(obs[inds], q[inds, k]) from (n, sum q x, sum q x^2)
            for k in range(self.K):
                G = self.var_emit[k]
                G._set_mf(*G._posterior_hypparams(st.neff[k], st.xbar[k], st.xsq[k]))
            return
        for k in range(self.K):
            G = self.var_emit[k]
            if not is_niw_gaussian(G):
                raise RuntimeError("fused batch update needs NIW / diagonal Gaussian or Categorical "
                                   "emissions; call infer(fused=False)")
            n = st.neff[k]
            if n > 0:
                xbar = st.xbar[k] / n
                sumsq = st.S[k] - n * np.outer(xbar, xbar)
            else:
                xbar, sumsq = None, None
            G.mu_mf, G.sigma_mf, G.kappa_mf, G.nu_mf = G._posterior_hypparams(n, xbar, sumsq)
            G.mu, G.sigma = G.mu_mf, G.sigma_mf / (G.nu_mf - self.D - 1)

    def global_update(self):
        """ Literal host M-step of the reference (hmmbatchcd.py:172-189)."""
        self.var_init = self.prior_init + self.var_x[0, :]
        self.var_tran = self.prior_tran.copy()
        for t in range(1, self.T):
            self.var_tran += np.outer(self.var_x[t - 1, :], self.var_x[t, :])
        inds = np.logical_not(self.mask)
        for k in range(self.K):
            self.var_emit[k].meanfieldupdate(self.obs[inds, :], self.var_x[inds, k])
