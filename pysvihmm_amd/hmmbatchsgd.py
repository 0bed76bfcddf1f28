"""Batch natural-gradient VB -- class surface of reference ``hmmbatchsgd.py``.
Full-data E-step on the MI355X each iteration, Robbins-Monro natural-gradient
M-step on the host (hmmbatchsgd.py:202-259)."""
from __future__ import division

import sys
import time

import numpy as np

from .hmmbase import VariationalHMMBase, is_niw_gaussian, is_diag_gaussian
from . import util

eps = 1e-9
tau0 = 1.
kappa0 = 0.7


class VBHMM(VariationalHMMBase):
    """Same constructor as reference hmmbatchsgd.py:64-66."""

    @staticmethod
    def make_param_dict(prior_init, prior_tran, prior_emit, tau=tau0,
                        kappa=kappa0, mask=None):
        return {'prior_init': prior_init, 'prior_tran': prior_tran,
                'prior_emit': prior_emit, 'mask': mask, 'tau': tau,
                'kappa': kappa}

    def __init__(self, obs, prior_init, prior_tran, prior_emit, tau=tau0,
                 kappa=kappa0, mask=None, init_init=None, init_tran=None,
                 epsilon=1e-8, maxit=100, verbose=False, sts=None, engine=None, device=0, dtype="f64"):
        super(VBHMM, self).__init__(obs, prior_init, prior_tran, prior_emit,
                                    mask=mask, init_init=init_init,
                                    init_tran=init_tran, verbose=verbose,
                                    sts=sts, engine=engine, device=device, dtype=dtype)
        self.batch = self.obs
        self.elbo = -np.inf
        self.tau = tau
        self.kappa = kappa
        self.lrate = tau ** (-kappa)
        self.epsilon = epsilon
        self.maxit = maxit
        self.batchfactor = 1.

        self.var_x = np.ones((self.T, self.K))
        self.var_x /= np.sum(self.var_x, axis=1)[:, np.newaxis]

        self.lalpha = np.empty((self.T, self.K))
        self.lbeta = np.empty((self.T, self.K))
        self.lliks = np.empty((self.T, self.K))
        self.mod_init = np.zeros(self.K)
        self.mod_tran = np.zeros((self.K, self.K))

    def infer(self, fused=True):
        """reference hmmbatchsgd.py:143-200 (no early stop, ``if False:`` :180)."""
        if type(self).local_update is not VariationalHMMBase.local_update \
                or type(self).global_update is not VBHMM.global_update:
            fused = False
        self.obs_full = self.obs.copy()
        self.obs[self.mask, :] = np.nan       # hmmbatchsgd.py:149 (NaN rows -> lliks 0)
        self._obs_dirty = True

        maxit = self.maxit
        self.elbo_vec = np.inf * np.ones(maxit)
        self.pred_logprob_mean = np.nan * np.ones(maxit)
        self.pred_logprob_std = np.nan * np.ones(maxit)
        self.iter_time = np.nan * np.ones(maxit)

        for it in range(maxit):
            start_time = time.time()
            self.lrate = (it + self.tau) ** (-self.kappa)
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

            self.elbo = lb
            self.elbo_vec[it] = lb
            if np.any(self.mask):
                if fused:
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

    def _natgrad_emissions(self, lrate, stats_for_k):
        for k in range(self.K):
            G = self.var_emit[k]
            if is_diag_gaussian(G):
                # the same blend in the diagonal family's natural parameters (the reference's
                # NIW arithmetic, util.py:28-60, has no counterpart for it)
                new = stats_for_k(G, k)
                eta = ((1. - lrate) * G.to_natural(G.mf_mu, G.mf_nus, G.mf_alphas, G.mf_betas)
                       + lrate * G.to_natural(*new))
                G._set_mf(*G.from_natural(eta))
                continue
            mu_mf, sigma_mf, kappa_mf, nu_mf = stats_for_k(G, k)
            nats_t = util.NIW_mf_natural_pars(mu_mf, sigma_mf, kappa_mf, nu_mf)
            nats_old = util.NIW_mf_natural_pars(G.mu_mf, G.sigma_mf, G.kappa_mf, G.nu_mf)
            nats_new = (1. - lrate) * nats_old + lrate * nats_t
            util.NIW_mf_moment_pars(G, *nats_new)

    def _global_update_from_stats(self, st):
        lrate = self.lrate
        self.var_init = self.prior_init + self._q0
        nats_old = self.var_tran - 1.
        nats_t = (self.prior_tran + st.A_raw) - 1.
        self.var_tran = ((1. - lrate) * nats_old + lrate * nats_t) + 1.

        def from_stats(G, k):
            if hasattr(st, "xsq"):
                return G._posterior_hypparams(st.neff[k], st.xbar[k], st.xsq[k])
            if not is_niw_gaussian(G):
                raise RuntimeError("fused batch update needs NIW Gaussian emissions")
            n = st.neff[k]
            if n > 0:
                xbar = st.xbar[k] / n
                return G._posterior_hypparams(n, xbar, st.S[k] - n * np.outer(xbar, xbar))
            return G._posterior_hypparams(n, None, None)
        self._natgrad_emissions(lrate, from_stats)

    def global_update(self, batch=None):
        """Literal host M-step (reference hmmbatchsgd.py:202-259)."""
        if batch is None:
            batch = self.obs
        lrate = self.lrate
        self.var_init = self.prior_init + self.var_x[0, :]
        nats_old = self.var_tran - 1.
        tran_mf = self.prior_tran.copy()
        for t in range(1, self.T):
            tran_mf += np.outer(self.var_x[t - 1, :], self.var_x[t, :])
        nats_t = tran_mf - 1.
        self.var_tran = ((1. - lrate) * nats_old + lrate * nats_t) + 1.
        inds = np.logical_not(self.mask)
        self._natgrad_emissions(
            lrate, lambda G, k: (G._posterior_hypparams(*G._get_weighted_statistics(batch[inds, :], self.var_x[inds, k]))
                                 if is_diag_gaussian(G) else util.NIW_meanfield(G, batch[inds, :], self.var_x[inds, k])))
