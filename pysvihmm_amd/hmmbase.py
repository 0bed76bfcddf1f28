"""VariationalHMMBase -- same class surface as the reference ``hmmbase.py``.

The E-step methods (``local_update``, ``forward_msgs``, ``backward_msgs``,
``full_local_update``, ``ffbs_fast``) run on the MI355X through the C ABI
(``pysvihmm_amd.engine.HipEngine``); everything cheap and O(K^2 + K D^2) stays
host-side NumPy with the reference's formulas: psi-expectations
(reference hmmbase.py:214-216, SciPy digamma), ELBO (hmmbase.py:145-199), metrics
(hmmbase.py:346-406).

State flows through instance attributes exactly like the reference (``obs, mask,
var_init, var_tran, var_emit, lliks, lalpha, lbeta, var_x, mod_init, mod_tran``).
The engine handle is never pickled (``__getstate__``).

``engine=`` lets a caller supply the engine object (tests inject the oracle to
exercise the host logic without a GPU).  The default is the HIP engine; there is
no CPU fallback: constructing it without the built library / a GPU raises.
"""
from __future__ import division

import abc
from copy import deepcopy

import numpy as np
import numpy.linalg as npl
import scipy.spatial.distance as dist
from numpy import newaxis as npa
from scipy.special import digamma, gammaln

from . import util
from . import _lib as L
from .distributions import Gaussian

# This is for taking logs of things so we don't get -inf (reference hmmbase.py:30)
eps = 1e-9
DBL_EPSILON = float(np.finfo(np.float64).eps)


def is_niw_gaussian(e):
    """Device fast path is keyed on the concrete emission family, mirroring the
    reference's ``type(self.var_emit[0]) is Gaussian`` dispatch
    (hmmsgd_metaobs.py:887,1050): ``Gaussian`` itself, subclasses that keep its
    ``expected_log_likelihood`` (an override must be honoured, so such objects take the generic
    plugin route: host ``expected_log_likelihood`` -> uploaded ``lliks``), or any class that opts in
    with ``svihmm_niw_fastpath = True``."""
    t = type(e)
    if getattr(t, "svihmm_niw_fastpath", False):
        return True
    return isinstance(e, Gaussian) and t.expected_log_likelihood is Gaussian.expected_log_likelihood


def is_diag_gaussian(e):
    """Diagonal-covariance family (``distributions.DiagonalGaussian`` and subclasses that keep
    its ``expected_log_likelihood``): the device runs it on 2 D + 1 features per row."""
    from .distributions import DiagonalGaussian
    t = type(e)
    return (isinstance(e, DiagonalGaussian)
            and t.expected_log_likelihood is DiagonalGaussian.expected_log_likelihood)


def dirichlet_elbo(prior, var):
    """sum over rows of E_q[log Dir(.|prior_r)] + H[Dir(.|var_r)] for row-wise Dirichlet factors
    (the ``*_energy + *_entropy`` terms of reference hmmbase.py:150-181 and
    hmmsgd_metaobs.py:277-292, with their ``eps`` placement inside gammaln / digamma)."""
    prior = np.asarray(prior, dtype=np.float64)
    var = np.asarray(var, dtype=np.float64)
    elog = digamma(var + eps) - digamma(var.sum(axis=1) + eps)[:, npa]      # E_q log theta
    lognorm = lambda a: gammaln(a.sum(axis=1) + eps) - gammaln(a + eps).sum(axis=1)
    energy = lognorm(prior) + ((prior - 1.) * elog).sum(axis=1)
    entropy = -(lognorm(var) + ((var - 1.) * elog).sum(axis=1))
    return float(energy.sum() + entropy.sum())


class VariationalHMMBase(object, metaclass=abc.ABCMeta):
    """Abstract base class for finite variational HMMs."""

    # Interface
    @abc.abstractmethod
    def global_update(self):
        pass

    @abc.abstractmethod
    def infer(self):
        """ Perform inference. """
        pass

    @staticmethod
    def make_param_dict(prior_init, prior_tran, prior_emit, mask=None):
        return {'prior_init': prior_init, 'prior_tran': prior_tran,
                'prior_emit': prior_emit, 'mask': mask}

    def set_mask(self, mask):
        if mask is None:
            self.mask = np.zeros(self.obs.shape[0], dtype='bool')
        else:
            self.mask = np.asarray(mask).astype('bool')
        self._obs_dirty = True

    def __init__(self, obs, prior_init, prior_tran, prior_emit, mask=None,
                 init_init=None, init_tran=None, verbose=False, sts=None,
                 engine=None, device=0, dtype="f64"):
        self.verbose = verbose
        self.sts = sts

        # Save the hyperparameters
        self.prior_init = deepcopy(np.asarray(prior_init)).astype('float64')
        self.prior_tran = deepcopy(np.asarray(prior_tran)).astype('float64')
        self.prior_emit = deepcopy(prior_emit)

        # Initialize global variational distributions.
        if init_init is None:
            self.var_init = self.prior_init / np.sum(self.prior_init)
        else:
            self.var_init = np.array(init_init, dtype='float64')
        if init_tran is None:
            self.var_tran = self.prior_tran / np.sum(self.prior_tran, axis=1)[:, np.newaxis]
        else:
            self.var_tran = np.array(init_tran, dtype='float64')

        # copy: mean and covariance of the prior objects are the (random) initial values
        self.var_emit = deepcopy(prior_emit)

        self.obs = obs
        self.K = self.prior_tran.shape[0]
        if obs.ndim == 1:
            self.T = obs.shape[0]
            self.D = 1
        elif obs.ndim == 2:
            self.T, self.D = obs.shape
        else:
            raise RuntimeError("obs must have 1 or 2 dimensions")
        self.set_mask(mask)

        self.elbo = -np.inf
        self._engine = engine
        self._device = device
        self._dtype = dtype       # "f64" (the reference's type) or "f32" (engine precision mode)
        self._obs_dirty = True
        self._lZ = None

    # -- engine plumbing -----------------------------------------------------------
    @property
    def engine(self):
        if self._engine is None:
            from .engine import HipEngine
            self._engine = HipEngine(self._device, dtype=getattr(self, "_dtype", "f64"))  # raises without library/GPU
            self._obs_dirty = True
        return self._engine

    def __getstate__(self):
        pend = self.__dict__.get("_pending_rows")
        if pend is not None:      # resolve device-resident attributes before the handle goes
            for name in list(pend[2]):
                getattr(self, name)
        d = dict(self.__dict__)
        d.pop('_pending_rows', None)
        d.pop('_prior_stack', None)
        d.pop('_host_lliks', None)
        d.pop('_obs_print', None)
        d['_engine'] = None       # device handles are not picklable
        d['_obs_dirty'] = True
        return d

    def __setstate__(self, d):
        self.__dict__.update(d)

    def set_data(self, obs, mask=None):
        self.obs = obs
        if mask is None:
            self.mask = np.zeros(self.obs.shape[0], dtype='bool')
        else:
            self.mask = mask
        self._obs_dirty = True

    def _upload_obs(self, force=False):
        """obs/mask -> HBM (once; again after set_data / in-place edits flagged by
        ``_obs_dirty``, or when another model has used the same engine since: the resident copy
        belongs to whoever uploaded last -- the engines clear ``_obs_owner`` on every upload).
        Coordinates are the engine's business: the resident copy is kept centred inside the
        handle, means and statistics cross the C ABI in the coordinates of ``self.obs``
        (include/svihmm.h, svihmm_set_obs)."""
        eng = self.engine
        if force or self._obs_dirty or getattr(eng, "_obs_owner", None) != id(self):
            eng.set_obs(self.obs, self.mask)
            eng._obs_owner = id(self)
            self._obs_dirty = False
            self._obs_print = self._obs_fingerprint()

    def _obs_fingerprint(self):
        """Identity + content probe of ``self.obs`` / ``self.mask``: buffer address, shape,
        strides and the bytes of a strided sample (every 509th element, the first and last 4096)
        and of the whole mask.  Only consulted when the caller opted in with
        ``assume_obs_unchanged = True`` (an attribute of the model): by default every ``infer()``
        uploads ``self.obs`` again, like the reference reads it afresh on every call -- a checksum
        of the whole buffer (0.26 s for 256 MB) costs 50 x the upload it would save (4.5 ms)."""
        import zlib
        o = np.asarray(self.obs)
        flat = o.reshape(-1) if o.flags.c_contiguous else None
        if flat is None:
            return None                   # (unusual layout: no probe, always upload)
        sample = np.concatenate((flat[:4096], flat[::509], flat[-4096:]))
        m = np.ascontiguousarray(self.mask)
        return (o.__array_interface__["data"][0], o.shape, o.strides, hash(sample.tobytes()),
                m.shape, zlib.crc32(m.view(np.uint8)))

    def _obs_unchanged(self):
        if not getattr(self, "assume_obs_unchanged", False):
            return False                  # default: upload again (in-place edits of any size count)
        fp = self.__dict__.get("_obs_print")
        return fp is not None and fp == self._obs_fingerprint()

    def _psi_expectations(self):
        """reference hmmbase.py:214-216."""
        self.mod_init = digamma(self.var_init + eps) - digamma(np.sum(self.var_init) + eps)
        tran_sum = np.sum(self.var_tran, axis=1)
        self.mod_tran = digamma(self.var_tran + eps) - digamma(tran_sum[:, npa] + eps)

    def _emission_arrays(self):
        ve = self.var_emit
        return (np.array([g.mu_mf for g in ve], dtype=np.float64),
                np.array([g.sigma_mf for g in ve], dtype=np.float64),
                np.array([float(g.kappa_mf) for g in ve]),
                np.array([float(g.nu_mf) for g in ve]))

    def _niw_fastpath(self):
        # (observations wider than the NIW kernels take go the generic route: lliks from the
        #  emitters' expected_log_likelihood on the host, recursions on the device)
        d = self.obs.shape[1] if self.obs.ndim == 2 else 1
        return d <= L.NIW_MAX_D and all(is_niw_gaussian(e) for e in self.var_emit)

    def _diag_fastpath(self):
        d = self.obs.shape[1] if self.obs.ndim == 2 else 1
        return (d <= L.DIAG_MAX_D and hasattr(self.engine, "set_emission_diag")
                and all(is_diag_gaussian(e) for e in self.var_emit))

    def _diag_arrays(self):
        ve = self.var_emit
        return tuple(np.array([getattr(g, n) for g in ve], dtype=np.float64)
                     for n in ("mf_mu", "mf_nus", "mf_alphas", "mf_betas"))

    def _cat_fastpath(self):
        """Categorical emissions over one integer-valued observation column (the device keeps
        the E log theta table and counts symbols; reference hmmsgd_metaobs.py:907-926)."""
        from .distributions import Categorical
        ve = self.var_emit
        return (all(type(e) is Categorical for e in ve)
                and len({e.num_parameters() for e in ve}) == 1
                and (self.obs.ndim == 1 or self.obs.shape[1] == 1))

    def _prior_arrays(self):
        """Stacked NIW prior hyperparameters (mu_0 [K,D], sigma_0 [K,D,D], kappa_0 [K], nu_0 [K]).
        The priors of a model do not change after construction (the reference never writes
        them either), so they are stacked once per emitter array."""
        ve = self.var_emit
        key = (id(ve), len(ve))
        c = self.__dict__.get("_prior_stack")
        if c is None or c[0] != key:
            c = (key, np.array([g.mu_0 for g in ve]), np.array([g.sigma_0 for g in ve]),
                 np.array([float(g.kappa_0) for g in ve]), np.array([float(g.nu_0) for g in ve]))
            self._prior_stack = c
        return c[1:]

    def _emit_vlb(self):
        """sum_k var_emit[k].get_vlb() (reference hmmbase.py:183-185), batched for NIW
        Gaussians."""
        ve = self.var_emit
        if self._niw_fastpath() and all(type(e).get_vlb is Gaussian.get_vlb for e in ve):
            from .distributions import niw_vlb_batch
            mu, sg, ka, nu = self._emission_arrays()
            mu0, sg0, ka0, nu0 = self._prior_arrays()
            terms = None
            eng = self.engine
            if hasattr(eng, "niw_vlb_terms"):
                # log det / trace / quadratic form of sigma_mf from the device (it factorises
                # sigma_mf for the E-step anyway): the host-side batched solve of K D x D systems
                # costs more than the device E-step of a 64-window minibatch
                key = (id(self), id(ve), mu0.shape)      # the engine's prior copy belongs to one model
                if getattr(eng, "_prior_owner", None) != key:
                    eng.set_emission_prior(mu0, sg0)
                    eng._prior_owner = key
                terms = eng.niw_vlb_terms(mu, sg, ka, nu)
            return float(np.sum(niw_vlb_batch(
                mu, sg, ka, nu, mu0, sg0, ka0, nu0, terms=terms)))
        tot = 0.
        for k in range(self.K):
            tot += ve[k].get_vlb()
        return tot

    def _push_globals(self):
        self.engine.set_globals(self.mod_init, self.mod_tran)

    def _push_emission(self, windows=None, Lm=None, nan_mask=False):
        """Make the emission log-likelihoods available on the device.

        NIW Gaussians: upload the K mean-field factors, lliks are evaluated by the
        emission kernel.  Any other plugin object: evaluate
        ``expected_log_likelihood`` on the host (reference hmmbase.py:219-220) and
        upload ``lliks``.  Returns the flag word for the engine calls."""
        if self._niw_fastpath():
            mu, sg, ka, nu = self._emission_arrays()
            self.engine.set_emission_niw(mu, sg, ka, nu)
            return L.MASK_AS_NAN if nan_mask else 0
        if self._diag_fastpath():
            # diagonal family: the K x D normal-inverse-gamma factors; lliks on 2 D + 1 features
            self.engine.set_emission_diag(*self._diag_arrays())
            return L.MASK_AS_NAN if nan_mask else 0
        if self._cat_fastpath():
            # Categorical: E log theta table (pybasicbayes Categorical.expected_log_likelihood),
            # looked up on the device
            table = np.stack([digamma(e.alpha_mf) - digamma(np.sum(e.alpha_mf)) for e in self.var_emit])
            self.engine.set_emission_cat(table)
            return L.MASK_AS_NAN if nan_mask else 0
        obs = self.obs if self.obs.ndim == 2 else self.obs[:, None]
        if windows is None:
            windows, Lm = [0], self.T
        ll = np.empty((len(windows), Lm, self.K))
        for b, s in enumerate(windows):
            x = obs[s:s + Lm]
            if nan_mask:
                x = x.copy()
                x[self.mask[s:s + Lm]] = np.nan
            for k, odist in enumerate(self.var_emit):
                ll[b, :, k] = np.nan_to_num(odist.expected_log_likelihood(x))
        self.engine.set_lliks(ll)
        self._host_lliks = ll
        return L.USE_HOST_LLIKS

    # -- ELBO ------------------------------------------------------------------------
    def lower_bound(self):
        """ Variational lower bound (reference hmmbase.py:145-199)."""
        # E_q[log p] - E_q[log q] of the Dirichlet factors (initial distribution, transition rows)
        pi_term = dirichlet_elbo(self.prior_init[None, :], self.var_init[None, :])
        A_term = dirichlet_elbo(self.prior_tran, self.var_tran)

        emit_vlb = self._emit_vlb()

        # "log Z" = sum over ALL t of LSE_k lalpha[t,k] (quirk Q4, hmmbase.py:194);
        # reduced on the device with the posterior kernel when available
        if self._lZ is not None:
            lZ = self._lZ
        else:
            lZ = np.sum(np.logaddexp.reduce(self.lalpha, axis=1))

        return pi_term + A_term + emit_vlb + lZ

    # -- E-step ------------------------------------------------------------------------
    def local_update(self, obs=None, mask=None):
        """Batch local update (reference hmmbase.py:201-229) on the device.
        Afterwards ``lliks, lalpha, lbeta, var_x, mod_init, mod_tran`` hold host
        arrays of shape [T,K] / [K] / [K,K]."""
        if obs is not None or mask is not None:
            self.set_data(self.obs if obs is None else obs,
                          self.mask if mask is None else mask)
        self._psi_expectations()
        self._upload_obs()
        self._push_globals()
        flags = self._push_emission()
        if (type(self).forward_msgs is not VariationalHMMBase.forward_msgs
                or type(self).backward_msgs is not VariationalHMMBase.backward_msgs):
            # a subclass replaced the message passes ("Override this for specialized behavior",
            # reference hmmbase.py:279,304): follow the reference's sequence literally
            self._local_update_literal([0], self.T, flags)
            return
        r = self.engine.forward_backward([0], self.T, flags=flags)
        self.lalpha = r["lalpha"][0]
        self.lbeta = r["lbeta"][0]
        self.var_x = r["var_x"][0]
        self._lZ = float(r["local_lb"][0])
        self.lliks = self.engine.read_intermediate("lliks", 1, self.T)[0]

    def _local_update_literal(self, starts, Lm, flags, metaobs=None):
        """lliks from the engine (or the uploaded host lliks), then ``self.forward_msgs()`` /
        ``self.backward_msgs()`` as overridable hooks, posterior on the host
        (reference hmmbase.py:219-229)."""
        if flags & L.USE_HOST_LLIKS:
            self.lliks = self._host_lliks[0]
        else:
            self.lliks = self.engine.loglik(starts, Lm, flags=flags)[0]
        if metaobs is None:
            self.forward_msgs()
            self.backward_msgs()
        else:
            self.forward_msgs(metaobs)
            self.backward_msgs(metaobs)
        v = self.lalpha + self.lbeta
        v = v - np.max(v, axis=1)[:, npa]
        v = np.exp(v)
        self.var_x = v / np.sum(v, axis=1)[:, npa]
        self._lZ = None

    def forward_msgs(self, obs=None, mask=None):
        """lalpha from ``self.lliks, self.mod_init, self.mod_tran``
        (reference hmmbase.py:266-295)."""
        self.engine.set_globals(self.mod_init, self.mod_tran)
        self.engine.set_lliks(np.ascontiguousarray(self.lliks)[None])
        r = self.engine.forward_backward(None, self.lliks.shape[0], flags=L.USE_HOST_LLIKS,
                                         want=("lalpha", "local_lb"), B=1)
        self.lalpha = r["lalpha"][0]
        self._lZ = float(r["local_lb"][0])

    def backward_msgs(self, obs=None, mask=None):
        """lbeta (reference hmmbase.py:297-320)."""
        self.engine.set_globals(self.mod_init, self.mod_tran)
        self.engine.set_lliks(np.ascontiguousarray(self.lliks)[None])
        r = self.engine.forward_backward(None, self.lliks.shape[0], flags=L.USE_HOST_LLIKS,
                                         want=("lbeta",), B=1)
        self.lbeta = r["lbeta"][0]

    def _batch_estep_stats(self):
        """Whole-chain E-step + expected sufficient statistics on the device, batch
        transition form (hmmbatchcd.py:182-184): used by the batch infer() loops so
        that only O(K^2 + K D^2) numbers cross PCIe per iteration."""
        self._psi_expectations()
        self._upload_obs()
        self._push_globals()
        flags = self._push_emission()
        st = self.engine.estep([0], self.T, flags=flags)  # no TRANS_WRAP: t=1..T-1
        self._lZ = float(st.lb[0])
        self._q0 = self.engine.read_rows("var_x", 0, 1)[0]
        return st

    def _fetch_local(self):
        """Pull the per-time-step arrays of the last device E-step to the host
        attributes (documented reference attributes)."""
        T = self.T
        self.lliks = self.engine.read_intermediate("lliks", 1, T)[0]
        self.lalpha = self.engine.read_intermediate("lalpha", 1, T)[0]
        self.lbeta = self.engine.read_intermediate("lbeta", 1, T)[0]
        self.var_x = self.engine.read_intermediate("var_x", 1, T)[0]

    def pred_logprob(self):
        """Mean predictive log-probability of the masked data
        (reference hmmbase.py:322-340)."""
        K = self.K
        obs = self.obs
        mask = self.mask
        nmiss = np.sum(mask)
        if nmiss == 0:
            return None
        logprob = np.zeros((nmiss, K))
        for k, odist in enumerate(self.var_emit):
            logprob[:, k] = (np.log(self.var_x[mask, k] + eps)
                             + odist.expected_log_likelihood(obs[mask, :]))
        return np.mean(np.logaddexp.reduce(logprob, axis=1))

    def full_local_update(self):
        self.local_update()
        return self.var_x

    def _full_estep_device(self):
        """Whole-chain E-step whose per-row results stay in HBM (for hamming_dist(None, ...))."""
        self._psi_expectations()
        self._upload_obs()
        self._push_globals()
        flags = self._push_emission()
        self.engine.forward_backward([0], self.T, flags=flags, want=())

    # -- FFBS (reference hmm_fast.pyx:43-124, bound at hmmbase.py:409-411) ---------------
    def ffbs_fast(self, var_init, lalpha_init=None, uniforms=None):
        """Forward-filter backward-sample.  Returns ``(z int64[T], lalpha[T,K])``.

        Follows the Cython variant: ``DBL_EPSILON`` instead of 1e-9 and transitions
        ``log(var_tran + DBL_EPSILON)`` (un-normalised; quirk Q6).  ``uniforms`` (one
        per time step) replaces libc ``rand()``, which cannot be reproduced on a
        device; default draws them from ``np.random``."""
        var_init = np.asarray(var_init, dtype=np.float64)
        A = self.var_tran
        mod_init = digamma(var_init + DBL_EPSILON) - digamma(np.sum(var_init) + DBL_EPSILON)
        logA = np.log(A + DBL_EPSILON)
        if uniforms is None:
            uniforms = np.random.random_sample(self.T)
        if lalpha_init is not None:
            # backward sampling only, from the supplied messages (hmm_fast.pyx:80-95: neither
            # the likelihoods nor the filter run; lalpha_init itself is returned)
            lalpha_init = np.asarray(lalpha_init, dtype=np.float64)
            if lalpha_init.shape != (self.T, self.K):
                raise RuntimeError("lalpha_init must have shape (T, K)")
            return self.engine.ffbs_sample(lalpha_init, logA, uniforms), lalpha_init
        self._upload_obs()
        self.engine.set_globals(mod_init, logA)
        flags = self._push_emission()
        z, lalpha = self.engine.ffbs(logA, uniforms, flags=flags)
        return z, lalpha

    # -- metrics (host) ----------------------------------------------------------------
    def hamming_dist(self, full_var_x, true_sts):
        """Hamming distance after the best label permutation (reference hmmbase.py:346-365).
        ``full_var_x=None``: the whole-chain E-step, the arg-max and the K x K count matrix
        run on the device and only the labels cross the bus (T = 1e6, K = 64: 4 MB instead
        of the 512 MB of var_x); the distance follows from the counts."""
        if full_var_x is None:
            true_sts = np.asarray(true_sts).ravel()
            self._full_estep_device()
            _, DM = self.engine.state_argmax(true_sts, want_z=False)
            best_match = util.match_from_counts(DM)
            hit = DM[np.arange(self.K), best_match].sum()
            return 1.0 - hit / float(true_sts.size), best_match
        state_sq = np.argmax(full_var_x, axis=1).astype(int)
        best_match = util.munkres_match(true_sts, state_sq, self.K)
        return dist.hamming(true_sts, best_match[state_sq]), best_match

    def KL_L2_gaussian(self, emit_true, permutation):
        """Evaluation metric (reference hmmbase.py:364-390): for every learned state ``k2`` paired
        with the true state ``k = permutation[k2]``, KL(N(learned) || N(true)) summed, and the
        summed L2 distance of the means."""
        kl_total = 0.
        l2_total = 0.
        for k2 in range(len(permutation)):
            tru, fit = emit_true[permutation[k2]], self.var_emit[k2]
            d = np.asarray(tru.mu) - np.asarray(fit.mu)
            l2_total += npl.norm(d)
            St, Sf = np.asarray(tru.sigma), np.asarray(fit.sigma)
            kl_total += .5 * (np.trace(npl.solve(St, Sf)) + d.dot(npl.solve(St, d)) - d.shape[0]
                              - (npl.slogdet(Sf)[1] - npl.slogdet(St)[1]))
        return kl_total, l2_total

    def A_dist(self, A_true, perm):
        A = self.var_tran / np.sum(self.var_tran, axis=1)[:, np.newaxis]
        A_true = A_true[np.ix_(perm, perm)]
        return npl.norm(A_true - A)
