"""HipEngine: thin object wrapper over the C ABI (one handle = one GPU).

This is the E-step "operator" the host classes call; it mirrors the steps of the
reference's ``local_update`` / ``intermediate_pars`` (hmmsgd_metaobs.py:487-519,
857-928) but on whole minibatches of windows at once.  Device handles never live
in a pickled ``__dict__`` (the reference pickles whole VBHMM objects,
cluster/run_cluster_simple.py:32-37).
"""
import ctypes as C

import numpy as np

from . import _lib as L


class PackedStats(object):
    """View of the packed statistics buffer
    ``[A_raw K*K | xbar K*D | neff K | S K*D*D | lb]``."""

    def __init__(self, buf, K, D):
        self.buf, self.K, self.D = buf, K, D
        o = 0
        self.A_raw = buf[o:o + K * K].reshape(K, K); o += K * K
        self.xbar = buf[o:o + K * D].reshape(K, D); o += K * D
        self.neff = buf[o:o + K]; o += K
        self.S = buf[o:o + K * D * D].reshape(K, D, D); o += K * D * D
        self.lb = buf[o:o + 1]

    @staticmethod
    def size(K, D):
        return K * K + K * D + K + K * D * D + 1


class PackedDiagStats(object):
    """View of the packed statistics of a diagonal-Gaussian E-step
    ``[A_raw K*K | xbar K*D | neff K | xsq K*D | lb]`` (xsq[k, d] = sum_t q[t,k] x[t,d]^2)."""

    def __init__(self, buf, K, D):
        self.buf, self.K, self.D = buf, K, D
        o = 0
        self.A_raw = buf[o:o + K * K].reshape(K, K); o += K * K
        self.xbar = buf[o:o + K * D].reshape(K, D); o += K * D
        self.neff = buf[o:o + K]; o += K
        self.xsq = buf[o:o + K * D].reshape(K, D); o += K * D
        self.lb = buf[o:o + 1]

    @staticmethod
    def size(K, D):
        return K * K + 2 * K * D + K + 1


class PackedCatStats(object):
    """View of the packed statistics of a Categorical-emission E-step
    ``[A_raw K*K | counts K*V | lb]`` (counts[k, v] = sum of var_x[t, k] over unmasked rows
    with symbol v)."""

    def __init__(self, buf, K, V):
        self.buf, self.K, self.V = buf, K, V
        self.A_raw = buf[:K * K].reshape(K, K)
        self.counts = buf[K * K:K * K + K * V].reshape(K, V)
        self.lb = buf[K * K + K * V:K * K + K * V + 1]

    @staticmethod
    def size(K, V):
        return K * K + K * V + 1


class HipEngine(object):
    """E-step engine on one MI355X.  Raises RuntimeError when the HIP library or
    a GPU is unavailable (no fallback)."""

    name = "hip"

    def __init__(self, device=0, dtype="f64"):
        self._lib = L.load()
        h = C.c_void_p()
        L.check(self._lib.svihmm_create(int(device), C.byref(h)), "svihmm_create")
        self._h = h
        if dtype not in ("f64", "f32", np.float64, np.float32):
            raise RuntimeError("dtype must be 'f64' or 'f32'")
        if dtype in ("f32", np.float32):
            self.set_precision("f32")
        self.device = int(device)
        self.T = self.D = self.K = 0
        self.V = 0            # > 0: Categorical emission with V symbols is active
        self.diag = False     # diagonal-Gaussian emission is active (its own packed layout)
        self._comm = False

    # -- lifecycle ------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.svihmm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __getstate__(self):
        raise RuntimeError("HipEngine holds device memory and cannot be pickled; "
                           "VBHMM objects drop it in __getstate__")

    def sync(self):
        L.check(self._lib.svihmm_sync(self._h), "svihmm_sync")

    def set_precision(self, dtype):
        """'f64' (default) or 'f32': fp32 storage of the scaled messages + fp32 statistics GEMM on
        the E-step fast path (see include/svihmm.h); inputs and outputs stay float64."""
        mode = {"f64": L.F64, "f32": L.F32}[dtype]
        L.check(self._lib.svihmm_set_precision(self._h, mode), "svihmm_set_precision")

    def precision(self):
        """(mode, whether the last E-step batch ran in the fp32 format)."""
        m, u = C.c_int32(), C.c_int32()
        L.check(self._lib.svihmm_get_precision(self._h, C.byref(m), C.byref(u)), "svihmm_get_precision")
        return ("f32" if m.value == L.F32 else "f64"), bool(u.value)

    def on_next_mutation(self, callback):
        """Register a one-shot callback that runs before the next call that uploads parameters
        or overwrites the E-step intermediates (``lliks / lalpha / lbeta / var_x`` rows in HBM):
        a host object that fetches those rows lazily resolves them while they are still valid."""
        self._on_mutate = callback

    def _pre_mutate(self):
        cb, self._on_mutate = getattr(self, "_on_mutate", None), None
        if cb is not None:
            cb()

    # -- inputs ----------------------------------------------------------------------
    def set_obs(self, obs, mask=None):
        self._pre_mutate()
        obs = np.asarray(obs, dtype=np.float64)
        if obs.ndim == 1:
            obs = obs[:, None]
        obs = np.ascontiguousarray(obs)
        T, D = obs.shape
        m = None
        if mask is not None:
            m = np.ascontiguousarray(np.asarray(mask).astype(np.uint8))
            if m.shape != (T,):
                raise RuntimeError("mask must have shape (T,)")
        L.check(self._lib.svihmm_set_obs(self._h, L.dptr(obs), T, D, L.u8ptr(m)), "svihmm_set_obs")
        self.T, self.D = T, D
        self._obs_owner = None        # whoever uploaded claims the resident copy afterwards

    def generate(self, tran, means, chols, T, seed=0):
        """Generate a synthetic sequence directly in HBM (reference ``gen_synthetic.generate_data``
        semantics; counter-based randomness, see ``include/svihmm.h``): it becomes the resident
        observation copy.  ``chols`` are lower Cholesky factors of the emission covariances."""
        self._pre_mutate()
        tran = L.as_f64(tran)
        K = tran.shape[0]
        cdf = np.cumsum(tran, axis=1)
        cdf = np.ascontiguousarray(cdf / cdf[:, -1:])
        means = L.as_f64(means)
        D = means.shape[1]
        chols = L.as_f64(chols, (K, D, D))
        L.check(self._lib.svihmm_generate(self._h, int(T), K, D, L.dptr(cdf), L.dptr(means), L.dptr(chols),
                                          int(seed) & 0xFFFFFFFFFFFFFFFF), "svihmm_generate")
        self.T, self.D = int(T), D
        self._obs_owner = None

    def read_generated(self, want_obs=True, want_sts=True):
        sts = np.empty(self.T, dtype=np.int32) if want_sts else None
        obs = np.empty((self.T, self.D)) if want_obs else None
        L.check(self._lib.svihmm_read_generated(
            self._h, None if sts is None else sts.ctypes.data_as(C.c_void_p), L.dptr(obs)),
            "svihmm_read_generated")
        return obs, sts

    def shift_obs(self, shift):
        """Move the centre the handle keeps the resident observations at by ``shift`` (a
        conditioning hint; results of every call stay in the caller's coordinates)."""
        self._pre_mutate()
        c = np.ascontiguousarray(shift, dtype=np.float64)
        if c.shape != (self.D,):
            raise RuntimeError("shift must have shape (D,)")
        L.check(self._lib.svihmm_shift_obs(self._h, L.dptr(c)), "svihmm_shift_obs")

    def get_shift(self):
        """The centre ``c`` of the resident copy (``obs_dev = obs - c``)."""
        c = np.empty(self.D)
        L.check(self._lib.svihmm_get_shift(self._h, L.dptr(c)), "svihmm_get_shift")
        return c

    def set_obs_blocks(self, blocks, T, D, mask=None):
        """Upload a sequence that arrives in row blocks (``gen_synthetic.read_data_mmap``,
        reference ``gen_synthetic.py:188-191``): ``blocks`` yields ``[n_i, D]`` arrays in
        order; rows beyond the last block keep whatever the allocation held."""
        self._pre_mutate()
        T, D = int(T), int(D)
        m = None
        if mask is not None:
            m = np.ascontiguousarray(np.asarray(mask).astype(np.uint8))
            if m.shape != (T,):
                raise RuntimeError("mask must have shape (T,)")
        L.check(self._lib.svihmm_alloc_obs(self._h, T, D, int(m is not None)), "svihmm_alloc_obs")
        self.T, self.D = T, D
        self._obs_owner = None
        row = 0
        for blk in blocks:
            blk = np.ascontiguousarray(np.asarray(blk, dtype=np.float64).reshape(-1, D))
            n = blk.shape[0]
            if n == 0:
                continue
            mp = None if m is None else m[row:row + n].ctypes.data_as(C.c_void_p)
            L.check(self._lib.svihmm_set_obs_rows(self._h, row, n, L.dptr(blk), mp), "svihmm_set_obs_rows")
            row += n
        return row

    def set_globals(self, mod_init, ltran):
        self._pre_mutate()
        ltran = L.as_f64(ltran)
        K = ltran.shape[0]
        mod_init = L.as_f64(mod_init, (K,))
        if ltran.shape != (K, K):
            raise RuntimeError("ltran must be square")
        L.check(self._lib.svihmm_set_globals(self._h, K, L.dptr(mod_init), L.dptr(ltran)),
                "svihmm_set_globals")
        self.K = K

    def set_emission_niw(self, mu, sigma, kappa, nu, check=True):
        """Upload the K NIW mean-field factors; the quadratic-form parameters are built on
        the device.  ``check=True`` waits for the factorisation so that a sigma that is
        not positive definite raises here; ``check=False`` returns immediately (the error
        then surfaces at the next synchronising call -- used in hot loops)."""
        self._pre_mutate()
        mu = L.as_f64(mu)
        K, D = mu.shape
        sigma = L.as_f64(sigma, (K, D, D))
        kappa = L.as_f64(kappa, (K,))
        nu = L.as_f64(nu, (K,))
        L.check(self._lib.svihmm_set_emission_niw(self._h, K, D, L.dptr(mu), L.dptr(sigma),
                                                  L.dptr(kappa), L.dptr(nu)),
                "svihmm_set_emission_niw")
        self.V = 0
        self.diag = False
        if check:
            L.check(self._lib.svihmm_sync(self._h), "svihmm_set_emission_niw")

    def set_emission_diag(self, mu, nus, alphas, betas, check=True):
        """Upload K diagonal-covariance Gaussian factors (normal-inverse-gamma mean field per
        dimension, all arrays [K, D]); see ``svihmm_set_emission_diag``.  Statistics then come
        back as ``PackedDiagStats``."""
        self._pre_mutate()
        mu = L.as_f64(mu)
        K, D = mu.shape
        nus, alphas, betas = (L.as_f64(a, (K, D)) for a in (nus, alphas, betas))
        L.check(self._lib.svihmm_set_emission_diag(self._h, K, D, L.dptr(mu), L.dptr(nus), L.dptr(alphas),
                                                   L.dptr(betas)), "svihmm_set_emission_diag")
        self.V = 0
        self.diag = True
        if check:
            L.check(self._lib.svihmm_sync(self._h), "svihmm_set_emission_diag")

    def set_emission_prior(self, mu0, sigma0):
        mu0 = L.as_f64(mu0)
        K, D = mu0.shape
        sigma0 = L.as_f64(sigma0, (K, D, D))
        L.check(self._lib.svihmm_set_emission_prior(self._h, K, D, L.dptr(mu0), L.dptr(sigma0)),
                "svihmm_set_emission_prior")

    def niw_vlb_terms(self, mu, sigma, kappa, nu):
        """(log det sigma_mf, tr(sigma_mf^-1 sigma_0), (mu_mf-mu_0)' sigma_mf^-1 (mu_mf-mu_0)), each
        [K], for the given NIW mean-field parameters (prior from set_emission_prior); does not
        disturb the E-step's parameter set."""
        mu = L.as_f64(mu)
        K, D = mu.shape
        sigma = L.as_f64(sigma, (K, D, D)); kappa = L.as_f64(kappa, (K,)); nu = L.as_f64(nu, (K,))
        out = np.empty((3, K))
        L.check(self._lib.svihmm_niw_vlb_terms(self._h, K, D, L.dptr(mu), L.dptr(sigma), L.dptr(kappa),
                                               L.dptr(nu), L.dptr(out)), "svihmm_niw_vlb_terms")
        return out[0], out[1], out[2]

    def set_lliks(self, lliks):
        self._pre_mutate()
        lliks = L.as_f64(lliks)
        B, Lm, K = lliks.shape
        L.check(self._lib.svihmm_set_lliks(self._h, L.dptr(lliks), B, Lm), "svihmm_set_lliks")

    # -- compute -----------------------------------------------------------------------
    @staticmethod
    def _starts(starts):
        return np.ascontiguousarray(np.asarray(starts, dtype=np.int64).ravel())

    def loglik(self, starts, Lm, flags=0):
        self._pre_mutate()
        st = self._starts(starts)
        out = np.empty((len(st), Lm, self.K))
        L.check(self._lib.svihmm_loglik(self._h, L.i64ptr(st), len(st), int(Lm), int(flags),
                                        L.dptr(out)), "svihmm_loglik")
        return out

    def forward_backward(self, starts, Lm, flags=0, want=("lalpha", "lbeta", "var_x", "local_lb"),
                         B=None):
        self._pre_mutate()
        st = None if starts is None else self._starts(starts)
        B = len(st) if st is not None else int(B)
        self._rows = B * int(Lm)
        res = {}
        bufs = {}
        for name in ("lalpha", "lbeta", "var_x"):
            bufs[name] = np.empty((B, Lm, self.K)) if name in want else None
        bufs["local_lb"] = np.empty(B) if "local_lb" in want else None
        L.check(self._lib.svihmm_forward_backward(
            self._h, L.i64ptr(st), B, int(Lm), int(flags), L.dptr(bufs["lalpha"]),
            L.dptr(bufs["lbeta"]), L.dptr(bufs["var_x"]), L.dptr(bufs["local_lb"])),
            "svihmm_forward_backward")
        for k, v in bufs.items():
            if v is not None:
                res[k] = v
        return res

    def estep(self, starts, Lm, flags=L.TRANS_WRAP, read=True, inner=None):
        """Whole-minibatch E-step -> PackedStats (or None with read=False: the
        statistics stay in HBM for allreduce()).  ``inner=(off, length)`` restricts
        the statistics to that segment of every window (buffered meta-observations)."""
        self._pre_mutate()
        st = self._starts(starts)
        out = np.empty(self._packed_len()) if read else None
        off, ln = (0, int(Lm)) if inner is None else (int(inner[0]), int(inner[1]))
        self._rows = len(st) * int(Lm)
        L.check(self._lib.svihmm_estep_minibatch_ex(self._h, L.i64ptr(st), len(st), int(Lm),
                                                    off, ln, int(flags), L.dptr(out)),
                "svihmm_estep_minibatch")
        return self._wrap_packed(out) if read else None

    def _packed_len(self):
        if self.V:
            return PackedCatStats.size(self.K, self.V)
        return PackedDiagStats.size(self.K, self.D) if self.diag else PackedStats.size(self.K, self.D)

    def _wrap_packed(self, buf):
        if self.V:
            return PackedCatStats(buf, self.K, self.V)
        return PackedDiagStats(buf, self.K, self.D) if self.diag else PackedStats(buf, self.K, self.D)

    def set_emission_cat(self, logp):
        """Categorical emissions: ``logp[k, v] = E_q log theta_k[v]`` (obs = symbol indices)."""
        self._pre_mutate()
        logp = L.as_f64(logp)
        K, V = logp.shape
        L.check(self._lib.svihmm_set_emission_cat(self._h, K, V, L.dptr(logp)), "svihmm_set_emission_cat")
        self.V = V
        self.diag = False

    def pred_logprob(self, starts, Lm, flags=L.MASK_AS_NAN):
        """Mean predictive log-probability of the masked rows of the windows and their number
        (reference pred_logprob / pred_logprob_full); ``(None, 0)`` when nothing is masked."""
        self._pre_mutate()
        st = self._starts(starts)
        self._rows = len(st) * int(Lm)
        out = np.empty(2)
        L.check(self._lib.svihmm_pred_logprob(self._h, L.i64ptr(st), len(st), int(Lm), int(flags),
                                              L.dptr(out)), "svihmm_pred_logprob")
        n = int(out[1])
        return (float(out[0]) if n > 0 else None), n

    def state_argmax(self, true_sts=None, want_z=True):
        """``np.argmax(var_x, axis=1)`` over the rows of the last E-step (window-major) and, with
        labels, the count matrix ``DM[pred, true]`` of ``util.munkres_match`` (reference
        ``hmmbase.py:346-355``, ``util.py:236-277``) -- decoded on the device, so only the
        int32 labels cross the bus.  Returns ``(z or None, DM or None)``."""
        n = getattr(self, "_rows", 0)
        if n <= 0:
            raise RuntimeError("state_argmax: no E-step has run on this engine")
        z = np.empty(n, dtype=np.int32) if want_z else None
        ts = conf = None
        if true_sts is not None:
            ts = np.ascontiguousarray(np.asarray(true_sts).ravel(), dtype=np.int32)
            if ts.size != n:
                raise ValueError("true_sts has %d labels for %d decoded rows" % (ts.size, n))
            conf = np.zeros((self.K, self.K), dtype=np.int64)
        vp = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
        L.check(self._lib.svihmm_state_argmax(self._h, vp(ts), vp(z), vp(conf)), "svihmm_state_argmax")
        return z, conf

    def read_packed(self):
        out = np.empty(self._packed_len())
        L.check(self._lib.svihmm_read_packed(self._h, L.dptr(out)), "svihmm_read_packed")
        return self._wrap_packed(out)

    def read_intermediate(self, what, B, Lm):
        idx = {"lliks": 0, "lalpha": 1, "lbeta": 2, "var_x": 3}[what]
        out = np.empty((B, Lm, self.K))
        L.check(self._lib.svihmm_read_intermediate(self._h, idx, L.dptr(out)),
                "svihmm_read_intermediate")
        return out

    def read_rows(self, what, row0, nrows):
        idx = {"lliks": 0, "lalpha": 1, "lbeta": 2, "var_x": 3}[what]
        out = np.empty((int(nrows), self.K))
        L.check(self._lib.svihmm_read_rows(self._h, idx, int(row0), int(nrows), L.dptr(out)),
                "svihmm_read_rows")
        return out

    def ffbs(self, logA, uniforms, flags=0, want_lalpha=True):
        self._pre_mutate()
        logA = L.as_f64(logA, (self.K, self.K))
        u = L.as_f64(uniforms, (self.T,))
        z = np.empty(self.T, dtype=np.int64)
        la = np.empty((self.T, self.K)) if want_lalpha else None
        L.check(self._lib.svihmm_ffbs(self._h, L.dptr(logA), L.dptr(u), int(flags),
                                      L.i64ptr(z), L.dptr(la)), "svihmm_ffbs")
        return z, la

    def ffbs_sample(self, lalpha, logA, uniforms):
        """Backward sampling only from the supplied forward messages (the ``lalpha_init``
        branch of the reference's FFBS, hmm_fast.pyx:80-95)."""
        self._pre_mutate()
        lalpha = L.as_f64(lalpha)
        T, K = lalpha.shape
        logA = L.as_f64(logA, (K, K))
        u = L.as_f64(uniforms, (T,))
        z = np.empty(T, dtype=np.int64)
        L.check(self._lib.svihmm_ffbs_sample(self._h, T, K, L.dptr(lalpha), L.dptr(logA), L.dptr(u),
                                             L.i64ptr(z)), "svihmm_ffbs_sample")
        return z

    # -- SVI loop with the variational state resident in HBM ---------------------------------
    def svi_begin(self, prior_tran, var_tran, prior, factors, prior_logpart, maxit, zsign=1.0):
        """Upload the state of ``hmmsgd_metaobs.VBHMM.infer``: transition prior / factor [K,K],
        NIW ``prior`` and current ``factors`` as (mu [K,D], sigma [K,D,D], kappa [K], nu [K]).
        Afterwards ``svi_iteration`` runs whole iterations (stationary init, psi-expectations,
        E-step, natural-gradient step, ELBO) on the device without anything coming back."""
        self._pre_mutate()
        var_tran = L.as_f64(var_tran)
        K = var_tran.shape[0]
        prior_tran = L.as_f64(prior_tran, (K, K))
        mu0 = L.as_f64(prior[0]); D = mu0.shape[1]
        arrs = [mu0, L.as_f64(prior[1], (K, D, D)), L.as_f64(prior[2], (K,)), L.as_f64(prior[3], (K,)),
                L.as_f64(prior_logpart, (K,)), L.as_f64(factors[0], (K, D)), L.as_f64(factors[1], (K, D, D)),
                L.as_f64(factors[2], (K,)), L.as_f64(factors[3], (K,))]
        L.check(self._lib.svihmm_svi_begin(self._h, K, D, L.dptr(prior_tran), L.dptr(var_tran),
                                           *([L.dptr(a) for a in arrs] + [int(maxit), float(zsign)])),
                "svihmm_svi_begin")
        self.K, self.V = K, 0
        self.diag = False
        self._svi_shape = (K, D)
        self._svi_family = "niw"

    def svi_begin_diag(self, prior_tran, var_tran, prior, factors, maxit):
        """The same loop for ``DiagonalGaussian`` emitters: ``prior`` / ``factors`` as
        (mu, nus, alphas, betas), each [K, D]."""
        self._pre_mutate()
        var_tran = L.as_f64(var_tran)
        K = var_tran.shape[0]
        prior_tran = L.as_f64(prior_tran, (K, K))
        D = np.asarray(prior[0]).shape[1]
        pb = np.ascontiguousarray(np.stack([L.as_f64(a, (K, D)) for a in prior]))
        fb = np.ascontiguousarray(np.stack([L.as_f64(a, (K, D)) for a in factors]))
        L.check(self._lib.svihmm_svi_begin_diag(self._h, K, D, L.dptr(prior_tran), L.dptr(var_tran), L.dptr(pb),
                                                L.dptr(fb), int(maxit)), "svihmm_svi_begin_diag")
        self.K, self.V = K, 0
        self.diag = True
        self._svi_shape = (K, D)
        self._svi_family = "diag"

    def svi_begin_cat(self, prior_tran, var_tran, alpha0, alpha, maxit):
        """The same loop for ``Categorical`` emitters over one symbol column: Dirichlet prior and
        factors [K, V]."""
        self._pre_mutate()
        var_tran = L.as_f64(var_tran)
        K = var_tran.shape[0]
        prior_tran = L.as_f64(prior_tran, (K, K))
        alpha0 = L.as_f64(alpha0); V = alpha0.shape[1]
        alpha = L.as_f64(alpha, (K, V))
        L.check(self._lib.svihmm_svi_begin_cat(self._h, K, V, L.dptr(prior_tran), L.dptr(var_tran), L.dptr(alpha0),
                                               L.dptr(alpha), int(maxit)), "svihmm_svi_begin_cat")
        self.K, self.V = K, V
        self.diag = False
        self._svi_shape = (K, V)
        self._svi_family = "cat"

    def svi_read_factors(self):
        """(var_tran, var_init, factors) with ``factors`` in the loop's family layout: NIW
        (mu, sigma, kappa, nu), diagonal (mu, nus, alphas, betas), Categorical alpha [K, V]."""
        K, W = self._svi_shape
        fam = getattr(self, "_svi_family", "niw")
        n = {"niw": K * W + K * W * W + 2 * K, "diag": 4 * K * W, "cat": K * W}[fam]
        vt, vi, blk = np.empty((K, K)), np.empty(K), np.empty(n)
        L.check(self._lib.svihmm_svi_read_factors(self._h, L.dptr(vt), L.dptr(vi), L.dptr(blk)),
                "svihmm_svi_read_factors")
        if fam == "diag":
            fac = tuple(blk.reshape(4, K, W))
        elif fam == "cat":
            fac = blk.reshape(K, W)
        else:
            o1, o2 = K * W, K * W + K * W * W
            fac = (blk[:o1].reshape(K, W), blk[o1:o2].reshape(K, W, W), blk[o2:o2 + K], blk[o2 + K:])
        return vt, vi, fac

    def svi_iteration(self, it, starts, nwin_total, Lm, flags, rho, bfactA, bfactE, inner=None):
        """Enqueue iteration ``it`` on the windows ``starts`` (asynchronous)."""
        self._pre_mutate()
        st = self._starts(starts)
        off, ln = (0, int(Lm)) if inner is None else (int(inner[0]), int(inner[1]))
        self._rows = len(st) * int(Lm)
        L.check(self._lib.svihmm_svi_iteration(self._h, int(it), L.i64ptr(st), len(st), int(nwin_total),
                                               int(Lm), off, ln, int(flags), float(rho), float(bfactA),
                                               float(bfactE)), "svihmm_svi_iteration")

    def svi_set_adagrad(self, ada_G):
        """AdaGrad accumulator of the transition factor joins the resident state (reference
        hmmsgd_metaobs.py:1036-1040); ``None`` switches back to the plain rho step."""
        if ada_G is None:
            L.check(self._lib.svihmm_svi_set_adagrad(self._h, None), "svihmm_svi_set_adagrad")
            return
        g = L.as_f64(ada_G, (self.K, self.K))
        L.check(self._lib.svihmm_svi_set_adagrad(self._h, L.dptr(g)), "svihmm_svi_set_adagrad")

    def svi_read_adagrad(self):
        out = np.empty((self.K, self.K))
        L.check(self._lib.svihmm_svi_read_adagrad(self._h, L.dptr(out)), "svihmm_svi_read_adagrad")
        return out

    def svi_read_elbo(self, n):
        """(elbo_vec[:n], device milliseconds of each iteration); waits for the device."""
        e = np.empty(int(n)); ms = np.empty(int(n))
        L.check(self._lib.svihmm_svi_read_elbo(self._h, int(n), L.dptr(e), L.dptr(ms)), "svihmm_svi_read_elbo")
        return e, ms

    def svi_recoveries(self):
        """How often the current loop left its device-side counters for stream events mid-way (svihmm_debug.h)."""
        n = C.c_int32()
        L.check(self._lib.svihmm_svi_recoveries(self._h, C.byref(n)), "svihmm_svi_recoveries")
        return int(n.value)

    def svi_read_state(self):
        """Current (var_tran, var_init, mu, sigma, kappa, nu); waits for the device."""
        K, D = self._svi_shape
        out = [np.empty((K, K)), np.empty(K), np.empty((K, D)), np.empty((K, D, D)), np.empty(K), np.empty(K)]
        L.check(self._lib.svihmm_svi_read_state(self._h, *[L.dptr(a) for a in out]), "svihmm_svi_read_state")
        return tuple(out)

    def read_globals(self):
        """(mod_init [K], ltran [K,K]) as the recursions currently hold them."""
        mi, lt = np.empty(self.K), np.empty((self.K, self.K))
        L.check(self._lib.svihmm_read_globals(self._h, L.dptr(mi), L.dptr(lt)), "svihmm_read_globals")
        return mi, lt

    # -- multi-GPU ------------------------------------------------------------------------
    def comm_unique_id(self):
        buf = C.create_string_buffer(128)
        L.check(self._lib.svihmm_comm_unique_id(buf), "svihmm_comm_unique_id")
        return buf.raw

    def comm_init(self, uid, rank, nranks):
        L.check(self._lib.svihmm_comm_init(self._h, uid, int(rank), int(nranks)),
                "svihmm_comm_init")
        self._comm = True

    def comm_count(self):
        """Number of ranks RCCL itself reports for the communicator (0 without one)."""
        n = C.c_int32()
        L.check(self._lib.svihmm_comm_count(self._h, C.byref(n)), "svihmm_comm_count")
        return n.value

    def allreduce_packed(self):
        L.check(self._lib.svihmm_allreduce_packed(self._h), "svihmm_allreduce_packed")

    def export_packed(self):
        """This handle's statistics as the all-reduce would put them on the wire (caller coordinates)."""
        out = np.empty(self._packed_len())
        L.check(self._lib.svihmm_export_packed(self._h, L.dptr(out)), "svihmm_export_packed")
        return out

    def import_packed(self, buf):
        """The reduced vector of a host-side exchange back into the handle (see export_packed)."""
        buf = L.as_f64(buf, (self._packed_len(),))
        L.check(self._lib.svihmm_import_packed(self._h, L.dptr(buf)), "svihmm_import_packed")

    def allreduce_host(self, arr, op="sum"):
        a = np.ascontiguousarray(np.asarray(arr, dtype=np.float64).ravel())
        L.check(self._lib.svihmm_allreduce_host(self._h, L.dptr(a), a.size,
                                                1 if op == "max" else 0),
                "svihmm_allreduce_host")
        return a.reshape(np.shape(arr))

    # -- measurement ------------------------------------------------------------------------
    def profile(self, on=True, only=None):
        """HIP events around the handle's launches; ``only``: names of the kernel slots to bracket
        (``svihmm_kernel_name``), the others run without events."""
        v = 1 if on else 0
        if on and only is not None:
            names = [self._lib.svihmm_kernel_name(i).decode() for i in range(L.NKERN)]
            v = L.PROF_SLOTS
            for n in only:
                v |= 1 << names.index(n)
        L.check(self._lib.svihmm_profile_enable(self._h, v), "profile_enable")

    def profile_reset(self):
        L.check(self._lib.svihmm_profile_reset(self._h), "profile_reset")

    def profile_read(self):
        ms = np.zeros(L.NKERN)
        cnt = np.zeros(L.NKERN, dtype=np.int64)
        L.check(self._lib.svihmm_profile_read(self._h, L.dptr(ms), L.i64ptr(cnt)), "profile_read")
        names = [self._lib.svihmm_kernel_name(i).decode() for i in range(L.NKERN)]
        return {n: (float(m), int(c)) for n, m, c in zip(names, ms, cnt) if c > 0}

    def last_kernel(self, slot_name):
        """Name of the kernel function the slot's last launch dispatched ("" when not recorded)."""
        names = [self._lib.svihmm_kernel_name(i).decode() for i in range(L.NKERN)]
        return self._lib.svihmm_last_kernel_name(self._h, names.index(slot_name)).decode()

    def set_variant(self, which, value):
        # ("svi_loop", slot 0 -- 1: the resident SVI loop on stream events instead of device-side counters;
        #  include/svihmm_debug.h lists every slot)
        idx = {"svi_loop": 0, "stats": 1, "fb": 2, "emission_mt": 3, "pipeline": 4, "emission_orbit": 5, "chain": 6}[which] if isinstance(which, str) else which
        L.check(self._lib.svihmm_set_variant(self._h, idx, int(value)), "set_variant")

    def selftest_mfma(self, A, B):
        A = L.as_f64(A, (16, 4)); B = L.as_f64(B, (4, 16))
        out = np.empty((16, 16))
        L.check(self._lib.svihmm_selftest_mfma(self._h, L.dptr(A), L.dptr(B), L.dptr(out)),
                "selftest_mfma")
        return out


def device_count():
    n = C.c_int()
    lib = L.load()
    rc = lib.svihmm_device_count(C.byref(n))
    return n.value if rc == 0 else 0
