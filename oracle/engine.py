"""ORACLE -- TEST INFRASTRUCTURE ONLY.

``OracleEngine`` exposes the same protocol as ``pysvihmm_amd.engine.HipEngine`` but
evaluates every call with the reference restatement (oracle/ref_numpy.py, and the
C port for speed).  Tests inject it into the host classes (``engine=``) to check
the host logic on a machine without a GPU, and use it as the checker for the HIP
engine.  The product never imports this module and has no CPU fallback.
"""
import numpy as np

from . import ref_numpy as R
from . import ref_c

MASK_AS_NAN, TRANS_WRAP, USE_HOST_LLIKS = 1, 2, 4


class _Packed(object):
    def __init__(self, buf, K, D):
        self.buf, self.K, self.D = buf, K, D
        o = 0
        self.A_raw = buf[o:o + K * K].reshape(K, K); o += K * K
        self.xbar = buf[o:o + K * D].reshape(K, D); o += K * D
        self.neff = buf[o:o + K]; o += K
        self.S = buf[o:o + K * D * D].reshape(K, D, D); o += K * D * D
        self.lb = buf[o:o + 1]


class _PackedDiag(object):
    def __init__(self, buf, K, D):
        self.buf, self.K, self.D = buf, K, D
        o = 0
        self.A_raw = buf[o:o + K * K].reshape(K, K); o += K * K
        self.xbar = buf[o:o + K * D].reshape(K, D); o += K * D
        self.neff = buf[o:o + K]; o += K
        self.xsq = buf[o:o + K * D].reshape(K, D); o += K * D
        self.lb = buf[o:o + 1]


class _PackedCat(object):
    def __init__(self, buf, K, V):
        self.buf, self.K, self.V = buf, K, V
        self.A_raw = buf[:K * K].reshape(K, K)
        self.counts = buf[K * K:K * K + K * V].reshape(K, V)
        self.lb = buf[K * K + K * V:]


class OracleEngine(object):
    name = "oracle"

    def __init__(self, device=0, use_c=True):
        self.use_c = use_c
        self.T = self.D = self.K = 0
        self._host_ll = None
        self._last = {}
        self._packed = None
        self._comm = None

    def close(self):
        pass

    def sync(self):
        pass

    def on_next_mutation(self, callback):
        self._on_mutate = callback

    def _pre_mutate(self):
        cb, self._on_mutate = getattr(self, "_on_mutate", None), None
        if cb is not None:
            cb()

    # -- inputs
    def set_obs(self, obs, mask=None):
        self._pre_mutate()
        obs = np.array(obs, dtype=np.float64)
        if obs.ndim == 1:
            obs = obs[:, None]
        self.obs = obs
        self.T, self.D = obs.shape
        self.mask = None if mask is None else np.asarray(mask).astype(bool).copy()
        self._obs_owner = None

    def shift_obs(self, shift):
        """The HIP engine's conditioning hint (svihmm_shift_obs): results of every call stay in
        the caller's coordinates, so for the checker -- which evaluates the reference's centred
        quadratic forms directly -- it is a no-op."""
        self._pre_mutate()

    def get_shift(self):
        return np.zeros(self.D)

    def set_obs_blocks(self, blocks, T, D, mask=None):
        obs = np.zeros((int(T), int(D)))
        row = 0
        for blk in blocks:
            blk = np.asarray(blk, dtype=np.float64).reshape(-1, int(D))
            obs[row:row + blk.shape[0]] = blk
            row += blk.shape[0]
        self.set_obs(obs, mask)
        return row

    def set_globals(self, mod_init, ltran):
        self._pre_mutate()
        self.mod_init = np.array(mod_init, dtype=np.float64)
        self.ltran = np.array(ltran, dtype=np.float64)
        self.K = self.ltran.shape[0]

    def _svi_on_upload(self, family, factors):
        """What an emission upload does to a running device loop on libsvihmm_hip.so
        (svihmm_set_emission_niw / _diag / _cat): a re-push of the loop's OWN family and shape
        replaces its factors (the loop's factor block is the handle's parameter block; a
        Categorical table upload leaves the Dirichlet factors alone), anything else ends the loop."""
        sv = getattr(self, "_svi", None)
        if sv is None or getattr(self, "_svi_internal", False):
            return
        fam = sv.get("family", "niw")
        same = fam == family
        if same and family == "cat":
            same = factors[0].shape == sv["mf"].shape
        elif same:
            same = all(np.shape(a) == np.shape(b) for a, b in zip(factors, sv["mf"]))
        if not same:
            self._svi = None
        elif family != "cat":
            sv["mf"] = [np.array(a, dtype=np.float64) for a in factors]

    def set_emission_niw(self, mu, sigma, kappa, nu, check=True):
        self._pre_mutate()
        self.V = 0
        self.diag = False
        self.em = tuple(np.array(a, dtype=np.float64) for a in (mu, sigma, kappa, nu))
        self._svi_on_upload("niw", self.em)
        for k in range(len(self.em[2])):
            np.linalg.cholesky(self.em[1][k])

    def set_emission_diag(self, mu, nus, alphas, betas, check=True):
        self._pre_mutate()
        self.V = 0
        self.diag = True
        self.emd = tuple(np.array(a, dtype=np.float64) for a in (mu, nus, alphas, betas))
        self._svi_on_upload("diag", self.emd)
        if not all(np.all(a > 0) for a in self.emd[1:]):
            raise RuntimeError("set_emission_diag: nus / alphas / betas must be positive")

    def set_emission_cat(self, logp):
        self._pre_mutate()
        self.diag = False
        self.cat = np.array(logp, dtype=np.float64)
        self.V = self.cat.shape[1]
        self._svi_on_upload("cat", (self.cat,))

    def set_lliks(self, lliks):
        self._pre_mutate()
        self._host_ll = np.array(lliks, dtype=np.float64)

    # -- compute
    def _window_ll(self, s, Lm, flags):
        x = self.obs[s:s + Lm]
        if (flags & MASK_AS_NAN) and self.mask is not None:
            x = x.copy()
            x[self.mask[s:s + Lm]] = np.nan
        if getattr(self, "V", 0):
            xv = x[:, 0]
            ok = ~np.isnan(xv)
            ll = np.zeros((len(xv), self.K))
            ll[ok] = self.cat[:, xv[ok].astype(int)].T
            return ll
        if getattr(self, "diag", False):
            return R.lliks_diag(x, *self.emd)
        f = ref_c.lliks_niw if self.use_c else R.lliks_niw
        return f(x, *self.em)

    def _check(self, starts, Lm):
        for s in starts:
            if s < 0 or s + Lm > self.T:
                raise RuntimeError("window out of range")

    def loglik(self, starts, Lm, flags=0):
        self._pre_mutate()
        starts = np.asarray(starts, dtype=np.int64).ravel()
        self._check(starts, Lm)
        return np.stack([self._window_ll(int(s), Lm, flags) for s in starts])

    def _lls(self, starts, Lm, flags, B):
        if flags & USE_HOST_LLIKS:
            if self._host_ll is None or self._host_ll.shape[:2] != (B, Lm):
                raise RuntimeError("no uploaded lliks of shape [B,Lm,K]")
            return self._host_ll
        return self.loglik(starts, Lm, flags)

    def forward_backward(self, starts, Lm, flags=0,
                         want=("lalpha", "lbeta", "var_x", "local_lb"), B=None):
        self._pre_mutate()
        st = None if starts is None else np.asarray(starts, dtype=np.int64).ravel()
        B = len(st) if st is not None else int(B)
        ll = self._lls(st, Lm, flags, B)
        fw = ref_c.forward if self.use_c else R.forward_msgs
        bw = ref_c.backward if self.use_c else R.backward_msgs
        la = np.stack([fw(ll[b], self.mod_init, self.ltran) for b in range(B)])
        lb = np.stack([bw(ll[b], self.ltran) for b in range(B)])
        q = np.stack([R.posterior(la[b], lb[b]) for b in range(B)])
        llb = np.array([R.local_lower_bound(la[b]) for b in range(B)])
        self._last = dict(lliks=ll, lalpha=la, lbeta=lb, var_x=q, local_lb=llb)
        return {k: self._last[k] for k in want}

    def estep(self, starts, Lm, flags=TRANS_WRAP, read=True, inner=None):
        self._pre_mutate()
        st = np.asarray(starts, dtype=np.int64).ravel()
        self._check(st, Lm)
        B = len(st)
        K, D = self.K, self.D
        if getattr(self, "V", 0):
            return self._estep_cat(st, Lm, flags, read, inner)
        if getattr(self, "diag", False):
            return self._estep_diag(st, Lm, flags, read, inner)
        buf = np.zeros(K * K + K * D + K + K * D * D + 1)
        P = _Packed(buf, K, D)
        if B == 0:
            self._packed = P
            return P if read else None
        self.forward_backward(st, Lm, flags)
        off, ln = (0, Lm) if inner is None else inner
        q = self._last["var_x"][:, off:off + ln]
        for b in range(B):
            s = int(st[b]) + off
            P.A_raw[:] += (R.transition_stat_wrap(q[b]) if flags & TRANS_WRAP
                           else R.transition_stat_batch(q[b]))
            inds = (np.ones(ln, bool) if self.mask is None
                    else np.logical_not(self.mask[s:s + ln]))
            x = self.obs[s:s + ln][inds]
            for k in range(K):
                xb, ne, Sk = R.niw_suffstats(x, q[b][inds, k])
                P.xbar[k] += xb; P.neff[k] += ne; P.S[k] += Sk
        P.lb[0] = self._last["local_lb"].sum()
        self._packed = P
        return P if read else None

    def state_argmax(self, true_sts=None, want_z=True):
        """hmmbase.py:346-355 (np.argmax(full_var_x, axis=1)) + the count matrix of
        util.py:236-277 (DM[pred, true])."""
        q = self._last["var_x"].reshape(-1, self.K)
        z = np.argmax(q, axis=1).astype(np.int32)
        conf = None
        if true_sts is not None:
            ts = np.asarray(true_sts).ravel().astype(np.int64)
            ok = (ts >= 0) & (ts < self.K)
            conf = np.zeros((self.K, self.K), dtype=np.int64)
            np.add.at(conf, (z[ok], ts[ok]), 1)
        return (z if want_z else None), conf

    def pred_logprob(self, starts, Lm, flags=MASK_AS_NAN):
        self._pre_mutate()
        st = np.asarray(starts, dtype=np.int64).ravel()
        if self.mask is None:
            return None, 0
        r = self.forward_backward(st, Lm, flags, want=("var_x",))
        tot, n = 0.0, 0
        for b, s in enumerate(st):
            m = self.mask[s:s + Lm]
            if not m.any():
                continue
            xm = self.obs[s:s + Lm][m]
            ll = (self.cat[:, xm[:, 0].astype(int)].T if getattr(self, "V", 0)
                  else R.lliks_niw(xm, *self.em))
            v = np.log(r["var_x"][b][m] + 1e-9) + ll
            tot += np.sum(np.logaddexp.reduce(v, axis=1))
            n += int(m.sum())
        return (tot / n if n else None), n

    def _estep_cat(self, st, Lm, flags, read, inner):
        K, V, B = self.K, self.V, len(st)
        buf = np.zeros(K * K + K * V + 1)
        P = _PackedCat(buf, K, V)
        if B:
            self.forward_backward(st, Lm, flags)
            off, ln = (0, Lm) if inner is None else inner
            q = self._last["var_x"][:, off:off + ln]
            for b in range(B):
                s = int(st[b]) + off
                P.A_raw[:] += (R.transition_stat_wrap(q[b]) if flags & TRANS_WRAP
                               else R.transition_stat_batch(q[b]))
                x = self.obs[s:s + ln, 0]
                ok = ~np.isnan(x)
                if self.mask is not None:
                    ok &= ~self.mask[s:s + ln]
                for v in range(V):
                    P.counts[:, v] += q[b][ok & (x == v)].sum(0)
            P.lb[0] = self._last["local_lb"].sum()
        self._packed = P
        return P if read else None

    def _estep_diag(self, st, Lm, flags, read, inner):
        K, D, B = self.K, self.D, len(st)
        P = _PackedDiag(np.zeros(K * K + 2 * K * D + K + 1), K, D)
        if B:
            self.forward_backward(st, Lm, flags)
            off, ln = (0, Lm) if inner is None else inner
            q = self._last["var_x"][:, off:off + ln]
            for b in range(B):
                s = int(st[b]) + off
                P.A_raw[:] += (R.transition_stat_wrap(q[b]) if flags & TRANS_WRAP
                               else R.transition_stat_batch(q[b]))
                inds = (np.ones(ln, bool) if self.mask is None else np.logical_not(self.mask[s:s + ln]))
                x = self.obs[s:s + ln][inds]
                for k in range(K):
                    sx, n, sxx = R.diag_suffstats(x, q[b][inds, k])
                    P.xbar[k] += sx; P.neff[k] += n; P.xsq[k] += sxx
            P.lb[0] = self._last["local_lb"].sum()
        self._packed = P
        return P if read else None

    def read_packed(self):
        return self._packed

    def read_intermediate(self, what, B, Lm):
        return self._last[what]

    def read_rows(self, what, row0, nrows):
        a = self._last[what]
        return a.reshape(-1, a.shape[-1])[row0:row0 + nrows].copy()

    def ffbs(self, logA, uniforms, flags=0, want_lalpha=True):
        self._pre_mutate()
        ll = self._window_ll(0, self.T, flags)
        la = R.forward_msgs(ll, self.mod_init, self.ltran)
        return self.ffbs_sample(la, logA, uniforms), la

    def ffbs_sample(self, la, logA, uniforms):
        """hmm_fast.pyx:97-122 (backward sampling from given forward messages)."""
        T, K = la.shape
        z = np.empty(T, dtype=np.int64)
        lp = la[T - 1]
        p = np.exp(lp - lp.max()); p /= p.sum()
        z[T - 1] = R.rand_discrete(p, uniforms[T - 1])
        for t in range(T - 2, -1, -1):
            lp = la[t] + logA[:, z[t + 1]]
            p = np.exp(lp - lp.max()); p /= p.sum()
            z[t] = R.rand_discrete(p, uniforms[t])
        return z

    # -- the SVI loop protocol of HipEngine.svi_*: reference arithmetic, iteration by iteration
    def svi_begin(self, prior_tran, var_tran, prior, factors, prior_logpart, maxit, zsign=1.0):
        self._pre_mutate()
        c = lambda a: np.array(a, dtype=np.float64)
        self._svi = dict(prior_tran=c(prior_tran), var_tran=c(var_tran), prior=[c(a) for a in prior],
                         mf=[c(a) for a in factors], elbo=np.full(int(maxit), np.nan),
                         conv="pybasicbayes" if zsign > 0 else "bishop", var_init=None)
        self.K = self._svi["var_tran"].shape[0]

    def svi_begin_diag(self, prior_tran, var_tran, prior, factors, maxit):
        self._pre_mutate()
        c = lambda a: np.array(a, dtype=np.float64)
        self._svi = dict(prior_tran=c(prior_tran), var_tran=c(var_tran), prior=[c(a) for a in prior],
                         mf=[c(a) for a in factors], elbo=np.full(int(maxit), np.nan), var_init=None, family="diag")
        self.K = self._svi["var_tran"].shape[0]

    def svi_begin_cat(self, prior_tran, var_tran, alpha0, alpha, maxit):
        self._pre_mutate()
        c = lambda a: np.array(a, dtype=np.float64)
        self._svi = dict(prior_tran=c(prior_tran), var_tran=c(var_tran), prior=c(alpha0), mf=c(alpha),
                         elbo=np.full(int(maxit), np.nan), var_init=None, family="cat")
        self.K = self._svi["var_tran"].shape[0]

    def svi_read_factors(self):
        sv = self._svi
        fam = sv.get("family", "niw")
        fac = sv["mf"].copy() if fam == "cat" else tuple(a.copy() for a in sv["mf"])
        return sv["var_tran"].copy(), sv["var_init"].copy(), fac

    def _svi_tran_step(self, A_raw, nwin_total, rho, bfactA):
        sv = self._svi
        A_inter = A_raw + nwin_total * (sv["prior_tran"] - 1.)
        if sv.get("ada_G") is not None:            # hmmsgd_metaobs.py:1036-1040
            nats_old = sv["var_tran"] - 1.
            sv["ada_G"] = sv["ada_G"] + nats_old ** 2
            ada = sv["ada_G"] ** .25
            sv["var_tran"] = ((1. - 1.0 / ada) * nats_old + (bfactA * A_inter) / ada) + 1.
        else:
            sv["var_tran"] = ((1. - rho) * (sv["var_tran"] - 1.) + rho * (bfactA * A_inter)) + 1.

    def _svi_iteration_family(self, it, starts, nwin_total, Lm, flags, rho, bfactA, bfactE, inner):
        """The element-wise families (hmmsgd_metaobs.py:1050-1084 with their natural parameters)."""
        from scipy.special import digamma
        from pysvihmm_amd.distributions import Categorical, DiagonalGaussian
        sv = self._svi
        K = self.K
        self._svi_internal = True
        try:
            if sv["family"] == "diag":
                self.set_emission_diag(*sv["mf"])
            else:
                a = sv["mf"]
                self.set_emission_cat(digamma(a) - digamma(a.sum(1))[:, None])
        finally:
            self._svi_internal = False
        self.estep(starts, Lm, flags=flags, read=False, inner=inner)
        self.allreduce_packed()
        st = self._packed
        self._svi_tran_step(st.A_raw, nwin_total, rho, bfactA)
        vlb = 0.
        if sv["family"] == "diag":
            D = sv["mf"][0].shape[1]
            new = [np.empty_like(a) for a in sv["mf"]]
            for k in range(K):
                n_old = DiagonalGaussian.to_natural(*[a[k] for a in sv["mf"]])
                n_0 = DiagonalGaussian.to_natural(*[a[k] for a in sv["prior"]])
                e = np.stack([st.xbar[k], np.full(D, st.neff[k]), st.xsq[k], np.full(D, st.neff[k])])
                res = DiagonalGaussian.from_natural((1. - rho) * n_old + rho * (n_0 + bfactE * e))
                for i in range(4):
                    new[i][k] = res[i]
                g = DiagonalGaussian(mu=res[0], sigmas=np.ones(D), mu_0=sv["prior"][0][k], nus_0=sv["prior"][1][k],
                                     alphas_0=sv["prior"][2][k], betas_0=sv["prior"][3][k])   # (no random draws)
                g._set_mf(*res)
                vlb += g.get_vlb()
            sv["mf"] = new
        else:
            a0 = sv["prior"]
            inter = nwin_total * (a0 - 1.) + st.counts
            sv["mf"] = ((1. - rho) * (sv["mf"] - 1.) + rho * bfactE * inter) + 1.
            for k in range(K):
                vlb += Categorical(weights=np.full(a0.shape[1], 1.0 / a0.shape[1]), alphav_0=a0[k],
                                   alpha_mf=sv["mf"][k]).get_vlb()       # (weights given: no random draw)
        sv["elbo"][it] = st.lb[0] + R.dirichlet_lower_bound(sv["prior_tran"], sv["var_tran"]) + vlb

    def svi_iteration(self, it, starts, nwin_total, Lm, flags, rho, bfactA, bfactE, inner=None):
        """hmmsgd_metaobs.py:351 .. 445 for one iteration (stationary init :413-418 by
        np.linalg.eig as the reference does, psi :502-504, the minibatch loop, global_update
        :1010-1069, global_lower_bound :273-296)."""
        from pysvihmm_amd.distributions import Gaussian
        sv = getattr(self, "_svi", None)
        if sv is None:
            raise RuntimeError("svihmm_svi_iteration: call svihmm_svi_begin first")
        sv["var_init"] = R.stationary_init(sv["var_tran"])
        mod_init, ltran = R.psi_expectations(sv["var_init"], sv["var_tran"])
        self.set_globals(mod_init, ltran)
        if sv.get("family", "niw") != "niw":
            return self._svi_iteration_family(it, starts, nwin_total, Lm, flags, rho, bfactA, bfactE, inner)
        self._svi_internal = True
        try:
            self.set_emission_niw(*sv["mf"])
        finally:
            self._svi_internal = False
        self.estep(starts, Lm, flags=flags, read=False, inner=inner)
        self.allreduce_packed()
        st = self._packed
        K = self.K
        A_inter = st.A_raw + nwin_total * (sv["prior_tran"] - 1.)
        if sv.get("ada_G") is not None:            # hmmsgd_metaobs.py:1036-1040
            nats_old = sv["var_tran"] - 1.
            sv["ada_G"] = sv["ada_G"] + nats_old ** 2
            ada = sv["ada_G"] ** .25
            sv["var_tran"] = ((1. - 1.0 / ada) * nats_old + (bfactA * A_inter) / ada) + 1.
        else:
            sv["var_tran"] = ((1. - rho) * (sv["var_tran"] - 1.) + rho * (bfactA * A_inter)) + 1.
        mu, sg, ka, nu = sv["mf"]
        mu0, sg0, ka0, nu0 = sv["prior"]
        vlb = 0.
        for k in range(K):
            n_old = R.niw_nat(mu[k], sg[k], ka[k], nu[k])
            n_0 = R.niw_nat(mu0[k], sg0[k], ka0[k], nu0[k])
            e = [st.xbar[k], st.neff[k], st.S[k], st.neff[k]]
            n_new = [(1. - rho) * n_old[i] + rho * (n_0[i] + bfactE * e[i]) for i in range(4)]
            mu[k], sg[k], ka[k], nu[k] = R.niw_moment(*n_new)
            g = Gaussian(mu=mu[k], sigma=np.eye(len(mu[k])), mu_0=mu0[k], sigma_0=sg0[k], kappa_0=ka0[k],
                         nu_0=nu0[k])
            g.mu_mf, g.sigma_mf, g.kappa_mf, g.nu_mf = mu[k], sg[k], ka[k], nu[k]
            vlb += g.get_vlb(sv["conv"])
        sv["elbo"][it] = st.lb[0] + R.dirichlet_lower_bound(sv["prior_tran"], sv["var_tran"]) + vlb

    def svi_set_adagrad(self, ada_G):
        self._svi["ada_G"] = None if ada_G is None else np.array(ada_G, dtype=np.float64)

    def svi_read_adagrad(self):
        return self._svi["ada_G"].copy()

    def read_globals(self):
        return self.mod_init.copy(), self.ltran.copy()

    def svi_read_elbo(self, n):
        if getattr(self, "_svi", None) is None:
            raise RuntimeError("svihmm_svi_read_elbo: bad arguments")
        return self._svi["elbo"][:n].copy(), np.zeros(n)

    def svi_read_state(self):
        sv = self._svi
        # (before the first iteration: the stationary vector of the initial var_tran, as the device loop's begin computes it)
        vi = sv["var_init"] if sv["var_init"] is not None else R.stationary_init(sv["var_tran"])
        return (sv["var_tran"].copy(), np.array(vi, dtype=np.float64)) + tuple(a.copy() for a in sv["mf"])

    # -- multi-process (host all-reduce through an injected communicator)
    def allreduce_packed(self):
        if self._comm is not None:
            self._comm.allreduce_inplace(self._packed.buf)
