"""Extended-precision referee for the natural-gradient global step (test infrastructure).

The class-level fuzz found a family of cases -- ``metaobs_half = 1, mb_sz = 1``: ONE window of three rows per
iteration, batch factors (T - 2L - 1) / (2 L S) of 350 .. 750 -- where the device loop and the oracle engine end up
2e-6 .. 7e-5 apart in sigma_mf.  Both run fp64; the step

    eta3 = sigma + kappa mu mu'       eta3' = (1 - rho) eta3 + rho (eta3_0 + bE S)       sigma' = eta3' - kappa' mu' mu''

(reference util.py:28-60, hmmsgd_metaobs.py:1048-1069) cancels: which side carries the error cannot be read off two
fp64 results.  This module restates the step in ``np.longdouble`` (x87 extended: 64-bit mantissa) and offers

  * ``step_ld``: ONE global step from given (pre-state, statistics) -- what an exactly rounded evaluation of the
    reference's formulas would return for those inputs;
  * ``RefereeEngine``: the oracle engine with its variational state kept in longdouble across iterations (the E-step
    still runs in fp64 from the rounded state: its 1e-13 errors are common to every side).

Only tests import this module.
"""
import numpy as np

from oracle.engine import OracleEngine
from oracle import ref_numpy as R

LD = np.longdouble


def have_extended_precision():
    return np.finfo(LD).eps < 1e-18


def niw_nat_ld(mu, sigma, kappa, nu):
    """reference util.py:28-37 in longdouble."""
    mu = np.asarray(mu, dtype=LD); sigma = np.asarray(sigma, dtype=LD)
    kappa = LD(kappa); nu = LD(nu)
    p = len(mu)
    return [kappa * mu, kappa, sigma + np.outer(mu, mu) * kappa, nu + 2 + p]


def niw_moment_ld(e1, e2, e3, e4):
    """reference util.py:40-60 in longdouble -> (mu, sigma, kappa, nu)."""
    p = len(e1)
    mu = e1 / e2
    return mu, e3 - np.outer(mu, mu) * e2, e2, e4 - 2 - p


def step_ld(pre, prior, stats, rho, bE):
    """One NIW global step (hmmsgd_metaobs.py:1048-1069) per state in longdouble.
    pre / prior = (mu [K,D], sigma [K,D,D], kappa [K], nu [K]); stats = (xbar [K,D], neff [K], S [K,D,D]).
    Returns the post-state as longdouble arrays."""
    mu, sg, ka, nu = pre
    mu0, sg0, ka0, nu0 = prior
    xbar, neff, S = stats
    K = len(ka)
    rho = LD(rho); bE = LD(bE)
    out = ([], [], [], [])
    for k in range(K):
        n_old = niw_nat_ld(mu[k], sg[k], ka[k], nu[k])
        n_0 = niw_nat_ld(mu0[k], sg0[k], ka0[k], nu0[k])
        e = [np.asarray(xbar[k], dtype=LD), LD(neff[k]), np.asarray(S[k], dtype=LD), LD(neff[k])]
        n_new = [(1 - rho) * n_old[i] + rho * (n_0[i] + bE * e[i]) for i in range(4)]
        for o, v in zip(out, niw_moment_ld(*n_new)):
            o.append(v)
    return tuple(np.array(o, dtype=LD) for o in out)


def tran_step_ld(var_tran, prior_tran, A_raw, nwin, rho, bA):
    """hmmsgd_metaobs.py:1022-1046 (no AdaGrad; quirk Q2: prior_tran - 1 in every window's A_i) in longdouble."""
    vt = np.asarray(var_tran, dtype=LD)
    A_inter = np.asarray(A_raw, dtype=LD) + LD(nwin) * (np.asarray(prior_tran, dtype=LD) - 1)
    return ((1 - LD(rho)) * (vt - 1) + LD(rho) * (LD(bA) * A_inter)) + 1


class RefereeEngine(OracleEngine):
    """OracleEngine whose resident SVI state lives in longdouble: E-step (fp64, C restatement) from the rounded state,
    global step in longdouble.  NIW family, no AdaGrad."""

    def svi_begin(self, prior_tran, var_tran, prior, factors, prior_logpart, maxit, zsign=1.0):
        OracleEngine.svi_begin(self, prior_tran, var_tran, prior, factors, prior_logpart, maxit, zsign)
        sv = self._svi
        sv["var_tran_ld"] = np.array(sv["var_tran"], dtype=LD)
        sv["mf_ld"] = [np.array(a, dtype=LD) for a in sv["mf"]]

    def svi_iteration(self, it, starts, nwin_total, Lm, flags, rho, bfactA, bfactE, inner=None):
        sv = self._svi
        sv["var_init"] = R.stationary_init(sv["var_tran"])
        mod_init, ltran = R.psi_expectations(sv["var_init"], sv["var_tran"])
        self.set_globals(mod_init, ltran)
        self._svi_internal = True
        try:
            self.set_emission_niw(*sv["mf"])
        finally:
            self._svi_internal = False
        self.estep(starts, Lm, flags=flags, read=False, inner=inner)
        st = self._packed
        sv["var_tran_ld"] = tran_step_ld(sv["var_tran_ld"], sv["prior_tran"], st.A_raw, nwin_total, rho, bfactA)
        sv["mf_ld"] = list(step_ld(sv["mf_ld"], sv["prior"], (st.xbar, st.neff, st.S), rho, bfactE))
        sv["var_tran"] = np.array(sv["var_tran_ld"], dtype=np.float64)
        sv["mf"] = [np.array(a, dtype=np.float64) for a in sv["mf_ld"]]
        sv["elbo"][it] = np.nan
