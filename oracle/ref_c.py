"""ctypes wrapper of oracle/ref_c.c (TEST INFRASTRUCTURE ONLY)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int64)
_up = C.POINTER(C.c_uint8)


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        l = C.CDLL(_SO)
        l.orc_digamma.restype = C.c_double
        l.orc_digamma.argtypes = [C.c_double]
        l.orc_posterior.restype = C.c_double
        l.orc_lliks_niw.argtypes = [_dp, C.c_int64, C.c_int, C.c_int, _dp, _dp, _dp, _dp, _dp]
        l.orc_forward.argtypes = [_dp, _dp, _dp, C.c_int64, C.c_int, _dp]
        l.orc_backward.argtypes = [_dp, _dp, C.c_int64, C.c_int, _dp]
        l.orc_posterior.argtypes = [_dp, _dp, C.c_int64, C.c_int, _dp]
        l.orc_estep_minibatch.argtypes = [_dp, _up, C.c_int64, C.c_int, _ip, C.c_int, C.c_int,
                                          C.c_int, _dp, _dp, _dp, _dp, _dp, _dp, C.c_uint, _dp]
        l.orc_estep_minibatch_mt.argtypes = [_dp, _up, C.c_int64, C.c_int, _ip, C.c_int, C.c_int,
                                             C.c_int, _dp, _dp, _dp, _dp, _dp, _dp, C.c_uint, C.c_int, _dp]
        _lib = l
    return _lib


def _d(a):
    return a.ctypes.data_as(_dp)


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def lliks_niw(x, mu, sigma, kappa, nu):
    x, mu, sigma, kappa, nu = map(_c, (x, mu, sigma, kappa, nu))
    n, D = x.shape
    K = mu.shape[0]
    out = np.empty((n, K))
    rc = lib().orc_lliks_niw(_d(x), n, D, K, _d(mu), _d(sigma), _d(kappa), _d(nu), _d(out))
    if rc:
        raise RuntimeError("orc_lliks_niw rc=%d" % rc)
    return out


def forward(ll, mod_init, ltran):
    ll, mod_init, ltran = map(_c, (ll, mod_init, ltran))
    T, K = ll.shape
    la = np.empty((T, K))
    lib().orc_forward(_d(ll), _d(mod_init), _d(ltran), T, K, _d(la))
    return la


def backward(ll, ltran):
    ll, ltran = map(_c, (ll, ltran))
    T, K = ll.shape
    lb = np.empty((T, K))
    lib().orc_backward(_d(ll), _d(ltran), T, K, _d(lb))
    return lb


def posterior(la, lb):
    la, lb = map(_c, (la, lb))
    T, K = la.shape
    q = np.empty((T, K))
    s = lib().orc_posterior(_d(la), _d(lb), T, K, _d(q))
    return q, s


def estep_minibatch(obs, mask, starts, Lm, mod_init, ltran, mu, sigma, kappa, nu, flags=2, threads=1):
    """``threads > 1``: the windows are dealt to that many OpenMP threads (bench.py's all-cores
    baseline); the default is the reference's single-threaded loop."""
    obs, mod_init, ltran, mu, sigma, kappa, nu = map(_c, (obs, mod_init, ltran, mu, sigma,
                                                          kappa, nu))
    T, D = obs.shape
    K = ltran.shape[0]
    st = np.ascontiguousarray(starts, dtype=np.int64)
    m = None if mask is None else np.ascontiguousarray(np.asarray(mask).astype(np.uint8))
    packed = np.empty(K * K + K * D + K + K * D * D + 1)
    mp = None if m is None else m.ctypes.data_as(_up)
    if threads > 1:
        rc = lib().orc_estep_minibatch_mt(_d(obs), mp, T, D, st.ctypes.data_as(_ip), len(st), int(Lm), K,
                                          _d(mod_init), _d(ltran), _d(mu), _d(sigma), _d(kappa), _d(nu),
                                          int(flags), int(threads), _d(packed))
    else:
        rc = lib().orc_estep_minibatch(_d(obs), mp, T, D, st.ctypes.data_as(_ip), len(st), int(Lm), K,
                                       _d(mod_init), _d(ltran), _d(mu), _d(sigma), _d(kappa), _d(nu),
                                       int(flags), _d(packed))
    if rc:
        raise RuntimeError("orc_estep_minibatch rc=%d" % rc)
    return packed
