"""Synthetic HMM data -- semantics of reference ``gen_synthetic.py``.

``generate_data`` keeps the reference's behaviour (start in state 0,
``np.random.choice(K, p=tran[cur])`` transitions, ``emit[cur].rvs()[0]`` emissions,
gen_synthetic.py:27-44) with the same global ``np.random`` stream consumption order,
so a seeded run reproduces the reference's sequence given the same emission class.
``generate_data_fast`` is a vectorised generator for the T=1e6..1e8 benchmark
sequences.  The mmap writer fixes the reference's shape bug (it allocates obs with
K instead of D columns and never writes the last row, gen_synthetic.py:165,173-180).
"""
from __future__ import division

import numpy as np

from .util import make_mask, make_mask_prediction


def _walk(tran, emit, T):
    K = tran.shape[0]
    curr_st = 0
    obs = [emit[0].rvs()[0]]
    sts = [0]
    for i in range(T - 1):
        curr_st = np.random.choice(K, p=tran[curr_st, :])
        sts.append(curr_st)
        obs.append(emit[curr_st].rvs()[0])
    return np.array(obs), np.array(sts)


def generate_data(tran, emit, T, miss=0., nmasks=1):
    obs, sts = _walk(tran, emit, T)
    masks = None
    if miss > 0.:
        masks = [make_mask(sts, miss) for i in range(nmasks)]
        if len(masks) == 1:
            masks = masks[0]
    return obs, sts, masks


def generate_data_smoothing(tran, emit, T, miss=0., left=0, nmasks=1):
    obs, sts = _walk(tran, emit, T)
    masks = None
    if miss > 0.:
        masks = [make_mask(sts, miss, left) for i in range(nmasks)]
        if len(masks) == 1:
            masks = masks[0]
    return obs, sts, masks


def generate_data_prediction(tran, emit, T, miss=0., nmasks=1):
    obs, sts = _walk(tran, emit, T)
    masks = None
    if miss > 0:
        masks = [make_mask_prediction(sts, miss) for i in range(nmasks)]
        if len(masks) == 1:
            masks = masks[0]
    return obs, sts, masks


def generate_data_fast(tran, means, chols, T, rng=None):
    """Vectorised Gaussian-HMM generator: state path by inverse-CDF on the rows of
    ``tran`` (chunked), emissions ``means[z] + eps @ chol[z]'``."""
    rng = np.random.default_rng() if rng is None else rng
    K, D = means.shape
    cdf = np.cumsum(tran, axis=1)
    cdf[:, -1] = 1.0
    u = rng.random(T)
    sts = np.empty(T, dtype=np.int64)
    cur = 0
    sts[0] = 0
    for t in range(1, T):
        cur = int(np.searchsorted(cdf[cur], u[t]))
        sts[t] = cur
    z = rng.normal(size=(T, D))
    obs = means[sts] + np.einsum('td,tkd->tk', z, chols[sts]) if chols.ndim == 3 else means[sts] + z
    return obs, sts


def generate_data_device(tran, means, chols, T, seed=0, engine=None, want_obs=False):
    """The same process generated in HBM by the engine (``svihmm_generate``): the sequence stays
    resident as the engine's observation copy -- T = 1e8 x D = 32 is 25.6 GB that never exists on
    the host -- and only the states (and the observations if asked) come back.
    ``chols``: lower Cholesky factors of the emission covariances [K,D,D].
    Returns ``(obs or None, sts int32[T], engine)``."""
    if engine is None:
        from .engine import HipEngine
        engine = HipEngine(0)
    engine.generate(tran, means, chols, T, seed)
    obs, sts = engine.read_generated(want_obs=want_obs, want_sts=True)
    return obs, sts, engine


def generate_data_mmap(tran, emit, T, obs_path='obs.dat', sts_path='sts.dat'):
    """Write a long sequence to disk with np.memmap (float64 obs [T,D], int32 sts [T,1])."""
    D = len(emit[0].rvs()[0])
    fpo = np.memmap(obs_path, dtype='float64', mode='w+', shape=(T, D))
    fps = np.memmap(sts_path, dtype='int32', mode='w+', shape=(T, 1))
    states = np.arange(tran.shape[0])
    curr_st = 0
    fps[0, :] = 0
    fpo[0, :] = emit[0].rvs()[0]
    for i in range(1, T):
        curr_st = np.random.choice(states, p=tran[curr_st, :])
        fps[i, :] = curr_st
        fpo[i, :] = emit[curr_st].rvs()[0]
    del fps
    del fpo


def read_data_mmap(N, T, size, obs_path='obs.dat'):
    fp = np.memmap(obs_path, dtype='float64', mode='r', shape=(T, N))
    for i in range(T // size):
        yield np.array(fp[i * size:(i + 1) * size, :])
