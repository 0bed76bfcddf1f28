"""Synthetic HMM data -- semantics of reference ``gen_synthetic.py``.

``generate_data`` keeps the reference's behaviour (start in state 0,
``np.random.choice(K, p=tran[cur])`` transitions, ``emit[cur].rvs()[0]`` emissions,
gen_synthetic.py:27-44) with the same global ``np.random`` stream consumption order,
so a seeded run reproduces the reference's sequence given the same emission class.
``generate_data_fast`` draws the same process run by run on the host (T up to ~1e7);
``generate_data_device`` generates it in HBM (T = 1e8 and beyond).  The mmap writer fixes the reference's shape bug (it allocates obs with
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
    """Host generator for long Gaussian-HMM sequences (same process as ``generate_data``: start
    in state 0, row-``tran`` transitions, ``means[z] + chol[z] n``), drawn run by run instead of
    step by step: a visit to state ``k`` lasts Geometric(1 - tran[k,k]) steps and leaves to
    ``j != k`` with probability ``tran[k,j] / (1 - tran[k,k])``.  Emissions are filled per
    state, so the peak memory is the output itself.  ``chols``: [K,D,D] lower factors, or
    anything with ``ndim != 3`` for unit covariances.  (For T >= 1e7 prefer
    ``generate_data_device``: the sequence is then generated in HBM and never exists on the host.)"""
    rng = np.random.default_rng() if rng is None else rng
    tran = np.asarray(tran, dtype=np.float64)
    K, D = means.shape
    stay = np.clip(np.diag(tran), 0.0, 1.0)
    leave = tran.copy()
    np.fill_diagonal(leave, 0.0)
    tot = leave.sum(axis=1)
    leave_cdf = np.cumsum(leave / np.where(tot > 0, tot, 1.0)[:, None], axis=1)
    sts = np.empty(T, dtype=np.int64)
    t, cur = 0, 0
    while t < T:
        run = T - t if (stay[cur] >= 1.0 or tot[cur] <= 0) else int(rng.geometric(1.0 - stay[cur]))
        sts[t:t + run] = cur
        t += run
        if t < T:
            cur = int(min(np.searchsorted(leave_cdf[cur], rng.random(), side='right'), K - 1))
    obs = rng.normal(size=(T, D))
    full = getattr(chols, "ndim", 0) == 3
    for k in range(K):
        rows = np.flatnonzero(sts == k)
        if rows.size:
            obs[rows] = (obs[rows].dot(chols[k].T) if full else obs[rows]) + means[k]
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
