#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by EXECUTING THE REFERENCE.

Runs only in the build container (needs /root/reference).  Nothing of the
reference's source is written into the repo: its modules are copied to a
temporary directory, made importable on Python 3 by mechanical edits (SURVEY.md
section 8c recipe), executed with seeded inputs, and only the numeric inputs /
outputs are saved as ``.npz`` fixtures.

Mechanical edits applied to the temporary copies:
  1. ``python3 -m lib2to3 -w -n`` (print / xrange / cPickle ...)
  2. hmmbase.py:410-411  ``types.MethodType(f, None, cls)`` -> plain attribute
  3. util.py:14,25,37,70,83  ragged ``np.array([...])`` -> ``dtype=object``
  4. hmm_fast.pyx:59,70  ``np.int_t``/``np.int_`` -> ``np.int64_t``/``np.int64``;
     hmm_fast.pyx:80 ``lalpha_init == None`` -> ``is None``; cythonize + gcc
  5. ``np.float_`` -> ``np.float64`` (NumPy 2)
  6. Categorical branch of ``intermediate_pars`` / ``intermediate_pars_buffer``
     (hmmsgd_metaobs.py:920-921, 996-997) -- the branch cannot execute as written on any NumPy
     we know of: ``z[[[np.arange(n)], [data]]] = 1`` is the long-removed nested-list form of
     ``z[np.arange(n), data] = 1`` (and ``data`` is an [n, 1] column there, because
     ``local_update`` indexes ``obs[lo:hi, :]`` and so needs 2-D observations), and ``w *= z``
     multiplies the [n] weights into the [n, C] one-hot IN PLACE (a broadcasting error).  The
     staged copy uses the tuple index on the flattened integer symbols and ``w = w[:, None] * z``
     -- what the surrounding comments ("mask w to only ...") and the downstream
     ``_get_weighted_statistics(data, w)`` call require.  The Categorical fixture
     records that the unpatched branch raises (``cat_intermediate_raises``) and the output of
     the repaired branch; everything else in it (local_update on Categorical lliks,
     ``global_update`` :1071-1084) is unmodified reference code.
The absent third-party package ``pybasicbayes`` is satisfied by this repo's own
emission classes (``pysvihmm_amd.distributions``); therefore the fixtures pin
everything DOWNSTREAM of ``lliks`` (pure reference arithmetic) and record
``lliks`` itself as an input.

Usage:  python tests/golden/make_golden.py
"""
import glob
import importlib
import os
import re
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
SEED = 8675309  # the reference's own experiment seed, cluster/exper_run_simple.py:172


def stage_reference(tmp):
    files = ["hmmbase.py", "hmmbatchcd.py", "hmmbatchsgd.py", "hmmsgd_metaobs.py",
             "util.py", "gen_synthetic.py", "munkres.py"]
    for f in files:
        shutil.copy(os.path.join(REF, f), tmp)
    subprocess.check_call([sys.executable, "-m", "lib2to3", "-w", "-n"] +
                          [os.path.join(tmp, f) for f in files],
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)

    def patch(fn, fun):
        p = os.path.join(tmp, fn)
        s = open(p).read()
        s2 = fun(s)
        assert s2 != s, "patch did not apply: " + fn
        open(p, "w").write(s2)

    patch("hmmbase.py", lambda s: re.sub(
        r"VariationalHMMBase\.ffbs_fast = \\\n\s*types\.MethodType\(hmm_fast\.FFBS, None, VariationalHMMBase\)",
        "VariationalHMMBase.ffbs_fast = hmm_fast.FFBS", s))
    patch("util.py", lambda s: re.sub(
        r"np\.array\(\[([^\]\n]*)\]\)", r"np.array([\1], dtype=object)", s))
    for f in files:
        p = os.path.join(tmp, f)
        s = open(p).read().replace("np.float_)", "np.float64)")
        open(p, "w").write(s)

    # keep an unrepaired copy of the metaobs module to show that its Categorical branch raises
    shutil.copy(os.path.join(tmp, "hmmsgd_metaobs.py"), os.path.join(tmp, "hmmsgd_metaobs_asis.py"))
    patch("hmmsgd_metaobs.py", lambda s: s.replace(
        "z[[[np.arange(data.shape[0])], [data]]] = 1",
        "z[np.arange(data.shape[0]), np.asarray(data).ravel().astype(int)] = 1").replace(
        "w *= z", "w = w[:, None] * z"))

    # Cython module
    pyx = open(os.path.join(REF, "hmm_fast.pyx")).read()
    pyx = pyx.replace("np.int_t", "np.int64_t").replace("dtype=np.int_)", "dtype=np.int64)")
    pyx = pyx.replace("lalpha_init == None", "lalpha_init is None")
    pyx = pyx.replace("xrange", "range")
    open(os.path.join(tmp, "hmm_fast.pyx"), "w").write(pyx)
    import sysconfig
    subprocess.check_call([sys.executable, "-m", "cython", "-3", "hmm_fast.pyx"], cwd=tmp)
    ext = sysconfig.get_config_var("EXT_SUFFIX")
    subprocess.check_call(
        ["gcc", "-O2", "-shared", "-fPIC", "-w",
         "-I" + np.get_include(), "-I" + sysconfig.get_paths()["include"],
         "hmm_fast.c", "-o", "hmm_fast" + ext, "-lm"], cwd=tmp)

    # third-party stand-in: this repo's emission classes
    os.makedirs(os.path.join(tmp, "pybasicbayes"))
    open(os.path.join(tmp, "pybasicbayes", "__init__.py"), "w").write("")
    open(os.path.join(tmp, "pybasicbayes", "distributions.py"), "w").write(
        "from pysvihmm_amd.distributions import Gaussian, Categorical\n")


def make_problem(K, D, T, rng, miss=0.0):
    """Synthetic Gaussian HMM in the shape of SURVEY.md 8(d)."""
    from pysvihmm_amd.distributions import Gaussian
    tran = 0.9 * np.eye(K) + 0.1 / max(K - 1, 1) * (1 - np.eye(K))
    tran /= tran.sum(1)[:, None]
    means = rng.normal(0, 3.0, size=(K, D))
    sts = np.empty(T, dtype=np.int64)
    cur = 0
    for t in range(T):
        sts[t] = cur
        cur = rng.choice(K, p=tran[cur])
    obs = means[sts] + rng.normal(size=(T, D))
    mask = rng.random(T) < miss
    mu_0 = obs.mean(0)
    sigma_0 = 0.75 * np.cov(obs.T).reshape(D, D)
    prior_emit = []
    for k in range(K):
        g = Gaussian(mu=means[k] + rng.normal(size=D), sigma=np.eye(D),
                     mu_0=mu_0, sigma_0=sigma_0, kappa_0=0.01, nu_0=D + 2)
        # a proper NIW mean-field factor as starting point
        A = rng.normal(size=(D, D))
        g.sigma_mf = sigma_0 + 0.1 * A.dot(A.T)
        g.mu_mf = g.mu.copy()
        g.kappa_mf = 0.01 + 3.0 * rng.random()
        g.nu_mf = D + 2 + 5.0 * rng.random()
        prior_emit.append(g)
    prior_emit = np.array(prior_emit)
    prior_tran = np.ones((K, K))
    prior_init = np.ones(K)
    init_tran = 1.0 + rng.random((K, K)) * T / K
    return dict(obs=obs, sts=sts, mask=mask, prior_emit=prior_emit,
                prior_tran=prior_tran, prior_init=prior_init, init_tran=init_tran)


def emit_arrays(var_emit):
    return dict(mu=np.array([g.mu_mf for g in var_emit]),
                sigma=np.array([g.sigma_mf for g in var_emit]),
                kappa=np.array([float(g.kappa_mf) for g in var_emit]),
                nu=np.array([float(g.nu_mf) for g in var_emit]))


def prior_arrays(var_emit):
    return dict(mu0=np.array([g.mu_0 for g in var_emit]),
                sigma0=np.array([g.sigma_0 for g in var_emit]),
                kappa0=np.array([float(g.kappa_0) for g in var_emit]),
                nu0=np.array([float(g.nu_0) for g in var_emit]))


def trace_metaobs(HM, name, K, D, T, L, S, maxit, miss, seed, ctor_kw=None, infer_kw=None, probes=()):
    """``ctor_kw`` / ``infer_kw``: extra constructor / ``infer`` arguments (adaptive L, growBuffer);
    ``probes``: (ind, halflength) pairs for direct ``get_local_messages`` calls after ``infer``."""
    ctor_kw = dict(ctor_kw or {})
    infer_kw = dict(infer_kw or {})
    rng = np.random.default_rng(seed)
    pb = make_problem(K, D, T, rng, miss)
    hmm = HM.VBHMM(pb["obs"].copy(), pb["prior_init"], pb["prior_tran"],
                   pb["prior_emit"], tau=1.0, kappa=0.7, metaobs_half=L, mb_sz=S,
                   mask=pb["mask"], init_tran=pb["init_tran"], maxit=maxit,
                   seed=seed % (2 ** 31), **ctor_kw)
    rec = dict(obs=pb["obs"], mask=pb["mask"], prior_tran=pb["prior_tran"],
               init_tran=pb["init_tran"], L=L, S=S, T=T, K=K, D=D, maxit=maxit,
               tau=1.0, kappa=0.7, seed=seed % (2 ** 31))
    rec.update({"prior_" + k: v for k, v in prior_arrays(hmm.var_emit).items()})
    rec.update({"init_" + k: v for k, v in emit_arrays(hmm.var_emit).items()})
    it_state = dict(it=-1, w=0)
    per_window = []
    per_iter = []

    orig_local = hmm.local_update
    orig_inter = hmm.intermediate_pars
    orig_glob = hmm.global_update

    def local_update(metaobs=None):
        pre = dict(var_init=hmm.var_init.copy(), var_tran=hmm.var_tran.copy(),
                   i1=metaobs.i1, i2=metaobs.i2)
        if ctor_kw or infer_kw:
            pre["iter"] = len(per_iter)
        pre.update(emit_arrays(hmm.var_emit))
        orig_local(metaobs=metaobs)
        pre.update(mod_init=hmm.mod_init.copy(), mod_tran=hmm.mod_tran.copy(),
                   lliks=hmm.lliks.copy(), lalpha=hmm.lalpha.copy(),
                   lbeta=hmm.lbeta.copy(), var_x=hmm.var_x.copy(),
                   local_lb=hmm.local_lower_bound())
        per_window.append(pre)

    def intermediate_pars(metaobs=None):
        A_i, e_i = orig_inter(metaobs)
        w = per_window[-1]
        w["A_i"] = A_i.copy()
        w["xbar"] = np.array([e[0] for e in e_i])
        w["neff"] = np.array([float(e[1]) for e in e_i])
        w["Sk"] = np.array([e[2] for e in e_i])
        return A_i, e_i

    def global_update(A_inter, emit_inter):
        d = dict(A_inter=A_inter.copy(),
                 E_xbar=np.array([e[0] for e in emit_inter]),
                 E_neff=np.array([float(e[1]) for e in emit_inter]),
                 E_S=np.array([e[2] for e in emit_inter]),
                 lrate=hmm.lrate)
        orig_glob(A_inter, emit_inter)
        d["var_tran_new"] = hmm.var_tran.copy()
        d.update({"new_" + k: v for k, v in emit_arrays(hmm.var_emit).items()})
        if getattr(hmm, "adagrad", False):
            d["ada_G_new"] = hmm.ada_G.copy()       # (:1036-1040: the accumulated squared gradients)
        per_iter.append(d)

    orig_buf = hmm.intermediate_pars_buffer
    orig_selL = hmm.select_L
    orig_selB = hmm.select_buffer
    chosen_L, chosen_buf = [], []

    def intermediate_pars_buffer(metaobs, bufferL, L_):
        A_i, e_i = orig_buf(metaobs, bufferL, L_)
        w = per_window[-1]
        w["A_i"] = A_i.copy()
        w["xbar"] = np.array([e[0] for e in e_i])
        w["neff"] = np.array([float(e[1]) for e in e_i])
        w["Sk"] = np.array([e[2] for e in e_i])
        return A_i, e_i

    def select_L(*a, **k):
        r = orig_selL(*a, **k)
        chosen_L.append((len(per_iter), int(r)))
        return r

    def select_buffer(*a, **k):
        r = orig_selB(*a, **k)
        chosen_buf.append((len(per_iter), int(r)))
        return r

    hmm.local_update = local_update
    hmm.intermediate_pars = intermediate_pars
    hmm.global_update = global_update
    if ctor_kw or infer_kw:
        hmm.intermediate_pars_buffer = intermediate_pars_buffer
        hmm.select_L = select_L
        hmm.select_buffer = select_buffer
        rec["ctor_growBuffer"] = int(bool(ctor_kw.get("growBuffer", False)))
        rec["ctor_bufferBudget"] = int(bool(ctor_kw.get("bufferBudget", False)))
        if ctor_kw.get("adagrad", False):
            rec["ctor_adagrad"] = 1
        for k_, v_ in infer_kw.items():
            rec["infer_" + k_] = v_
    hmm.infer(**infer_kw)
    rec["elbo_vec"] = hmm.elbo_vec.copy()
    nw = len(per_window)
    ragged = len({w["lliks"].shape for w in per_window}) > 1
    for key in per_window[0]:
        if ragged and key in ("lliks", "lalpha", "lbeta", "var_x"):
            continue       # window length changes between iterations (adaptive L / buffer)
        rec["w_" + key] = np.array([w[key] for w in per_window])
    for key in per_iter[0]:
        rec["it_" + key] = np.array([d[key] for d in per_iter])
    if ctor_kw or infer_kw:
        rec["chosen_L"] = np.array(chosen_L, dtype=np.int64).reshape(-1, 2)      # (iteration, L)
        rec["chosen_buffer"] = np.array(chosen_buf, dtype=np.int64).reshape(-1, 2)
        # iteration index of every recorded window (the minibatch size may change: bufferBudget)
        rec["w_iter"] = np.array([w["iter"] for w in per_window])
        rec["w_len"] = np.array([w["lliks"].shape[0] for w in per_window])
        for j, (ind, hl) in enumerate(probes):
            rec["probe%d_ind" % j] = ind
            rec["probe%d_half" % j] = hl
            rec["probe%d_var_x" % j] = hmm.get_local_messages(ind, hl)
    rec["windows_per_iter"] = nw // maxit
    # full-sequence E-step with NaN masking (hmmsgd_metaobs.py:1147-1205)
    hmm.local_update = orig_local
    rec["full_var_init"] = hmm.var_init.copy()
    rec["full_var_x"] = hmm.full_local_update()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **rec)
    print("wrote", name, "windows:", nw)


def trace_batch(MOD, name, K, D, T, maxit, miss, seed, sgd):
    rng = np.random.default_rng(seed)
    pb = make_problem(K, D, T, rng, miss)
    np.random.seed(seed % (2 ** 31))
    kw = dict(mask=pb["mask"], init_tran=pb["init_tran"] / pb["init_tran"].sum(1)[:, None] * 5,
              maxit=maxit)
    if sgd:
        kw.update(tau=1.0, kappa=0.7)
    hmm = MOD.VBHMM(pb["obs"].copy(), pb["prior_init"], pb["prior_tran"],
                    pb["prior_emit"], **kw)
    rec = dict(obs=pb["obs"], mask=pb["mask"], prior_tran=pb["prior_tran"],
               prior_init=pb["prior_init"], init_tran=kw["init_tran"],
               var_init0=hmm.var_init.copy(), K=K, D=D, T=T, maxit=maxit, sgd=int(sgd))
    rec.update({"prior_" + k: v for k, v in prior_arrays(hmm.var_emit).items()})
    rec.update({"init_" + k: v for k, v in emit_arrays(hmm.var_emit).items()})
    its = []
    orig_local = hmm.local_update
    orig_glob = hmm.global_update

    def local_update(obs=None, mask=None):
        orig_local()
        its.append(dict(lliks=hmm.lliks.copy(), lalpha=hmm.lalpha.copy(),
                        lbeta=hmm.lbeta.copy(), var_x=hmm.var_x.copy(),
                        mod_init=hmm.mod_init.copy(), mod_tran=hmm.mod_tran.copy()))

    def global_update(*a, **k):
        orig_glob(*a, **k)
        d = its[-1]
        d["var_tran_new"] = hmm.var_tran.copy()
        d["var_init_new"] = hmm.var_init.copy()
        d.update({"new_" + kk: v for kk, v in emit_arrays(hmm.var_emit).items()})

    hmm.local_update = local_update
    hmm.global_update = global_update
    hmm.infer()
    rec["elbo_vec"] = np.asarray(hmm.elbo_vec, dtype=float)
    for key in its[0]:
        rec["it_" + key] = np.array([d[key] for d in its])
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **rec)
    print("wrote", name, "iters:", len(its), "elbo", rec["elbo_vec"][:3])


def trace_ffbs(HB, HM, name, K, D, T, seed):
    rng = np.random.default_rng(seed)
    pb = make_problem(K, D, T, rng, 0.0)
    hmm = HM.VBHMM(pb["obs"].copy(), pb["prior_init"], pb["prior_tran"],
                   pb["prior_emit"], metaobs_half=2, mb_sz=1,
                   init_tran=pb["init_tran"], maxit=1, seed=1)
    var_init = np.ones(K) / K + 0.1 * rng.random(K)
    z, lalpha = hmm.ffbs_fast(var_init)
    rec = dict(obs=pb["obs"], var_init=var_init, var_tran=hmm.var_tran.copy(),
               lalpha=lalpha, z=np.asarray(z), K=K, D=D, T=T)
    rec.update(emit_arrays(hmm.var_emit))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **rec)
    print("wrote", name)


def trace_categorical(HM, HM_asis, name, K, V, T, L, S, seed):
    """Categorical emissions (hmmsgd_metaobs.py:907-926, 1071-1084).  ``infer`` itself raises for
    them (``util.NIW_zero_nat_pars`` at :402-403), so the loop body is driven by hand exactly as
    :405-439 does: stationary init, ``local_update``, ``intermediate_pars``, accumulation,
    ``global_update``."""
    from pysvihmm_amd.distributions import Categorical
    rng = np.random.default_rng(seed)
    tran = 0.85 * np.eye(K) + 0.15 / (K - 1) * (1 - np.eye(K))
    emis = rng.dirichlet(0.3 * np.ones(V), size=K)
    sts = np.empty(T, dtype=np.int64)
    cur = 0
    for t in range(T):
        sts[t] = cur
        cur = rng.choice(K, p=tran[cur])
    obs = np.array([rng.choice(V, p=emis[z]) for z in sts], dtype=np.int64)[:, None]   # [T, 1] column
    mask = rng.random(T) < 0.1
    alphav_0 = 0.5 + rng.random(V)
    prior_emit = np.array([Categorical(alphav_0=alphav_0, alpha_mf=alphav_0 + 5.0 * rng.random(V),
                                       weights=np.ones(V) / V) for _ in range(K)])
    init_tran = 1.0 + rng.random((K, K)) * T / K
    mk = lambda mod: mod.VBHMM(obs.copy(), np.ones(K), np.ones((K, K)), prior_emit, tau=1.0, kappa=0.7,
                               metaobs_half=L, mb_sz=S, mask=mask, init_tran=init_tran, maxit=2,
                               seed=seed % (2 ** 31))
    hmm = mk(HM)
    rec = dict(obs=obs, mask=mask, K=K, V=V, T=T, L=L, S=S, tau=1.0, kappa=0.7, init_tran=init_tran,
               alphav_0=alphav_0, init_alpha_mf=np.array([g.alpha_mf for g in hmm.var_emit]))
    # (a) the unrepaired branch raises
    asis = mk(HM_asis)
    asis.local_update(metaobs=HM_asis.MetaObs(10, 10 + 2 * L))
    try:
        asis.intermediate_pars(HM_asis.MetaObs(10, 10 + 2 * L))
        rec["cat_intermediate_raises"] = 0
    except Exception as e:
        rec["cat_intermediate_raises"] = 1
        rec["cat_intermediate_error"] = type(e).__name__
    # (b) two hand-driven iterations of the loop body
    np.random.seed(seed % (2 ** 31))
    wins, its = [], []
    for it in range(2):
        hmm.lrate = (it + hmm.tau) ** (-hmm.kappa)
        minibatch = hmm.metaobs_fun(hmm.T, L, S)
        A_inter = np.zeros_like(hmm.var_tran)
        emit_inter = [np.zeros(V) for _ in range(K)]
        lb = 0.
        for data in minibatch:
            A_mean = hmm.var_tran / np.sum(hmm.var_tran, axis=1)[:, None]
            ew, ev = np.linalg.eig(A_mean.T)
            hmm.var_init = np.abs(ev[:, np.argsort(ew)[::-1][0]])
            hmm.local_update(metaobs=data)
            A_i, e_i = hmm.intermediate_pars(data)
            A_inter += A_i
            for k in range(K):
                emit_inter[k] += e_i[k]
            lb += hmm.local_lower_bound()
            wins.append(dict(i1=data.i1, i2=data.i2, lliks=hmm.lliks.copy(), lalpha=hmm.lalpha.copy(),
                             lbeta=hmm.lbeta.copy(), var_x=hmm.var_x.copy(), A_i=A_i.copy(),
                             e_i=np.array(e_i), local_lb=hmm.local_lower_bound(),
                             var_init=hmm.var_init.copy()))
        d = dict(A_inter=A_inter.copy(), emit_inter=np.array(emit_inter), lrate=hmm.lrate, lb=lb,
                 var_tran_old=hmm.var_tran.copy(), alpha_old=np.array([g.alpha_mf for g in hmm.var_emit]))
        hmm.global_update(A_inter, emit_inter)
        d["var_tran_new"] = hmm.var_tran.copy()
        d["alpha_new"] = np.array([g.alpha_mf for g in hmm.var_emit])
        d["weights_new"] = np.array([g.weights for g in hmm.var_emit])
        its.append(d)
    for key in wins[0]:
        rec["w_" + key] = np.array([w[key] for w in wins])
    for key in its[0]:
        rec["it_" + key] = np.array([d[key] for d in its])
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **rec)
    print("wrote", name, "raises:", rec["cat_intermediate_raises"], rec.get("cat_intermediate_error"))


def main(only=None):
    """`only`: regenerate just the fixtures whose names are given (the others stay untouched)."""
    sys.path.insert(0, REPO)
    tmp = tempfile.mkdtemp(prefix="pysvihmm_ref_")
    try:
        stage_reference(tmp)
        sys.path.insert(0, tmp)
        HB = importlib.import_module("hmmbase")
        HM = importlib.import_module("hmmsgd_metaobs")
        CD = importlib.import_module("hmmbatchcd")
        SG = importlib.import_module("hmmbatchsgd")
        if only:
            global trace_metaobs, trace_batch, trace_ffbs, trace_categorical
            keep = {f.__name__: f for f in (trace_metaobs, trace_batch, trace_ffbs, trace_categorical)}
            gate = lambda f: (lambda M, name, *a, **k: f(M, name, *a, **k) if name in only else None)
            trace_metaobs, trace_batch, trace_categorical = (gate(keep[n]) for n in
                                                             ("trace_metaobs", "trace_batch", "trace_categorical"))
            trace_ffbs = lambda HB_, HM_, name, *a, **k: keep["trace_ffbs"](HB_, HM_, name, *a, **k) if name in only else None
        else:
            for f in glob.glob(os.path.join(HERE, "*.npz")):
                os.remove(f)
        trace_metaobs(HM, "metaobs_K2_D2_L4", 2, 2, 200, 4, 3, 3, 0.0, SEED)
        trace_metaobs(HM, "metaobs_K4_D2_L10_mask", 4, 2, 400, 10, 4, 3, 0.1, SEED + 1)
        trace_metaobs(HM, "metaobs_K16_D8_L16", 16, 8, 600, 16, 3, 2, 0.05, SEED + 2)
        trace_metaobs(HM, "metaobs_K64_D32_L8", 64, 32, 500, 8, 2, 2, 0.0, SEED + 3)
        trace_batch(CD, "batchcd_K4_D2_T300", 4, 2, 300, 3, 0.1, SEED + 4, False)
        trace_batch(SG, "batchsgd_K4_D3_T250", 4, 3, 250, 3, 0.1, SEED + 5, True)
        trace_ffbs(HB, HM, "ffbs_K5_D3_T120", 5, 3, 120, SEED + 6)
        # adaptive window length (select_L, :521-569) and buffered meta-observations
        # (select_buffer :579-661, intermediate_pars_buffer :932-1008, buffer_budget :571-577)
        trace_metaobs(HM, "adaptive_K4_D2", 4, 2, 400, 3, 3, 4, 0.1, SEED + 7,
                      infer_kw=dict(adaptive=True, perIter=2, epsilon=1e-7, minHalfL=2, Lincrement=1,
                                    Lcutoff=30), probes=((57, 4), (200, 9)))
        trace_metaobs(HM, "growbuf_K4_D2", 4, 2, 400, 5, 3, 4, 0.1, SEED + 8,
                      ctor_kw=dict(growBuffer=True),
                      infer_kw=dict(perIter=2, epsilon=1e-3, Lincrement=2, Lcutoff=40), probes=((120, 7),))
        trace_metaobs(HM, "growbuf_budget_K3_D2", 3, 2, 500, 4, 5, 3, 0.0, SEED + 9,
                      ctor_kw=dict(growBuffer=True, bufferBudget=True),
                      infer_kw=dict(perIter=1, epsilon=1e-2, Lincrement=1, Lcutoff=25))
        HMA = importlib.import_module("hmmsgd_metaobs_asis")
        trace_categorical(HM, HMA, "categorical_K3_V5", 3, 5, 300, 6, 4, SEED + 10)
        # AdaGrad-scaled transition step (hmmsgd_metaobs.py:1036-1040), round 4
        trace_metaobs(HM, "adagrad_K4_D2", 4, 2, 400, 8, 4, 5, 0.05, SEED + 11, ctor_kw=dict(adagrad=True))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main(only=set(sys.argv[1:]) or None)
