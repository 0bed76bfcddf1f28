#!/usr/bin/env python3
"""bench.py -- SVI-HMM E-step throughput on MI355X (BASELINE.json metric).

Metric: obs-state updates/s = (time steps processed) x K / wall-seconds of one SVI
E-step, K=64 Gaussian HMM.

Workload (BASELINE.json configs[2], the configuration the metric is quoted on):
K=64, D=32 full-covariance NIW-Gaussian HMM, T=1,000,000 observations resident in
HBM (generated there by svihmm_generate: gen_synthetic.generate_data semantics, sticky
0.9 transitions, means ~ N(0, 25 I), unit covariances, seed 8675309 + rank),
meta-observations of half-length L=128 (Lm=257).  One *step* = one SVI E-step that touches
every observation once: floor(T/Lm)=3891 tiled windows: upload of the current globals
(psi-expectations, NIW factors) -> emission expected log-likelihood -> forward/backward
-> posteriors -> expected sufficient statistics -> [RCCL all-reduce at N>1] -> D2H of the
packed statistics.

Timing: W warm-up steps, then R repetitions of a block of EXACTLY K steps, each block
bracketed by barrier + stream synchronisation on both sides and reduced with MAX over
ranks; the reported ms_per_step / value are the MEDIAN block (every block is listed in
"ms_per_step_reps").

Multi-GPU (configs[3]): one process per GPU, each with its own T=1M sequence, weak scaling,
one RCCL all-reduce of the packed statistics per step.  Launch either with
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` (RANK /
LOCAL_RANK / WORLD_SIZE from the environment) or plainly as `python bench.py --gpus N`:
without WORLD_SIZE in the environment the script starts the N ranks itself (and fails if
fewer than N GPUs are visible).  "ranks" in the output is ncclCommCount of the communicator.

Side figures (world size 1, outside the headline timed region): the literal minibatch=64
E-step, one full SVI iteration through the class surface (hmmsgd_metaobs.VBHMM.infer: E-step +
global step + ELBO), the whole-chain E-step and FFBS, configs[4] (K=256, D=64), and the CPU
baselines (the reference algorithm restated in C on 1 core and on all cores, and its NumPy
restatement on 1 core) on bounded samples of the same workload.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)
# RCCL prints a version banner to stdout under NCCL_DEBUG=VERSION (set on the GPU boxes);
# the contract is ONE JSON line, so quieten it before librccl is loaded with the HIP library.
if os.environ.get("NCCL_DEBUG", "").upper() in ("VERSION", "INFO"):
    del os.environ["NCCL_DEBUG"]
os.environ.setdefault("NCCL_DEBUG_FILE", "/dev/stderr")   # whatever RCCL still says: not on stdout
# one node by contract (--gpus N of ONE node, rendezvous on 127.0.0.1): RCCL's bootstrap sockets on
# the loopback interface -- the GPU boxes have no network, and an unrelated interface must not be picked
os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")

K, D, T, LHALF = 64, 32, 1000000, 128
LM = 2 * LHALF + 1
SEED = 8675309            # the reference's own experiment seed (cluster/exper_run_simple.py:172)
FP64_PEAK_TFLOPS = 78.6   # MI355X datasheet fp64 vector = matrix peak (guide has no fp64 row)
HBM_PEAK_GBS = 8000.0     # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec
EPS = 1e-9


def true_process(rank):
    """True parameters of the synthetic sequence: legacy seeded stream."""
    rs = np.random.RandomState(SEED + rank)
    tran = 0.9 * np.eye(K) + 0.1 / (K - 1) * (1.0 - np.eye(K))
    means = rs.normal(0.0, 5.0, size=(K, D))
    chols = np.broadcast_to(np.eye(D), (K, D, D)).copy()
    return rs, tran, means, chols


def variational_state(rs, means, obs_head, Kq=K, Dq=D, Tq=T):
    """Variational state the timed E-step runs from: var_tran = 1 + U(0,1) T/K,
    NIW means = true means + N(0,1), scale matrices around 0.75 cov(obs)."""
    from scipy.special import digamma
    var_tran = 1.0 + rs.random_sample((Kq, Kq)) * Tq / Kq
    A_mean = var_tran / var_tran.sum(1)[:, None]
    pi = np.linalg.solve(A_mean.T - np.eye(Kq) + 1.0, np.ones(Kq))     # Perron vector (stationary init)
    var_init = pi / np.sqrt(pi.dot(pi))
    mod_init = digamma(var_init + EPS) - digamma(var_init.sum() + EPS)
    ltran = digamma(var_tran + EPS) - digamma(var_tran.sum(1)[:, None] + EPS)
    mu = means + rs.normal(size=(Kq, Dq))
    sigma0 = 0.75 * np.cov(obs_head.T)
    sig = np.empty((Kq, Dq, Dq))
    for k in range(Kq):
        a = rs.normal(size=(Dq, Dq))
        sig[k] = sigma0 + 0.1 * a.dot(a.T)
    kappa = 0.01 + 50.0 * rs.random_sample(Kq)
    nu = Dq + 2 + 50.0 * rs.random_sample(Kq)
    return dict(mod_init=mod_init, ltran=ltran, mu=mu, sigma=sig, kappa=kappa, nu=nu, sigma0=sigma0)


def algorithmic_flops(rows):
    """fp64 flops per launch of each kernel ('algorithmic work')."""
    F = (D + 1) * (D + 2) // 2            # 561 augmented features
    return {
        "emission": 2.0 * rows * F * K,             # ll = Phi[rows,F] . theta[F,K]
        "stats": 2.0 * rows * (F + K) * K,          # Phi^T q  and  q_prev^T q
        "forward_backward": 2.0 * rows * 2 * K * K,  # two K x K mat-vecs per row
    }


def effective_cores():
    """Host cores this process may actually use: the cgroup CPU quota when there is one (the GPU
    boxes expose 256 hardware threads but grant the container 16), else the affinity mask."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except Exception:
            pass
    return n


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def rank_envs(n, port, base=None):
    """Launch environment of each of the n ranks of one node (one process per GPU; rendezvous on
    127.0.0.1; dmabuf IPC for RCCL)."""
    envs = []
    for r in range(n):
        env = dict(os.environ if base is None else base, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SVIHMM_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        envs.append(env)
    return envs


def error_record(n_gpus, msg, **extra):
    """The ONE JSON line of a run that could not produce a measurement: same leading keys as a result
    line, value null, and the reason -- so that a failed N-GPU run leaves a record instead of nothing."""
    rec = {"metric": "obs-state updates/sec (T*K/s) per SVI E-step, K=64 Gaussian HMM", "value": None,
           "unit": "updates/s", "n_gpus": n_gpus, "higher_is_better": True, "error": msg}
    rec.update(extra)
    return rec


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks (one process per GPU)
    ourselves; rank 0's stdout (the JSON line) passes through.  Exit code = worst child.  A rank that
    dies takes the job down (its peers would wait in a collective for ever): the others get
    SVIHMM_PEER_GRACE_S seconds (default 20), then SIGTERM; if rank 0 printed no JSON line the launcher
    prints the error record with every rank's exit code."""
    # (first contact with a new box: a missing / unloadable libsvihmm_hip.so or a HIP runtime that reports no
    #  device must still leave the ONE JSON line, not a traceback)
    try:
        from pysvihmm_amd.engine import device_count
        ndev = device_count()
    except BaseException as e:
        if isinstance(e, KeyboardInterrupt):
            raise
        msg = "cannot count HIP devices: %r" % (e,)
        sys.stderr.write("bench.py: %s\n" % msg)
        print(json.dumps(error_record(args.gpus, msg + " -- 0 HIP device(s) usable")))
        sys.stdout.flush()
        return 2
    if ndev < args.gpus:
        sys.stderr.write("bench.py: --gpus %d but only %d HIP device(s) visible\n" % (args.gpus, ndev))
        print(json.dumps(error_record(args.gpus, "--gpus %d but only %d HIP device(s) visible" % (args.gpus, ndev))))
        return 2
    procs = []
    for r, env in enumerate(rank_envs(args.gpus, free_port())):
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=(r == 0)))
    import threading
    seen = {"json": False}

    def pump():
        for line in procs[0].stdout:
            if line.lstrip().startswith("{"):
                seen["json"] = True
            sys.stdout.write(line)
            sys.stdout.flush()
    th = threading.Thread(target=pump, daemon=True)
    th.start()
    grace = float(os.environ.get("SVIHMM_PEER_GRACE_S", "20"))
    failed_at = None
    while any(p.poll() is None for p in procs):
        if failed_at is None and any(p.poll() not in (None, 0) for p in procs):
            failed_at = time.time()
        if failed_at is not None and time.time() - failed_at > grace:
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            break
        time.sleep(0.05)
    codes = []
    for p in procs:
        try:
            codes.append(p.wait(timeout=15))
        except subprocess.TimeoutExpired:
            p.kill()
            codes.append(p.wait())
    th.join(timeout=5)
    rc = next((c for c in codes if c), 0)
    if not seen["json"]:
        print(json.dumps(error_record(args.gpus, "no result line from rank 0", rank_exit_codes=codes)))
        sys.stdout.flush()
        rc = rc or 1
    return rc


def packed_size(Kq, Dq):
    return Kq * Kq + Kq * Dq + Kq + Kq * Dq * Dq + 1


def headline_record(args, strong, world, ranks_seen, have_comm, block, per_rank, prof, rows, gen_ms, side):
    """The ONE JSON line from the measured blocks (seconds per block of args.steps steps, already
    MAX-reduced over ranks) and the HIP-event profile of the timed region."""
    dt = float(np.median(block))
    ms_per_step = dt / args.steps * 1e3
    tot_rows = rows if strong else rows * world
    value = tot_rows * K * args.steps / dt
    flops = algorithmic_flops(rows)
    nlaunch = args.reps * args.steps
    kern = {}
    for name, (ms, cnt) in prof.items():
        kern[name] = {"ms_per_launch": ms / cnt, "launches": cnt}
        if name in flops:
            kern[name]["tflops"] = flops[name] / (ms / cnt * 1e-3) / 1e12
    dom = max((k for k in kern if k in flops), key=lambda k: kern[k]["ms_per_launch"])
    achieved_ev = kern[dom]["tflops"]
    # counters and the profiler's clock come from the committed profile set (tools/profile_round.sh):
    # HBM bytes per launch / per step (FETCH_SIZE, WRITE_SIZE passes) and rocprofv3's average duration
    traffic = step_traffic = prof_us = None
    prof_src = pmc_src = None
    try:
        pj = json.load(open(os.path.join(REPO, "profiles", "pmc_traffic.json")))
        traffic = pj.get(dom, {}).get("hbm_bytes_per_launch")
        step_traffic = pj.get("_step_total", {}).get("hbm_bytes_per_step")
        pmc_src = pj.get("_note", "").split("source ")[-1] or None
    except Exception:
        pass
    try:
        kj = json.load(open(os.path.join(REPO, "profiles", "kernel_stats.json")))
        prof_us = kj.get(dom, {}).get("avg_us")
        prof_src = kj.get("_note", "").split("source ")[-1] or None
    except Exception:
        pass
    # the committed profile is only quoted when it still describes THIS build: same kernel function as the one
    # the run dispatched (svihmm_last_kernel_name) and an average duration within 10 % of the live HIP-event
    # figure; otherwise the line says "stale" and carries the live figure
    prof_kernel = None
    stale = None
    try:
        prof_kernel = kj.get(dom, {}).get("kernel")
    except Exception:
        pass
    live_kernel = (side or {}).get("_dispatched", {}).get(dom)
    ev_us = kern[dom]["ms_per_launch"] * 1e3
    if prof_us:
        if live_kernel and prof_kernel and live_kernel != prof_kernel:
            stale = "profile is of %s, this run dispatched %s" % (prof_kernel, live_kernel)
        elif abs(prof_us - ev_us) > 0.10 * ev_us:
            stale = "profile average %.1f us differs from this run's HIP-event average %.1f us by more than 10 %%" % (prof_us, ev_us)
    if stale:
        prof_us_used = None
    else:
        prof_us_used = prof_us
    achieved_prof = flops[dom] / (prof_us_used * 1e-6) / 1e12 if prof_us_used else None
    # algorithmic HBM bytes of the whole step: obs read once + packed stats out
    alg_bytes = rows * D * 8.0 + packed_size(K, D) * 8.0
    step_flops = sum(flops.values())
    # bytes one launch of the dominant kernel has to move: statistics = obs + ah + bh rows in, 128 x (F' + K) x K
    # partial sums out; emission = obs in, Eh out; sweeps = Eh in, ah + bh out
    kern_alg_bytes = {"stats": rows * (D + 2 * K) * 8.0 + 128 * 640 * K * 8.0,
                      "emission": rows * (D + K) * 8.0,
                      "forward_backward": rows * 3 * K * 8.0}.get(dom, alg_bytes)
    # roofline.frac is the figure a reader recomputes from profiles/ (rocprofv3's average of the
    # dominant kernel); the live HIP-event figure of THIS run is frac_events.  Without a committed
    # profile set the live figure is all there is.
    achieved = achieved_prof if achieved_prof else achieved_ev
    res = {
        "metric": "obs-state updates/sec (T*K/s) per SVI E-step, K=64 Gaussian HMM",
        "value": value, "unit": "updates/s", "n_gpus": args.gpus, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
        "scaling": args.scaling, "vs_baseline": None, "dtype": "f64",
        "data": "synthetic (generated in HBM by svihmm_generate, %.1f ms)" % gen_ms,
        "reps": args.reps, "timing": "median of %d blocks of %d steps (max over ranks per block)"
                                     % (args.reps, args.steps),
        "timed_region_s": float(np.sum(block)),
        "ms_per_step_reps": [float(b / args.steps * 1e3) for b in block],
        "ranks": ranks_seen, "ranks_source": "ncclCommCount" if have_comm else "no communicator",
        "config": {"workload": "configs[2]: K=64 D=32 full-cov NIW-Gaussian HMM, T=1e6 "
                               "resident, metaobs L=128 (Lm=257), one E-step over all "
                               "3891 tiled windows (T*K=6.4e7 updates) per GPU per step",
                   "K": K, "D": D, "T": T, "Lm": LM, "windows_per_step": rows // LM,
                   "sequences": 1 if strong else world,
                   "parallelism": ("one sequence resident on every GPU, windows[rank::N] per rank, one "
                                   "all-reduce of the packed statistics per step") if strong else
                                  "windows sharded; 1 sequence/GPU"},
        "roofline": {"bound": "mfma", "kernel": dom, "achieved": achieved,
                     "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": achieved / FP64_PEAK_TFLOPS,
                     "frac_source": ("rocprofv3 --kernel-trace --stats average of the kernel, %s (%.1f us); checked against "
                                     "this run: same kernel function, HIP-event average %.1f us" % (prof_src, prof_us, ev_us))
                                    if achieved_prof else
                                    ("stale: %s -- frac is this run's HIP-event figure" % stale) if stale else
                                    "HIP events of this run (no profiles/kernel_stats.json)",
                     "kernel_function": live_kernel or prof_kernel,
                     "achieved_events": achieved_ev, "frac_events": achieved_ev / FP64_PEAK_TFLOPS,
                     "whole_step_frac": step_flops / (ms_per_step * 1e-3) / 1e12 / FP64_PEAK_TFLOPS,
                     # two scopes, each with its own pair: `traffic` / `traffic_ratio` = the dominant kernel's HBM
                     # bytes per launch (PMC counters) over the bytes that launch has to move (its inputs obs + ah + bh
                     # once, its partials out); `traffic_step` / `traffic_ratio_step` = all kernels of one step over
                     # the step's algorithmic bytes (obs in once + packed statistics out)
                     "traffic": traffic,
                     "traffic_ratio": (traffic / kern_alg_bytes) if traffic else None,
                     "traffic_scope": "dominant kernel, bytes per launch; *_step: whole step (all kernels)",
                     "traffic_step": step_traffic,
                     "traffic_ratio_step": (step_traffic / alg_bytes) if step_traffic else None,
                     "clock": "frac: the profiler's clock (reproducible from profiles/); frac_events: HIP events "
                              "around every launch of the kernel in the timed region (%d launches; a few %% "
                              "shorter than under the profiler; only this kernel is bracketed there, the other "
                              "entries of `kernels` come from a short separate pass); whole_step_frac: algorithmic "
                              "flop of emission + sweeps + statistics / ms_per_step / peak" % nlaunch,
                     "note": "fp64: v_mfma_f64_16x16x4_f64; peak = MI355X datasheet fp64 "
                             "(matrix = vector = 78.6 TF); traffic_ratio = counter bytes / "
                             "algorithmic bytes of the same scope (%s)" % pmc_src},
        "roofline_hbm": {"bound": "hbm", "achieved": alg_bytes / (ms_per_step * 1e-3) / 1e9,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": alg_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "traffic_per_step": step_traffic,
                         "note": "algorithmic bytes (obs read once + stats) / step time; the "
                                 "path is compute-bound (SURVEY.md 8d)"},
        "kernels": kern,
    }
    if per_rank is not None:
        res["per_rank_ms_per_step"] = per_rank
    if "allreduce" in kern:
        res["allreduce"] = {"ms_per_step": kern["allreduce"]["ms_per_launch"], "bytes": packed_size(K, D) * 8,
                            "note": "ncclAllReduce(sum, f64) of the packed statistics on the handle's stream "
                                    "(HIP events around the call)"}
    res.update({k: v for k, v in (side or {}).items() if not k.startswith("_")})
    res["roofline"]["regimes"] = regimes(side or {})
    return res


def regimes(side):
    """Numbers only, under `roofline` (the driver's record keeps sub-keys of `roofline` and `config`): the
    other regimes the side figures measure -- the literal minibatch = 64 E-step, the S=64 SVI iteration through
    the class surface (fp64 and fp32 mode), the fp32-mode epoch step, configs[4]; ms per step / iteration and the
    fraction of the fp64 peak (fp32-mode entries: of the bf16 dense peak for the kernels named in the side
    record), plus the wall time of ONE infer(maxit=100) call with its fixed part broken down."""
    def num(x):
        return float(x) if isinstance(x, (int, float)) and np.isfinite(x) else None
    out = {}
    r = side.get("minibatch_s64") or {}
    if "ms_per_step" in r:
        out["minibatch_s64"] = {"ms": num(r["ms_per_step"]), "frac": num(r.get("roofline", {}).get("frac"))}
    for key in ("svi_iteration_s64", "svi_iteration_s64_f32"):
        r = side.get(key) or {}
        if "ms" in r:
            out[key] = {"ms": num(r["ms"]), "frac": num(r.get("roofline", {}).get("frac")),
                        "iter_time_median_ms": num(r.get("iter_time_median_ms"))}
            if "infer_wall_ms_maxit100" in r:
                out[key]["infer_wall_ms_maxit100"] = num(r["infer_wall_ms_maxit100"])
                out[key]["infer_wall_breakdown_ms"] = {k: num(v) for k, v in (r.get("infer_wall_breakdown_ms") or {}).items()}
            if "max_rel_err_vs_f64_final_var_tran" in r:
                out[key]["max_rel_err_vs_f64"] = num(r["max_rel_err_vs_f64_final_var_tran"])
    r = side.get("f32_mode") or {}
    if "ms_per_step" in r:
        out["f32_mode"] = {"ms": num(r["ms_per_step"]),
                           "frac_bf16_emission": num(r.get("roofline_emission", {}).get("frac")),
                           "frac_bf16_stats": num(r.get("roofline_stats", {}).get("frac")),
                           "max_rel_err_vs_f64": num(r.get("max_rel_err_vs_f64_statistics"))}
    r = side.get("c5_k256_d64") or {}
    if "ms" in r:
        out["c5_k256_d64"] = {"ms": num(r["ms"]), "frac": num(r.get("roofline", {}).get("frac")),
                              "f32_ms": num((r.get("f32_mode") or {}).get("ms")),
                              "f32_max_rel_err_vs_f64": num((r.get("f32_mode") or {}).get("max_rel_err_vs_f64_statistics"))}
    r = side.get("c2_k16_d8") or {}
    if "epoch_step" in r:
        out["c2_k16_d8"] = {"epoch_ms": num(r["epoch_step"]["ms"]), "full_chain_ms": num(r["full_chain_estep"]["ms"]),
                            "ffbs_ms": num(r["ffbs_fast"]["ms"])}
    return out


REQUIRED = {"metric": str, "value": float, "unit": str, "n_gpus": int, "steps": int, "warmup": int,
            "ms_per_step": float, "higher_is_better": bool, "scaling": str, "dtype": str, "data": str,
            "config": dict, "roofline": dict, "ranks": int}


def validate_record(res, n_gpus):
    """Schema of the line the driver parses (contract of the task + the multi-GPU extras)."""
    for k, t in REQUIRED.items():
        assert k in res, "missing key %s" % k
        assert isinstance(res[k], t), "%s: %r is not %s" % (k, res[k], t.__name__)
    assert "vs_baseline" in res and res["vs_baseline"] is None
    assert res["n_gpus"] == n_gpus and res["ranks"] == n_gpus
    assert "workload" in res["config"] and "model" not in res["config"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in res["roofline"], "roofline.%s" % k
    if n_gpus > 1:
        assert len(res["per_rank_ms_per_step"]) == n_gpus
        assert "ms_per_step" in res["allreduce"]
        ss = res["strong_scaling"]
        for leg in ("epoch", "minibatch_s64"):
            for k in ("windows", "windows_per_rank", "ms_per_step", "value", "allreduce_ms_per_step"):
                assert k in ss[leg], "strong_scaling.%s.%s" % (leg, k)
    json.loads(json.dumps(res))
    return True


def dry_run(args):
    """`bench.py --gpus N --dry-run`: everything of an N-rank run that needs no device -- the N launch
    environments (one process per GPU, loopback rendezvous, rank / local-rank / world-size, dmabuf
    IPC), the window partitions of both scaling modes, and the record assembly on synthetic timings,
    validated against the schema.  Prints the (synthetic, marked) line; exit code 0 = consistent."""
    n = args.gpus
    envs = rank_envs(n, free_port(), base={})
    assert len(envs) == n
    ports = {e["MASTER_PORT"] for e in envs}
    assert len(ports) == 1 and all(e["MASTER_ADDR"] == "127.0.0.1" for e in envs)
    assert [int(e["RANK"]) for e in envs] == list(range(n)) == [int(e["LOCAL_RANK"]) for e in envs]
    assert all(int(e["WORLD_SIZE"]) == n and e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" for e in envs)
    B = T // LM
    all_starts = np.arange(B, dtype=np.int64) * LM
    st64 = (np.arange(64, dtype=np.int64) * (T // 64)) % (T - LM)
    for st in (all_starts, st64):       # strong scaling: the ranks' shares tile the batch exactly once
        parts = [st[r::n] for r in range(n)]
        assert sum(len(p) for p in parts) == len(st)
        assert np.array_equal(np.sort(np.concatenate(parts)), np.sort(st))
    rows = B * LM
    strong = args.scaling == "strong"
    # synthetic timings shaped like a real run (NOT measurements)
    rng = np.random.default_rng(0)
    block = args.steps * 3.2e-3 * (1.0 + 0.01 * rng.random(args.reps))
    prof = {"emission": (1.16 * args.reps * args.steps, args.reps * args.steps),
            "forward_backward": (0.45 * args.reps * args.steps, args.reps * args.steps),
            "stats": (1.35 * args.reps * args.steps, args.reps * args.steps)}
    side = {}
    per_rank = None
    if n > 1:
        prof["allreduce"] = (0.05 * args.reps * args.steps, args.reps * args.steps)
        per_rank = [3.2] * n
        leg = lambda w: {"windows": int(w), "windows_per_rank": int((w + n - 1) // n), "ms_per_step": 1.0,
                         "value": 1.0, "unit": "updates/s", "scaling": "strong", "allreduce_ms_per_step": 0.05}
        if not strong and not args.no_strong:
            side["strong_scaling"] = {"epoch": leg(B), "minibatch_s64": leg(64), "note": "synthetic"}
    res = headline_record(args, strong, n, n, n > 1, block, per_rank, prof, rows, 0.0, side)
    res["data"] = "DRY RUN: synthetic timings, no device touched -- not a measurement"
    if n > 1 and strong:
        res["strong_scaling"] = {"epoch": leg(B), "minibatch_s64": leg(64)}     # (schema check only)
    validate_record(res, n)
    print(json.dumps(res))
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--reps", type=int, default=0,
                    help="repetitions of the --steps block (median reported); default: as many as make the "
                         "headline region span >= 5 s of GPU time (80 blocks of 20 steps at ~3.1 ms), so that "
                         "a coarse utilisation sampler sees it")
    ap.add_argument("--dry-run", action="store_true",
                    help="no devices: build every rank's launch environment for --gpus N, run the record "
                         "assembly on synthetic timings and validate the schema of the line (tests)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak (default, BASELINE configs[3]): one independent T=1M sequence per GPU; strong: ONE "
                         "sequence, the epoch's 3891 windows dealt windows[rank::N] (north_star: 'minibatches shard "
                         "across the GPUs'), one all-reduce per step.  At N>1 the weak run also reports the strong "
                         "figure as a side record")
    ap.add_argument("--no-strong", action="store_true", help="skip the strong-scaling side record of an N>1 weak run")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-side", action="store_true", help="skip the side figures (profiling runs)")
    args = ap.parse_args()

    if args.reps <= 0:
        args.reps = max(10, int(np.ceil(5.0 / (max(args.steps, 1) * 3.2e-3))))
    if args.dry_run:
        raise SystemExit(dry_run(args))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(launch_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # A rank that fails (communicator init, a device error, a peer that went away and made the launcher
    # terminate us) still leaves ONE JSON line on rank 0's stdout: the error record.
    state = {"printed": False}
    if rank == 0 and world > 1:
        import signal

        def on_term(signum, frame):
            if not state["printed"]:
                print(json.dumps(error_record(args.gpus, "terminated by the launcher (signal %d): a peer rank failed "
                                                         "or the job was cancelled" % signum)))
                sys.stdout.flush()
            os._exit(143)
        signal.signal(signal.SIGTERM, on_term)
    try:
        run_rank(args, state)
    except BaseException as e:
        if rank == 0 and not state["printed"] and not isinstance(e, KeyboardInterrupt):
            msg = str(e) if isinstance(e, SystemExit) else repr(e)
            print(json.dumps(error_record(args.gpus, msg, rank=rank)))
            sys.stdout.flush()
        raise


def run_rank(args, state):
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus))

    # the product path: HIP library through the C ABI (raises if unavailable)
    from pysvihmm_amd.engine import HipEngine, PackedStats
    from pysvihmm_amd import _lib as L
    eng = HipEngine(local_rank)

    # N>1: one process per GPU, RCCL over xGMI through the C ABI.  No torch in the
    # process: the ncclUniqueId is exchanged through a file keyed by the launcher's pid,
    # barriers and the max-over-ranks timing are RCCL all-reduces on the handle's stream
    # (+ hipStreamSynchronize) -- the same bracket as dist.barrier()+cuda.synchronize().
    use_comm = world > 1 or os.environ.get("SVIHMM_FORCE_COMM") == "1"  # (world-1 rehearsal)
    comm = None
    ranks_seen = 1
    if use_comm:
        from pysvihmm_amd.comm import RcclComm, file_uid_exchange
        # bounded rendezvous: a rank that never shows up must not hold the others past the driver's patience
        ex = file_uid_exchange(rank, timeout=float(os.environ.get("SVIHMM_RENDEZVOUS_TIMEOUT", "90")))
        comm = RcclComm(eng, rank, world, ex)
        ranks_seen = eng.comm_count()
        if ranks_seen != world:
            raise SystemExit("RCCL communicator has %d ranks, expected %d" % (ranks_seen, world))

    # the sequence: generated in HBM (the path's own f3 row), resident before the timed region
    # (strong scaling: every rank holds the SAME sequence and works on its share of the windows)
    strong = args.scaling == "strong"
    seq = 0 if strong else rank
    rs, tran, means, chols = true_process(seq)
    t0 = time.perf_counter()
    eng.generate(tran, means, chols, T, seed=SEED + seq)
    eng.sync()
    gen_ms = (time.perf_counter() - t0) * 1e3
    need_host_obs = rank == 0 and world == 1 and not (args.no_side and args.no_cpu_baseline)
    if need_host_obs:
        obs_host, sts_host = eng.read_generated()
        head = obs_host[:20000]
    else:
        obs_host = sts_host = None
        head = eng.read_generated(want_sts=False)[0][:20000]
    pb = variational_state(rs, means, head)
    B = T // LM
    all_starts = np.arange(B, dtype=np.int64) * LM
    starts = all_starts[rank::world] if strong else all_starts
    rows = B * LM

    def step(st=starts):
        # (NIW factors first: their upload -> Cholesky -> theta chain is what the emission GEMM
        #  waits for; the globals are only needed by the sweeps after it)
        eng.set_emission_niw(pb["mu"], pb["sigma"], pb["kappa"], pb["nu"], check=False)
        eng.set_globals(pb["mod_init"], pb["ltran"])
        eng.estep(st, LM, flags=L.TRANS_WRAP, read=False)
        if use_comm:
            eng.allreduce_packed()
        return eng.read_packed()

    def barrier():
        eng.sync()
        if comm is not None:
            comm.barrier(eng)
        eng.sync()

    # warm-up with HIP events around EVERY kernel: finds the dominant kernel (events between two
    # dependent kernels cost dispatch time -- 0.055 ms per 3.15 ms step with all slots,
    # profiles/r04a_overlap_probe.txt -- so the timed region brackets only that one)
    eng.profile(True)
    eng.profile_reset()
    for _ in range(max(args.warmup, 1)):
        step()
    wprof = eng.profile_read()
    eng.profile(False)
    fl0 = algorithmic_flops(B * LM)
    dom_slot = max((k for k in wprof if k in fl0), key=lambda k: wprof[k][0] / wprof[k][1])
    only = [dom_slot] + (["allreduce"] if use_comm else [])
    eng.profile(True, only=only)
    eng.profile_reset()
    block = np.zeros(args.reps)
    for r in range(args.reps):
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = step()
        barrier()
        block[r] = time.perf_counter() - t0
    prof = eng.profile_read()
    eng.profile(False)
    dispatched = {}
    if hasattr(eng, "last_kernel"):
        dispatched = {dom_slot: eng.last_kernel(dom_slot)}
    # the other kernels of the step: a short separate pass with every slot bracketed (outside the
    # timed region; the dominant kernel keeps its timed-region figure)
    eng.profile(True)
    eng.profile_reset()
    for _ in range(min(args.steps, 10)):
        step()
    allprof = eng.profile_read()
    eng.profile(False)
    for k, v in allprof.items():
        prof.setdefault(k, v)
    mine = block.copy()
    per_rank = None
    if comm is not None:
        block = eng.allreduce_host(block, "max")          # MAX over ranks, block by block
        slots = np.zeros(world)
        slots[rank] = np.median(mine) / args.steps * 1e3
        per_rank = [float(v) for v in eng.allreduce_host(slots, "sum")]
    assert np.all(np.isfinite(out.buf)), "non-finite statistics"
    # sanity: posteriors sum to one => wrap transition statistic sums to the row count
    tot_rows = rows if strong else rows * world
    assert abs(out.A_raw.sum() / tot_rows - 1.0) < 1e-9, out.A_raw.sum() / tot_rows

    side = {}
    if world == 1 and not args.no_side:
        side = side_figures(eng, L, pb, step, barrier, obs_host, args)
    side["_dispatched"] = dispatched
    if comm is not None and not strong and not args.no_strong:
        # the other partitioning of DESIGN 6, measured in the same job: ONE sequence on every GPU
        # (rank 0's), the epoch's windows and the S=64 minibatch dealt round-robin over the ranks
        side["strong_scaling"] = strong_scaling(eng, L, comm, rank, world, args, barrier)

    if rank == 0:
        res = headline_record(args, strong, world, ranks_seen, comm is not None, block, per_rank, prof, rows,
                              gen_ms, side)
        print(json.dumps(res))
        sys.stdout.flush()
        state["printed"] = True
    if comm is not None:
        comm.barrier(eng)
        if rank == 0:
            try:
                os.remove(ex.path)
            except OSError:
                pass
    eng.close()


def strong_scaling(eng, L, comm, rank, world, args, barrier):
    """One sequence, windows dealt over the ranks (DESIGN 6 mode (i)).  Every rank regenerates
    rank 0's sequence in its own HBM (counter-based generator: bit-identical copies)."""
    rs, tran, means, chols = true_process(0)
    eng.generate(tran, means, chols, T, seed=SEED)
    head = eng.read_generated(want_sts=False)[0][:20000]
    pb = variational_state(rs, means, head)
    B = T // LM
    all_starts = np.arange(B, dtype=np.int64) * LM
    st64 = (np.arange(64, dtype=np.int64) * (T // 64)) % (T - LM)
    out = {}
    for name, st_all, reps in (("epoch", all_starts, 5), ("minibatch_s64", st64, 5)):
        mine = st_all[rank::world]

        def step():
            eng.set_emission_niw(pb["mu"], pb["sigma"], pb["kappa"], pb["nu"], check=False)
            eng.set_globals(pb["mod_init"], pb["ltran"])
            eng.estep(mine, LM, flags=L.TRANS_WRAP, read=False)     # (an empty shard yields zero statistics)
            eng.allreduce_packed()
            return eng.read_packed()
        for _ in range(3):
            res = step()
        eng.profile(True, only=["allreduce"]); eng.profile_reset()
        blocks = np.zeros(reps)
        for r in range(reps):
            barrier()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                res = step()
            barrier()
            blocks[r] = time.perf_counter() - t0
        prof = eng.profile_read(); eng.profile(False)
        blocks = eng.allreduce_host(blocks, "max")
        dt = float(np.median(blocks)) / args.steps
        nrows = len(st_all) * LM
        assert abs(res.A_raw.sum() / nrows - 1.0) < 1e-9          # every rank holds the statistics of ALL windows
        ar = prof.get("allreduce", (0.0, 0))
        out[name] = {"windows": int(len(st_all)), "windows_per_rank": int(len(mine)), "ms_per_step": dt * 1e3,
                     "value": nrows * K / dt, "unit": "updates/s", "scaling": "strong",
                     "allreduce_ms_per_step": ar[0] / max(ar[1], 1)}
    out["note"] = ("one T=1e6 sequence resident on all %d GPUs, windows[rank::%d] per rank, one all-reduce per step; "
                   "value = rows of the WHOLE batch x K / time (max over ranks)" % (world, world))
    return out


def median_time(fn, sync, n, warm=2):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(n):
        sync()
        t0 = time.perf_counter()
        fn()
        sync()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts))


def side_figures(eng, L, pb, step, barrier, obs_host, args):
    """World size 1 only, outside the headline timed region."""
    res = {}
    # 1. the literal minibatch=64 E-step (configs[2]: mb_sz=64), same resident data
    st64 = (np.arange(64, dtype=np.int64) * (T // 64)) % (T - LM)
    for _ in range(3):
        step(st64)
    n64 = max(args.steps, 20)
    blocks = []
    for _ in range(5):
        barrier()
        t0 = time.perf_counter()
        for _ in range(n64):
            step(st64)
        barrier()
        blocks.append((time.perf_counter() - t0) / n64)
    dt64 = float(np.median(blocks))
    fl64 = sum(algorithmic_flops(64 * LM).values())
    eng.profile(True); eng.profile_reset()
    for _ in range(10):
        step(st64)
    p64 = eng.profile_read(); eng.profile(False)
    k64 = {k: v[0] / v[1] for k, v in p64.items()}
    f64k = algorithmic_flops(64 * LM)
    res["minibatch_s64"] = {"windows": 64, "ms_per_step": dt64 * 1e3, "value": 64 * LM * K / dt64,
                            "unit": "updates/s", "note": "E-step + statistics of 64 windows (engine calls only)",
                            "kernels_ms": k64,
                            "roofline": {"bound": "mfma", "kernel": "whole step (emission + sweeps + statistics)",
                                         "achieved": fl64 / dt64 / 1e12, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                                         "frac": fl64 / dt64 / 1e12 / FP64_PEAK_TFLOPS,
                                         "per_kernel_frac": {k: f64k[k] / (k64[k] * 1e-3) / 1e12 / FP64_PEAK_TFLOPS
                                                             for k in f64k if k in k64},
                                         "note": "algorithmic fp64 flop of 64 windows x 257 rows / wall; a 257-step dependent "
                                                 "chain at one wave per SIMD: latency-, not throughput-bound (DESIGN 4, 'The S = 64 iteration')"}}

    # 1b. the same epoch step in the fp32 mode (scaled messages stored as float, statistics GEMM on
    #     v_mfma_f32_16x16x4_f32, emission as the centred quadratic form on the bf16 matrix pipe with
    #     three-term operands = fp32 accuracy; the sweeps' arithmetic stays fp64):
    #     a second figure beside the fp64 headline, never the headline
    try:
        ref64 = step().buf.copy()
        eng.set_precision("f32")
        for _ in range(3):
            out32 = step()
        used = eng.precision()[1]
        blocks = []
        for _ in range(5):          # (timed without kernel events, like the headline: they cost ~0.05 ms per step)
            barrier()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                out32 = step()
            barrier()
            blocks.append((time.perf_counter() - t0) / args.steps)
        eng.profile(True); eng.profile_reset()
        for _ in range(min(args.steps, 10)):
            step()
        p32 = eng.profile_read(); eng.profile(False)
        dt32 = float(np.median(blocks))
        rows = (T // LM) * LM
        scale = np.maximum(np.abs(ref64), 1e-6 * rows)
        res["f32_mode"] = {"ms_per_step": dt32 * 1e3, "value": rows * K / dt32, "unit": "updates/s",
                           "dtype": "f32 storage of Eh/ah/bh; centred emission AND the statistics GEMM on bf16 MFMA with every operand as three bf16 terms (fp32 accumulators = fp32 arithmetic); f64 recursion arithmetic",
                           "ran_in_f32_format": bool(used),
                           "max_rel_err_vs_f64_statistics": float(np.max(np.abs(out32.buf - ref64) / scale)),
                           "kernels_ms": {k: v[0] / v[1] for k, v in p32.items()}}
        # the mode's emission kernel (k_emission_bf16x3): 18 v_mfma_f32_32x32x16_bf16 per pair of states and
        # 32-row tile = 9216 K flop per row on the bf16 pipe; the centred triangular form it evaluates is
        # K D (D + 1) flop per row of fp32 arithmetic
        em = p32.get("emission")
        if em and em[1] > 0 and K <= 64 and D <= 32:
            ems = em[0] / em[1] * 1e-3
            res["f32_mode"]["roofline_emission"] = {
                "bound": "mfma", "kernel": "k_emission_bf16x3", "achieved": rows * K * 9216.0 / ems / 1e12,
                "peak": 2500.0, "unit": "TFLOP/s", "frac": rows * K * 9216.0 / ems / 1e12 / 2500.0,
                "fp32_equivalent_tflops": rows * K * D * (D + 1.0) / ems / 1e12,
                "note": "bf16 dense MFMA peak (MI355X_MICROARCH.md); x and U as three bf16 terms, six products in fp32 "
                        "accumulators; fp32_equivalent = the centred triangular quadratic form's K D (D+1) flop per row "
                        "(fp32-input MFMA peak: 157 TF/s)"}
        # the statistics GEMM of the mode (k_stats_bf16x3): 20 feature tiles x 2 state tiles x 6 products of
        # v_mfma_f32_32x32x16_bf16 per 16 rows on the bf16 pipe; 2 (F + K) K flop per row of fp32 arithmetic
        sk = p32.get("stats")
        if sk and sk[1] > 0 and K == 64 and D == 32:
            sms = sk[0] / sk[1] * 1e-3
            Fq = (D + 1) * (D + 2) // 2
            res["f32_mode"]["roofline_stats"] = {
                "bound": "mfma", "kernel": "k_stats_bf16x3", "achieved": rows / 16.0 * 20 * 12 * 32768.0 / sms / 1e12,
                "peak": 2500.0, "unit": "TFLOP/s", "frac": rows / 16.0 * 20 * 12 * 32768.0 / sms / 1e12 / 2500.0,
                "fp32_equivalent_tflops": rows * 2.0 * (Fq + K) * K / sms / 1e12,
                "note": "bf16 dense MFMA peak; both operands as three bf16 terms, six products in fp32 accumulators; "
                        "fp32_equivalent = 2 (F + K) K flop per row (fp32-input MFMA peak: 157 TF/s)"}
        # the mode's tolerance (north_star: 1e-3) holds in the bench itself
        assert res["f32_mode"]["max_rel_err_vs_f64_statistics"] < 1e-3, res["f32_mode"]["max_rel_err_vs_f64_statistics"]
    except AssertionError:
        raise
    except Exception as e:
        res["f32_mode"] = {"error": repr(e)}
    finally:
        eng.set_precision("f64")

    # 2. whole-chain paths on the same resident sequence
    DE = np.finfo(np.float64).eps
    eng.set_emission_niw(pb["mu"], pb["sigma"], pb["kappa"], pb["nu"])

    def full_chain():
        eng.set_globals(pb["mod_init"], pb["ltran"])
        eng.estep([0], T, flags=0, read=False)
        eng.read_packed()
    t_fc = median_time(full_chain, eng.sync, 5, warm=1)
    logA = np.log(np.exp(pb["ltran"]) + DE)
    u = np.random.default_rng(1).random(T)
    eng.set_globals(pb["mod_init"], logA)
    t_ff = median_time(lambda: eng.ffbs(logA, u, want_lalpha=False), eng.sync, 5, warm=1)
    res["whole_chain"] = {
        "full_chain_estep": {"T": T, "ms": t_fc * 1e3, "value": T * K / t_fc, "unit": "updates/s",
                             "note": "one window = the whole sequence (exact blocked scan), E-step + statistics"},
        "ffbs": {"T": T, "ms": t_ff * 1e3, "value": T * K / t_ff, "unit": "updates/s",
                 "note": "forward filter (blocked scan) + backward sampling (composed draw maps), "
                         "z[T] back on the host"}}
    eng.set_globals(pb["mod_init"], pb["ltran"])

    # 3. CPU baselines on bounded samples of the headline workload (before the wide model
    #    replaces the resident sequence); the GPU-vs-port cross-checks inside are ASSERTED
    if not args.no_cpu_baseline and obs_host is not None:
        res.update(cpu_baselines(eng, L, pb, step, obs_host))

    # 4. one SVI iteration through the class surface (north_star: "local_update/global_update
    #    loop"): hmmsgd_metaobs.VBHMM.infer, E-step + global natural-gradient step + ELBO.  The
    #    class shares the handle (it uploads its own copy of the sequence); the C ABI is
    #    coordinate-invariant, and the step after it is checked against the one before.
    if obs_host is not None:
        before = step().buf.copy()
        try:
            res["svi_iteration_s64"] = svi_iteration(eng, obs_host)
        except Exception as e:       # a side figure must not take the headline down
            res["svi_iteration_s64"] = {"error": repr(e)}
        # the same loop in the fp32 mode (the loop enters the fp32 format once var_tran's lower bound allows:
        # from its second iteration for the class's default initialisation); final state compared with the
        # fp64 run's at the mode's tolerance
        try:
            eng.set_precision("f32")
            r32 = svi_iteration(eng, obs_host)
            r32["ran_in_f32_format"] = bool(eng.precision()[1])
            r64 = res["svi_iteration_s64"]
            if "_final" in r64 and "_final" in r32:
                a, b = r32.pop("_final"), r64["_final"]
                r32["max_rel_err_vs_f64_final_var_tran"] = float(np.max(np.abs(a - b) / (np.abs(b) + 1e-6 * np.abs(b).max())))
                assert r32["max_rel_err_vs_f64_final_var_tran"] < 1e-3, r32["max_rel_err_vs_f64_final_var_tran"]
            res["svi_iteration_s64_f32"] = r32
        except AssertionError:
            raise
        except Exception as e:
            res["svi_iteration_s64_f32"] = {"error": repr(e)}
        finally:
            eng.set_precision("f64")
            res.get("svi_iteration_s64", {}).pop("_final", None)
        eng.set_obs(obs_host, None)
        after = step().buf
        drift = float(np.max(np.abs(after - before) / (1e-9 + np.abs(before))))
        res.setdefault("svi_iteration_s64", {})["raw_step_after_class_max_rel_diff"] = drift
        assert drift < 1e-9, "raw E-step on the shared handle changed after the class ran: %g" % drift

    # 4b. configs[1]: K=16, D=8, T=1e5 -- epoch step, full-chain E-step, ffbs_fast beside the C port's
    #     Cython-variant forward filter on one core (replaces the resident sequence)
    try:
        res["c2_k16_d8"] = config1_record(eng, L, with_cpu=not args.no_cpu_baseline)
    except Exception as e:
        res["c2_k16_d8"] = {"error": repr(e)}
    # 5. configs[4]: K=256, D=64 full covariance, T=1e6, epoch sweep of 3891 windows
    try:
        res["c5_k256_d64"] = wide_model(eng, L)
    except Exception as e:
        res["c5_k256_d64"] = {"error": repr(e)}
    return res


def svi_iteration(eng, obs_host):
    from pysvihmm_amd import hmmsgd_metaobs
    from pysvihmm_amd.distributions import Gaussian
    head = obs_host[:20000]
    mu0, sg0 = head.mean(0), 0.75 * np.cov(head.T)
    np.random.seed(0)
    prior = np.array([Gaussian(mu_0=mu0, sigma_0=sg0, kappa_0=0.01, nu_0=D + 2) for _ in range(K)])

    def run(maxit):
        hmm = hmmsgd_metaobs.VBHMM(obs_host, np.ones(K), np.ones((K, K)), prior, tau=1.0, kappa=0.7,
                                   metaobs_half=LHALF, mb_sz=64, maxit=maxit, seed=1, engine=eng)
        t0 = time.perf_counter()
        hmm.infer()
        return time.perf_counter() - t0, hmm
    run(5)
    # (the call's fixed part -- uploads, the final read-back -- is ~12 ms and jitters by several ms from call to
    #  call: 2000 iterations between the two lengths and medians instead of minima keep that below 0.003 ms per
    #  iteration; tools/svi_wall_vs_device.py shows the wall time linear in maxit with the device's own
    #  per-iteration times as slope.  Rounds 2-5 differenced infer(70) - infer(10): +-0.02 ms of noise.)
    n1, n2 = 100, 2100
    r1 = [run(n1) for _ in range(3)]
    t1 = float(np.median([t[0] for t in r1]))
    # the call a user of the reference's default maxit = 100 sees, and where its fixed part goes
    wall_keys = ("upload_obs", "svi_begin", "submit_iterations", "wait_and_read_state", "last_window")
    wall100 = {k: float(np.median([getattr(t[1], "infer_wall_ms", {}).get(k, 0.0) for t in r1])) for k in wall_keys}
    wall100["iterations_device"] = float(np.median([np.sum(t[1].iter_time) * 1e3 for t in r1]))
    t2s = [run(n2) for _ in range(3)]
    t2 = float(np.median([t[0] for t in t2s]))
    hmm = t2s[-1][1]
    per_it = (t2 - t1) / (n2 - n1)
    assert np.all(np.isfinite(hmm.elbo_vec))
    fl = sum(algorithmic_flops(64 * LM).values())
    return {"ms": per_it * 1e3, "value": 64 * LM * K / per_it, "unit": "updates/s",
            "iter_time_median_ms": float(np.median(hmm.iter_time[5:]) * 1e3),
            "infer_wall_ms_maxit100": t1 * 1e3, "infer_wall_breakdown_ms": wall100,
            "infer_wall_note": "one infer(maxit=100) call, median of 3: upload_obs = the reference's unconditional re-read "
                               "of self.obs (256 MB pageable host -> HBM; `assume_obs_unchanged = True` skips it), svi_begin = "
                               "prior / factor uploads + first theta + globals, submit_iterations = host time of the 100 "
                               "svihmm_svi_iteration calls (the device runs behind), wait_and_read_state = rest of the device "
                               "time + state / ELBO read-back, last_window = lliks / lalpha / lbeta / var_x of the last window; "
                               "iterations_device = sum of iter_time (device stamps)",
            "_final": hmm.var_tran.copy(),
            "roofline": {"bound": "mfma", "kernel": "whole iteration (E-step kernels' algorithmic flop / wall)",
                         "achieved": fl / per_it / 1e12, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": fl / per_it / 1e12 / FP64_PEAK_TFLOPS,
                         "note": "five dependent launches on a 257-step chain: latency-bound (DESIGN 4, 'The S = 64 iteration')"},
            "note": "hmmsgd_metaobs.VBHMM.infer, mb_sz=64, L=128: wall per iteration (minibatch sampling + "
                    "E-step + global step + ELBO), from infer(maxit=%d) - infer(maxit=%d); iter_time = "
                    "E-step + global step as the reference clocks it" % (n2, n1)}


def cpu_baselines(eng, L, pb, step, obs_host):
    """The reference algorithm on the host cores, bounded samples of the same workload."""
    from oracle import ref_c, ref_numpy
    res = {}
    B = T // LM
    starts = np.arange(B, dtype=np.int64) * LM
    ncore = effective_cores()
    nhw = os.cpu_count() or 1
    par = (pb["mod_init"], pb["ltran"], pb["mu"], pb["sigma"], pb["kappa"], pb["nu"])
    # (i) plain-C port, 1 core
    nwin = 80    # ~3 s of single-core work
    t0 = time.perf_counter()
    ref = ref_c.estep_minibatch(obs_host, None, starts[:nwin], LM, *par, flags=2)
    cdt = time.perf_counter() - t0
    chk = step(starts[:nwin])
    err = float(np.max(np.abs(chk.buf - ref) / (1e-9 + np.abs(ref))))
    res["cpu_baseline"] = {
        "value": nwin * LM * K / cdt, "unit": "updates/s", "cores": 1, "kind": "port", "port_kind": "c_port",
        "sample": "first %d of the %d windows of the same workload (%.1f s); plain-C restatement of the "
                  "reference's single-threaded K^2 log-add-exp recursions; the container may use %d of "
                  "the host's %d hardware threads" % (nwin, B, cdt, ncore, nhw),
        "gpu_vs_port_max_rel_err": err}
    assert err < 1e-6, "GPU vs C port on %d windows: max rel err %g" % (nwin, err)
    # (ii) the same port with the windows dealt to all host cores (OpenMP)
    nthr = ncore
    nwin_all = min(B, max(nthr * 250, 320))      # <= 10 s at 37 ms per window and core
    t0 = time.perf_counter()
    ref_all = ref_c.estep_minibatch(obs_host, None, starts[:nwin_all], LM, *par, flags=2, threads=nthr)
    adt = time.perf_counter() - t0
    res["cpu_baseline_all_cores"] = {
        "value": nwin_all * LM * K / adt, "unit": "updates/s", "cores": nthr, "kind": "port", "port_kind": "c_port",
        "sample": "first %d windows (%.1f s), OpenMP over windows on %d threads = the container's CPU "
                  "quota (host: %d hardware threads)" % (nwin_all, adt, nthr, nhw)}
    if nwin_all >= B:
        chk = step()
        err_all = float(np.max(np.abs(chk.buf - ref_all) / (1e-9 + np.abs(ref_all))))
        res["cpu_baseline_all_cores"]["gpu_vs_port_max_rel_err_all_windows"] = err_all
        assert err_all < 1e-6, "GPU vs C port on all %d windows: max rel err %g" % (B, err_all)
    # (iii) the NumPy restatement (the reference's own expressions), 1 core
    nwin_np = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 2.0 and nwin_np < B:
        s = int(starts[nwin_np])
        ll = ref_numpy.lliks_niw(obs_host[s:s + LM], pb["mu"], pb["sigma"], pb["kappa"], pb["nu"])
        la = ref_numpy.forward_msgs(ll, pb["mod_init"], pb["ltran"])
        lb = ref_numpy.backward_msgs(ll, pb["ltran"])
        q = ref_numpy.posterior(la, lb)
        ref_numpy.transition_stat_wrap(q)
        for k in range(K):
            ref_numpy.niw_suffstats(obs_host[s:s + LM], q[:, k])
        ref_numpy.local_lower_bound(la)
        nwin_np += 1
    ndt = time.perf_counter() - t0
    res["cpu_baseline_numpy"] = {
        "value": nwin_np * LM * K / ndt, "unit": "updates/s", "cores": 1, "kind": "port",
        "port_kind": "numpy_restatement (reference expressions: SURVEY 8(d)(i)'s 'reference CPU path')",
        "sample": "first %d windows (%.1f s); NumPy restatement of the reference's local_update + "
                  "intermediate_pars expressions (np.logaddexp.reduce folds, np.outer loop)" % (nwin_np, ndt)}
    return res


def cpu_baseline_ffbs(obs, p1, mi_cy, logA, la_gpu, z_gpu, K1, T1):
    """configs[1]'s CPU leg: the C port's expected log-likelihoods + forward filter in the Cython variant
    (hmm_fast.pyx:74-93) on ONE host core, and the check of the GPU's lalpha against it (asserted)."""
    from oracle import ref_c
    t0 = time.perf_counter()
    ll = ref_c.lliks_niw(obs, p1["mu"], p1["sigma"], p1["kappa"], p1["nu"])
    t1 = time.perf_counter()
    la = ref_c.forward(ll, mi_cy, logA)
    t2 = time.perf_counter()
    err = float(np.max(np.abs(la_gpu - la) / (1.0 + np.abs(la))))
    assert err < 1e-9, "ffbs lalpha vs the C port's Cython-variant forward filter: %g" % err
    assert z_gpu.shape == (T1,) and z_gpu.min() >= 0 and z_gpu.max() < K1
    return {"value": T1 * K1 / (t2 - t0), "unit": "updates/s", "cores": 1, "kind": "port",
            "ms": (t2 - t0) * 1e3, "ms_lliks": (t1 - t0) * 1e3, "ms_forward_filter": (t2 - t1) * 1e3,
            "sample": "the whole T=1e5 sequence: expected log-likelihoods + the forward filter of "
                      "hmm_fast.pyx:80-93 (K^2 log-add-exp per step, log(var_tran + DBL_EPSILON)) in plain C on "
                      "one core; the O(T K) backward sampling pass is not included",
            "gpu_lalpha_vs_port_max_rel_err": err}


def config1_record(eng, L, with_cpu=True):
    """BASELINE configs[1]: K=16, D=8 full-covariance NIW-Gaussian HMM, T=1e5, "single-GPU E-step vs
    Cython hmm_fast": the windowed epoch step, the full-chain E-step (`full_local_update`) and
    `ffbs_fast` (forward filter + backward sampling, `z` and `lalpha` back on the host), with the C port's
    forward filter IN THE CYTHON VARIANT -- mod_init with DBL_EPSILON, transitions log(var_tran +
    DBL_EPSILON) un-normalised, /root/reference/hmm_fast.pyx:74-93 -- timed on one host core beside it
    and used as the check of the returned lalpha (asserted)."""
    from scipy.special import digamma
    K1, D1, T1 = 16, 8, 100000
    DE = np.finfo(np.float64).eps
    rs = np.random.RandomState(SEED + 1)
    tran = 0.9 * np.eye(K1) + 0.1 / (K1 - 1) * (1.0 - np.eye(K1))
    means = rs.normal(0.0, 5.0, size=(K1, D1))
    chols = np.broadcast_to(np.eye(D1), (K1, D1, D1)).copy()
    eng.generate(tran, means, chols, T1, seed=SEED + 1)
    obs = eng.read_generated(want_sts=False)[0]
    p1 = variational_state(rs, means, obs[:20000], K1, D1, T1)
    B1 = T1 // LM
    st = np.arange(B1, dtype=np.int64) * LM
    eng.set_emission_niw(p1["mu"], p1["sigma"], p1["kappa"], p1["nu"])

    def epoch():
        eng.set_emission_niw(p1["mu"], p1["sigma"], p1["kappa"], p1["nu"], check=False)
        eng.set_globals(p1["mod_init"], p1["ltran"])
        eng.estep(st, LM, flags=L.TRANS_WRAP, read=False)
        return eng.read_packed()

    def full_chain():
        eng.set_globals(p1["mod_init"], p1["ltran"])
        eng.estep([0], T1, flags=0, read=False)
        return eng.read_packed()
    out = epoch()
    assert np.all(np.isfinite(out.buf)) and abs(out.A_raw.sum() / (B1 * LM) - 1.0) < 1e-9
    t_ep = median_time(epoch, eng.sync, 20, warm=3)
    fc = full_chain()
    assert np.all(np.isfinite(fc.buf)) and abs(fc.neff.sum() / T1 - 1.0) < 1e-9
    t_fc = median_time(full_chain, eng.sync, 10, warm=2)
    # ffbs_fast as hmmbase.ffbs_fast issues it (Cython variant of the globals)
    var_tran = 1.0 + rs.random_sample((K1, K1)) * T1 / K1
    var_init = rs.random_sample(K1) + 0.1
    mi_cy = digamma(var_init + DE) - digamma(var_init.sum() + DE)
    logA = np.log(var_tran + DE)
    u = np.random.default_rng(2).random(T1)
    holder = {}

    def ffbs():
        eng.set_globals(mi_cy, logA)
        holder["z"], holder["la"] = eng.ffbs(logA, u, want_lalpha=True)
    t_ff = median_time(ffbs, eng.sync, 10, warm=2)
    res = {"K": K1, "D": D1, "T": T1, "Lm": LM,
           "epoch_step": {"windows": B1, "ms": t_ep * 1e3, "value": B1 * LM * K1 / t_ep, "unit": "updates/s",
                          "note": "parameter upload + E-step over all 389 tiled windows + statistics read-back"},
           "full_chain_estep": {"ms": t_fc * 1e3, "value": T1 * K1 / t_fc, "unit": "updates/s",
                                "note": "full_local_update: one window = the whole sequence (exact blocked scan), "
                                        "E-step + statistics"},
           "ffbs_fast": {"ms": t_ff * 1e3, "value": T1 * K1 / t_ff, "unit": "updates/s",
                         "note": "hmm_fast.FFBS semantics: forward filter + backward sampling, z[T] and lalpha[T,K] "
                                 "(12.8 MB) back on the host"}}
    if with_cpu:
        res["cpu_baseline_ffbs"] = cpu_baseline_ffbs(obs, p1, mi_cy, logA, holder["la"], holder["z"], K1, T1)
    return res


def wide_model(eng, L):
    """configs[4]: K=256, D=64 full-covariance NIW-Gaussian HMM, T=1e6 (host-generated)."""
    from pysvihmm_amd.gen_synthetic import generate_data_fast
    Kw, Dw = 256, 64
    rs = np.random.RandomState(SEED + 4)
    rng = np.random.default_rng(SEED + 4)
    tran = 0.9 * np.eye(Kw) + 0.1 / (Kw - 1) * (1.0 - np.eye(Kw))
    means = rs.normal(0.0, 5.0, size=(Kw, Dw))
    obs, _ = generate_data_fast(tran, means, None, T, rng)
    pw = variational_state(rs, means, obs[:20000], Kw, Dw, T)
    eng.set_obs(obs, None)
    del obs
    Bw = T // LM
    st = np.arange(Bw, dtype=np.int64) * LM

    def one():
        eng.set_globals(pw["mod_init"], pw["ltran"])
        eng.set_emission_niw(pw["mu"], pw["sigma"], pw["kappa"], pw["nu"], check=False)
        eng.estep(st, LM, flags=L.TRANS_WRAP, read=False)
        return eng.read_packed()
    out = one()
    rows = Bw * LM
    assert np.all(np.isfinite(out.buf)) and abs(out.A_raw.sum() / rows - 1.0) < 1e-9
    eng.profile(True)
    eng.profile_reset()
    n = 5
    dt = median_time(one, eng.sync, n, warm=1)
    prof = eng.profile_read()
    eng.profile(False)
    nstep = n + 1                       # steps since profile_reset (one warm-up + n timed)
    F = (Dw + 1) * (Dw + 2) // 2
    fl = rows * (2.0 * F * Kw + 2.0 * (F + Kw) * Kw + 2 * 2.0 * Kw * Kw)
    rec = {"ms": dt * 1e3, "value": rows * Kw / dt, "unit": "updates/s", "tflops": fl / dt / 1e12,
           "K": Kw, "D": Dw, "T": T, "Lm": LM, "windows_per_step": Bw,
           "kernels_ms": {k: v[0] / nstep for k, v in prof.items()},
           "kernel_launches_per_step": {k: v[1] / nstep for k, v in prof.items()},
           "roofline": {"bound": "mfma", "kernel": "whole step (fp64 MFMA kernels)", "achieved": fl / dt / 1e12,
                        "peak": 78.6, "unit": "TFLOP/s", "frac": fl / dt / 1e12 / 78.6},
           "note": "configs[4] epoch sweep, fp64; tflops = algorithmic MFMA flops (emission 2FK + statistics "
                   "2(F+K)K + sweeps 4K^2 per row) / wall; kernels_ms = device time per STEP summed over the "
                   "slot's launches (emission = GEMM + scaling pass, stats = feature + transition blocks)"}
    # the same step in the fp32 mode (round 5: k_emission_bf16x3d, k_scale_ll_f32, k_sweeps_lin2<float>,
    # k_stats_bf16x3w); north_star's fp32 tolerance asserted against the fp64 statistics of this run
    try:
        ref64 = out.buf.copy()
        eng.set_precision("f32")
        o32 = one()
        used = eng.precision()[1]
        eng.profile(True); eng.profile_reset()
        dt32 = median_time(one, eng.sync, n, warm=1)
        p32 = eng.profile_read(); eng.profile(False)
        scale = np.maximum(np.abs(ref64), 1e-6 * rows)
        err = float(np.max(np.abs(o32.buf - ref64) / scale))
        # bf16 MFMA work issued: emission 60 v_mfma_f32_32x32x16_bf16 per pair of states and 32-row tile, statistics
        # (68 feature + 8 transition tiles) x 8 state tiles x 6 per 16 rows; fp32-equivalent = the arithmetic the
        # mode replaces: K D (D + 1) + 2 (F + K) K + 4 K^2 flop per row
        bf_em = rows / 32.0 * (Kw / 2) * 60 * 32768.0
        bf_st = rows / 16.0 * ((F + 31) // 32 + Kw // 32) * (Kw // 32) * 6 * 32768.0
        kms = {k: v[0] / nstep for k, v in p32.items()}
        rec["f32_mode"] = {"ms": dt32 * 1e3, "value": rows * Kw / dt32, "unit": "updates/s", "ran_in_f32_format": bool(used),
                           "max_rel_err_vs_f64_statistics": err, "kernels_ms": kms,
                           "fp32_equivalent_tflops": rows * (Kw * Dw * (Dw + 1.0) + 2.0 * (F + Kw) * Kw + 4.0 * Kw * Kw) / dt32 / 1e12,
                           "roofline": {"bound": "mfma", "kernel": "k_emission_bf16x3d + k_stats_bf16x3w (bf16 pipe)",
                                        "achieved": (bf_em + bf_st) / ((kms.get("emission", 0) + kms.get("stats", 0)) * 1e-3 + 1e-30) / 1e12,
                                        "peak": 2500.0, "unit": "TFLOP/s",
                                        "frac": (bf_em + bf_st) / ((kms.get("emission", 0) + kms.get("stats", 0)) * 1e-3 + 1e-30) / 1e12 / 2500.0}}
        assert err < 1e-3, err
    except AssertionError:
        raise
    except Exception as e:
        rec["f32_mode"] = {"error": repr(e)}
    finally:
        eng.set_precision("f64")
    return rec


if __name__ == "__main__":
    main()
