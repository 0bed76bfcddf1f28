#!/usr/bin/env python3
"""bench.py -- SVI-HMM E-step throughput on MI355X (BASELINE.json metric).

Metric: obs-state updates/s = (time steps processed) x K / wall-seconds of one SVI
E-step, K=64 Gaussian HMM.

Workload (BASELINE.json configs[2], the configuration the metric is quoted on):
K=64, D=32 full-covariance NIW-Gaussian HMM, T=1,000,000 observations resident in
HBM, meta-observations of half-length L=128 (Lm=257).  One *step* = one SVI E-step
that touches every observation once: floor(T/Lm)=3891 tiled windows (the "T*K"
figure of SURVEY.md 8d): upload of the current globals (psi-expectations, NIW
factors) -> emission expected log-likelihood -> log-domain forward/backward ->
posteriors -> expected sufficient statistics -> [RCCL all-reduce at N>1] -> D2H of
the packed statistics.  The strict "minibatch=64" latency case (64 windows per
step) is reported beside it in "minibatch_s64".

Multi-GPU (configs[3]): one process per GPU, each with its own T=1M sequence
(seed 8675309+rank), weak scaling, one all-reduce of the packed statistics/step.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
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

K, D, T, LHALF = 64, 32, 1000000, 128
LM = 2 * LHALF + 1
SEED = 8675309
FP64_PEAK_TFLOPS = 78.6   # MI355X datasheet fp64 vector = matrix peak (guide has no fp64 row)
HBM_PEAK_GBS = 8000.0     # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec


def synth(rank):
    """Synthetic Gaussian-HMM sequence + variational state."""
    from scipy.special import digamma
    rng = np.random.default_rng(SEED + rank)
    means = rng.normal(0.0, 5.0, size=(K, D))
    # sticky chain (0.9 self-transition), start in state 0, vectorised by dwell times
    sts = np.empty(T, dtype=np.int64)
    t, cur = 0, 0
    while t < T:
        dwell = rng.geometric(0.1)
        sts[t:t + dwell] = cur
        t += dwell
        cur = (cur + 1 + rng.integers(0, K - 1)) % K
    obs = means[sts] + rng.normal(size=(T, D))
    var_tran = 1.0 + rng.random((K, K)) * T / K
    A_mean = var_tran / var_tran.sum(1)[:, None]
    ew, ev = np.linalg.eig(A_mean.T)
    var_init = np.abs(ev[:, np.argsort(ew)[::-1][0]]).real
    eps = 1e-9
    mod_init = digamma(var_init + eps) - digamma(var_init.sum() + eps)
    ltran = digamma(var_tran + eps) - digamma(var_tran.sum(1)[:, None] + eps)
    mu = means + rng.normal(size=(K, D))
    sigma0 = 0.75 * np.cov(obs[:20000].T)
    sig = np.empty((K, D, D))
    for k in range(K):
        a = rng.normal(size=(D, D))
        sig[k] = sigma0 + 0.1 * a.dot(a.T)
    kappa = 0.01 + 50.0 * rng.random(K)
    nu = D + 2 + 50.0 * rng.random(K)
    return dict(obs=obs, mod_init=mod_init, ltran=ltran, mu=mu, sigma=sig, kappa=kappa, nu=nu)


def algorithmic_flops(rows):
    """fp64 flops per launch of each kernel ('algorithmic work')."""
    F = (D + 1) * (D + 2) // 2            # 561 augmented features
    return {
        "emission": 2.0 * rows * F * K,             # ll = Phi[rows,F] . theta[F,K]
        "stats": 2.0 * rows * (F + K) * K,          # Phi^T q  and  q_prev^T q
        "forward_backward": 2.0 * rows * 2 * K * K,  # two K x K mat-vecs per row
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
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
    if use_comm:
        from pysvihmm_amd.comm import RcclComm, file_uid_exchange
        ex = file_uid_exchange(rank)
        comm = RcclComm(eng, rank, world, ex)

    pb = synth(rank)
    eng.set_obs(pb["obs"], None)            # resident in HBM before the timed region
    B = T // LM
    starts = np.arange(B, dtype=np.int64) * LM
    rows = B * LM

    def step(st=starts):
        eng.set_globals(pb["mod_init"], pb["ltran"])
        eng.set_emission_niw(pb["mu"], pb["sigma"], pb["kappa"], pb["nu"], check=False)
        eng.estep(st, LM, flags=L.TRANS_WRAP, read=False)
        if use_comm:
            eng.allreduce_packed()
        return eng.read_packed()

    def barrier():
        eng.sync()
        if comm is not None:
            comm.barrier(eng)
        eng.sync()

    for _ in range(args.warmup):
        step()
    eng.profile(True)
    eng.profile_reset()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier()
    dt = time.perf_counter() - t0
    prof = eng.profile_read()
    eng.profile(False)
    if comm is not None:
        dt = float(eng.allreduce_host(np.array([dt]), "max")[0])   # MAX over ranks
    assert np.all(np.isfinite(out.buf)), "non-finite statistics"
    # sanity: posteriors sum to one => wrap transition statistic sums to the row count
    tot_rows = rows * world
    assert abs(out.A_raw.sum() / tot_rows - 1.0) < 1e-9, out.A_raw.sum() / tot_rows

    ms_per_step = dt / args.steps * 1e3
    value = tot_rows * K * args.steps / dt

    # strict minibatch=64 latency case (same data, 64 windows/step), rank 0 view
    st64 = (np.arange(64, dtype=np.int64) * (T // 64)) % (T - LM)
    for _ in range(3):
        step(st64)
    barrier()
    t0 = time.perf_counter()
    n64 = max(args.steps, 20)
    for _ in range(n64):
        step(st64)
    barrier()
    dt64 = (time.perf_counter() - t0) / n64

    # whole-chain paths on the same resident sequence (world size 1 only: side figures, outside
    # the timed region above): the full-sequence E-step (one window of T rows) and FFBS
    chain = None
    if world == 1:
        DE = np.finfo(np.float64).eps
        t_fc = t_ff = 1e9
        for rep in range(3):
            eng.set_globals(pb["mod_init"], pb["ltran"])
            eng.sync(); t0 = time.perf_counter()
            eng.estep([0], T, flags=0, read=False); eng.read_packed()
            t_fc = min(t_fc, time.perf_counter() - t0)
        logA = np.log(np.exp(pb["ltran"]) + DE)
        u = np.random.default_rng(1).random(T)
        eng.set_globals(pb["mod_init"], logA)
        for rep in range(3):
            eng.sync(); t0 = time.perf_counter()
            eng.ffbs(logA, u, want_lalpha=False)
            t_ff = min(t_ff, time.perf_counter() - t0)
        chain = {"full_chain_estep": {"T": T, "ms": t_fc * 1e3, "value": T * K / t_fc, "unit": "updates/s",
                                      "note": "one window = the whole sequence (exact blocked scan), "
                                              "E-step + statistics"},
                 "ffbs": {"T": T, "ms": t_ff * 1e3, "value": T * K / t_ff, "unit": "updates/s",
                          "note": "forward filter (blocked scan) + backward sampling (composed "
                                  "draw maps), z[T] back on the host"}}
        eng.set_globals(pb["mod_init"], pb["ltran"])

    if rank == 0:
        flops = algorithmic_flops(rows)
        kern = {}
        for name, (ms, cnt) in prof.items():
            kern[name] = {"ms_per_launch": ms / cnt, "launches": cnt}
            if name in flops:
                kern[name]["tflops"] = flops[name] / (ms / cnt * 1e-3) / 1e12
        dom = max((k for k in kern if k in flops), key=lambda k: kern[k]["ms_per_launch"])
        achieved = kern[dom]["tflops"]
        traffic = None
        pmc_path = os.path.join(REPO, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc_path):
            try:
                traffic = json.load(open(pmc_path)).get(dom, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        # algorithmic HBM bytes of the whole step: obs read once + packed stats out
        alg_bytes = rows * D * 8.0 + PackedStats.size(K, D) * 8.0
        res = {
            "metric": "obs-state updates/sec (T*K/s) per SVI E-step, K=64 Gaussian HMM",
            "value": value, "unit": "updates/s", "n_gpus": args.gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "configs[2]: K=64 D=32 full-cov NIW-Gaussian HMM, T=1e6 "
                                   "resident, metaobs L=128 (Lm=257), one E-step over all "
                                   "3891 tiled windows (T*K=6.4e7 updates) per GPU per step",
                       "K": K, "D": D, "T": T, "Lm": LM, "windows_per_step": B,
                       "sequences": world, "parallelism": "windows sharded; 1 sequence/GPU"},
            "roofline": {"bound": "mfma", "kernel": dom, "achieved": achieved,
                         "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / FP64_PEAK_TFLOPS, "traffic": traffic,
                         "note": "fp64: v_mfma_f64_16x16x4_f64; peak = MI355X datasheet fp64 "
                                 "(matrix = vector = 78.6 TF); measured microbench ceiling on "
                                 "this part is lower, see DESIGN.md"},
            "roofline_hbm": {"bound": "hbm", "achieved": alg_bytes / (ms_per_step * 1e-3) / 1e9,
                             "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": alg_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             "note": "algorithmic bytes (obs read once + stats) / step time; the "
                                     "path is compute-bound (SURVEY.md 8d)"},
            "kernels": kern,
            "minibatch_s64": {"windows": 64, "ms_per_step": dt64 * 1e3,
                              "value": 64 * LM * K / dt64, "unit": "updates/s"},
        }
        if chain:
            res["whole_chain"] = chain
        if not args.no_cpu_baseline:
            # the reference algorithm restated in C (oracle/ref_c.c), 1 core, bounded sample
            from oracle import ref_c
            nwin = 320   # ~12 s of single-core work (the guidance asks for 10-30 s)
            t0 = time.perf_counter()
            ref = ref_c.estep_minibatch(pb["obs"], None, starts[:nwin], LM, pb["mod_init"],
                                        pb["ltran"], pb["mu"], pb["sigma"], pb["kappa"],
                                        pb["nu"], flags=2)
            cdt = time.perf_counter() - t0
            res["cpu_baseline"] = {
                "value": nwin * LM * K / cdt, "unit": "updates/s", "cores": 1, "kind": "port",
                "sample": "first %d of the %d windows of the same workload (%.1f s); plain-C "
                          "restatement of the reference's single-threaded K^2 log-add-exp "
                          "recursions, host has %d cores" % (nwin, B, cdt, os.cpu_count())}
            # and a cross-check of the GPU result on that sample
            chk = step(starts[:nwin]) if world == 1 else None
            if chk is not None:
                err = np.max(np.abs(chk.buf - ref) / (1e-9 + np.abs(ref)))
                res["cpu_baseline"]["gpu_vs_port_max_rel_err"] = float(err)
        print(json.dumps(res))
    if comm is not None:
        comm.barrier(eng)
        if rank == 0:
            try:
                os.remove(ex.path)
            except OSError:
                pass
    eng.close()


if __name__ == "__main__":
    main()
