"""The bench workload for the probe scripts: sequence generated in HBM on the given engine,
variational state of SURVEY.md 8d (same seeded stream as bench.py rank 0)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def bench_problem(eng, want_obs=False):
    rs, tran, means, chols = bench.true_process(0)
    eng.generate(tran, means, chols, bench.T, seed=bench.SEED)
    obs = eng.read_generated(want_sts=False)[0]
    pb = bench.variational_state(rs, means, obs[:20000])
    if want_obs:
        pb["obs"] = obs
    return pb
