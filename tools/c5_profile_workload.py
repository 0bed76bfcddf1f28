"""configs[4] epoch step alone (K=256, D=64, T=1e6, 3891 windows) for the rocprofv3 passes of
tools/profile_round.sh: kernel stats, FETCH_SIZE / WRITE_SIZE, SQ counters of the wide kernels
(k_emission_mfma<4,2,false>, k_scale_ll, k_sweeps_lin2<8,true>, k_lin_posterior,
k_stats_mfma4<5,2,2,5,...>, the transition-block launch)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from pysvihmm_amd.engine import HipEngine
from pysvihmm_amd.gen_synthetic import generate_data_fast
from pysvihmm_amd import _lib as L
Kw, Dw, T, LM = 256, 64, bench.T, bench.LM
rs = np.random.RandomState(bench.SEED + 4)
rng = np.random.default_rng(bench.SEED + 4)
tran = 0.9 * np.eye(Kw) + 0.1 / (Kw - 1) * (1.0 - np.eye(Kw))
means = rs.normal(0.0, 5.0, size=(Kw, Dw))
obs, _ = generate_data_fast(tran, means, None, T, rng)
pw = bench.variational_state(rs, means, obs[:20000], Kw, Dw, T)
e = HipEngine(0)
e.set_obs(obs, None)
st = np.arange(T // LM, dtype=np.int64) * LM
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    e.set_globals(pw["mod_init"], pw["ltran"])
    e.set_emission_niw(pw["mu"], pw["sigma"], pw["kappa"], pw["nu"], check=False)
    e.estep(st, LM, flags=L.TRANS_WRAP, read=False)
    out = e.read_packed()
assert abs(out.A_raw.sum() / (len(st) * LM) - 1.0) < 1e-9
# round 5: the same step in the fp32 mode (k_emission_bf16x3d<true>, k_scale_ll_f32, k_sweeps_lin2<..., float>,
# k_stats_bf16x3w) -- its kernels carry other names, so one trace holds both
e.set_precision("f32")
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    e.set_globals(pw["mod_init"], pw["ltran"])
    e.set_emission_niw(pw["mu"], pw["sigma"], pw["kappa"], pw["nu"], check=False)
    e.estep(st, LM, flags=L.TRANS_WRAP, read=False)
    out = e.read_packed()
assert e.precision()[1] and abs(out.A_raw.sum() / (len(st) * LM) - 1.0) < 1e-5
e.close()
