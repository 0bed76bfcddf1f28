#!/bin/bash
# round 6: LDS layout fixes -- parity suites that touch the three kernels, bench, SQ counters of the epoch step
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_lin_sweeps.py tests/test_gpu_emission_orbit.py tests/test_gpu_f32.py tests/test_gpu_configs.py -x -q -m gpu 2>&1 | tail -8 > $OUT/r06b_pytest.log
cat $OUT/r06b_pytest.log
timeout 600 python bench.py --no-cpu-baseline 2> $OUT/r06b_bench.err | tail -1 > $OUT/r06b_bench.json
python - <<'PY'
import json
r = json.loads(open("gpurun_out/r06b_bench.json").read())
print("ms_per_step", r["ms_per_step"], "kernels", {k: round(v["ms_per_launch"], 4) for k, v in r["kernels"].items()})
print(json.dumps(r["roofline"]["regimes"]))
PY
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_sq
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS -d /tmp/prof_sq -o sq -- python $ROOT/bench.py --steps 5 --warmup 2 --reps 2 --no-cpu-baseline --no-side > /tmp/prof_sq.log 2>&1
python $ROOT/tools/rocpd_summary.py $(find /tmp/prof_sq -name "*.db" | head -1) > $OUT/r06b_sq_counters.txt 2>&1
grep -E 'k_stats_mfma4|k_sweeps_lin|k_emission_orbit' $OUT/r06b_sq_counters.txt | head -40
