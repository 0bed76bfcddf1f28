#!/bin/bash
# round 6: emission k-loop without branches -- parity of everything that runs the 16-row emission tile, iteration timing, trace
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_theta_split.py tests/test_gpu_fused.py tests/test_gpu_emission_orbit.py tests/test_gpu_classes.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -5 > $OUT/r06i_pytest.log
cat $OUT/r06i_pytest.log
for v in "" "" "f32"; do timeout 200 python tools/r4_svi_probe.py $v 2>&1 | tail -2; done
timeout 300 python tools/r6_fused_check.py 2>&1 | head -4
bash tools/r5_svi_trace.sh > $OUT/r06i_svi_iteration_trace.txt 2>&1
grep -E "k_emission|k_sweep_stats|k_svi_step|k_finalize|period|k_svi_globals" $OUT/r06i_svi_iteration_trace.txt | head -20
