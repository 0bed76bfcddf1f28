#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_classes.py tests/test_gpu_diag.py tests/test_categorical.py tests/test_gpu_fused.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|assert" | head -5
for v in "" "" "f32"; do timeout 200 python tools/r4_svi_probe.py $v 2>&1 | tail -2; done
bash tools/r5_svi_trace.sh > $OUT/r06j_svi_iteration_trace.txt 2>&1
grep -E "k_emission|k_sweep_stats|k_svi_step|k_finalize|period|k_svi_globals|k_svi_vlb|k_svi_elbo|span" $OUT/r06j_svi_iteration_trace.txt | head -20
