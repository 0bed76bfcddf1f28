#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_fused.py -x -q -m gpu 2>&1 | tail -15 > $OUT/r06g_pytest.log
cat $OUT/r06g_pytest.log
timeout 120 python tools/r6_fused_trace.py 64 3 > $OUT/r06g_fused_trace.txt 2>&1
cat $OUT/r06g_fused_trace.txt
timeout 300 python tools/r6_fused_check.py 2>&1 | head -4 > $OUT/r06g_fused_check.txt
cat $OUT/r06g_fused_check.txt
