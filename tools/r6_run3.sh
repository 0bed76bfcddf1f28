#!/bin/bash
# round 6: emission tiles inside the fused E-step launch -- parity, timing both ways, stamps, the S = 64 iteration
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_fused.py -x -q -m gpu 2>&1 | tail -15 > $OUT/r06g_pytest.log
cat $OUT/r06g_pytest.log
timeout 300 python tools/r6_fused_check.py > $OUT/r06g_fused_check.txt 2>&1
cat $OUT/r06g_fused_check.txt
timeout 120 python tools/r6_fused_trace.py 64 3 > $OUT/r06g_fused_trace.txt 2>&1
cat $OUT/r06g_fused_trace.txt
timeout 300 python tools/r4_svi_probe.py 2>&1 | tail -12 > $OUT/r06g_svi_probe.txt
cat $OUT/r06g_svi_probe.txt
