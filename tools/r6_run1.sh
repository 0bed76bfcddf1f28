#!/bin/bash
# round 6, first GPU contact: LDS probe (timing + counters), the loop-hardening tests, a bench line
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
$ROOT/tools/probe/lds_probe > $OUT/r06a_lds_probe.txt 2>&1
rm -rf /tmp/prof_lds
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS -d /tmp/prof_lds -o lds -- $ROOT/tools/probe/lds_probe > /tmp/prof_lds.log 2>&1
python $ROOT/tools/rocpd_summary.py $(find /tmp/prof_lds -name "*.db" | head -1) > $OUT/r06a_lds_probe_pmc.txt 2>&1
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_classes.py -x -q -m gpu 2>&1 | tail -15 > $OUT/r06a_pytest_classes.log
timeout 600 python bench.py 2> $OUT/r06a_bench.err | tail -1 > $OUT/r06a_bench.json
tail -c 1500 $OUT/r06a_bench.json
cat $OUT/r06a_pytest_classes.log
cat $OUT/r06a_lds_probe.txt
