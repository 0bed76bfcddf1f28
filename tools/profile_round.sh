#!/bin/bash
# Collect the rocprofv3 evidence for one profile set (run on the GPU box through gpurun):
#   tools/profile_round.sh r01e
# kernel trace + stats, FETCH_SIZE / WRITE_SIZE in separate PMC passes, SQ counters, bench line.
# Output: gpurun_out/<tag>_*.txt / .json (copy what should be judged into profiles/).
set -u
TAG=${1:-rXX}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# 2 blocks of 5 steps of the headline workload + the side figures (their kernels are told apart by
# their launch grids: 3891-window epoch step, 64-window minibatch / SVI iteration, whole chain,
# configs[4] K=256 D=64); PMC passes skip the side figures except the SVI iteration
BENCH="python $ROOT/bench.py --steps 5 --warmup 2 --reps 2 --no-cpu-baseline"
run() {  # name, rocprof args...
  local name=$1; shift
  rm -rf /tmp/prof_$name
  rocprofv3 "$@" -d /tmp/prof_$name -o $TAG -- $BENCH > /tmp/prof_$name.log 2>&1
  local db=$(find /tmp/prof_$name -name "*.db" | head -1)
  python $ROOT/tools/rocpd_summary.py $db
}
run kt --kernel-trace --stats > $OUT/${TAG}_kernel_stats.txt
python $ROOT/tools/svi_trace.py $(find /tmp/prof_kt -name "*.db" | head -1) > $OUT/${TAG}_svi_iteration_trace.txt
# HBM counters on the headline workload alone (no side figures), fp64 and fp32 mode separately
BENCH_ALL=$BENCH
BENCH="python $ROOT/bench.py --steps 5 --warmup 2 --reps 2 --no-cpu-baseline --no-side"
{ run fetch --kernel-trace --pmc FETCH_SIZE; run write --kernel-trace --pmc WRITE_SIZE; } > $OUT/${TAG}_pmc_hbm.txt
BENCH="python $ROOT/tools/f32_profile_workload.py"
{ run fetch32 --kernel-trace --pmc FETCH_SIZE; run write32 --kernel-trace --pmc WRITE_SIZE; } > $OUT/${TAG}_pmc_hbm_f32.txt
run sq32 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS > $OUT/${TAG}_sq_counters_f32.txt
# configs[4] (K=256, D=64): the wide kernels alone
BENCH="python $ROOT/tools/c5_profile_workload.py"
run kt5 --kernel-trace --stats > $OUT/${TAG}_c5_kernel_stats.txt
{ run fetch5 --kernel-trace --pmc FETCH_SIZE; run write5 --kernel-trace --pmc WRITE_SIZE; } > $OUT/${TAG}_c5_pmc_hbm.txt
run sq5 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS > $OUT/${TAG}_c5_sq_counters.txt
BENCH=$BENCH_ALL
run sq --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS > $OUT/${TAG}_sq_counters.txt
# the committed summaries bench.py reads (roofline.frac on the profiler's clock, roofline.traffic)
cd $ROOT/tools && python kernel_stats_to_json.py $OUT/${TAG}_kernel_stats.txt | sed "s#$OUT/#profiles/#g" > $OUT/kernel_stats.json
python pmc_to_traffic.py $OUT/${TAG}_pmc_hbm.txt $OUT/${TAG}_pmc_hbm_f32.txt | sed "s#$OUT/#profiles/#g" > $OUT/pmc_traffic.json
cp $OUT/kernel_stats.json $OUT/pmc_traffic.json $ROOT/profiles/ 2>/dev/null
cd $ROOT && python bench.py 2>/dev/null | tail -1 > $OUT/${TAG}_bench.json
tail -c 600 $OUT/${TAG}_bench.json
