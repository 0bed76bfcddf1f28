#!/bin/bash
# rocprofv3 kernel statistics of the 64-window device-resident SVI loop (tools/r4_svi_probe.py):
# the kernels of the iteration chain with their average durations.  Run on the GPU box through gpurun.
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_w
rocprofv3 --kernel-trace --stats -d /tmp/prof_w -o w -- python $GRAFT_REPO_ROOT/tools/r4_svi_probe.py > /tmp/prof_w.log 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $(find /tmp/prof_w -name "*.db" | head -1) | grep -E "k_wave_lin4|k_emission_orbit<4, 4, 1>|k_niw_to_theta_wave<32>|k_stats_mfma4.*grid +206|k_finalize" | head
