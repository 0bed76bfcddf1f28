#!/bin/bash
# rocprofv3 kernel trace of the 64-window device-resident SVI loop (tools/r4_svi_probe.py with the
# given arguments, e.g. "f32 5:4"): timeline of one iteration + the loop's period.  Through gpurun.
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_t
rocprofv3 --kernel-trace -d /tmp/prof_t -o t -- python $GRAFT_REPO_ROOT/tools/r4_svi_probe.py "$@" > /tmp/prof_t.log 2>&1
DB=$(find /tmp/prof_t -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/svi_trace.py $DB
python $GRAFT_REPO_ROOT/tools/svi_iteration_period.py $DB
