#!/bin/bash
# rocprofv3 kernel stats of one bench.py run on the GPU box; copies the stats CSV and the kernel sequence of the last full step
# (scripts/step_trace.py) back - the raw trace is large and stays on the box.
#   scripts/prof_bench.sh <tag> [bench args...]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/$tag -o p -- python3 $GRAFT_REPO_ROOT/bench.py "$@" > $GRAFT_REPO_ROOT/gpurun_out/$tag.log 2>&1
rc=$?
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/$tag
find /tmp/$tag -name "*stats*.csv" -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/$tag/ \;
tr=$(find /tmp/$tag -name "*kernel_trace.csv" | head -1)
[ -n "$tr" ] && python3 $GRAFT_REPO_ROOT/scripts/step_trace.py $tr $GRAFT_REPO_ROOT/gpurun_out/$tag/step_trace.txt
grep '"metric"' $GRAFT_REPO_ROOT/gpurun_out/$tag.log | cut -c1-260
exit $rc
