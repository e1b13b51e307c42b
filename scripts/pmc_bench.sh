#!/bin/bash
# one rocprofv3 --pmc pass over bench.py; copies the per-kernel summary back.   scripts/pmc_bench.sh <tag> "<counters>" [bench args...]
tag=$1; ctr=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/$tag -o p -- python3 $GRAFT_REPO_ROOT/bench.py "$@" > $GRAFT_REPO_ROOT/gpurun_out/$tag.log 2>&1
rc=$?
python3 $GRAFT_REPO_ROOT/scripts/pmc_summary.py /tmp/$tag > $GRAFT_REPO_ROOT/gpurun_out/$tag.txt
exit $rc
