#!/bin/bash
# HBM-side traffic of one training step: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: they do not fit one pass) over bench.py,
# per-kernel sums of the last full step -> gpurun_out/<tag>_{fetch,write}.json.   scripts/pmc_traffic.sh <tag> [bench args...]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/$tag
  timeout -s KILL 240 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/$tag -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras "$@" > $GRAFT_REPO_ROOT/gpurun_out/${tag}_$c.log 2>&1 || exit 1
  python3 $GRAFT_REPO_ROOT/scripts/pmc_step.py /tmp/$tag $c $GRAFT_REPO_ROOT/gpurun_out/${tag}_$c.json || exit 1
done
