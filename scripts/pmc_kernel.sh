#!/bin/bash
# Counters of ONE kernel of the training step: one rocprofv3 --pmc pass per quoted counter group over a short bench.py run, restricted
# to the kernels matching <regex>; prints the per-dispatch mean of every counter (last 3 dispatches).
#   scripts/pmc_kernel.sh <tag> <kernel regex> "<counters of pass 1>" ["<counters of pass 2>" ...]   (bench args via BENCH_ARGS)
tag=$1; rx=$2; shift 2
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/${tag}_pmc.txt
: > $out
for group in "$@"; do
  rm -rf /tmp/$tag
  timeout -s KILL 240 rocprofv3 --kernel-trace --pmc $group --kernel-include-regex "$rx" --output-format csv -d /tmp/$tag -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras $BENCH_ARGS > $GRAFT_REPO_ROOT/gpurun_out/${tag}_pass.log 2>&1 || { echo "pass failed: $group"; tail -5 $GRAFT_REPO_ROOT/gpurun_out/${tag}_pass.log; exit 1; }
  python3 - /tmp/$tag >> $out <<'PY'
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
per = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    per.setdefault(r["Counter_Name"], collections.OrderedDict()).setdefault(int(r["Dispatch_Id"]), [r["Kernel_Name"][:60], 0.0])[1] += float(r["Counter_Value"])
for c, d in per.items():
    last = list(d.values())[-3:]
    print(f"{c:34s} {sum(v for _, v in last) / len(last):16.1f}   ({len(d)} dispatches; {last[-1][0]})")
PY
done
cat $out
