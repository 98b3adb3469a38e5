#!/bin/bash
# usage: tools/kprof.sh <tag> <bench args...>  -- per-kernel average durations of a bench run (rocprofv3 kernel trace)
R=$GRAFT_REPO_ROOT; tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kprof_$tag -o bench -- python3 $R/bench.py --no-cpu-baseline --no-e2e "$@" > $R/gpurun_out/kprof_$tag.log 2>&1
python3 - <<PY
import csv
print("== $tag", "$*")
for r in csv.DictReader(open("$R/gpurun_out/kprof_$tag/bench_kernel_stats.csv")):
    if "rocclr" in r["Name"]: continue
    print("%-46s calls=%3s avg=%9.1f us  %5s%%" % (r["Name"].split("(")[0][:46], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
grep -a -o '"value": [0-9.]*' $R/gpurun_out/kprof_$tag.log
