#!/bin/bash
# usage: tools/ktimeline.sh <tag> [bench args] -- kernel timeline of the two-job bench pass: start / end of every launch of one step
# (who runs beside whom), from a rocprofv3 kernel trace
R=$GRAFT_REPO_ROOT; tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/ktl_$tag -o bench -- python3 $R/bench.py --no-cpu-baseline --no-e2e --steps 4 --warmup 2 "$@" > $R/gpurun_out/ktl_$tag.log 2>&1
python3 - <<PY
import csv
rows = []
for r in csv.DictReader(open("$R/gpurun_out/ktl_$tag/bench_kernel_trace.csv")):
    n = r["Kernel_Name"]
    if "k_ht_" not in n and "k_idwt" not in n: continue
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), n.split("(")[0].replace("void htj2k::", "").replace("htj2k::", "")[:34], int(r["Grid_Size_X"])))
rows.sort()
# the timed two-job steps come first in the run: print launches 36 .. 72 (steps 3-5 of 2 jobs x 6 kernels)
t0 = rows[36][0]
for s, e, q, n, g in rows[36:72]:
    print("%9.1f %9.1f  %7.1f us  q%-3s %-34s grid %d" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, n, g))
PY
