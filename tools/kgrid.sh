#!/bin/bash
# usage: tools/kgrid.sh <tag> <bench args...> -- average duration per (kernel, grid size) of a bench run: tells the IDWT levels apart
R=$GRAFT_REPO_ROOT; tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/kgrid_$tag -o bench -- python3 $R/bench.py --no-cpu-baseline --no-e2e --jobs 1 "$@" > $R/gpurun_out/kgrid_$tag.log 2>&1
python3 - <<PY
import csv, collections
acc = collections.OrderedDict()
for r in csv.DictReader(open("$R/gpurun_out/kgrid_$tag/bench_kernel_trace.csv")):
    n = r["Kernel_Name"].split("(")[0][:60]
    if "rocclr" in n or "fill" in n.lower(): continue
    k = (n, r["Grid_Size_X"] + " vgpr " + r["VGPR_Count"] + " lds " + r["LDS_Block_Size"])
    a = acc.setdefault(k, [0, 0.0])
    a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print("== $tag", "$*")
for (n, g), (c, t) in acc.items():
    print("%-62s grid=%9s calls=%4d avg=%9.1f us" % (n, g, c, t / c))
PY
grep -a -o '"value": [0-9.]*' $R/gpurun_out/kgrid_$tag.log | head -1
