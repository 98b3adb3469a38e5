#!/bin/bash
# usage: tools/kgrid_cfg.sh <tag> <gpu_configs.py filters...> -- average duration per (kernel, grid size) of a tools/gpu_configs.py run
R=$GRAFT_REPO_ROOT; tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/kcfg_$tag -o cfg -- python3 $R/tools/gpu_configs.py "$@" > $R/gpurun_out/kcfg_$tag.log 2>&1
python3 - <<PY
import csv, collections
acc = collections.OrderedDict()
for r in csv.DictReader(open("$R/gpurun_out/kcfg_$tag/cfg_kernel_trace.csv")):
    n = r["Kernel_Name"].split("(")[0][:60]
    if "rocclr" in n or "fill" in n.lower(): continue
    k = (n, r["Grid_Size_X"])
    a = acc.setdefault(k, [0, 0.0])
    a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for (n, g), (c, t) in acc.items():
    print("%-62s grid=%9s calls=%4d avg=%9.1f us" % (n, g, c, t / c))
PY
