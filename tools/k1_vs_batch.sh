#!/bin/bash
# k_ht_vlc time against the number of frames per job: does the kernel run in whole "rounds" of resident waves?
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for b in "$@"; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/k1b_$b -o bench -- python3 $R/bench.py --no-cpu-baseline --no-e2e --steps 6 --warmup 2 --batch $b > $R/gpurun_out/k1b_$b.log 2>&1
  python3 - <<PY
import csv
b=$b
rows={r["Name"].split("(")[0]: float(r["AverageNs"])/1e3 for r in csv.DictReader(open("$R/gpurun_out/k1b_$b/bench_kernel_stats.csv"))}
k1=[v for k,v in rows.items() if "k_ht_vlc" in k][0]; k2=[v for k,v in rows.items() if "decode_pair" in k][0]; k0=[v for k,v in rows.items() if "unstuff" in k][0]
waves=(b*6321+63)//64
print("batch %3d  waves %5d (%.2f x 2560)  k_ht_vlc %7.1f us = %5.2f us/frame   pair %5.2f us/frame  unstuff %5.2f us/frame" % (b, waves, waves/2560, k1, k1/b, k2/b, k0/b))
PY
done
