#!/bin/bash
# usage: tools/pmc_quick.sh <tag> <counter> -- one PMC pass of the one-job bench, per-kernel means (A/B of a single counter)
R=$GRAFT_REPO_ROOT; tag=$1; ctr=$2
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $R/gpurun_out/pmcq_$tag -o b -- python3 $R/bench.py --no-e2e --no-cpu-baseline --steps 3 --warmup 1 --jobs 1 > $R/gpurun_out/pmcq_$tag.log 2>&1 || { tail -3 $R/gpurun_out/pmcq_$tag.log; exit 1; }
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmcq_$tag/b_counter_collection.csv | grep "k_ht_\|k_idwt"
