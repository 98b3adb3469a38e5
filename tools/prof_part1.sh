#!/bin/bash
# usage: tools/prof_part1.sh  -- rocprofv3 kernel stats of `bench.py --part1 80` and the PMC instruction mix of k_mq_decode;
# summaries land in gpurun_out/part1_prof/ (copy what should be judged into profiles/)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/part1_prof; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python3 $R/bench.py --part1 80 --no-e2e --no-cpu-baseline --steps 5 > $O/bench.log 2>&1 || { tail -5 $O/bench.log; exit 1; }
cp $O/stats/bench_kernel_stats.csv $O/part1_bench_kernel_stats.csv
tail -1 $O/bench.log > $O/part1_bench.json
bash $R/tools/mq_pmc.sh > $O/part1_mq_pmc.csv 2>&1
cat $O/part1_mq_pmc.csv | grep 25344
