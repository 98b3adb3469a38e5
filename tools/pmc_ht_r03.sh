#!/bin/bash
# SQ counters of every kernel of the bench step -> gpurun_out/r03_ht_sq.csv (copied to profiles/ by hand).
# usage: tools/pmc_ht_r03.sh [extra bench args]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_ht_r03; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
echo "kernel,grid,counter,dispatches,mean_value" > $R/gpurun_out/r03_ht_sq.csv
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" \
           "SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/$tag -o b -- python3 $R/bench.py --no-e2e --no-cpu-baseline --steps 3 --warmup 1 --jobs 1 "$@" > $O/$tag.log 2>&1 || { tail -3 $O/$tag.log; exit 1; }
  python3 $R/tools/pmc_summary.py $O/$tag/b_counter_collection.csv | grep "k_ht_\|k_idwt" >> $R/gpurun_out/r03_ht_sq.csv
  echo "done $tag"
done
