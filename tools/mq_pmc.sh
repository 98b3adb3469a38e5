#!/bin/bash
# usage: tools/mq_pmc.sh  -- instruction mix of k_mq_decode on one 4-frame Part-1 job (rocprofv3 PMC passes)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/mq_pmc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU"; do
  tag=$(echo $set | cut -d' ' -f1)
  TIMING=1 BATCH=4 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/$tag -o mq -- python3 $R/tools/gpu_mq.py > $O/$tag.log 2>&1 || { tail -5 $O/$tag.log; exit 1; }
  python3 $R/tools/pmc_summary.py $O/$tag/mq_counter_collection.csv | grep -i "mq_decode"
done
