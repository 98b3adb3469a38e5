#!/bin/bash
# usage (on the GPU box): bash tools/prof_round.sh r02
# rocprofv3 passes behind profiles/<tag>_*: kernel trace + stats of the bench command, FETCH_SIZE and WRITE_SIZE in
# separate counter passes (TCC slots: they do not fit together), the same two counters on the copy-shape microbenchmark
# to calibrate the gfx950 FETCH_SIZE halving, the SQ counters of the HT kernels, and kernel stats + FETCH / WRITE of the
# other configurations (tools/gpu_configs.py: C3, C4).  tools/make_profiles.py <tag> turns the output into profiles/<tag>_*.
# The bench command runs with --jobs 1: every launch of a kernel then has the shape of bench.py's one-job pass, the one
# `roofline` and `stage_ms_per_step_one_job` come from, so that the per-kernel averages of the trace can be held against them
# (the default --jobs 2 pass launches the same kernels on half batches as well).
set -o pipefail
tag=${1:-r03}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$tag
mkdir -p $O
[ -x $R/tools/ubench/membw ] || make -C $R ubench > /dev/null 2>&1 || exit 1
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --steps 10 --warmup 2 --jobs 1 --no-cpu-baseline --no-e2e"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- $BENCH > $O/bench_trace.log 2>&1 || exit 1
echo "trace done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o bench -- $BENCH > $O/bench_fetch.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o bench -- $BENCH > $O/bench_write.log 2>&1 || exit 1
echo "bench pmc done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/cal_fetch -o membw -- $R/tools/ubench/membw > $O/membw_fetch.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/cal_write -o membw -- $R/tools/ubench/membw > $O/membw_write.log 2>&1 || exit 1
$R/tools/ubench/membw > $O/membw.txt 2>&1
echo "calibration done"
CFG="python3 $R/tools/gpu_configs.py C3 C4"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/cfg_trace -o cfg -- $CFG > $O/cfg_trace.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/cfg_fetch -o cfg -- $CFG > $O/cfg_fetch.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/cfg_write -o cfg -- $CFG > $O/cfg_write.log 2>&1 || exit 1
echo "configs done"
python3 $R/tools/gpu_configs.py > $O/configs.log 2>&1 && cp $R/gpurun_out/configs.json $O/configs.json
bash $R/tools/pmc_ht_${tag}.sh > $O/pmc_ht.log 2>&1 && cp $R/gpurun_out/${tag}_ht_sq.csv $O/ht_sq.csv
echo "sq done"
find $O -name "*.csv" | wc -l
