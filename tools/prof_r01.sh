#!/bin/bash
# rocprofv3 passes behind profiles/r01_*: kernel trace + stats of the bench command, then FETCH_SIZE and
# WRITE_SIZE in separate counter passes (TCC slots: they do not fit together), then the same two counters on the
# copy-shape microbenchmark to calibrate the gfx950 FETCH_SIZE halving on this access width (8 B per lane).
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r01
mkdir -p $O
[ -x $R/tools/ubench/membw ] || make -C $R ubench > /dev/null 2>&1 || exit 1
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-e2e"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- $BENCH > $O/bench_trace.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o bench -- $BENCH > $O/bench_fetch.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o bench -- $BENCH > $O/bench_write.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/cal_fetch -o membw -- $R/tools/ubench/membw > $O/membw_fetch.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/cal_write -o membw -- $R/tools/ubench/membw > $O/membw_write.log 2>&1 || exit 1
find $O -name "*.csv" | head -30
