#!/bin/bash
# Runs on the GPU box (through gpurun): kernel-trace statistics and, in separate passes, the HBM
# traffic counters of the default bench.py workload.  Output goes to gpurun_out/prof_<tag>/.
set -e
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PSBA_BENCH_NO_CFG5=1 PSBA_BENCH_NO_CLUSTERED=1  # the extras of the default line are not part of the profiled workload
ARGS="$REPO/bench.py --steps 20 --warmup 3 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace -- python3 $ARGS > $OUT/bench_trace.json 2> $OUT/trace.log
rocprofv3 --pmc FETCH_SIZE -f csv -d $OUT/pmc_fetch -- python3 $ARGS > $OUT/bench_fetch.json 2> $OUT/fetch.log
rocprofv3 --pmc WRITE_SIZE -f csv -d $OUT/pmc_write -- python3 $ARGS > $OUT/bench_write.json 2> $OUT/write.log
find $OUT -name "*.csv" | head -20
