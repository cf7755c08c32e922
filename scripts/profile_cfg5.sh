#!/bin/bash
# Runs on the GPU box (through gpurun): kernel-trace statistics of the cfg5 workload (2000 cameras,
# dense 12000 x 12000 S) and, in a separate counters-only pass, the MFMA counters of its Cholesky.
# Output goes to gpurun_out/prof_<tag>/.
set -e
TAG=${1:-r02_cfg5}
PTS=${2:-200000}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$REPO/bench.py --workload cfg5 --cfg5-points $PTS --steps 4 --warmup 1 --segment 2 --spread-segments 0 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace -- python3 $ARGS > $OUT/bench_trace.json 2> $OUT/trace.log
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES -f csv -d $OUT/pmc_mfma -- python3 $ARGS > $OUT/bench_mfma.json 2> $OUT/mfma.log
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -f csv -d $OUT/pmc_l2 -- python3 $ARGS > $OUT/bench_l2.json 2> $OUT/l2.log
find $OUT -name "*.csv" | head -20
