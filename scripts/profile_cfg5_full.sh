#!/bin/bash
# Runs on the GPU box (through gpurun): BASELINE configs[4] at FULL size (2000 cameras x 2 M points x 20 M
# observations) -- kernel-trace statistics, then counters-only passes: the MFMA counters of the dense
# factorization and FETCH_SIZE / WRITE_SIZE (K2's k_schur_lds at a size where W no longer fits any cache).
# Output goes to gpurun_out/prof_<tag>/.
set -e
TAG=${1:-r04_cfg5}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$REPO/bench.py --workload cfg5 --cfg5-points 2000000 --steps 3 --warmup 0 --segment 3 --spread-segments 0 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace -- python3 $ARGS > $OUT/bench_trace.json 2> $OUT/trace.log
echo trace done
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES -f csv -d $OUT/pmc_mfma -- python3 $ARGS > $OUT/bench_mfma.json 2> $OUT/mfma.log
echo mfma done
rocprofv3 --pmc FETCH_SIZE -f csv -d $OUT/pmc_fetch -- python3 $ARGS > $OUT/bench_fetch.json 2> $OUT/fetch.log
echo fetch done
rocprofv3 --pmc WRITE_SIZE -f csv -d $OUT/pmc_write -- python3 $ARGS > $OUT/bench_write.json 2> $OUT/write.log
find $OUT -name "*_kernel_stats.csv" | head -3
