#!/bin/bash
# Runs on the GPU box (through gpurun): a few more PMC passes over the default bench.py workload
# (LDS conflicts, VALU activity, wait cycles).  Separate passes, counters only.
# Output: gpurun_out/prof_<tag>/pmc_<name>/...
set -e
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PSBA_BENCH_NO_CFG5=1 PSBA_BENCH_NO_CLUSTERED=1  # the extras of the default line are not part of the profiled workload
ARGS="$REPO/bench.py --steps 10 --warmup 1 --no-cpu-baseline"
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS -f csv -d $OUT/pmc_lds -- python3 $ARGS > /dev/null 2> $OUT/pmc_lds.log
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES -f csv -d $OUT/pmc_valu -- python3 $ARGS > /dev/null 2> $OUT/pmc_valu.log
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_INSTS_LDS -f csv -d $OUT/pmc_wait -- python3 $ARGS > /dev/null 2> $OUT/pmc_wait.log
find $OUT -name "*counter_collection.csv" | head
