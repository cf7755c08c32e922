#!/bin/bash
# GPU box: GRBM_GUI_ACTIVE (cycles the chip was busy) per launch of the cfg5 workload's kernels; with the
# kernel-trace durations of profile_cfg5.sh this gives the clock the kernels actually ran at.
set -e
TAG=${1:-clock}
PTS=${2:-200000}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$REPO/bench.py --workload cfg5 --cfg5-points $PTS --steps 4 --warmup 1 --segment 2 --spread-segments 0 --no-cpu-baseline"
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT -f csv -d $OUT/pmc_clk -- python3 $ARGS > $OUT/bench_clk.json 2> $OUT/clk.log
