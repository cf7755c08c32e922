#!/bin/bash
# kernel-trace of a short bench run; output under gpurun_out/prof_<tag>/trace
TAG=${1:-z}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -f csv -d $REPO/gpurun_out/prof_$TAG/trace -- python3 $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $REPO/gpurun_out/prof_$TAG/bench.json 2> $REPO/gpurun_out/prof_$TAG/log.txt
ls $REPO/gpurun_out/prof_$TAG/trace/*/
