#!/bin/bash
# Kernel-trace timeline of one cfg5 factorization (200 k points) -> gpurun_out/tl_<tag>.txt (scripts/chol_timeline.py).
# Usage (through gpurun): bash scripts/chol_trace.sh tag [VAR=value ...]
TAG=$1; shift
for kv in "$@"; do export "$kv"; done
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -f csv -d $REPO/gpurun_out/tl_$TAG -- python3 $REPO/bench.py --workload cfg5 --cfg5-points 200000 --steps 2 --warmup 0 --segment 2 --spread-segments 0 --no-cpu-baseline --no-extras > $REPO/gpurun_out/tl_$TAG.json 2> $REPO/gpurun_out/tl_$TAG.err
cd $REPO
f=$(ls gpurun_out/tl_$TAG/*/*kernel_trace.csv | head -1)
python scripts/chol_timeline.py $f 100000 > gpurun_out/tl_$TAG.txt
rm -rf gpurun_out/tl_$TAG
wc -l gpurun_out/tl_$TAG.txt
