#!/bin/bash
# usage: scripts/prof.sh <tag>  - kernel-trace stats of the default bench run (on the GPU box)
set -e
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
rm -rf $OUT && mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --steps 2 --warmup 1 --cpu-seconds 0 > $OUT/bench.json 2> $OUT/bench.err
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
cat $OUT/kernel_stats.csv | head -12
cat $OUT/bench.json
