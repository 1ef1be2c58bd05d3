#!/bin/bash
# usage: scripts/prof.sh <tag> [bench.py args]  - kernel-trace stats of a bench.py run (on the GPU box)
set -e
TAG=${1:-r02}; shift || true
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
rm -rf $OUT && mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --steps 2 --warmup 1 --cpu-seconds 0 "$@" > $OUT/bench.json 2> $OUT/bench.err
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
head -12 $OUT/kernel_stats.csv
cat $OUT/bench.json
