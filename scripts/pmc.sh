#!/bin/bash
# usage: scripts/pmc.sh <tag> "<counters>" [render_n args]  - one PMC pass on the GPU box
TAG=$1; CTRS=$2; shift 2
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
rm -rf $OUT && mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -s KILL 120 rocprofv3 --pmc $CTRS --output-format csv -d $OUT -- python3 scripts/render_n.py "$@" > $OUT/run.log 2>&1
F=$(find $OUT -name "*counter_collection.csv" | head -1)
python3 - "$F" <<'PY'
import csv, sys, collections
f = sys.argv[1]
acc = collections.defaultdict(lambda: [0.0, 0])
for row in csv.DictReader(open(f)):
    k = row['Kernel_Name']
    if 'rt_' not in k or 'resolve' in k: continue
    name = k.split('(')[0].replace('void ', '')
    acc[(name, row['Counter_Name'])][0] += float(row['Counter_Value']); acc[(name, row['Counter_Name'])][1] += 1
for (name, c), (v, n) in sorted(acc.items()):
    print("%-36s %-28s sum %.6g  per-dispatch %.6g (n=%d)" % (name, c, v, v / n, n))
PY
