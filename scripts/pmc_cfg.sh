#!/bin/bash
# usage: scripts/pmc_cfg.sh <tag> "<counters>" <config> <traversal> [frames] [pipeline]  - one PMC pass on the GPU box
# of scripts/render_cfg.py (a benchmark configuration rendered as bench.py's timed region renders it);
# prints every rt_ kernel's counters, mean per dispatch, into gpurun_out/pmc_<tag>.txt as well.
TAG=$1; CTRS=$2; shift 2
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
rm -rf $OUT && mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -s KILL 170 rocprofv3 --pmc $CTRS --output-format csv -d $OUT -- python3 scripts/render_cfg.py "$@" > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 1; }
F=$(find $OUT -name "*counter_collection.csv" | head -1)
python3 - "$F" <<'PY' | tee $GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG.txt
import csv, sys, collections
f = sys.argv[1]
acc = collections.defaultdict(list)
for row in csv.DictReader(open(f)):
    k = row['Kernel_Name']
    if 'rt_' not in k or 'resolve' in k: continue
    name = k.split('(')[0].replace('void ', '')
    acc[(name, row['Counter_Name'])].append(float(row['Counter_Value']))
for (name, c), v in sorted(acc.items()):
    # the first dispatch is the warm-up launch; a ragged last launch holds fewer frames: report the largest of the rest
    w = v[1:] or v
    print("%-36s %-28s sum %.6g  per-dispatch %.6g (n=%d)" % (name, c, sum(w), max(w), len(w)))
PY
