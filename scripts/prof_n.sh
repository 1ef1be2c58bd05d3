#!/bin/bash
# usage: scripts/prof_n.sh <tag> [render_n args]  - per-kernel times of a render_n run (on the GPU box)
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/profn_$TAG
rm -rf $OUT && mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 scripts/render_n.py "$@" > $OUT/run.log 2>&1
F=$(find $OUT -name "*kernel_trace.csv" | head -1)
python3 - "$F" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if 'rt_' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# print the sequence of the last frame: kernels after the last 'round 0'-like gap
seq = [(r['Kernel_Name'].split('(')[0].replace('void ', ''), int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows]
acc = collections.defaultdict(lambda: [0, 0])
for n, s, e in seq:
    acc[n][0] += e - s; acc[n][1] += 1
for n, (t, c) in acc.items():
    print("%-50s calls %4d total %.3f ms avg %.1f us" % (n, c, t / 1e6, t / c / 1e3))
print("last 24 launches (name, duration us, gap to previous us):")
prev = None
for n, s, e in seq[-24:]:
    print("  %-46s %9.1f %8.1f" % (n, (e - s) / 1e3, (s - prev) / 1e3 if prev else 0.0)); prev = e
PY
