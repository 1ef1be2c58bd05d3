#!/bin/bash
# usage (on the GPU box): scripts/spill_sweep.sh <config> <lib.so|default> ...   - for every build of the library: time per frame
# of the near-first flavour on the configuration (64 frames, best of 3) and WRITE_SIZE / FETCH_SIZE per launch of 8 frames
CFG=$1; shift
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for LIB in "$@"; do
  if [ "$LIB" != default ]; then export RT_AMD_LIBRARY=$GRAFT_REPO_ROOT/$LIB; else unset RT_AMD_LIBRARY; fi
  T=$(timeout -k 10 120 python3 - "$CFG" <<'PY'
import sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import raytracer_amd as R
from raytracer_amd.workloads import build_workload
wl = build_workload(sys.argv[1])
r = R.Renderer(0); r.upload_scene(wl["scene"].buffers()); r.resize(wl["width"], wl["height"]); r.set_option(R.RT_OPT_TRAVERSAL, 1)
r.render_frames(wl["pc"], 0, 1, 8); r.sync()
best = 1e9
for rep in range(3):
    r.render_frames(wl["pc"], 0, 1, 64); r.sync(); best = min(best, r.last_render_ms()[1])
print("%.3f" % best)
PY
) || exit 1
  for C in WRITE_SIZE FETCH_SIZE; do
    OUT=$GRAFT_REPO_ROOT/gpurun_out/sweep_tmp; rm -rf $OUT; mkdir -p $OUT
    timeout -s KILL 120 rocprofv3 --pmc $C --output-format csv -d $OUT -- python3 scripts/render_cfg.py $CFG 1 16 > $OUT/run.log 2>&1 || exit 1
    V=$(python3 - "$OUT" $C <<'PY'
import csv, sys, glob, os
f = glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True)[0]
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if r["Counter_Name"] == sys.argv[2] and "pairkernel" in r["Kernel_Name"]]
print("%.3f" % (max(v[1:] or v) * 1024 / 1e9))
PY
)
    eval "$C=$V"
  done
  echo "$LIB: $T ms/frame, WRITE_SIZE $WRITE_SIZE GB, FETCH_SIZE x2 $(python3 -c "print('%.3f' % (2*$FETCH_SIZE))") GB per launch of 8 frames (config $CFG, near-first)"
done
