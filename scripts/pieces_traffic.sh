cd $GRAFT_REPO_ROOT
for pf in "0 8" "1 8" "1 32" "0 32"; do
  set -- $pf
  export RT_PIECES=$1 RT_FPL=$2
  for C in WRITE_SIZE FETCH_SIZE; do
    OUT=$GRAFT_REPO_ROOT/gpurun_out/sweep_tmp; rm -rf $OUT; mkdir -p $OUT
    (cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; timeout -s KILL 120 rocprofv3 --pmc $C --output-format csv -d $OUT -- python3 scripts/render_cfg.py 1 1 32 > $OUT/run.log 2>&1) || exit 1
    python3 - "$OUT" $C $1 $2 <<'PY'
import csv, sys, glob, os, collections
f = glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] == sys.argv[2] and "rt_" in r["Kernel_Name"]:
        acc[r["Kernel_Name"].split("(")[0].replace("void ","")].append(float(r["Counter_Value"]))
for k, v in acc.items():
    w = v[1:] or v     # first = warm-up launch
    print("pieces %s fpl %s %s %s: total over the 32 frames %.3f GB (%d dispatches)" % (sys.argv[3], sys.argv[4], sys.argv[2], k, sum(w) * 1024 / 1e9, len(w)))
PY
  done
  tail -1 $OUT/run.log
done
