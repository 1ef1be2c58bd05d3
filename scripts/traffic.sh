#!/bin/bash
# usage: scripts/traffic.sh <config> <traversal> [frames]  (on the GPU box)
# HBM traffic per launch of the dominant kernel: rocprofv3 FETCH_SIZE and WRITE_SIZE in separate --pmc
# passes (they do not fit one pass), FETCH_SIZE doubled (gfx950 counts 64 B per 128-B request). Writes
# gpurun_out/traffic_<config>_<traversal>.json, an entry for profiles/hbm_traffic.json that carries the
# sha256 over the kernel, layout, tree-builder and host sources, so that bench.py only uses it for the sources it was measured on.
CFG=$1; TRAV=$2; FRAMES=${3:-0}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/traffic_${CFG}_${TRAV}
rm -rf $OUT && mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 scripts/render_cfg.py $CFG $TRAV $FRAMES > $OUT/fetch.log 2>&1 || exit 1
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 scripts/render_cfg.py $CFG $TRAV $FRAMES > $OUT/write.log 2>&1 || exit 1
python3 - "$OUT" "$CFG" "$TRAV" <<'PY'
import csv, sys, glob, json, hashlib, os, collections
out, cfg, trav = sys.argv[1], sys.argv[2], int(sys.argv[3])
root = os.environ["GRAFT_REPO_ROOT"]
def collect(sub, counter):
    f = glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True)[0]
    per = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if row["Counter_Name"] != counter or "rt_" not in k or "resolve" in k: continue
        per[k.split("(")[0].replace("void ", "")].append(float(row["Counter_Value"]))
    return per
fetch, write = collect("fetch", "FETCH_SIZE"), collect("write", "WRITE_SIZE")
frames_line = [l for l in open(os.path.join(out, "fetch.log")).read().splitlines() if "frames per launch" in l][-1].strip()
fpl = int(frames_line.split(",")[-1].split()[0])
h = hashlib.sha256()
for name in ("rt_kernels.hip", "rt_team.inl", "rt_device.h", "rt_bvh.cpp", "rt_host.cpp"):      # bench.py: kernel_source_hash()
    h.update(open(os.path.join(root, "raytracer_amd/csrc", name), "rb").read())
sha = h.hexdigest()
entries = []
for k in fetch:
    # the first dispatch is the warm-up launch; full launches of the timed call follow
    f = fetch[k][1:] or fetch[k]; w = write.get(k, [0.0])[1:] or write.get(k, [0.0])
    full_f = max(f); full_w = max(w)           # a ragged last launch holds fewer frames
    name = k.split("<")[0]
    entries.append({"kernel": name, "config": cfg, "traversal": trav, "frames_per_launch": fpl, "sources_sha256": sha,
                    "fetch_size_kb": full_f, "write_size_kb": full_w,
                    "bytes_per_launch": 2.0 * full_f * 1024.0 + full_w * 1024.0,
                    "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), FETCH_SIZE x 2 (gfx950), largest launch of the run; " + frames_line})
# a launch in pieces is two kernels: the pieces' sum belongs to the launch
by = {e["kernel"]: e for e in entries}
for name, e in by.items():
    other = by.get(name.replace("rt_pairkernel", "rt_sum_pieces"))
    if "rt_pairkernel" in name and other is not None:
        e["sum_kernel_bytes_per_launch"] = other["bytes_per_launch"]
        e["bytes_per_launch"] += other["bytes_per_launch"]
json.dump({"entries": entries}, open(out + ".json", "w"), indent=1)
print(json.dumps(entries, indent=1))
PY
