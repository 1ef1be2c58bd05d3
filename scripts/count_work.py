"""Counts the work the two-samples-per-lane kernels really do on a benchmark configuration: box tests, triangle
tests, GetHit calls and record fetches of the traversal flavour that is asked for, from the counters compiled into
raytracer_amd/librt_amd_count.so (the same kernels with -DRT_WORK_COUNTERS; never the library that is timed).
usage: count_work.py <config> <traversal 0|1> [frames]   -> one JSON line"""
import sys, os, json, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["RT_AMD_LIBRARY"] = os.path.join(ROOT, "raytracer_amd", "librt_amd_count.so")
import raytracer_amd as R
from raytracer_amd.workloads import build_workload

config = sys.argv[1] if len(sys.argv) > 1 else "1"
trav = int(sys.argv[2]) if len(sys.argv) > 2 else 1
wl = build_workload(config)
frames = int(sys.argv[3]) if len(sys.argv) > 3 and int(sys.argv[3]) > 0 else wl["frames"]
r = R.Renderer(0)
r.upload_scene(wl["scene"].buffers())
r.resize(wl["width"], wl["height"])
r.set_option(R.RT_OPT_PIPELINE, 3)
r.set_option(R.RT_OPT_TRAVERSAL, trav)
lib = R.load_library()
out = (C.c_uint64 * 10)()
r.sync()
if lib.rt_debug_work_counters(out) != 0:
    raise SystemExit("rt_debug_work_counters failed")
r.render_frames(wl["pc"], 0, 1, frames)
r.sync()
if lib.rt_debug_work_counters(out) != 0:
    raise SystemExit("rt_debug_work_counters failed")
print(json.dumps({"config": config, "traversal": trav, "frames": frames, "box_tests": int(out[0]), "triangle_tests": int(out[1]),
                  "rays": int(out[2]), "record_fetches": int(out[3]),
                  # share of the lanes that are active where a wave passes (near-first kernels on four-child records)
                  "lane_share": {"visits": round(out[5] / max(1, out[4]), 4), "leaf_triangles": round(out[7] / max(1, out[6]), 4),
                                 "walls": round(out[9] / max(1, out[8]), 4)},
                  "lane_slots": {"visits": int(out[4]), "leaf_triangles": int(out[6]), "walls": int(out[8])}}))
