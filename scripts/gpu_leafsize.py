"""Near-first traversal: time per frame against the leaf size of the rebuilt trees (RT_OPT_SAH_MAX_LEAF).
usage: gpu_leafsize.py [config]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracer_amd as R
from raytracer_amd.workloads import build_workload
wl = build_workload(sys.argv[1] if len(sys.argv) > 1 else "1")
pc = wl["pc"]
for leaf in (1, 2, 3, 4, 6, 8, 12, 16):
    r = R.Renderer(0)
    r.set_option(R.RT_OPT_SAH_MAX_LEAF, leaf)
    r.upload_scene(wl["scene"].buffers()); r.resize(wl["width"], wl["height"])
    r.set_option(R.RT_OPT_TRAVERSAL, 1)
    r.render_frames(pc, 0, 1, 8); r.sync()
    best = min((r.render_frames(pc, 0, 1, 16), r.sync(), r.last_render_ms()[1])[2] for _ in range(3))
    print("max leaf %2d: %.3f ms/frame, records %d, stack depth %d" % (leaf, best, r.get_option(R.RT_INFO_FAST_RECORDS), r.get_option(R.RT_INFO_FAST_STACK_DEPTH)), flush=True)
    r.close()
