"""Renders a benchmark configuration the way bench.py's timed region does (for profiling runs).
usage: render_cfg.py <config 0|1|2|3|4s> <traversal 0|1> [frames] [pipeline] [fast]
environment: RT_PIECES (RT_OPT_LAUNCH_PIECES), RT_FPL (RT_OPT_FRAMES_PER_LAUNCH)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracer_amd as R
from raytracer_amd.workloads import build_workload
config = sys.argv[1] if len(sys.argv) > 1 else "1"
trav = int(sys.argv[2]) if len(sys.argv) > 2 else 1
wl = build_workload(config)
frames = int(sys.argv[3]) if len(sys.argv) > 3 and int(sys.argv[3]) > 0 else wl["frames"]
r = R.Renderer(0); r.upload_scene(wl["scene"].buffers()); r.resize(wl["width"], wl["height"])
if len(sys.argv) > 4 and int(sys.argv[4]) >= 0:
    r.set_option(R.RT_OPT_PIPELINE, int(sys.argv[4]))
if len(sys.argv) > 5:
    r.set_option(R.RT_OPT_FAST_MATH, int(sys.argv[5]))
r.set_option(R.RT_OPT_TRAVERSAL, trav)
if os.environ.get("RT_PIECES"):
    r.set_option(R.RT_OPT_LAUNCH_PIECES, int(os.environ["RT_PIECES"]))
if os.environ.get("RT_FPL"):
    r.set_option(R.RT_OPT_FRAMES_PER_LAUNCH, int(os.environ["RT_FPL"]))
pc = wl["pc"]
r.render_frames(pc, 0, 1, min(frames, 8)); r.sync()        # warm-up launch (first in the profile)
r.clear_accumulation()
r.render_frames(pc, 0, 1, frames); r.sync()
print("config %s traversal %d: %d frames, %.3f ms/frame, %d frames per launch" % (config, trav, frames, r.last_render_ms()[1], r.get_option(R.RT_OPT_FRAMES_PER_LAUNCH)))
