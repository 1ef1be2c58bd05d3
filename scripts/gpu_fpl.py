"""Time per frame against RT_OPT_FRAMES_PER_LAUNCH (64 frames of configs[1], near-first and reference order)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracer_amd as R
from raytracer_amd.workloads import build_workload
wl = build_workload(sys.argv[1] if len(sys.argv) > 1 else "1"); pc = wl["pc"]
r = R.Renderer(0); r.upload_scene(wl["scene"].buffers()); r.resize(wl["width"], wl["height"])
for trav in (1, 0):
    r.set_option(R.RT_OPT_TRAVERSAL, trav)
    for fpl in (2, 4, 8, 16, 32, 64):
        r.set_option(R.RT_OPT_FRAMES_PER_LAUNCH, fpl)
        r.render_frames(pc, 0, 1, 8); r.sync()
        best = min((r.render_frames(pc, 0, 1, 64), r.sync(), r.last_render_ms()[1])[2] for _ in range(2))
        print("traversal %d, %2d frames per launch: %.3f ms/frame" % (trav, fpl, best), flush=True)
