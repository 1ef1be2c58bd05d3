"""Diagnostic (library built with EXTRA_DEVFLAGS=-DRT_SECTION_TIMERS): walker occupancy and path-wave waiting of pipeline 4."""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracer_amd as R
from raytracer_amd.workloads import build_workload
wl = build_workload(sys.argv[1] if len(sys.argv) > 1 else "1")
r = R.Renderer(0); r.upload_scene(wl["scene"].buffers()); r.resize(wl["width"], wl["height"])
r.set_option(R.RT_OPT_PIPELINE, 4)
lib = R.load_library(); out = (C.c_uint64 * 8)()
for trav in (0, 1):
    r.set_option(R.RT_OPT_TRAVERSAL, trav)
    r.render_frames(wl["pc"], 0, 1, 8); r.sync(); lib.rt_debug_sections(out)
    r.render_frames(wl["pc"], 0, 1, 8); r.sync(); ms = r.last_render_ms()[1]
    lib.rt_debug_sections(out)
    v = [int(x) for x in out]
    print("traversal %d: %.3f ms/frame; per frame: walker iterations %.3g, lanes per iteration %.1f, idle polls %.3g, jobs %.3g, steps per job %.1f" % (
        trav, ms, v[0] / 8, v[1] / max(v[0], 1), v[2] / 8, v[3] / 8, v[1] / max(v[3], 1)))
    print("   path waves: %.1f %% of their cycles waiting for walkers; walker wave lifetime %.3g cycles vs iterations %.3g -> %.0f cycles per iteration (incl. idle)" % (
        100.0 * v[4] / max(v[5], 1), v[6], v[0] + v[2], v[6] / max(v[0] + v[2], 1)))
