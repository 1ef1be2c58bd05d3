"""Diagnostic (library built with EXTRA_DEVFLAGS=-DRT_SECTION_TIMERS, named by RT_AMD_LIBRARY): wave-level cycles per
section of the two-samples-per-lane kernel (pipeline 3), strict flavour. usage: gpu_sections_pair.py [config]"""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracer_amd as R
from raytracer_amd.workloads import build_workload
wl = build_workload(sys.argv[1] if len(sys.argv) > 1 else "1")
r = R.Renderer(0); r.upload_scene(wl["scene"].buffers()); r.resize(wl["width"], wl["height"]); r.set_option(R.RT_OPT_PIPELINE, 3); r.set_option(R.RT_OPT_LAUNCH_PIECES, 0)      # (the timers are read out at the end of a launch in one piece)
pc = wl["pc"]
lib = R.load_library()
out = (C.c_uint64 * 8)()
names = ["kernel total", "GetHit total", "walls, regular rays", "runs of BVH parents, regular rays", "shading (both samples)",
         "  of the runs: pre-pass over the parents", "runs of BVH parents, shadow rays", "walls, shadow rays"]
for trav in (1, 0):
    r.set_option(R.RT_OPT_TRAVERSAL, trav)
    r.render_frames(pc, 0, 1, 8); r.sync()
    lib.rt_debug_sections(out)
    r.render_frames(pc, 0, 1, 8); r.sync()
    ms = r.last_render_ms()[1]
    lib.rt_debug_sections(out)
    v = [int(x) for x in out]
    print("traversal %d: %.3f ms/frame (timers on)" % (trav, ms))
    for k in range(8):
        print("   %-42s %6.2f %% of wave cycles" % (names[k], 100.0 * v[k] / v[0]))
    print("   %-42s %6.2f %%" % ("GetHit minus runs and walls", 100.0 * (v[1] - v[2] - v[3] - v[6] - v[7]) / v[0]))
    print("   %-42s %6.2f %%" % ("outside GetHit and shading", 100.0 * (v[0] - v[1] - v[4]) / v[0]))
