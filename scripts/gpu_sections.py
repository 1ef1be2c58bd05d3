"""Diagnostic (library built with EXTRA_DEVFLAGS=-DRT_SECTION_TIMERS): wave-level cycles per section of the
lock-step kernel (pipeline 0), strict flavour. usage: gpu_sections.py [variant full|walls|bvh]"""
import sys, os, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracer_amd as R
W, H = 2560, 1369
s = R.Scene(); s.init_default()
r = R.Renderer(0); r.upload_scene(s.buffers()); r.resize(W, H); r.set_option(R.RT_OPT_PIPELINE, 0)
pc = R.default_push_constants(); R.camera_to_push_constants(R.default_camera(), W, H, pc); pc.parentBVHCount = 8
lib = R.load_library()
out = (C.c_uint64 * 8)()
names = ["kernel total", "trace total", "leaf parents (walls)", "BVH walks", "finish_hit", "material_scatter", "-", "-"]
for trav in (0, 1):
    r.set_option(R.RT_OPT_TRAVERSAL, trav)
    r.render_frames(pc, 0, 1, 2); r.sync()
    lib.rt_debug_sections(out)
    r.render_frames(pc, 0, 1, 8); r.sync()
    ms = r.last_render_ms()[1]
    lib.rt_debug_sections(out)
    v = [int(x) for x in out]
    print("traversal %d: %.3f ms/frame (timers on)" % (trav, ms))
    for k in range(6):
        print("   %-22s %6.2f %% of wave cycles" % (names[k], 100.0 * v[k] / v[0]))
    print("   %-22s %6.2f %%" % ("trace minus walls, walks", 100.0 * (v[1] - v[2] - v[3]) / v[0]))
    print("   %-22s %6.2f %%" % ("outside trace/finish/scatter", 100.0 * (v[0] - v[1] - v[4] - v[5]) / v[0]))
