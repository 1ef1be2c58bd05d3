"""Time of single-frame dispatches (rt_render_frame, 1 ray per pixel): pipelines 3 and 0, both traversal orders."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracer_amd as R
from raytracer_amd.workloads import build_workload
wl = build_workload(sys.argv[1] if len(sys.argv) > 1 else "1"); pc = wl["pc"]
r = R.Renderer(0); r.upload_scene(wl["scene"].buffers()); r.resize(wl["width"], wl["height"])
import hashlib
for pipe in (3, 0):
    for trav in (1, 0):
        r.set_option(R.RT_OPT_PIPELINE, pipe); r.set_option(R.RT_OPT_TRAVERSAL, trav)
        r.clear_accumulation()
        for f in range(4):
            pc.frame = f; r.render_frame(pc)
        r.sync()
        t0 = time.perf_counter()
        for f in range(4, 20):
            pc.frame = f; r.render_frame(pc)
        r.sync()
        dt = (time.perf_counter() - t0) / 16
        print("pipeline %d traversal %d: %.3f ms per single-frame dispatch  accum %s" % (pipe, trav, dt * 1e3, hashlib.sha256(r.read_image(R.RT_IMAGE_ACCUM).tobytes()).hexdigest()[:8]), flush=True)
