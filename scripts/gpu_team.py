"""Experiment: pipeline 4 (path waves + walker waves) against the lock-step kernels: bit-identity and ms/frame.
usage: gpu_team.py [default|heightfield|blobs] [W H] [walkers...]"""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracer_amd as R
from raytracer_amd.workloads import build_workload
which = sys.argv[1] if len(sys.argv) > 1 else "default"
wl = build_workload({"default": "1", "heightfield": "2", "blobs": "4s"}[which])
W, H = wl["width"], wl["height"]
pc = wl["pc"]
if len(sys.argv) > 3 and int(sys.argv[2]) > 0:
    W, H = int(sys.argv[2]), int(sys.argv[3])
    assert which == "default"
    R.camera_to_push_constants(R.default_camera(), W, H, pc)
walkers = [int(a) for a in sys.argv[4:]] or [4]
frames = 8
r = R.Renderer(0); r.upload_scene(wl["scene"].buffers()); r.resize(W, H)
pc.frame = 0
st = r.frame_stats(pc)
print("%s %dx%d rays/frame %d" % (which, W, H, st.rays), flush=True)
for trav in (0, 1):
    r.set_option(R.RT_OPT_TRAVERSAL, trav)
    ref = None
    for pipe, nw in [(0, 0), (3, 0)] + [(4, w) for w in walkers]:
        r.set_option(R.RT_OPT_PIPELINE, pipe)
        if nw:
            r.set_option(R.RT_OPT_TEAM_WALKERS, nw)
        r.clear_accumulation(); r.render_frames(pc, 0, 1, 2); r.sync()
        r.clear_accumulation(); r.render_frames(pc, 0, 1, frames); r.sync()
        ms = r.last_render_ms()[1]
        acc = r.read_image(R.RT_IMAGE_ACCUM); out = r.read_image(R.RT_IMAGE_DRAW)
        if ref is None:
            ref = (acc, out)
        same = bool((acc.view(np.uint32) == ref[0].view(np.uint32)).all() and (out.view(np.uint32) == ref[1].view(np.uint32)).all())
        print("traversal %d pipeline %d walkers %d: %.3f ms/frame %.0f Mrays/s  identical to pipeline 0: %s" % (trav, pipe, nw, ms, st.rays / ms / 1e3, same), flush=True)
