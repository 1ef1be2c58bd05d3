"""Renders N frames (for profiling runs). usage: render_n.py [frames] [W H] [fast] [pipeline] [variant]
variant: full | walls | bvh; then traversal (0 reference order, 1 near-first)"""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracer_amd as R
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (2560, 1369)
fast = int(sys.argv[4]) if len(sys.argv) > 4 else 0
pipe = int(sys.argv[5]) if len(sys.argv) > 5 else 1
variant = sys.argv[6] if len(sys.argv) > 6 else "full"
trav = int(sys.argv[7]) if len(sys.argv) > 7 else 0
s = R.Scene(); s.init_default(); arrays = s.buffers().to_arrays()
keep = {"full": list(range(8)), "walls": [2, 3, 4, 5, 6, 7], "bvh": [0, 1]}[variant]
a = [x.copy() for x in arrays]
parents = a[0].reshape(8, 64)[keep].copy(); parents.view(np.int32).reshape(len(keep), 16)[:, 12] = np.arange(len(keep))
a[0] = parents.reshape(-1); a[1] = a[1].reshape(8, 80)[keep].copy().reshape(-1)
r = R.Renderer(0); r.upload_scene(a); r.resize(W, H); r.set_option(R.RT_OPT_FAST_MATH, fast); r.set_option(R.RT_OPT_PIPELINE, pipe); r.set_option(R.RT_OPT_TRAVERSAL, trav)
pc = R.default_push_constants(); R.camera_to_push_constants(R.default_camera(), W, H, pc); pc.parentBVHCount = len(keep)
r.render_frames(pc, 0, 1, n); r.sync()
print("ms/frame", r.last_render_ms()[1])
