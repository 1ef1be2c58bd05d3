"""Renders N frames of the default scene (for profiling runs). usage: render_n.py [frames] [W H] [fast]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracer_amd as R
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (2560, 1369)
fast = int(sys.argv[4]) if len(sys.argv) > 4 else 0
s = R.Scene(); s.init_default()
r = R.Renderer(0); r.upload_scene(s.buffers()); r.resize(W, H); r.set_option(R.RT_OPT_FAST_MATH, fast)
pc = R.default_push_constants(); R.camera_to_push_constants(R.default_camera(), W, H, pc); pc.parentBVHCount = 8
r.render_frames(pc, 0, 1, n); r.sync()
print("ms/frame", r.last_render_ms()[1])
