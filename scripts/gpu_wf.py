"""Timing sweep of the wavefront pipeline's knobs. usage: gpu_wf.py [W H]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracer_amd as R
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2560, 1369)
s = R.Scene(); s.init_default()
r = R.Renderer(0); r.upload_scene(s.buffers()); r.resize(W, H)
pc = R.default_push_constants(); R.camera_to_push_constants(R.default_camera(), W, H, pc); pc.parentBVHCount = 8
r.set_option(R.RT_OPT_PIPELINE, 2)
def run(rounds, refill, waves):
    r.set_option(R.RT_OPT_WF_ROUNDS, rounds); r.set_option(R.RT_OPT_WF_REFILL, refill); r.set_option(R.RT_OPT_WAVES_PER_CU, waves)
    r.render_frames(pc, 0, 1, 4); r.sync()
    r.render_frames(pc, 0, 1, 8); r.sync()
    print("rounds %2d refill %2d waves/CU %2d : %.2f ms/frame" % (rounds, refill, waves, r.last_render_ms()[1]), flush=True)
for cfg in [(5, 16, 12), (5, 32, 12), (5, 64, 12), (5, 64, 8), (5, 64, 16), (3, 64, 12), (8, 64, 12), (1, 64, 12), (0, 64, 12)]:
    run(*cfg)
