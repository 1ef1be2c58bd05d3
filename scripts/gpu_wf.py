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
def run(rounds, refill, waves, walk=24, parts=4):
    r.set_option(R.RT_OPT_WF_ROUNDS, rounds); r.set_option(R.RT_OPT_WF_REFILL, refill); r.set_option(R.RT_OPT_WAVES_PER_CU, waves); r.set_option(R.RT_OPT_WF_WALK_WAVES, walk); r.set_option(R.RT_OPT_WF_PARTS, parts)
    r.render_frames(pc, 0, 1, 4); r.sync()
    r.render_frames(pc, 0, 1, 8); r.sync()
    print("rounds %2d refill %2d waves/CU %2d walk waves %2d parts %d : %.2f ms/frame" % (rounds, refill, waves, walk, parts, r.last_render_ms()[1]), flush=True)
cfgs = [tuple(int(x) for x in a.split(',')) for a in sys.argv[3:]] or [(5, 16, 12), (5, 8, 12), (5, 32, 12), (5, 16, 14), (5, 16, 16), (5, 8, 14), (4, 16, 12)]
for cfg in cfgs:
    run(*cfg)
