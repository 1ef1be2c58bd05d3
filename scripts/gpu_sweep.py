"""Sweep of the round-selection knobs of the wave-scheduled kernel."""
import sys, os, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracer_amd as R
W, H = 2560, 1369
s = R.Scene(); s.init_default()
r = R.Renderer(0); r.upload_scene(s.buffers()); r.resize(W, H); r.set_option(R.RT_OPT_PIPELINE, 1)
pc = R.default_push_constants(); R.camera_to_push_constants(R.default_camera(), W, H, pc); pc.parentBVHCount = 8
def run(bo, bt, bs, qmin, qdiv):
    for o, v in zip(range(7, 12), (bo, bt, bs, qmin, qdiv)):
        r.set_option(o, v)
    r.render_frames(pc, 0, 1, 4); r.sync()
    r.render_frames(pc, 0, 1, 8); r.sync()
    ms = r.last_render_ms()[1]
    pc.frame = 0
    d = r.schedule_stats(pc)
    le = [d[2 * k + 1] / max(1, d[2 * k]) for k in range(2)]
    print("bias T/S %3d %3d quorum min %2d div %d : %.2f ms  iterations OS %.2fM T %.2fM lanes %.0f/%.0f" % (
        bt, bs, qmin, qdiv, ms, d[0] / 1e6, d[2] / 1e6, le[0], le[1]), flush=True)
configs = [(0,-32,0,4,8), (0,-48,0,4,8), (0,-64,0,4,8), (0,-64,0,2,16), (0,-64,0,8,4), (0,-64,0,1,64), (0,-48,0,2,16), (0,-64,0,16,2), (0,-64,0,6,6)]
for c in configs:
    run(*c)
