"""Ad-hoc GPU check: default scene, every pipeline vs oracle at a small size, plus timing."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import raytracer_amd as R
import oracle_lib as O

pipes = [int(a) for a in sys.argv[1:]] or [0, 1, 2, 3]
W, H = 320, 180
s = R.Scene(); s.init_default(); sb = s.buffers(); arrays = sb.to_arrays()
r = R.Renderer(0)
r.upload_scene(sb); r.resize(W, H)
pc = R.default_push_constants(); cam = R.default_camera(); R.camera_to_push_constants(cam, W, H, pc); pc.parentBVHCount = 8
acc_o = np.zeros((H, W, 4), np.float32)
for f in range(2):
    pc.frame = f
    o = O.render(arrays, pc.as_bytes(), W, H, accum=acc_o, counters=True); acc_o = o["accum"]
for pipe in pipes:
    r.set_option(R.RT_OPT_PIPELINE, pipe)
    for fast in (0, 1):
        r.set_option(R.RT_OPT_FAST_MATH, fast)
        r.clear_accumulation()
        for f in range(2):
            pc.frame = f
            r.render_frame(pc)
        acc = r.read_image(R.RT_IMAGE_ACCUM); out = r.read_image(R.RT_IMAGE_DRAW)
        print("pipeline=%d fast=%d accum bit-equal pixels: %.4f%%  draw bit-equal %.4f%%" % (
            pipe, fast, 100.0 * np.mean(np.all(acc.view(np.uint32) == acc_o.view(np.uint32), axis=2)),
            100.0 * np.mean(np.all(out.view(np.uint32) == o["out"].view(np.uint32), axis=2))))
    r.set_option(R.RT_OPT_FAST_MATH, 0)
    pc.frame = 1
    st = r.frame_stats(pc)
    print("   stats rays %d node %d tri %d | oracle %d %d %d" % (st.rays, st.bvhNodeTests, st.triangleTests, o["rays"].sum(), o["node"].sum(), o["tri"].sum()))
    for mode in (3, 4):
        pc.renderMode = mode; pc.frame = 0
        r.render_frame(pc); g = r.read_image(R.RT_IMAGE_DRAW)
        oo = O.render(arrays, pc.as_bytes(), W, H)["out"]
        print("   mode %d bit-equal %.4f%%" % (mode, 100.0 * np.mean(np.all(g.view(np.uint32) == oo.view(np.uint32), axis=2))))
    pc.renderMode = 0
W2, H2 = 2560, 1369
r.resize(W2, H2); R.camera_to_push_constants(cam, W2, H2, pc)
for pipe in pipes:
    r.set_option(R.RT_OPT_PIPELINE, pipe)
    for fast in (0, 1):
        r.set_option(R.RT_OPT_FAST_MATH, fast)
        r.render_frames(pc, 0, 1, 12); r.sync()
        r.clear_accumulation()
        r.render_frames(pc, 0, 1, 16); r.sync()
        ms_total, ms_per = r.last_render_ms()
        pc.frame = 0
        st = r.frame_stats(pc)
        print("pipeline=%d fast=%d 2560x1369: %.2f ms/frame -> %.1f Mrays/s" % (pipe, fast, ms_per, st.rays / (ms_per * 1e-3) / 1e6))
