"""Ad-hoc GPU check of pipeline 3 (two samples per lane): bit-equality with pipeline 0, then timing."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracer_amd as R
s = R.Scene(); s.init_default(); sb = s.buffers()
r = R.Renderer(0); r.upload_scene(sb)
cam = R.default_camera()
for (W, H, rpp, first, stride, count, acc) in [(320, 180, 1, 0, 1, 8, 1), (200, 120, 3, 2, 3, 5, 1), (136, 88, 2, 1, 2, 7, 0), (320, 180, 1, 0, 1, 3, 1)]:
    r.resize(W, H)
    pc = R.default_push_constants(); R.camera_to_push_constants(cam, W, H, pc); pc.parentBVHCount = 8
    pc.raysPerPixel = rpp; pc.accumulateFrames = acc
    imgs = []
    for pipe, q in ((0, 2), (3, 2)):
        r.set_option(R.RT_OPT_PIPELINE, pipe)
        r.clear_accumulation()
        r.render_frames(pc, first, stride, count); r.sync()
        imgs.append((r.read_image(R.RT_IMAGE_ACCUM), r.read_image(R.RT_IMAGE_DRAW)))
    print("%dx%d rpp=%d frames %d+%d*k x%d acc=%d: accum equal %.4f%% draw equal %.4f%%" % (W, H, rpp, first, stride, count, acc,
          min(100.0 * np.mean(np.all(imgs[0][0].view(np.uint32) == im[0].view(np.uint32), axis=2)) for im in imgs[1:]),
          min(100.0 * np.mean(np.all(imgs[0][1].view(np.uint32) == im[1].view(np.uint32), axis=2)) for im in imgs[1:])), flush=True)
W, H = 2560, 1369
r.resize(W, H)
pc = R.default_push_constants(); R.camera_to_push_constants(cam, W, H, pc); pc.parentBVHCount = 8
st = r.frame_stats(pc)
for pipe, q in ((0, 2), (3, 2), (0, 2), (3, 2)):
    r.set_option(R.RT_OPT_PIPELINE, pipe)
    r.render_frames(pc, 0, 1, 16); r.sync()
    r.clear_accumulation()
    r.render_frames(pc, 0, 1, 32); r.sync()
    ms = r.last_render_ms()[1]
    print("pipeline=%d q=%d 2560x1369: %.3f ms/frame -> %.1f Mrays/s" % (pipe, q, ms, st.rays / (ms * 1e-3) / 1e6), flush=True)
