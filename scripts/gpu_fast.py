"""Experiment: near-first traversal (RT_OPT_TRAVERSAL = 1) against the reference order: ms/frame and
the share of pixels whose accumulation value is bit-identical. usage: gpu_fast.py [default|heightfield|blobs] [leaf sizes...]"""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import raytracer_amd as R
from raytracer_amd import scenefile as SF
which = sys.argv[1] if len(sys.argv) > 1 else "default"
leafs = [int(a) for a in sys.argv[2:]] or [4]
frames = 8
if which == "default":
    W, H = 2560, 1369
    scene = R.Scene(); scene.init_default()
    pc = R.default_push_constants(); R.camera_to_push_constants(R.default_camera(), W, H, pc); pc.parentBVHCount = 8
else:
    out_dir = os.path.join(os.environ.get("TMPDIR", "/tmp"), "rt_generated")
    desc = SF.config_heightfield(out_dir) if which == "heightfield" else SF.config_blob_grid(out_dir)
    scene, cam, pc = SF.build_scene(desc, out_dir)
    W, H = desc["width"], desc["height"]
r = R.Renderer(0); r.upload_scene(scene.buffers()); r.resize(W, H)
pc.frame = 0
st = r.frame_stats(pc)
print("%s %dx%d: rays/frame %d, reference node tests %.1f tri tests %.1f per ray; records %d triangles %d depth %d" % (
    which, W, H, st.rays, st.bvhNodeTests / st.rays, st.triangleTests / st.rays, r.get_option(R.RT_INFO_RECORDS),
    r.get_option(R.RT_INFO_TRIANGLES), r.get_option(R.RT_INFO_STACK_DEPTH)), flush=True)
ref = {}
for pipe in (3, 0):
    r.set_option(R.RT_OPT_PIPELINE, pipe)
    for trav, leaf in [(0, 0)] + [(1, l) for l in leafs]:
        r.set_option(R.RT_OPT_TRAVERSAL, trav)
        if trav:
            r.set_option(R.RT_OPT_SAH_MAX_LEAF, leaf)
            t0 = time.time()
            info = (r.get_option(R.RT_INFO_FAST_OBJECTS), r.get_option(R.RT_INFO_FAST_RECORDS), r.get_option(R.RT_INFO_FAST_TRIANGLES), r.get_option(R.RT_INFO_FAST_STACK_DEPTH))
            build = time.time() - t0
        r.clear_accumulation(); r.render_frames(pc, 0, 1, 2); r.sync()
        r.clear_accumulation(); r.render_frames(pc, 0, 1, frames); r.sync()
        ms = r.last_render_ms()[1]
        img = r.read_image(R.RT_IMAGE_ACCUM)
        if trav == 0:
            ref[pipe] = img
            print("pipeline %d reference order: %.3f ms/frame  %.0f Mrays/s" % (pipe, ms, st.rays / ms / 1e3), flush=True)
        else:
            a, b = img.view(np.uint32), ref[pipe].view(np.uint32)
            same = np.all(a == b, axis=2)
            d = np.abs(img[..., :3] - ref[pipe][..., :3])
            tol = np.all(d <= 1e-4 + 1e-4 * np.abs(ref[pipe][..., :3]), axis=2) | same
            print("pipeline %d near-first leaf<=%d: %.3f ms/frame  %.0f Mrays/s  identical pixels %.5f%%  within tol %.5f%% (after %d frames)  built %s in %.2fs" % (
                pipe, leaf, ms, st.rays / ms / 1e3, 100.0 * same.mean(), 100.0 * tol.mean(), frames, info, build), flush=True)
