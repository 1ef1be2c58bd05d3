"""Times the generated benchmark scenes on the GPU. usage: bench_configs.py [heightfield|blobs] [frames] [pipelines...]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracer_amd as R
from raytracer_amd import scenefile as SF
which = sys.argv[1] if len(sys.argv) > 1 else "heightfield"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 8
pipes = [int(a) for a in sys.argv[3:]] or [0, 1, 2]
out_dir = os.path.join(os.environ.get("TMPDIR", "/tmp"), "rt_generated")
t0 = time.time()
desc = SF.config_heightfield(out_dir) if which == "heightfield" else SF.config_blob_grid(out_dir)
scene, cam, pc = SF.build_scene(desc, out_dir)
W, H = desc["width"], desc["height"]
tris = sum(scene.model_info(os.path.join(out_dir, m))[0] for m in sorted({o["model"] for o in desc["objects"]}))
print("%s: %d objects, %d stored triangles, %dx%d, scene built in %.1f s" % (desc["name"], len(desc["objects"]), tris, W, H, time.time() - t0), flush=True)
r = R.Renderer(0); r.upload_scene(scene.buffers()); r.resize(W, H)
pc.frame = 0
st = r.frame_stats(pc)
print("frame 0: rays %d node tests %d triangle tests %d (%.1f / %.1f per ray)" % (st.rays, st.bvhNodeTests, st.triangleTests, st.bvhNodeTests / st.rays, st.triangleTests / st.rays), flush=True)
ref = None
for p in pipes:
    r.set_option(R.RT_OPT_PIPELINE, p)
    r.clear_accumulation()
    r.render_frames(pc, 0, 1, 2); r.sync()
    r.clear_accumulation()
    r.render_frames(pc, 0, 1, frames); r.sync()
    ms = r.last_render_ms()[1]
    img = r.read_image(R.RT_IMAGE_ACCUM)
    same = "" if ref is None else " accum identical to pipeline %d: %s" % (pipes[0], bool((img.view("u4") == ref.view("u4")).all()))
    if ref is None: ref = img
    print("pipeline %d: %.2f ms/frame  %.1f Mrays/s%s" % (p, ms, st.rays / ms / 1e3, same), flush=True)
