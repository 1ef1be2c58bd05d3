"""Times builds of the library against each other on configs[1] (2560x1369, default scene).
usage: gpu_variants.py <lib.so|default> ...   - every library is timed in a process of its own"""
import sys, os, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

if len(sys.argv) > 2 and sys.argv[1] == "--one":
    sys.path.insert(0, ROOT)
    import raytracer_amd as R
    from raytracer_amd import api
    if sys.argv[2] != "default":
        api.LIB_PATH = os.path.join(ROOT, sys.argv[2])
    W, H = 2560, 1369
    s = R.Scene(); s.init_default()
    r = R.Renderer(0); r.upload_scene(s.buffers().to_arrays()); r.resize(W, H)
    pc = R.default_push_constants(); R.camera_to_push_constants(R.default_camera(), W, H, pc); pc.parentBVHCount = 8
    import hashlib
    out = []
    for pipe, trav in ((3, 1), (3, 0), (0, 1), (0, 0)):
        r.set_option(R.RT_OPT_PIPELINE, pipe); r.set_option(R.RT_OPT_TRAVERSAL, trav)
        r.clear_accumulation()
        r.render_frames(pc, 0, 1, 8); r.sync()
        best = 1e9
        for rep in range(3):
            r.render_frames(pc, 0, 1, 64); r.sync()
            best = min(best, r.last_render_ms()[1])
        out.append("p%d t%d %.3f %s" % (pipe, trav, best, hashlib.sha256(r.read_image(R.RT_IMAGE_ACCUM).tobytes()).hexdigest()[:8]))
    depth = [r.get_option(o) for o in (R.RT_INFO_STACK_DEPTH, R.RT_INFO_FAST_STACK_DEPTH)]
    print("%-28s ms/frame: %s   stack depth %s" % (sys.argv[2], "  ".join(out), depth), flush=True)
    sys.exit(0)

for lib in sys.argv[1:]:
    rc = subprocess.call([sys.executable, os.path.abspath(__file__), "--one", lib], timeout=170)
    if rc != 0:
        print("%s: exit %d, stopping" % (lib, rc), flush=True)
        sys.exit(1)
