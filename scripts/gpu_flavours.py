"""Times builds of the library against each other: both traversal flavours of pipeline 3 on benchmark configurations
(the configuration's frame count, best of 3) with a digest of the images.
usage: gpu_flavours.py "<configs>" <lib.so|default> ...   - every library is timed in a process of its own"""
import sys, os, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

if len(sys.argv) > 3 and sys.argv[1] == "--one":
    sys.path.insert(0, ROOT)
    if sys.argv[3] != "default":
        os.environ["RT_AMD_LIBRARY"] = os.path.join(ROOT, sys.argv[3])
    import hashlib
    import raytracer_amd as R
    from raytracer_amd.workloads import build_workload
    out = []
    for cfg in sys.argv[2].split():
        wl = build_workload(cfg)
        frames = min(wl["frames"], 64)
        r = R.Renderer(0); r.upload_scene(wl["scene"].buffers()); r.resize(wl["width"], wl["height"])
        for trav in (1, 0):
            r.set_option(R.RT_OPT_TRAVERSAL, trav)
            r.clear_accumulation()
            r.render_frames(wl["pc"], 0, 1, frames); r.sync()
            sha = hashlib.sha256(r.read_image(R.RT_IMAGE_ACCUM).tobytes()).hexdigest()[:8]
            best = 1e9
            reps = max(3, int(400.0 / max(1.0, frames * 2.0)))       # short configurations are repeated more often (clocks, noise)
            for rep in range(reps):
                r.render_frames(wl["pc"], 0, 1, frames); r.sync()
                best = min(best, r.last_render_ms()[1])
            out.append("cfg %s t%d %.3f %s" % (cfg, trav, best, sha))
        r.close()
    print("%-44s %s" % (sys.argv[3], "   ".join(out)), flush=True)
    sys.exit(0)

for lib in sys.argv[2:]:
    rc = subprocess.call([sys.executable, os.path.abspath(__file__), "--one", sys.argv[1], lib], timeout=280)
    if rc != 0:
        print("%s: exit %d, stopping" % (lib, rc), flush=True)
        sys.exit(1)
