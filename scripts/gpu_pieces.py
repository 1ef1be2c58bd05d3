"""Launches in one piece against launches in pieces of two frames (RT_OPT_LAUNCH_PIECES), for several frames-per-launch
settings, on the benchmark configurations: ms per frame (best of 3 renders of the configuration's frame count) and
whether the images are the same. usage: gpu_pieces.py [configs, default "1 2 4s"] [traversals, default "1 0"]"""
import sys, os, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracer_amd as R
from raytracer_amd import workloads

configs = sys.argv[1].split() if len(sys.argv) > 1 else ["1", "2", "4s"]
travs = [int(t) for t in (sys.argv[2].split() if len(sys.argv) > 2 else ["1", "0"])]
for cfg in configs:
    w = workloads.build_workload(cfg)
    W, H, pc, frames = w["width"], w["height"], w["pc"], min(w["frames"], 64)
    r = R.Renderer(0)
    r.upload_scene(w["scene"].buffers())
    r.resize(W, H)
    for trav in travs:
        r.set_option(R.RT_OPT_TRAVERSAL, trav)
        line, shas = [], set()
        for pieces, fpl in ((0, 8), (1, 8), (1, 16), (1, 32), (1, 64)):
            if fpl > frames and fpl != 8:
                continue
            r.set_option(R.RT_OPT_LAUNCH_PIECES, pieces)
            r.set_option(R.RT_OPT_FRAMES_PER_LAUNCH, fpl)
            r.clear_accumulation()
            r.render_frames(pc, 0, 1, frames); r.sync()
            shas.add(hashlib.sha256(r.read_image(R.RT_IMAGE_ACCUM).tobytes() + r.read_image(R.RT_IMAGE_DRAW).tobytes()).hexdigest()[:8])
            best = 1e9
            for rep in range(3):
                r.render_frames(pc, 0, 1, frames); r.sync()
                best = min(best, r.last_render_ms()[1])
            line.append("%s x%d: %.3f" % ("pieces" if pieces else "one piece", fpl, best))
        print("config %s traversal %d (%d frames), ms/frame: %s; images %s" % (cfg, trav, frames, "  ".join(line), "identical" if len(shas) == 1 else "DIFFER " + str(shas)), flush=True)
    del r
