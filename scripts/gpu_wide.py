"""Four-child records against two-child records (near-first traversal, pipeline 3) on the benchmark
configurations: how many pixels of an 8-frame accumulation differ, and the time per frame of each.
usage: gpu_wide.py [configs, default "1 2 4s"] [lib.so]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import raytracer_amd as R
from raytracer_amd import api, workloads

configs = sys.argv[1].split() if len(sys.argv) > 1 else ["1", "2", "4s"]
if len(sys.argv) > 2:
    api.LIB_PATH = os.path.join(ROOT, sys.argv[2])
for cfg in configs:
    w = workloads.build_workload(cfg)
    W, H, pc = w["width"], w["height"], w["pc"]
    r = R.Renderer(0)
    r.upload_scene(w["scene"].buffers().to_arrays())
    r.resize(W, H)
    r.set_option(R.RT_OPT_PIPELINE, 3)
    r.set_option(R.RT_OPT_TRAVERSAL, 1)
    imgs, times = {}, {}
    for wide in (0, 2, 1):
        r.set_option(R.RT_OPT_WIDE_RECORDS, wide)
        r.clear_accumulation()
        r.render_frames(pc, 0, 1, 8); r.sync()
        imgs[wide] = r.read_image(R.RT_IMAGE_ACCUM).copy()
        best = 1e9
        for rep in range(3):
            r.render_frames(pc, 8, 1, 32); r.sync()
            best = min(best, r.last_render_ms()[1])
        times[wide] = best
    a, b = imgs[0].reshape(-1, 4), imgs[1].reshape(-1, 4)
    differ = int(np.any(a != b, axis=1).sum())
    tol = int(np.any(np.abs(a - b) > 1e-4 + 1e-4 * np.abs(a), axis=1).sum())
    c, d = imgs[2].reshape(-1, 4), imgs[1].reshape(-1, 4)
    print("config %s: four-child records with per-run box quartets %.3f ms/frame; with the hierarchy over the parents %.3f; pixels differing between those two %d (%d beyond tolerance)" % (
        cfg, times[2], times[1], int(np.any(c != d, axis=1).sum()), int(np.any(np.abs(c - d) > 1e-4 + 1e-4 * np.abs(c), axis=1).sum())), flush=True)
    print("config %s %dx%d: two-child %.3f ms/frame, four-child %.3f ms/frame; pixels differing after 8 frames %d of %d (%d beyond 1e-4 tolerance); "
          "four-child records %d, stack need %d, two-child records %d depth %d" % (
              cfg, W, H, times[0], times[1], differ, a.shape[0], tol, r.get_option(R.RT_INFO_WIDE_RECORDS), r.get_option(R.RT_INFO_WIDE_STACK_NEED),
              r.get_option(R.RT_INFO_FAST_RECORDS), r.get_option(R.RT_INFO_FAST_STACK_DEPTH)), flush=True)
    del r
