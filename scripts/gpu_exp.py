"""Experiment: cost split between the wall objects and the BVH objects of the default scene."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import raytracer_amd as R
W, H = 2560, 1369
s = R.Scene(); s.init_default(); arrays = s.buffers().to_arrays()
def variant(keep):
    a = [x.copy() for x in arrays]
    parents = a[0].reshape(8, 64)[keep].copy()
    pv = parents.view(np.int32).reshape(len(keep), 16)
    objs = a[1].reshape(8, 80)[keep].copy()
    pv[:, 12] = np.arange(len(keep))
    a[0] = parents.reshape(-1); a[1] = objs.reshape(-1)
    return a
r = R.Renderer(0); r.resize(W, H)
pc = R.default_push_constants(); R.camera_to_push_constants(R.default_camera(), W, H, pc)
for name, keep in (("full", list(range(8))), ("walls only", [2, 3, 4, 5, 6, 7]), ("dragon+ico only", [0, 1]), ("dragon+walls", [1,2,3,4,5,6,7]), ("walls then dragon+ico", [2,3,4,5,6,7,0,1])):
    a = variant(keep)
    r.upload_scene(a); pc.parentBVHCount = len(keep)
    for pipe, trav in ((3, 0), (3, 1), (0, 1)):
        r.set_option(R.RT_OPT_PIPELINE, pipe)
        r.set_option(R.RT_OPT_TRAVERSAL, trav)
        r.render_frames(pc, 0, 1, 2); r.sync()
        r.render_frames(pc, 0, 1, 6); r.sync()
        ms = r.last_render_ms()[1]
        pc.frame = 0
        st = r.frame_stats(pc)
        print("%-22s pipeline %d trav %d: %.2f ms/frame  rays %d node %d tri %d  -> %.0f Mrays/s" % (name, pipe, trav, ms, st.rays, st.bvhNodeTests, st.triangleTests, st.rays / ms / 1e3))
