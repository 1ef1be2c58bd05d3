"""Tile split rehearsed on one GPU: every rank's share of the tiles (64x64, serpentine, round-robin) rendered
one after the other, timed and counted. max / mean of the shares is what load balance alone allows an
N-GPU tile split to reach. usage: gpu_tile_balance.py [config] [world ...]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracer_amd as R
from raytracer_amd.workloads import build_workload
cfg = sys.argv[1] if len(sys.argv) > 1 else "4s"
worlds = [int(a) for a in sys.argv[2:]] or [2, 4, 8]
wl = build_workload(cfg); pc = wl["pc"]
r = R.Renderer(0); r.upload_scene(wl["scene"].buffers()); r.resize(wl["width"], wl["height"]); r.set_option(R.RT_OPT_TRAVERSAL, 1)
frames = min(wl["frames"], 16)
r.render_frames(pc, 0, 1, 8); r.sync()
r.render_frames(pc, 0, 1, frames); r.sync()
whole = r.last_render_ms()[0]
print("config %s, %d frames, whole image on one GPU: %.2f ms" % (cfg, frames, whole), flush=True)
for world in worlds:
    ms, rays = [], []
    for rank in range(world):
        r.set_option(R.RT_OPT_TILE_SIZE, 64); r.set_option(R.RT_OPT_TILE_WORLD, world); r.set_option(R.RT_OPT_TILE_RANK, rank)
        r.render_frames(pc, 0, 1, frames); r.sync()
        ms.append(r.last_render_ms()[0])
        p0 = pc.copy(); p0.frame = 0
        rays.append(r.frame_stats(p0).rays)
    r.set_option(R.RT_OPT_TILE_WORLD, 1); r.set_option(R.RT_OPT_TILE_RANK, 0)
    mean = sum(ms) / world
    print("world %d: share times %s ms; slowest %.2f, mean %.2f -> balance %.3f; speed-up over one GPU at best %.2f of %d; rays per share max/mean %.3f" % (
        world, " ".join("%.2f" % m for m in ms), max(ms), mean, mean / max(ms), whole / max(ms), world, max(rays) / (sum(rays) / world)), flush=True)
