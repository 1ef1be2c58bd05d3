"""Tile split rehearsed on one GPU: every rank's share of the tiles (64x64, serpentine, round-robin) rendered
one after the other, timed and counted. max / mean of the shares is what load balance alone allows an
N-GPU tile split to reach. Launches in one piece (RT_OPT_LAUNCH_PIECES 0) and in pieces of two frames (1), and
whether both give the same images. usage: gpu_tile_balance.py [config] [world ...]"""
import hashlib
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
digest = {}
fpl = int(os.environ.get("FRAMES_PER_LAUNCH", "8"))
r.set_option(R.RT_OPT_FRAMES_PER_LAUNCH, fpl)
print("frames per launch %d" % fpl)
for pieces in (0, 1):
    r.set_option(R.RT_OPT_LAUNCH_PIECES, pieces)
    r.clear_accumulation()
    r.render_frames(pc, 0, 1, 8); r.sync()
    r.clear_accumulation()
    r.render_frames(pc, 0, 1, frames); r.sync()
    whole = r.last_render_ms()[0]
    digest[pieces] = hashlib.sha256(r.read_image(R.RT_IMAGE_ACCUM).tobytes() + r.read_image(R.RT_IMAGE_DRAW).tobytes()).hexdigest()[:12]
    print("config %s, %d frames, launches in %s: whole image on one GPU %.2f ms, images %s" % (cfg, frames, "pieces of two frames" if pieces else "one piece", whole, digest[pieces]), flush=True)
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
        print("  world %d: share times %s ms; slowest %.2f, mean %.2f -> balance %.3f; speed-up over one GPU at best %.2f of %d; rays per share max/mean %.3f" % (
            world, " ".join("%.2f" % m for m in ms), max(ms), mean, mean / max(ms), whole / max(ms), world, max(rays) / (sum(rays) / world)), flush=True)
print("images identical:", digest[0] == digest[1])
