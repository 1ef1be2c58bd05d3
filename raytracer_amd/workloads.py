"""The benchmark configurations as (scene, push constants, size, frame count): the reference's default
scene (InitializeScene, HardwareRenderer.cpp:574-649) at the sizes the baseline names, and the two
generated scenes of scenefile.py. Used by bench.py and the profiling scripts, so that both time the
same thing."""
import os
import tempfile

from . import api
from . import scenefile

CONFIGS = ("1", "2", "3", "4s", "0")


def build_workload(config, out_dir=None):
    """-> dict(name, scene, pc, width, height, frames, data, workload, parents).
    config: "0" default scene 1280x720, 1 frame; "1" default scene 2560x1369, 64 frames; "2" generated
    100 352-triangle height field 1920x1080, 16 frames; "3" default scene 3840x2160, 1024 frames;
    "4s" generated 999 424-triangle mixed-material scene 1920x1080, 16 frames."""
    config = str(config)
    if config not in CONFIGS:
        raise ValueError("unknown config %r (one of %s)" % (config, ", ".join(CONFIGS)))
    if config in ("0", "1", "3"):
        W, H, frames = {"0": (1280, 720, 1), "1": (2560, 1369, 64), "3": (3840, 2160, 1024)}[config]
        scene = api.Scene()
        scene.init_default()
        pc = api.default_push_constants()
        api.camera_to_push_constants(api.default_camera(), W, H, pc)
        pc.parentBVHCount = scene.object_count()
        return {"name": "configs[%s]" % config, "scene": scene, "pc": pc, "width": W, "height": H, "frames": frames, "parents": pc.parentBVHCount,
                "data": "reference default scene (hard-coded InitializeScene: 8 objects, 8794 triangles; dragon_lowres.obj stands in for the undistributed dragon.obj)",
                "workload": "configs[%s]: default scene %dx%d, %d spp as %d accumulated frames of 1 ray/pixel, maxBounces 3" % (config, W, H, frames, frames)}
    out_dir = out_dir or os.path.join(os.environ.get("TMPDIR", tempfile.gettempdir()), "rt_generated_%d" % os.getuid())
    if config == "2":
        desc = scenefile.config_heightfield(out_dir, frames=16)
        what = "generated height field of 100 352 triangles over a floor quad, open sky (scenefile.config_heightfield, seed 1234)"
    else:
        desc = scenefile.config_blob_grid(out_dir, frames=16)
        what = "64 generated objects of 15 616 triangles (999 424 stored triangles), diffuse / metal / glass, over a floor quad (scenefile.config_blob_grid, seed 4321)"
    scene, cam, pc = scenefile.build_scene(desc, out_dir)
    W, H, frames = desc["width"], desc["height"], desc["frames"]
    return {"name": "configs[%s]" % config[0], "scene": scene, "pc": pc, "width": W, "height": H, "frames": frames, "parents": pc.parentBVHCount,
            "data": "synthetic: " + what,
            "workload": "configs[%s]%s: %s, %dx%d, %d spp as %d accumulated frames of 1 ray/pixel, maxBounces 3" % (
                config[0], " scene on one GPU share" if config == "4s" else "", desc["name"], W, H, frames, frames)}
