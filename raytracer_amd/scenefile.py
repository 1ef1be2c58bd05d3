"""Scene descriptions on disk and deterministic OBJ generators.

The reference builds its only scene in code (InitializeScene, HardwareRenderer.cpp:574-649) and
edits it through ImGui panels. A scene description here is a JSON document that carries exactly the
fields those panels edit - SceneObject (HardwareRenderer.h:77-92: model path, position, rotation in
degrees, scale, material), GPUMaterial (RaytracerTypes.h:127-138), CameraSettings
(HardwareRenderer.h:68-75) and the settings half of RaytracePushConstants
(RaytracerTypes.h:22-52) - so that every benchmark configuration can be rebuilt from a file:

    {"width": 1920, "height": 1080, "frames": 16,
     "materials": [{"albedo": [r,g,b], "smoothness": s, "fuzziness": f, "emission": e,
                    "refractionIndex": n, "absorbtion": [r,g,b]}, ...],
     "objects":   [{"model": "relative/or/absolute.obj", "position": [x,y,z],
                    "rotation": [degX,degY,degZ], "scale": [x,y,z], "material": index}, ...],
     "camera":    {"position": [x,y,z], "lookDirection": [x,y,z], "fov": deg,
                   "focusDistance": d, "defocusAngle": deg},
     "settings":  {"raysPerPixel": 1, "maxBounces": 3, "sunDirection": [...], ...}}

The generators write OBJ files (positions, vertex normals, triangles) from a 32-bit linear
congruential sequence, so a given seed gives the same bytes on every machine.
"""
import json
import math
import os

import numpy as np

from . import api

_MATERIAL_KEYS = ("albedo", "smoothness", "fuzziness", "emission", "refractionIndex", "absorbtion")
_SETTING_KEYS = ("raysPerPixel", "maxBounces", "accumulateFrames", "sunDirection", "sunIntensity", "sunColour",
                 "renderMode", "triangleTestThreshold", "bvhNodeTestThreshold", "depthDebugScale")


class Lcg:
    """x <- 1664525 x + 1013904223 (mod 2^32); uniform() in [0,1) from the high 24 bits."""

    def __init__(self, seed):
        self.x = seed & 0xFFFFFFFF

    def next(self):
        self.x = (1664525 * self.x + 1013904223) & 0xFFFFFFFF
        return self.x

    def uniform(self, lo=0.0, hi=1.0):
        return lo + (hi - lo) * ((self.next() >> 8) / 16777216.0)


def _write_obj(path, positions, normals, faces, comment):
    """positions, normals: (N,3) float arrays; faces: (M,3) int arrays of 0-based vertex indices
    (the same index addresses position and normal)."""
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    lines = ["# " + comment]
    lines += ["v %.6f %.6f %.6f" % tuple(p) for p in positions]
    lines += ["vn %.6f %.6f %.6f" % tuple(n) for n in normals]
    f1 = faces + 1
    lines += ["f %d//%d %d//%d %d//%d" % (a, a, b, b, c, c) for a, b, c in f1]
    tmp = path + ".tmp%d" % os.getpid()
    with open(tmp, "w") as fh:
        fh.write("\n".join(lines))
        fh.write("\n")
    os.replace(tmp, path)


def write_heightfield_obj(path, cells=224, seed=1234, extent=4.0, amplitude=0.45):
    """A cells x cells height field over [-extent, extent]^2 (2*cells^2 triangles; 224 -> 100 352):
    six sinusoids with phases and directions from the sequence plus a little per-vertex noise.
    Heights stay within +-extent so the reference's integer-snapped model box
    (HardwareRenderer.cpp:1240-1262) stays sane."""
    rng = Lcg(seed)
    n = cells + 1
    xs = np.linspace(-extent, extent, n)
    X, Z = np.meshgrid(xs, xs, indexing="xy")
    Y = np.zeros_like(X)
    for k in range(6):
        ang = rng.uniform(0.0, 2.0 * math.pi)
        freq = (k + 1) * rng.uniform(0.35, 0.75)
        phase = rng.uniform(0.0, 2.0 * math.pi)
        Y += (amplitude / (k + 1)) * np.sin(freq * (math.cos(ang) * X + math.sin(ang) * Z) + phase)
    noise = np.array([rng.uniform(-1.0, 1.0) for _ in range(n * n)]).reshape(n, n)
    Y += 0.004 * noise
    step = xs[1] - xs[0]
    dYdx = np.gradient(Y, step, axis=1)
    dYdz = np.gradient(Y, step, axis=0)
    N = np.stack([-dYdx, np.ones_like(Y), -dYdz], axis=-1)
    N /= np.linalg.norm(N, axis=-1, keepdims=True)
    P = np.stack([X, Y, Z], axis=-1).reshape(-1, 3)
    idx = np.arange(n * n).reshape(n, n)
    a, b, c, d = idx[:-1, :-1].ravel(), idx[:-1, 1:].ravel(), idx[1:, :-1].ravel(), idx[1:, 1:].ravel()
    faces = np.concatenate([np.stack([a, c, b], 1), np.stack([b, c, d], 1)], 0)   # counter-clockwise seen from +y
    order = np.argsort(np.concatenate([np.arange(len(a)) * 2, np.arange(len(a)) * 2 + 1]), kind="stable")
    _write_obj(path, P, N.reshape(-1, 3), faces[order], "height field %dx%d cells, seed %d" % (cells, cells, seed))
    return len(faces)


def write_blob_obj(path, seed, segments=128, rings=62, radius=0.8, amplitude=0.12):
    """A displaced UV sphere (2*segments*(rings-1) triangles; 128 x 62 -> 15 616): radius modulated
    by five spherical waves from the sequence. Normals are the normalised positions' gradients
    (numerical), outward."""
    rng = Lcg(seed)
    waves = [(rng.uniform(1.0, 5.0), rng.uniform(1.0, 5.0), rng.uniform(0.0, 2.0 * math.pi), amplitude / (k + 1)) for k in range(5)]

    def r_of(theta, phi):
        r = np.full_like(theta, radius)
        for ft, fp, ph, amp in waves:
            r = r + amp * np.sin(np.round(ft) * theta + ph) * np.sin(np.round(fp) * phi)
        return r

    def point(theta, phi):
        r = r_of(theta, phi)
        return np.stack([r * np.sin(phi) * np.cos(theta), r * np.cos(phi), r * np.sin(phi) * np.sin(theta)], -1)

    theta = np.arange(segments) * (2.0 * math.pi / segments)
    phi = np.arange(1, rings) * (math.pi / rings)
    T, Ph = np.meshgrid(theta, phi, indexing="xy")          # (rings-1, segments)
    P = point(T, Ph)
    eps = 1e-3
    dT = point(T + eps, Ph) - point(T - eps, Ph)
    dP = point(T, Ph + eps) - point(T, Ph - eps)
    N = np.cross(dT, dP)
    N /= np.maximum(np.linalg.norm(N, axis=-1, keepdims=True), 1e-20)
    flip = np.sum(N * P, -1, keepdims=True) < 0.0
    N = np.where(flip, -N, N)
    top = point(np.zeros(1), np.zeros(1))
    bottom = point(np.zeros(1), np.full(1, math.pi))
    positions = np.concatenate([P.reshape(-1, 3), top, bottom], 0)
    normals = np.concatenate([N.reshape(-1, 3), [[0.0, 1.0, 0.0]], [[0.0, -1.0, 0.0]]], 0)
    grid = np.arange((rings - 1) * segments).reshape(rings - 1, segments)
    nxt = np.roll(grid, -1, axis=1)
    a, b, c, d = grid[:-1].ravel(), nxt[:-1].ravel(), grid[1:].ravel(), nxt[1:].ravel()
    body = np.concatenate([np.stack([a, b, c], 1), np.stack([b, d, c], 1)], 0)
    ti, bi = len(positions) - 2, len(positions) - 1
    cap_top = np.stack([np.full(segments, ti), nxt[0], grid[0]], 1)
    cap_bottom = np.stack([np.full(segments, bi), grid[-1], nxt[-1]], 1)
    faces = np.concatenate([cap_top, body, cap_bottom], 0)
    _write_obj(path, positions, normals, faces, "displaced sphere %dx%d, seed %d" % (segments, rings, seed))
    return len(faces)


def write_quad_obj(path):
    """The unit floor quad used by the generated scenes (two triangles in the y = 0 plane, +y normal)."""
    P = np.array([[-0.5, 0.0, -0.5], [0.5, 0.0, -0.5], [0.5, 0.0, 0.5], [-0.5, 0.0, 0.5]])
    N = np.tile([[0.0, 1.0, 0.0]], (4, 1))
    _write_obj(path, P, N, np.array([[0, 2, 1], [0, 3, 2]]), "unit quad")
    return 2


def material(albedo=(1.0, 1.0, 1.0), smoothness=0.0, fuzziness=0.0, emission=0.0, refractionIndex=0.0, absorbtion=(0.0, 0.0, 0.0)):
    return {"albedo": list(albedo), "smoothness": smoothness, "fuzziness": fuzziness, "emission": emission,
            "refractionIndex": refractionIndex, "absorbtion": list(absorbtion)}


def config_heightfield(out_dir, cells=224, width=1920, height=1080, frames=16, seed=1234):
    """The 100k-primitive benchmark configuration: one generated height field of ~100k triangles
    over the floor quad, open sky."""
    write_heightfield_obj(os.path.join(out_dir, "heightfield_%d_%d.obj" % (cells, seed)), cells, seed)
    write_quad_obj(os.path.join(out_dir, "quad.obj"))
    return {
        "name": "heightfield", "width": width, "height": height, "frames": frames,
        "materials": [material((0.55, 0.55, 0.5)), material((0.75, 0.45, 0.3))],
        "objects": [
            {"model": "quad.obj", "position": [0, -1.0, 0], "rotation": [0, 0, 0], "scale": [40, 1, 40], "material": 0},
            {"model": "heightfield_%d_%d.obj" % (cells, seed), "position": [0, 0.6, 0], "rotation": [0, 0, 0], "scale": [1, 1, 1], "material": 1},
        ],
        "camera": {"position": [0, 2, 12], "lookDirection": [0, -0.1, -1], "fov": 60.0, "focusDistance": 10.0, "defocusAngle": 0.0},
        "settings": {"raysPerPixel": 1, "maxBounces": 3},
    }


def config_blob_grid(out_dir, side=4, segments=128, rings=62, width=1920, height=1080, frames=4, seed=4321):
    """The 1M-primitive benchmark configuration: side^3 distinct generated objects (4^3 x 15 616 = 999 424 stored
    triangles; models are cached by path in the reference, so only distinct files grow the
    buffers) on a grid, materials cycling diffuse / metal / dielectric, over the floor quad."""
    rng = Lcg(seed)
    write_quad_obj(os.path.join(out_dir, "quad.obj"))
    mats = [material((0.5, 0.5, 0.5))]
    objects = [{"model": "quad.obj", "position": [0, -1.2, 0], "rotation": [0, 0, 0], "scale": [60, 1, 60], "material": 0}]
    spacing = 2.2
    k = 0
    for iy in range(side):
        for iz in range(side):
            for ix in range(side):
                name = "blob_%d_%dx%d_%03d.obj" % (seed, segments, rings, k)
                write_blob_obj(os.path.join(out_dir, name), seed * 1000 + k, segments, rings)
                kind = k % 3
                colour = (rng.uniform(0.3, 0.95), rng.uniform(0.3, 0.95), rng.uniform(0.3, 0.95))
                if kind == 0:
                    mats.append(material(colour))
                elif kind == 1:
                    mats.append(material(colour, smoothness=1.0, fuzziness=rng.uniform(0.0, 0.4)))
                else:
                    mats.append(material((1.0, 1.0, 1.0), refractionIndex=1.5, absorbtion=(0.3 * (1 - colour[0]), 0.3 * (1 - colour[1]), 0.3 * (1 - colour[2]))))
                pos = [(ix - (side - 1) / 2) * spacing, iy * spacing, (iz - (side - 1) / 2) * spacing]
                rot = [rng.uniform(0, 360), rng.uniform(0, 360), rng.uniform(0, 360)]
                objects.append({"model": name, "position": pos, "rotation": rot, "scale": [1, 1, 1], "material": len(mats) - 1})
                k += 1
    reach = side * spacing
    return {
        "name": "blob_grid", "width": width, "height": height, "frames": frames,
        "materials": mats, "objects": objects,
        "camera": {"position": [0.9 * reach, 0.75 * reach, 1.6 * reach], "lookDirection": [-0.42, -0.22, -1.0], "fov": 50.0,
                   "focusDistance": 10.0, "defocusAngle": 0.0},
        "settings": {"raysPerPixel": 1, "maxBounces": 3},
    }


def save_scene(desc, path):
    with open(path, "w") as fh:
        json.dump(desc, fh, indent=1)


def load_scene(path):
    with open(path) as fh:
        desc = json.load(fh)
    for key in ("materials", "objects", "camera"):
        if key not in desc:
            raise ValueError("scene file %s has no '%s'" % (path, key))
    return desc


def material_struct(m):
    unknown = set(m) - set(_MATERIAL_KEYS)
    if unknown:
        raise ValueError("unknown material field(s): %s" % ", ".join(sorted(unknown)))
    return api.default_material(albedo=tuple(m.get("albedo", (1.0, 1.0, 1.0))), smoothness=m.get("smoothness", 0.0),
                                fuzziness=m.get("fuzziness", 0.0), emission=m.get("emission", 0.0),
                                refractiveIndex=m.get("refractionIndex", 0.0), absorbtion=tuple(m.get("absorbtion", (0.0, 0.0, 0.0))))


def build_scene(desc, base_dir="."):
    """Scene description -> (api.Scene with BufferSceneData done, CameraSettings, PushConstants
    with camera, settings and parentBVHCount filled for desc's width x height)."""
    scene = api.Scene()
    mats = [material_struct(m) for m in desc["materials"]]
    for m in mats:
        scene.add_material(m)
    for o in desc["objects"]:
        if not 0 <= int(o["material"]) < len(mats):
            raise ValueError("object material index %r out of range" % (o["material"],))
        path = o["model"] if os.path.isabs(o["model"]) else os.path.join(base_dir, o["model"])
        scene.load_model(path)   # cached by path, as LoadModel is (HardwareRenderer.cpp:1164)
        scene.add_object(path, tuple(o.get("position", (0, 0, 0))), tuple(o.get("rotation", (0, 0, 0))),
                         tuple(o.get("scale", (1, 1, 1))), mats[int(o["material"])])
    scene.buffer_scene_data()
    cam = api.default_camera()
    c = desc["camera"]
    for i in range(3):
        cam.cameraPosition[i] = float(c["position"][i])
        cam.cameraLookDirection[i] = float(c["lookDirection"][i])
    cam.cameraFov = float(c.get("fov", cam.cameraFov))
    cam.focusDistance = float(c.get("focusDistance", cam.focusDistance))
    cam.defocusAngle = float(c.get("defocusAngle", cam.defocusAngle))
    pc = api.default_push_constants()
    for k, val in desc.get("settings", {}).items():
        if k not in _SETTING_KEYS:
            raise ValueError("unknown setting '%s'" % k)
        if isinstance(val, (list, tuple)):
            for i, x in enumerate(val):
                getattr(pc, k)[i] = float(x)
        else:
            setattr(pc, k, val)
    api.camera_to_push_constants(cam, int(desc.get("width", 1280)), int(desc.get("height", 720)), pc)
    pc.parentBVHCount = len(desc["objects"])
    return scene, cam, pc
