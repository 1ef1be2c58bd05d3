"""Scene recipes shared by the tests: each returns a list of (model file, position, rotation deg,
scale, material index) plus a material table, and can be built with either scene builder."""
import os

import numpy as np

import oracle_lib as O


def mat(albedo=(1, 1, 1), smoothness=0.0, fuzziness=0.0, emission=0.0, ior=0.0, absorb=(0, 0, 0)):
    return np.array([albedo[0], albedo[1], albedo[2], smoothness, fuzziness, emission, ior, 0.0,
                     absorb[0], absorb[1], absorb[2], 0.0], np.float32)


def mixed_recipe():
    """Open-sky scene with every material class, non-uniform and sub-unit scales, a model without
    normals (teapot), a model with a 287-triangle leaf (cottage) and quads (cube)."""
    materials = [mat((0.5, 0.5, 0.5)), mat((1, 1, 1), ior=1.5, absorb=(0.2, 0.05, 0.01)), mat((1, 1, 1), smoothness=1.0),
                 mat((1.0, 0.71, 0.29), smoothness=0.8, fuzziness=0.4), mat((0.9, 0.9, 1.0), emission=4.0),
                 mat((0.8, 0.2, 0.2))]
    objects = [
        ("flat_quad.obj", (0, 0, 0), (0, 0, 0), (12, 12, 12), 0),
        ("cube.obj", (-2.5, 1.0, 0.5), (0, 30, 0), (1, 1, 1), 1),
        ("smooth_sphere.obj", (2.5, 1.0, 0.0), (0, 0, 0), (1.0, 1.0, 1.0), 2),
        ("teapot.obj", (0.0, 0.0, -2.5), (0, -40, 0), (0.6, 0.6, 0.6), 3),
        ("icosphere.obj", (0.0, 4.5, 1.0), (10, 20, 30), (0.5, 0.25, 0.5), 4),
        ("cottage_obj.obj", (-5.0, 0.0, -6.0), (0, 60, 0), (0.15, 0.15, 0.15), 5),
        ("dragon_lowres.obj", (4.0, 1.2, -3.0), (0, 120, 0), (0.5, 0.5, 0.5), 1),
    ]
    return materials, objects


def build_oracle(materials, objects, models_dir=O.MODELS_DIR):
    s = O.OracleScene()
    for m in materials:
        s.add_material(m)
    for name, *_ in objects:
        s.load_model(os.path.join(models_dir, name))
    for name, p, r, sc, mi in objects:
        s.add_object(os.path.join(models_dir, name), p, r, sc, mi)
    s.buffer_scene_data()
    return s


def build_product(materials, objects, models_dir=O.MODELS_DIR):
    import raytracer_amd as R
    s = R.Scene()
    mats = []
    for m in materials:
        mm = R.Material.from_buffer_copy(np.asarray(m, np.float32).tobytes())
        mats.append(mm)
        s.add_material(mm)
    for name, *_ in objects:
        s.load_model(os.path.join(models_dir, name))
    for name, p, r, sc, mi in objects:
        s.add_object(os.path.join(models_dir, name), p, r, sc, mats[mi])
    s.buffer_scene_data()
    return s


def make_pc(arrays, W, H, cam_pos=(0, 1, 5), look=(0, 0, -1), fov=90.0, focus=10.0, defocus=0.0, **fields):
    pc = O.default_pc(arrays[0].nbytes // 64)
    O.camera_push_constants(O.pc_bytes(pc), cam_pos, look, fov, focus, defocus, W, H)
    for k, v in fields.items():
        pc[k] = v
    return pc
