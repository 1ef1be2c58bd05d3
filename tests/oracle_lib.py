"""ctypes access to the CPU oracle (oracle/liboracle.so). Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
MODELS_DIR = os.path.join(ROOT, "assets", "models")
_LIB = None

# element size in bytes of each of the 18 bindings
BINDING_ELEM_BYTES = [64, 80, 48] + [16] * 9 + [16, 16] + [4] * 4
BINDING_DTYPES = [np.uint8, np.uint8, np.uint8] + [np.float32] * 9 + [np.float32, np.float32] + [np.int32] * 4


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(ORACLE_DIR, "liboracle.so")
        if not os.path.exists(path):
            subprocess.check_call(["make", "-C", ORACLE_DIR, "liboracle.so"])
        L = C.CDLL(path)
        L.orc_seed_from_coords.restype = C.c_uint32
        L.orc_seed_from_coords.argtypes = [C.c_uint32] * 4
        L.orc_rng_draw.restype = C.c_uint32
        L.orc_rng_draw.argtypes = [C.c_uint32, C.POINTER(C.c_uint32)]
        L.orc_sample_square.argtypes = [C.c_uint32, C.POINTER(C.c_float)]
        L.orc_random_unit_vector.argtypes = [C.c_uint32, C.POINTER(C.c_float)]
        L.orc_math_apply.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]
        L.orc_scene_new.restype = C.c_void_p
        L.orc_scene_free.argtypes = [C.c_void_p]
        L.orc_load_model.argtypes = [C.c_void_p, C.c_char_p]
        L.orc_add_material.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_add_object.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_buffer_scene_data.argtypes = [C.c_void_p]
        L.orc_default_scene.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
        L.orc_scene_binding.restype = C.c_void_p
        L.orc_scene_binding.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_size_t)]
        L.orc_model_info.argtypes = [C.c_void_p, C.c_char_p] + [C.POINTER(C.c_int)] * 4
        L.orc_camera_push_constants.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float,
                                                C.c_int, C.c_int, C.c_void_p]
        L.orc_render.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_obj_corners.restype = C.c_long
        L.orc_obj_corners.argtypes = [C.c_char_p, C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.POINTER(C.c_float))]
        L.orc_free.argtypes = [C.c_void_p]
        _LIB = L
    return _LIB


def seed_from_coords(x, y, frame, i):
    return lib().orc_seed_from_coords(x, y, frame, i)


def rng_draw(state):
    m = C.c_uint32()
    s = lib().orc_rng_draw(state, C.byref(m))
    return s, m.value


def math_apply(which, x):
    """include/rt_math.h over an array: which = 'exp2' | 'exp' | 'log2' | 'acos' | 'div10000'."""
    x = np.ascontiguousarray(x, np.float32)
    y = np.empty_like(x)
    lib().orc_math_apply({'exp2': 0, 'exp': 1, 'log2': 2, 'acos': 3, 'div10000': 4}[which], x.ctypes.data, y.ctypes.data, x.size)
    return y


def sample_square(seed):
    out = (C.c_float * 2)()
    lib().orc_sample_square(seed, out)
    return float(out[0]), float(out[1])


def random_unit_vector(state):
    """RandomUnitVector (Random.glsl:68-78) from a generator state -> (x, y, z)."""
    out = (C.c_float * 3)()
    lib().orc_random_unit_vector(state, out)
    return float(out[0]), float(out[1]), float(out[2])


def obj_corners(path):
    p = C.POINTER(C.c_float)()
    n = C.POINTER(C.c_float)()
    c = lib().orc_obj_corners(path.encode(), C.byref(p), C.byref(n))
    if c < 0:
        raise RuntimeError("oracle OBJ reader failed: %d" % c)
    pos = np.ctypeslib.as_array(p, shape=(c, 3)).copy() if c else np.zeros((0, 3), np.float32)
    nrm = np.ctypeslib.as_array(n, shape=(c, 3)).copy() if c else np.zeros((0, 3), np.float32)
    lib().orc_free(p)
    lib().orc_free(n)
    return pos, nrm


class OracleScene:
    def __init__(self):
        self.h = lib().orc_scene_new()

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_scene_free(self.h)
            self.h = None

    def load_model(self, path):
        r = lib().orc_load_model(self.h, path.encode())
        if r != 0:
            raise RuntimeError("orc_load_model(%s) = %d" % (path, r))

    def add_material(self, mat48):
        a = np.ascontiguousarray(mat48, dtype=np.float32)
        assert a.nbytes == 48
        return lib().orc_add_material(self.h, a.ctypes.data)

    def add_object(self, model, pos, rot, scale, material_index):
        p = np.asarray(pos, np.float32); r = np.asarray(rot, np.float32); s = np.asarray(scale, np.float32)
        return lib().orc_add_object(self.h, model.encode(), p.ctypes.data, r.ctypes.data, s.ctypes.data, material_index)

    def buffer_scene_data(self):
        if lib().orc_buffer_scene_data(self.h) != 0:
            raise RuntimeError("orc_buffer_scene_data failed")

    def default_scene(self, models_dir=MODELS_DIR, dragon="dragon_lowres.obj"):
        if lib().orc_default_scene(self.h, models_dir.encode(), dragon.encode()) != 0:
            raise RuntimeError("orc_default_scene failed")

    def model_info(self, path):
        v = [C.c_int() for _ in range(4)]
        if lib().orc_model_info(self.h, path.encode(), *[C.byref(x) for x in v]) != 0:
            raise RuntimeError("unknown model")
        return tuple(x.value for x in v)  # tris, nodes, depth, maxLeaf

    def binding(self, b):
        """numpy copy of storage buffer b (raw bytes for the struct bindings 0-2)."""
        cnt = C.c_size_t()
        ptr = lib().orc_scene_binding(self.h, b, C.byref(cnt))
        nbytes = cnt.value * BINDING_ELEM_BYTES[b]
        if not ptr or nbytes == 0:
            return np.zeros(0, BINDING_DTYPES[b])
        raw = (C.c_uint8 * nbytes).from_address(ptr)
        return np.frombuffer(bytes(raw), dtype=BINDING_DTYPES[b]).copy()

    def bindings(self):
        return [self.binding(b) for b in range(18)]


def camera_push_constants(pc_bytes, cam_pos=(0, 1, 5), look=(0, 0, -1), fov=90.0, focus=10.0, defocus=0.0, W=1280, H=720):
    """Fill the camera part of a 144-byte push-constant array (np.uint8[144]) in place."""
    p = np.asarray(cam_pos, np.float32); l = np.asarray(look, np.float32)
    lib().orc_camera_push_constants(p.ctypes.data, l.ctypes.data, fov, focus, defocus, W, H, pc_bytes.ctypes.data)
    return pc_bytes


def render(bindings, pc_bytes, W, H, accum=None, window=None, counters=False, nthreads=None):
    """One oracle dispatch. bindings: list of 18 numpy arrays. Returns dict with
    accum, out (H,W,4 float32) and optionally tri/node/rays (H,W int32)."""
    if accum is None:
        accum = np.zeros((H, W, 4), np.float32)
    accum = np.ascontiguousarray(accum, np.float32)
    out = np.zeros((H, W, 4), np.float32)
    x0, y0, x1, y1 = window if window else (0, 0, W, H)
    ptrs = (C.c_void_p * 18)(*[b.ctypes.data if b.size else None for b in bindings])
    tri = node = rays = None
    if counters:
        tri = np.zeros((H, W), np.int32); node = np.zeros((H, W), np.int32); rays = np.zeros((H, W), np.int32)
    nthreads = nthreads or min(16, os.cpu_count() or 1)
    lib().orc_render(ptrs, pc_bytes.ctypes.data, W, H, x0, y0, x1, y1, accum.ctypes.data, out.ctypes.data,
                     tri.ctypes.data if counters else None, node.ctypes.data if counters else None,
                     rays.ctypes.data if counters else None, nthreads)
    return {"accum": accum, "out": out, "tri": tri, "node": node, "rays": rays}


PC_DTYPE = np.dtype([
    ("pixel00Location", np.float32, 3), ("raysPerPixel", np.int32),
    ("pixelDeltaU", np.float32, 3), ("maxBounces", np.int32),
    ("pixelDeltaV", np.float32, 3), ("defocusAngle", np.float32),
    ("cameraPosition", np.float32, 3), ("parentBVHCount", np.int32),
    ("defocusDiskU", np.float32, 3), ("accumulateFrames", np.int32),
    ("defocusDiskV", np.float32, 3), ("frame", np.int32),
    ("sunDirection", np.float32, 3), ("sunIntensity", np.float32),
    ("sunColour", np.float32, 3), ("renderMode", np.int32),
    ("triangleTestThreshold", np.int32), ("bvhNodeTestThreshold", np.int32),
    ("depthDebugScale", np.float32), ("padding2", np.int32)])
assert PC_DTYPE.itemsize == 144


def default_pc(parent_count=0):
    """Push constants with the defaults of RaytracerTypes.h:22-52 (camera part zero)."""
    pc = np.zeros(1, PC_DTYPE)
    pc["raysPerPixel"] = 1; pc["maxBounces"] = 3; pc["accumulateFrames"] = 1; pc["frame"] = 0
    pc["sunDirection"] = (-0.5, -1.0, -0.3); pc["sunIntensity"] = 5.0
    pc["sunColour"] = (1.0, 0.95, 0.85); pc["renderMode"] = 0
    pc["triangleTestThreshold"] = 150; pc["bvhNodeTestThreshold"] = 120; pc["depthDebugScale"] = 50.0
    pc["parentBVHCount"] = parent_count
    return pc


def pc_bytes(pc):
    return pc.view(np.uint8).reshape(144)
