"""ctypes bindings of include/rt_abi.h plus two small classes, Scene and Renderer, that
mirror the scene / dispatch surface of the reference's HardwareRenderer
(src/Renderers/Hardware/HardwareRenderer.h:282-286 LoadModel / SetRefreshAccumulation /
SetDoRender / SetRenderFrames, HardwareRenderer.cpp:490-500 AddSceneObject, :651-771
BufferSceneData, :793-838 DispatchRayTracingCommands, :840-848 RefreshAccumulation,
:1030-1061 ProduceRender)."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# RT_AMD_LIBRARY: another build of the same library (a diagnostic or experimental one)
LIB_PATH = os.environ.get("RT_AMD_LIBRARY") or os.path.join(_HERE, "librt_amd.so")
MODELS_DIR = os.path.join(os.path.dirname(_HERE), "assets", "models")

RT_IMAGE_DRAW, RT_IMAGE_ACCUM = 0, 1
RT_OPT_FAST_MATH, RT_OPT_PIPELINE, RT_OPT_TILE_RANK, RT_OPT_TILE_WORLD, RT_OPT_TILE_SIZE, RT_OPT_WAVES_PER_CU = 1, 2, 3, 4, 5, 6
RT_OPT_SCHED_BIAS_OBJECT, RT_OPT_SCHED_BIAS_TRAV, RT_OPT_SCHED_BIAS_SHADE, RT_OPT_SCHED_QUORUM_MIN, RT_OPT_SCHED_QUORUM_DIV = 7, 8, 9, 10, 11
RT_OPT_WF_ROUNDS, RT_OPT_WF_REFILL, RT_OPT_WF_WALK_WAVES, RT_OPT_WF_PARTS = 12, 13, 14, 15
RT_OPT_FRAMES_PER_LAUNCH = 16
RT_OPT_TRAVERSAL, RT_OPT_SAH_MAX_LEAF, RT_OPT_TEAM_WALKERS = 17, 18, 19
RT_OPT_WIDE_RECORDS = 20
RT_OPT_LAUNCH_PIECES = 21
RT_INFO_WIDE_RECORDS, RT_INFO_WIDE_STACK_NEED, RT_INFO_TLAS_RECORDS = 107, 108, 109
RT_INFO_STACK_DEPTH, RT_INFO_RECORDS, RT_INFO_TRIANGLES = 100, 101, 102
RT_INFO_FAST_OBJECTS, RT_INFO_FAST_RECORDS, RT_INFO_FAST_TRIANGLES, RT_INFO_FAST_STACK_DEPTH = 103, 104, 105, 106


class RtError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("rt_amd error %d: %s" % (code, message))
        self.code = code


class PushConstants(C.Structure):
    """rt_push_constants (RaytracerTypes.h:22-52)."""
    _fields_ = [
        ("pixel00Location", C.c_float * 3), ("raysPerPixel", C.c_int32),
        ("pixelDeltaU", C.c_float * 3), ("maxBounces", C.c_int32),
        ("pixelDeltaV", C.c_float * 3), ("defocusAngle", C.c_float),
        ("cameraPosition", C.c_float * 3), ("parentBVHCount", C.c_int32),
        ("defocusDiskU", C.c_float * 3), ("accumulateFrames", C.c_int32),
        ("defocusDiskV", C.c_float * 3), ("frame", C.c_int32),
        ("sunDirection", C.c_float * 3), ("sunIntensity", C.c_float),
        ("sunColour", C.c_float * 3), ("renderMode", C.c_int32),
        ("triangleTestThreshold", C.c_int32), ("bvhNodeTestThreshold", C.c_int32),
        ("depthDebugScale", C.c_float), ("padding2", C.c_int32)]

    def as_bytes(self):
        return np.frombuffer(bytes(self), dtype=np.uint8).copy()

    def copy(self):
        return PushConstants.from_buffer_copy(bytes(self))


class CameraSettings(C.Structure):
    """rt_camera_settings (HardwareRenderer.h:68-75)."""
    _fields_ = [("cameraPosition", C.c_float * 3), ("cameraLookDirection", C.c_float * 3),
                ("cameraFov", C.c_float), ("focusDistance", C.c_float), ("defocusAngle", C.c_float)]


class Material(C.Structure):
    """rt_gpu_material (RaytracerTypes.h:127-138)."""
    _fields_ = [("albedo", C.c_float * 3), ("smoothness", C.c_float), ("fuzziness", C.c_float),
                ("emission", C.c_float), ("refractiveIndex", C.c_float), ("padding1", C.c_float),
                ("absorbtion", C.c_float * 3), ("padding3", C.c_float)]


class BufferView(C.Structure):
    _fields_ = [("data", C.c_void_p), ("count", C.c_size_t)]


_BINDING_NAMES = ["parentBVH", "objects", "materials", "triangleV0s", "triangleV1s", "triangleV2s", "triangleN0s",
                  "triangleN1s", "triangleN2s", "triangleUV0s", "triangleUV1s", "triangleUV2s", "aabbMins", "aabbMaxs",
                  "bvhNodeLeftChildren", "bvhNodeRightChildren", "bvhNodeTriangleStartIndices", "bvhNodeTriangleCounts"]
BINDING_ELEM_BYTES = [64, 80, 48] + [16] * 9 + [16, 16] + [4] * 4
BINDING_DTYPES = [np.uint8] * 3 + [np.float32] * 11 + [np.int32] * 4


class SceneBuffers(C.Structure):
    """rt_scene_buffers: the 18 storage buffers in binding order."""
    _fields_ = [(n, BufferView) for n in _BINDING_NAMES]

    @classmethod
    def from_arrays(cls, arrays):
        """arrays: list of 18 numpy arrays (raw bytes for bindings 0-2). Keeps them alive."""
        sb = cls()
        sb._keep = [np.ascontiguousarray(a) for a in arrays]
        for i, (name, a) in enumerate(zip(_BINDING_NAMES, sb._keep)):
            count = a.nbytes // BINDING_ELEM_BYTES[i]
            setattr(sb, name, BufferView(a.ctypes.data if a.nbytes else None, count))
        return sb

    def to_arrays(self):
        """numpy copies of the 18 buffers."""
        out = []
        for i, name in enumerate(_BINDING_NAMES):
            v = getattr(self, name)
            nbytes = v.count * BINDING_ELEM_BYTES[i]
            if not v.data or nbytes == 0:
                out.append(np.zeros(0, BINDING_DTYPES[i]))
                continue
            raw = (C.c_uint8 * nbytes).from_address(v.data)
            out.append(np.frombuffer(bytes(raw), dtype=BINDING_DTYPES[i]).copy())
        return out


class FrameStats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("bvhNodeTests", C.c_uint64), ("triangleTests", C.c_uint64), ("samples", C.c_uint64)]


_lib = None


def load_library():
    """Load librt_amd.so (built in-tree by raytracer_amd/csrc/Makefile). Fails loudly."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RtError(-100, "%s is missing: build it with `make -C raytracer_amd/csrc` (there is no fallback path)" % LIB_PATH)
    # PyTorch wheels bundle their own libamdhip64.so (same SONAME as the system one, but requested
    # under a different file name). Two HIP runtimes in one process cannot both own the GPU, so
    # let torch load its copy first: the loader then binds this library to that same runtime.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, ci, cf = C.c_void_p, C.c_int, C.c_float
    P = C.POINTER
    sigs = {
        "rt_create": (ci, [ci, P(vp)]), "rt_destroy": (None, [vp]), "rt_set_stream": (ci, [vp, vp]),
        "rt_set_option": (ci, [vp, ci, ci]), "rt_get_option": (ci, [vp, ci, P(ci)]),
        "rt_upload_scene": (ci, [vp, P(SceneBuffers)]), "rt_resize": (ci, [vp, ci, ci]),
        "rt_bind_images": (ci, [vp, vp, vp, ci, ci]), "rt_clear_accumulation": (ci, [vp]),
        "rt_render_frame": (ci, [vp, P(PushConstants)]), "rt_render_frames": (ci, [vp, P(PushConstants), ci, ci, ci]),
        "rt_sync": (ci, [vp]), "rt_read_image": (ci, [vp, ci, vp, C.c_size_t]), "rt_write_accum": (ci, [vp, vp, C.c_size_t]),
        "rt_resolve": (ci, [vp, ci]), "rt_frame_stats_compute": (ci, [vp, P(PushConstants), P(FrameStats)]),
        "rt_frame_stats_noalloc": (ci, [vp, P(PushConstants), vp]),
        "rt_comm_unique_id": (ci, [vp]), "rt_comm_init": (ci, [vp, vp, ci, ci]), "rt_comm_destroy": (ci, [vp]),
        "rt_comm_info": (ci, [vp, P(ci), P(ci)]), "rt_merge_accum": (ci, [vp]), "rt_gather_tiles": (ci, [vp]),
        "rt_merge_accum_local": (ci, [P(vp), ci]), "rt_gather_tiles_local": (ci, [P(vp), ci]),
        "rt_schedule_stats": (ci, [vp, P(PushConstants), P(C.c_uint64)]),
        "rt_last_render_ms": (ci, [vp, P(cf), P(cf)]), "rt_image_size": (ci, [vp, P(ci), P(ci)]),
        "rt_last_error": (C.c_char_p, [vp]), "rt_version": (C.c_char_p, []),
        "rt_camera_to_push_constants": (ci, [P(CameraSettings), ci, ci, P(PushConstants)]),
        "rt_default_push_constants": (None, [P(PushConstants)]), "rt_default_camera": (None, [P(CameraSettings)]),
        "rt_default_material": (None, [P(Material)]),
        "rt_scene_create": (vp, []), "rt_scene_destroy": (None, [vp]), "rt_scene_last_error": (C.c_char_p, [vp]),
        "rt_scene_add_material": (ci, [vp, P(Material)]), "rt_scene_material_index": (ci, [vp, P(Material)]),
        "rt_scene_load_model": (ci, [vp, C.c_char_p]),
        "rt_scene_add_object": (ci, [vp, C.c_char_p, P(cf), P(cf), P(cf), P(Material)]),
        "rt_scene_set_object": (ci, [vp, ci, P(cf), P(cf), P(cf), ci]), "rt_scene_set_material": (ci, [vp, ci, P(Material)]),
        "rt_scene_clear_objects": (ci, [vp]), "rt_scene_clear_materials": (ci, [vp]),
        "rt_scene_object_count": (ci, [vp]), "rt_scene_material_count": (ci, [vp]),
        "rt_scene_get_object": (ci, [vp, ci, C.c_char_p, C.c_size_t, P(cf), P(cf), P(cf), P(ci)]),
        "rt_scene_get_material": (ci, [vp, ci, P(Material)]),
        "rt_scene_buffer_data": (ci, [vp]), "rt_scene_get_buffers": (ci, [vp, P(SceneBuffers)]),
        "rt_scene_init_default": (ci, [vp, C.c_char_p, C.c_char_p]),
        "rt_scene_model_info": (ci, [vp, C.c_char_p, P(ci), P(ci), P(ci), P(ci)]),
        "rt_rgba32f_to_rgba8": (None, [vp, vp, C.c_size_t]),
        "rt_write_png_rgba8": (ci, [C.c_char_p, vp, ci, ci]), "rt_save_png": (ci, [vp, C.c_char_p]),
        "rt_produce_render": (ci, [vp, P(PushConstants), ci, C.c_char_p, P(C.c_double)]),
        "rt_math_apply": (ci, [vp, ci, vp, vp, C.c_size_t]),
    }
    for name, (res, args) in sigs.items():
        fn = getattr(L, name)   # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


EXPORTED_SYMBOLS = [
    "rt_create", "rt_destroy", "rt_set_stream", "rt_set_option", "rt_get_option", "rt_upload_scene", "rt_resize",
    "rt_bind_images", "rt_clear_accumulation", "rt_render_frame", "rt_render_frames", "rt_sync", "rt_read_image",
    "rt_write_accum", "rt_resolve", "rt_frame_stats_compute", "rt_schedule_stats", "rt_last_render_ms", "rt_image_size", "rt_last_error",
    "rt_version", "rt_camera_to_push_constants", "rt_default_push_constants", "rt_default_camera", "rt_default_material",
    "rt_scene_create", "rt_scene_destroy", "rt_scene_last_error", "rt_scene_add_material", "rt_scene_material_index",
    "rt_scene_load_model", "rt_scene_add_object", "rt_scene_set_object", "rt_scene_set_material", "rt_scene_buffer_data",
    "rt_scene_get_buffers", "rt_scene_init_default", "rt_scene_model_info", "rt_rgba32f_to_rgba8", "rt_write_png_rgba8",
    "rt_save_png", "rt_produce_render", "rt_math_apply", "rt_scene_object_count", "rt_scene_material_count",
    "rt_scene_get_object", "rt_scene_get_material", "rt_frame_stats_noalloc", "rt_comm_unique_id", "rt_comm_init",
    "rt_comm_destroy", "rt_comm_info", "rt_scene_clear_objects", "rt_scene_clear_materials", "rt_merge_accum", "rt_gather_tiles", "rt_merge_accum_local", "rt_gather_tiles_local"]


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


def default_push_constants():
    pc = PushConstants()
    load_library().rt_default_push_constants(C.byref(pc))
    return pc


def default_camera():
    cam = CameraSettings()
    load_library().rt_default_camera(C.byref(cam))
    return cam


def default_material(albedo=(1.0, 1.0, 1.0), **kw):
    m = Material()
    load_library().rt_default_material(C.byref(m))
    m.albedo = _f3(albedo)
    for k, v in kw.items():
        if k == "absorbtion":
            m.absorbtion = _f3(v)
        else:
            setattr(m, k, v)
    return m


def camera_to_push_constants(cam, width, height, pc):
    rc = load_library().rt_camera_to_push_constants(C.byref(cam), width, height, C.byref(pc))
    if rc != 0:
        raise RtError(rc, load_library().rt_last_error(None).decode())
    return pc


def rgba32f_to_rgba8(img):
    img = np.ascontiguousarray(img, np.float32)
    out = np.zeros(img.shape, np.uint8)
    load_library().rt_rgba32f_to_rgba8(img.ctypes.data, out.ctypes.data, img.size // 4)
    return out


def comm_unique_id():
    """rt_comm_unique_id: 128 bytes that rank 0 creates and every rank passes to Renderer.comm_init."""
    buf = (C.c_ubyte * 128)()
    rc = load_library().rt_comm_unique_id(buf)
    if rc != 0:
        raise RtError(rc, load_library().rt_last_error(None).decode())
    return bytes(buf)


def _ctx_array(renderers):
    return (C.c_void_p * len(renderers))(*[r.h for r in renderers])


def merge_accum_local(renderers):
    """rt_merge_accum_local: sum the accumulation images of several contexts of this process."""
    rc = load_library().rt_merge_accum_local(_ctx_array(renderers), len(renderers))
    if rc != 0:
        raise RtError(rc, load_library().rt_last_error(None).decode())


def gather_tiles_local(renderers):
    """rt_gather_tiles_local: exchange the owned tiles among several contexts of this process."""
    rc = load_library().rt_gather_tiles_local(_ctx_array(renderers), len(renderers))
    if rc != 0:
        raise RtError(rc, load_library().rt_last_error(None).decode())


def write_png_rgba8(path, rgba8):
    """rt_write_png_rgba8: (H, W, 4) uint8 -> PNG file."""
    img = np.ascontiguousarray(rgba8, np.uint8)
    if img.ndim != 3 or img.shape[2] != 4:
        raise ValueError("expected an (H, W, 4) uint8 image")
    rc = load_library().rt_write_png_rgba8(str(path).encode(), img.ctypes.data, img.shape[1], img.shape[0])
    if rc != 0:
        raise RtError(rc, "cannot write %s" % path)


class Scene:
    """Host-side scene with the reference's LoadModel / AddSceneObject / BufferSceneData semantics."""

    def __init__(self):
        self.lib = load_library()
        self.h = self.lib.rt_scene_create()
        if not self.h:
            raise RtError(-3, "rt_scene_create failed")

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.rt_scene_destroy(self.h)
            self.h = None

    def _check(self, rc):
        if rc < 0:
            raise RtError(rc, self.lib.rt_scene_last_error(self.h).decode())
        return rc

    def add_material(self, m):
        return self._check(self.lib.rt_scene_add_material(self.h, C.byref(m)))

    def material_index(self, m):
        return self.lib.rt_scene_material_index(self.h, C.byref(m))

    def set_material(self, index, m):
        self._check(self.lib.rt_scene_set_material(self.h, index, C.byref(m)))

    def load_model(self, path):
        self._check(self.lib.rt_scene_load_model(self.h, path.encode()))

    def add_object(self, model, position=(0, 0, 0), rotation=(0, 0, 0), scale=(1, 1, 1), material=None):
        return self._check(self.lib.rt_scene_add_object(self.h, model.encode(), _f3(position), _f3(rotation), _f3(scale),
                                                        C.byref(material) if material is not None else None))

    def set_object(self, index, position, rotation, scale, material_index):
        self._check(self.lib.rt_scene_set_object(self.h, index, _f3(position), _f3(rotation), _f3(scale), material_index))

    def clear_objects(self):
        self._check(self.lib.rt_scene_clear_objects(self.h))

    def clear_materials(self):
        self._check(self.lib.rt_scene_clear_materials(self.h))

    def object_count(self):
        return self.lib.rt_scene_object_count(self.h)

    def material_count(self):
        return self.lib.rt_scene_material_count(self.h)

    def get_object(self, index):
        """-> dict(model, position, rotation, scale, material_index): one entry of m_sceneObjects."""
        name = C.create_string_buffer(4096)
        pos, rot, scl, mat = _f3((0, 0, 0)), _f3((0, 0, 0)), _f3((0, 0, 0)), C.c_int(0)
        self._check(self.lib.rt_scene_get_object(self.h, index, name, len(name), pos, rot, scl, C.byref(mat)))
        return {"model": name.value.decode(), "position": tuple(pos), "rotation": tuple(rot), "scale": tuple(scl),
                "material_index": mat.value}

    def get_material(self, index):
        m = Material()
        self._check(self.lib.rt_scene_get_material(self.h, index, C.byref(m)))
        return m

    def buffer_scene_data(self):
        self._check(self.lib.rt_scene_buffer_data(self.h))

    def init_default(self, models_dir=MODELS_DIR, dragon_file="dragon_lowres.obj"):
        self._check(self.lib.rt_scene_init_default(self.h, models_dir.encode(), dragon_file.encode()))

    def model_info(self, path):
        v = [C.c_int() for _ in range(4)]
        self._check(self.lib.rt_scene_model_info(self.h, path.encode(), *[C.byref(x) for x in v]))
        return tuple(x.value for x in v)

    def buffers(self):
        sb = SceneBuffers()
        self._check(self.lib.rt_scene_get_buffers(self.h, C.byref(sb)))
        sb._owner = self
        return sb


class Renderer:
    """One GPU context: scene in HBM, two W x H rgba32f images, one HIP stream."""

    def __init__(self, device=0):
        self.lib = load_library()
        h = C.c_void_p()
        rc = self.lib.rt_create(device, C.byref(h))
        if rc != 0:
            raise RtError(rc, self.lib.rt_last_error(None).decode())
        self.h = h
        self.width = self.height = 0

    def close(self):
        if getattr(self, "h", None):
            self.lib.rt_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def _check(self, rc):
        if rc != 0:
            raise RtError(rc, self.lib.rt_last_error(self.h).decode())

    def set_option(self, option, value):
        self._check(self.lib.rt_set_option(self.h, option, value))

    def get_option(self, option):
        v = C.c_int(0)
        self._check(self.lib.rt_get_option(self.h, option, C.byref(v)))
        return v.value

    def set_stream(self, stream_handle):
        self._check(self.lib.rt_set_stream(self.h, stream_handle))

    def upload_scene(self, buffers):
        if not isinstance(buffers, SceneBuffers):
            buffers = SceneBuffers.from_arrays(buffers)
        self._check(self.lib.rt_upload_scene(self.h, C.byref(buffers)))

    def resize(self, width, height):
        self._check(self.lib.rt_resize(self.h, width, height))
        self.width, self.height = width, height

    def bind_images(self, draw_ptr, accum_ptr, width, height):
        self._check(self.lib.rt_bind_images(self.h, draw_ptr, accum_ptr, width, height))
        self.width, self.height = width, height

    def clear_accumulation(self):
        self._check(self.lib.rt_clear_accumulation(self.h))

    def render_frame(self, pc):
        self._check(self.lib.rt_render_frame(self.h, C.byref(pc)))

    def render_frames(self, pc, first_frame, stride, count):
        self._check(self.lib.rt_render_frames(self.h, C.byref(pc), first_frame, stride, count))

    def sync(self):
        self._check(self.lib.rt_sync(self.h))

    def read_image(self, which=RT_IMAGE_DRAW):
        out = np.empty((self.height, self.width, 4), np.float32)
        self._check(self.lib.rt_read_image(self.h, which, out.ctypes.data, out.nbytes))
        return out

    def write_accum(self, img):
        img = np.ascontiguousarray(img, np.float32)
        self._check(self.lib.rt_write_accum(self.h, img.ctypes.data, img.nbytes))

    def resolve(self, total_frames):
        self._check(self.lib.rt_resolve(self.h, total_frames))

    def frame_stats(self, pc):
        st = FrameStats()
        self._check(self.lib.rt_frame_stats_compute(self.h, C.byref(pc), C.byref(st)))
        return st

    def frame_stats_noalloc(self, pc, out_ptr):
        """rt_frame_stats_noalloc: enqueue only; out_ptr is the address of 32 bytes the GPU can write
        (four uint64: rays, bvhNodeTests, triangleTests, samples), valid after sync()."""
        self._check(self.lib.rt_frame_stats_noalloc(self.h, C.byref(pc), out_ptr))

    # ---- several GPUs (one process per GPU): RCCL communicator owned by the library ----
    def comm_init(self, unique_id, rank, world):
        """rt_comm_init; unique_id: the 128 bytes rank 0 got from comm_unique_id()."""
        buf = (C.c_ubyte * 128).from_buffer_copy(bytes(unique_id))
        self._check(self.lib.rt_comm_init(self.h, buf, rank, world))

    def comm_destroy(self):
        self._check(self.lib.rt_comm_destroy(self.h))

    def comm_info(self):
        a, b = C.c_int(), C.c_int()
        self._check(self.lib.rt_comm_info(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def merge_accum(self):
        self._check(self.lib.rt_merge_accum(self.h))

    def gather_tiles(self):
        self._check(self.lib.rt_gather_tiles(self.h))

    def schedule_stats(self, pc):
        out = (C.c_uint64 * 16)()
        self._check(self.lib.rt_schedule_stats(self.h, C.byref(pc), out))
        return [int(v) for v in out]

    def save_png(self, path):
        """WriteDrawImageToFile: the draw image as an 8-bit RGBA PNG."""
        self._check(self.lib.rt_save_png(self.h, str(path).encode()))

    def produce_render(self, pc, render_frames, png_path=None):
        """ProduceRender: frames pc.frame .. render_frames-1, then the PNG; returns the wall-clock
        seconds the reference prints. pc.frame is advanced."""
        sec = C.c_double()
        self._check(self.lib.rt_produce_render(self.h, C.byref(pc), int(render_frames),
                                               str(png_path).encode() if png_path is not None else None, C.byref(sec)))
        return sec.value

    def math_apply(self, which, x):
        """The device's rt_math.h functions over an array: which = 'exp2' | 'exp' | 'log2' | 'acos' | 'div10000'."""
        x = np.ascontiguousarray(x, np.float32)
        y = np.empty_like(x)
        self._check(self.lib.rt_math_apply(self.h, {"exp2": 0, "exp": 1, "log2": 2, "acos": 3, "div10000": 4}[which], x.ctypes.data, y.ctypes.data, x.size))
        return y

    def last_render_ms(self):
        a, b = C.c_float(), C.c_float()
        self._check(self.lib.rt_last_render_ms(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value
