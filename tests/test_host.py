"""CPU tests of the product's host side: the C-ABI library loads and exports every symbol the
header declares, its scene builder reproduces the oracle's arrays bit for bit, and it refuses to
render without a GPU (no fallback path)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import oracle_lib as O
import raytracer_amd as R
from raytracer_amd import api
import scenes

ROOT = O.ROOT


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "rt_abi.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = R.load_library()
    names = _declared_functions()
    assert len(names) >= 35
    for n in names:
        assert hasattr(lib, n), "librt_amd.so does not export %s" % n
    assert sorted(api.EXPORTED_SYMBOLS) == names


def test_struct_sizes():
    assert C.sizeof(R.PushConstants) == 144 and C.sizeof(R.Material) == 48
    assert C.sizeof(R.CameraSettings) == 36 and C.sizeof(R.SceneBuffers) == 18 * 16
    assert R.PushConstants.sunDirection.offset == 96 and R.PushConstants.renderMode.offset == 124


def test_default_scene_arrays_equal_oracle(default_scene_arrays):
    s = R.Scene()
    s.init_default()
    got = s.buffers().to_arrays()
    assert len(got) == 18
    for i, (a, b) in enumerate(zip(got, default_scene_arrays)):
        assert a.tobytes() == b.tobytes(), "binding %d differs" % i


def test_mixed_scene_arrays_equal_oracle():
    materials, objects = scenes.mixed_recipe()
    a = scenes.build_product(materials, objects).buffers().to_arrays()
    b = scenes.build_oracle(materials, objects).bindings()
    for i, (x, y) in enumerate(zip(a, b)):
        assert x.tobytes() == y.tobytes(), "binding %d differs" % i


@pytest.mark.parametrize("model", ["flat_quad.obj", "icosphere.obj", "dragon_lowres.obj", "teapot.obj",
                                   "smooth_sphere.obj", "cube.obj", "cottage_obj.obj"])
def test_model_structure_equals_oracle(model):
    p = os.path.join(O.MODELS_DIR, model)
    s = R.Scene(); s.load_model(p)
    o = O.OracleScene(); o.load_model(p)
    assert s.model_info(p) == o.model_info(p)


@pytest.mark.parametrize("W,H,fov,focus,defocus,pos,look", [
    (1280, 720, 90.0, 10.0, 0.0, (0, 1, 5), (0, 0, -1)),
    (2560, 1369, 90.0, 10.0, 0.0, (0, 1, 5), (0, 0, -1)),
    (1920, 1080, 60.0, 7.5, 2.0, (0.3, 2.0, 12.0), (0.1, -0.1, -1.0)),
    (333, 77, 35.0, 3.25, 0.7, (-4.0, 0.5, 1.0), (0.7, 0.2, -0.4))])
def test_camera_push_constants_equal_oracle(W, H, fov, focus, defocus, pos, look):
    cam = R.CameraSettings((C.c_float * 3)(*pos), (C.c_float * 3)(*look), fov, focus, defocus)
    pc = R.default_push_constants()
    R.camera_to_push_constants(cam, W, H, pc)
    opc = O.default_pc(0)
    O.camera_push_constants(O.pc_bytes(opc), pos, look, fov, focus, defocus, W, H)
    assert bytes(pc) == O.pc_bytes(opc).tobytes()


def test_material_lookup_and_cache():
    s = R.Scene()
    red = api.default_material((0.8, 0.1, 0.1))
    assert s.material_index(red) == -1          # GetMaterialIndex: unknown material -> -1
    assert s.add_material(api.default_material((0.5, 0.5, 0.5))) == 0
    assert s.add_material(red) == 1
    assert s.material_index(red) == 1
    p = os.path.join(O.MODELS_DIR, "cube.obj")
    s.load_model(p); s.load_model(p)            # cached by path: second load adds nothing
    s.add_object(p, (0, 0, 0), (0, 0, 0), (1, 1, 1), red)
    s.add_object(p, (3, 0, 0), (0, 45, 0), (2, 2, 2), red)
    s.buffer_scene_data()
    arr = s.buffers().to_arrays()
    assert arr[3].size == 4 * 12 and arr[0].nbytes == 2 * 64
    objs = arr[1].view(np.int32).reshape(2, 20)
    assert list(objs[:, 2]) == [1, 1] and list(objs[:, 0]) == [0, 0]


def test_errors_are_reported_not_thrown():
    s = R.Scene()
    with pytest.raises(R.RtError) as e:
        s.load_model("/nonexistent/model.obj")
    assert "does not exist" in str(e.value)
    with pytest.raises(R.RtError) as e:
        s.load_model("/tmp/model.stl")
    assert "Only .obj" in str(e.value)
    s.add_object("/not/loaded.obj")
    with pytest.raises(R.RtError):
        s.buffer_scene_data()


def test_rgba8_conversion():
    img = np.array([[[0.0, 0.5, 1.0, 0.0], [-1.0, 2.0, 0.999, 0.3]]], np.float32)
    out = R.rgba32f_to_rgba8(img)
    assert out.tolist() == [[[0, 128, 255, 255], [0, 255, 255, 255]]]


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(R.RtError) as e:
        R.Renderer(0)
    assert e.value.code == -2 and "no CPU path" in str(e.value)


def decode_png(path):
    """Minimal PNG reader for 8-bit RGBA, filter type 0 (what rt_write_png_rgba8 writes), with the
    chunk CRCs and the zlib checksum verified."""
    import struct, zlib
    data = open(path, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, chunks = 8, []
    while pos < len(data):
        n, kind = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        crc, = struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])
        assert zlib.crc32(kind + body) == crc
        chunks.append((kind, body)); pos += 12 + n
    assert [k for k, _ in chunks][0] == b"IHDR" and chunks[-1][0] == b"IEND"
    w, h, depth, colour, comp, flt, lace = struct.unpack(">IIBBBBB", chunks[0][1])
    assert (depth, colour, comp, flt, lace) == (8, 6, 0, 0, 0)
    raw = zlib.decompress(b"".join(b for k, b in chunks if k == b"IDAT"))   # checks the Adler-32
    rows = np.frombuffer(raw, np.uint8).reshape(h, 1 + 4 * w)
    assert not rows[:, 0].any()
    return rows[:, 1:].reshape(h, w, 4)


@pytest.mark.parametrize("W,H", [(1, 1), (7, 3), (300, 70), (200, 90)])
def test_png_writer_round_trip(tmp_path, W, H):
    """The file WriteDrawImageToFile's PNG step produces decodes to the bytes passed in; 300x70 and
    200x90 cross the 65535-byte stored-block limit."""
    rng = np.random.default_rng(W * 1000 + H)
    img = rng.integers(0, 256, (H, W, 4), dtype=np.uint8)
    path = str(tmp_path / "x.png")
    api.write_png_rgba8(path, img)
    assert np.array_equal(decode_png(path), img)
    with pytest.raises(api.RtError):
        api.write_png_rgba8(str(tmp_path / "no_such_dir" / "x.png"), img)


def test_scene_lists_can_be_read_back():
    """rt_scene_get_object / rt_scene_get_material return what InitializeScene and later edits put in
    (the scene editor's view of m_sceneObjects / m_sceneMaterials, SceneEditorUI.hpp:23-97)."""
    import raytracer_amd as R
    s = R.Scene(); s.init_default()
    assert s.object_count() == 8 and s.material_count() == 9
    objs = [s.get_object(i) for i in range(8)]
    assert sum(o["model"].endswith("flat_quad.obj") for o in objs) == 6
    assert any(o["model"].endswith("dragon_lowres.obj") for o in objs)
    assert all(0 <= o["material_index"] < 9 for o in objs)
    m = s.get_material(0)
    assert tuple(m.albedo) == (0.5, 0.5, 0.5)            # the ground material comes first (HardwareRenderer.cpp:576-581)
    s.set_object(3, (1.0, 2.0, 3.0), (10.0, 20.0, 30.0), (2.0, 2.0, 2.0), 4)
    o = s.get_object(3)
    assert o["position"] == (1.0, 2.0, 3.0) and o["rotation"] == (10.0, 20.0, 30.0) and o["scale"] == (2.0, 2.0, 2.0)
    assert o["material_index"] == 4
    with pytest.raises(R.RtError):
        s.get_object(8)
    with pytest.raises(R.RtError):
        s.get_material(-1)
