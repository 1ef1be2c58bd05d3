"""The C++ example host (examples/offline_render.cpp) uses the library through the C ABI alone."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "examples", "offline_render")
MODELS = os.path.join(ROOT, "assets", "models")


def fnv1a(data):
    h = 1469598103934665603
    for chunk in np.frombuffer(data, np.uint8).reshape(-1):
        h = ((h ^ int(chunk)) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def test_example_builds_and_fails_loudly_without_a_gpu(tmp_path):
    subprocess.run(["make", "-C", os.path.join(ROOT, "examples")], check=True, capture_output=True)
    assert os.path.exists(EXE)
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the run is covered by the gpu test")
    r = subprocess.run([EXE, MODELS, str(tmp_path / "x.png"), "64", "36", "1"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "no HIP device" in r.stderr and not (tmp_path / "x.png").exists()


@pytest.mark.gpu
@pytest.mark.parametrize("pipeline", [0, 2], ids=["lockstep", "wavefront"])
def test_example_renders_the_same_image_as_the_python_binding(tmp_path, pipeline):
    from test_host import decode_png
    import raytracer_amd as R
    W, H, frames = 160, 90, 3
    png = str(tmp_path / "render.png")
    r = subprocess.run([EXE, MODELS, png, str(W), str(H), str(frames), str(pipeline)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "Render took" in r.stdout and "frame counter %d" % frames in r.stdout
    checksum = int(r.stdout.split("image fnv1a ")[1].split()[0], 16)
    s = R.Scene(); s.init_default()
    rd = R.Renderer(0)
    try:
        rd.upload_scene(s.buffers()); rd.resize(W, H)
        pc = R.default_push_constants(); R.camera_to_push_constants(R.default_camera(), W, H, pc); pc.parentBVHCount = 8
        rd.render_frames(pc, 0, 1, frames)
        img = rd.read_image(R.RT_IMAGE_DRAW)
    finally:
        rd.close()
    assert fnv1a(img.tobytes()) == checksum
    assert np.array_equal(decode_png(png), R.rgba32f_to_rgba8(img))


# ---- examples/viewer_loop.cpp: the HardwareRenderer-shaped C++ class of include/rt_renderer.hpp ----

VIEWER = os.path.join(ROOT, "examples", "viewer_loop")


def parse_steps(stdout):
    steps = {}
    for line in stdout.splitlines():
        w = line.split()
        if w and w[0] == "step":
            steps[int(w[1])] = {"frame": int(w[3]), "pos": tuple(float(x) for x in w[5:8]), "dir": tuple(float(x) for x in w[9:12]),
                                "locked": int(w[13]), "refresh": int(w[15])}
    return steps


def test_camera_controller_known_answers():
    """CameraController.cpp:11-101 by hand: W is ignored until Escape locks the mouse; locked, W for 0.5 s at speed 16
    moves 8 along the look direction; a mouse delta of 1.8 at sensitivity 50 over 1 s turns yaw from -90 to 0 degrees
    (look along +x); D for 0.25 s strafes 4 along cross(look, up) = +z."""
    subprocess.run(["make", "-C", os.path.join(ROOT, "examples")], check=True, capture_output=True)
    r = subprocess.run([VIEWER, "--camera-only"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    s = parse_steps(r.stdout)
    assert s[4]["pos"] == (0.0, 1.0, 5.0) and s[4]["locked"] == 0
    assert s[5]["pos"] == (0.0, 1.0, -3.0) and s[5]["locked"] == 1 and s[5]["dir"] == (0.0, 0.0, -1.0)
    assert s[6]["dir"] == (1.0, 0.0, 0.0) and s[6]["pos"] == (0.0, 1.0, -3.0)
    assert s[7]["pos"] == (0.0, 1.0, 1.0) and s[7]["dir"] == (1.0, 0.0, 0.0)


@pytest.mark.gpu
def test_viewer_loop_session_matches_the_oracle(tmp_path):
    """The frame counter follows MainLoop / RefreshAccumulation / ProduceRender (HardwareRenderer.cpp:840-848, 921-922,
    1002-1005, 1039-1043), and the PNG the session writes equals the CPU oracle's render of the final camera and of
    the edited scene (one material changed, one object moved, one object removed, one material and one object
    appended), converted with the reference's float -> 8-bit rule."""
    from test_host import decode_png
    import oracle_lib as O
    import scenes
    import raytracer_amd as R
    png = str(tmp_path / "session.png")
    r = subprocess.run([VIEWER, MODELS, png], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    s = parse_steps(r.stdout)
    # three idle frames accumulate; every camera move, the resize and the scene edit restart at frame 0 (+1 after the
    # step); the render request runs the counter up to SetRenderFrames' 8
    assert [s[k]["frame"] for k in range(1, 12)] == [1, 2, 3, 4, 1, 1, 1, 1, 2, 3, 8]
    assert all(s[k]["refresh"] == 0 for k in range(1, 12))
    assert "size 256 144 objects 8 materials 10" in r.stdout and "Render took" in r.stdout
    W, H = 256, 144
    # the edited scene, rebuilt for the oracle from the default scene's lists
    sc = R.Scene(); sc.init_default()
    mats = [np.frombuffer(bytes(sc.get_material(i)), np.float32).copy() for i in range(sc.material_count())]
    mats[1][0:3] = (0.9, 0.2, 0.2)
    mats.append(scenes.mat((0.2, 0.3, 0.9)))
    objs = []
    for i in range(sc.object_count()):
        o = sc.get_object(i)
        objs.append([os.path.basename(o["model"]), tuple(np.float32(v) for v in o["position"]), o["rotation"], o["scale"], o["material_index"]])
    p = np.array(objs[6][1], np.float32); p[1] += np.float32(0.5); objs[6][1] = tuple(p)
    del objs[7]
    objs.append(["cube.obj", (3.0, 1.0, 2.0), (0.0, 30.0, 0.0), (1.0, 1.0, 1.0), 9])
    arrays = scenes.build_oracle(mats, [tuple(o) for o in objs]).bindings()
    opc = scenes.make_pc(arrays, W, H, cam_pos=s[11]["pos"], look=s[11]["dir"])
    acc = np.zeros((H, W, 4), np.float32)
    for f in range(8):
        opc["frame"] = f
        res = O.render(arrays, O.pc_bytes(opc), W, H, accum=acc)
        acc = res["accum"]
    assert np.array_equal(decode_png(png), R.rgba32f_to_rgba8(res["out"]))


import ctypes
C_float3 = ctypes.c_float * 3
