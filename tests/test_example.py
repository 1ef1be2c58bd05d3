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
