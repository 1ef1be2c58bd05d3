"""CPU tests of the oracle itself: known-answer values derived by hand from the reference's
Random.glsl, camera numbers implied by the shipped defaults, BVH structure counts, the OBJ
reader against the reference's own tinyobj (when the reference checkout is present) and the
committed golden vectors. Parity of the reference itself is unpinned (it ships no tests)."""
import hashlib
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as O

ROOT = O.ROOT
GOLDEN = os.path.join(ROOT, "tests", "golden")

# (x, y, frame, i) -> seed0, then three RandomFloatInRange draws as (state, state % 10000);
# computed from Random.glsl:4-12,28-41 with 32-bit wrap-around.
RNG_KAT = [
    ((0, 0, 0, 0), 0, [(663891101, 1101), (4140173061, 3061), (2480156085, 6085)]),
    ((1, 0, 0, 0), 3170179127, [(2109965997, 5997), (2109715639, 5639), (225364207, 4207)]),
    ((0, 1, 0, 0), 2349025055, [(2713974323, 4323), (3126664487, 4487), (2736548458, 8458)]),
    ((0, 0, 1, 0), 914087646, [(569070589, 589), (1332350803, 803), (3287257422, 7422)]),
    ((0, 0, 0, 1), 1994134165, [(844823004, 3004), (996515553, 5553), (2474642504, 2504)]),
    ((640, 360, 0, 0), 1001611554, [(382360336, 336), (1525823518, 3518), (1938090153, 153)]),
    ((1279, 719, 0, 0), 2449670263, [(2969752474, 2474), (3013857053, 7053), (931996402, 6402)]),
    ((2559, 1368, 63, 0), 1797293054, [(1902098679, 8679), (1747099992, 9992), (257855184, 5184)]),
    ((1, 2, 3, 4), 807802641, [(3331819512, 9512), (2500518630, 8630), (4264680753, 753)]),
]


def _py_seed(x, y, f, i):
    M = 0xFFFFFFFF
    s = (x * 374761393 + y * 668265263 + f * 1442695040 + i * 88963407) & M
    s = ((s ^ (s >> 16)) * 0x85ebca6b) & M
    s = ((s ^ (s >> 13)) * 0xc2b2ae35) & M
    return s ^ (s >> 16)


def _py_draw(s):
    M = 0xFFFFFFFF
    s = (s + 1) & M
    s = ((s ^ 61) ^ (s >> 16)) & M
    s = (s * 9) & M
    s = s ^ (s >> 4)
    s = (s * 0x27d4eb2d) & M
    s = s ^ (s >> 15)
    return s


@pytest.mark.parametrize("coords,seed0,chain", RNG_KAT)
def test_rng_known_answers(coords, seed0, chain):
    assert _py_seed(*coords) == seed0          # the table itself is consistent with the GLSL
    s = O.seed_from_coords(*coords)
    assert s == seed0
    ps = seed0
    for state, mod in chain:
        s, m = O.rng_draw(s)
        ps = _py_draw(ps)
        assert (s, m) == (state, mod)
        assert ps == state


def test_rng_random_states_match_python_model():
    rng = np.random.default_rng(7)
    for x, y, f, i in rng.integers(0, 4000, size=(200, 4)):
        s = O.seed_from_coords(int(x), int(y), int(f), int(i))
        assert s == _py_seed(int(x), int(y), int(f), int(i))
        assert O.rng_draw(s)[0] == _py_draw(s)


def test_sample_square_kat():
    # SampleSquare does one extra seed+=1 before its two draws (Random.glsl:43-49)
    ox, oy = O.sample_square(1001611554)
    assert ox == np.float32(np.float32(0.6570) - np.float32(0.5))
    assert oy == np.float32(np.float32(0.1372) - np.float32(0.5))


def test_camera_defaults_1280x720():
    pc = O.default_pc()
    O.camera_push_constants(O.pc_bytes(pc), W=1280, H=720)
    # vh = 2*tan(45deg)*10 = 20, vw = 20*16/9; dU = vw/1280, dV = -vh/720; UL = (-vw/2, 1+10, 5-10)
    np.testing.assert_allclose(pc["pixelDeltaU"][0], [20.0 * 16 / 9 / 1280, 0, 0], rtol=2e-7, atol=0)
    np.testing.assert_allclose(pc["pixelDeltaV"][0], [0, -20.0 / 720, 0], rtol=2e-7, atol=0)
    np.testing.assert_allclose(pc["pixel00Location"][0], [-17.7777778 + 0.0138889, 11 - 0.0138889, -5.0], rtol=1e-6)
    np.testing.assert_array_equal(pc["cameraPosition"][0], [0, 1, 5])
    np.testing.assert_array_equal(pc["defocusDiskU"][0], [0, 0, 0])
    assert pc["defocusAngle"][0] == 0


STRUCTURE = {  # model -> (triangles, child nodes, depth below the parent, largest leaf)
    "flat_quad.obj": (2, 0, 0, 2), "icosphere.obj": (80, 98, 6, 2), "dragon_lowres.obj": (8712, 8806, 22, 119),
    "teapot.obj": (6320, 5594, 20, 36), "smooth_sphere.obj": (3840, 4478, 14, 5), "cube.obj": (12, 2, 1, 6),
    "cottage_obj.obj": (478, 10, 5, 287)}


@pytest.mark.parametrize("model", sorted(STRUCTURE))
def test_bvh_structure(model):
    s = O.OracleScene()
    p = os.path.join(O.MODELS_DIR, model)
    s.load_model(p)
    assert s.model_info(p) == STRUCTURE[model]


def test_default_scene_layout(default_scene_arrays):
    b = default_scene_arrays
    assert b[0].nbytes == 8 * 64 and b[1].nbytes == 8 * 80 and b[2].nbytes == 9 * 48
    assert b[3].size == 4 * 8794 and b[12].size == 4 * 8904 and b[14].size == 8904
    parents = b[0].view(np.int32).reshape(8, 16)
    # model load order quad, icosphere, dragon fixes the global offsets (HardwareRenderer.cpp:629-636)
    assert list(parents[0, 8:12]) == [0, 1, 2, 80]          # icosphere: nodes 0.., triangles 2..81
    assert list(parents[1, 8:12]) == [98, 99, 82, 8712]     # dragon: nodes 98.., triangles 82..
    assert list(parents[2, 8:12]) == [-1, -1, 0, 2]         # quad: leaf parent, triangles 0..1
    assert list(parents[:, 12]) == list(range(8))
    # every wall gets a 10x10x10 world box (flat quad's zero extent is widened to +-1, then scaled by 5)
    pf = b[0].view(np.float32).reshape(8, 16)
    for i in range(2, 8):
        np.testing.assert_allclose(pf[i, 4:7] - pf[i, 0:3], [10, 10, 10], atol=1e-5)


def test_obj_reader_matches_committed_hashes():
    want = {}
    for line in open(os.path.join(GOLDEN, "obj_corner_hashes.txt")):
        name, n, h = line.split()
        want[name] = (int(n), h)
    for name, (n, h) in want.items():
        pos, nrm = O.obj_corners(os.path.join(O.MODELS_DIR, name))
        assert pos.shape[0] == n
        assert hashlib.sha256(pos.tobytes() + nrm.tobytes()).hexdigest() == h


@pytest.mark.skipif(not os.path.exists("/root/reference/contrib/Generic/include/tiny_obj_loader.h"),
                    reason="reference checkout not present (GPU box)")
def test_obj_reader_matches_reference_tinyobj(tmp_path):
    """The one piece of the reference that builds here: its vendored OBJ parser."""
    subprocess.check_call(["make", "-C", O.ORACLE_DIR, "ref"], stdout=subprocess.DEVNULL)
    exe = os.path.join(O.ORACLE_DIR, "_ref", "ref_tinyobj_dump")
    for name in sorted(os.listdir(O.MODELS_DIR)):
        if not name.endswith(".obj"):
            continue
        path = os.path.join(O.MODELS_DIR, name)
        out = str(tmp_path / "dump.bin")
        subprocess.check_call([exe, path, out])
        raw = open(out, "rb").read()
        n = int(np.frombuffer(raw[:8], np.int64)[0])
        ref = np.frombuffer(raw[8:], np.float32).reshape(n, 6)
        pos, nrm = O.obj_corners(path)
        assert pos.shape[0] == n
        assert np.array_equal(pos.view(np.uint32), np.ascontiguousarray(ref[:, :3]).view(np.uint32))
        assert np.array_equal(nrm.view(np.uint32), np.ascontiguousarray(ref[:, 3:]).view(np.uint32))


def test_oracle_matches_golden_full_frame(default_scene_arrays):
    g = np.load(os.path.join(GOLDEN, "default_128x72_f0f1.npz"))
    W, H = 128, 72
    pc = O.default_pc(8)
    pb = O.pc_bytes(pc)
    O.camera_push_constants(pb, W=W, H=H)
    acc = np.zeros((H, W, 4), np.float32)
    for f in range(2):
        pc["frame"] = f
        r = O.render(default_scene_arrays, pb, W, H, accum=acc, counters=True)
        acc = r["accum"]
    assert np.array_equal(r["accum"].view(np.uint32), g["accum"].view(np.uint32))
    assert np.array_equal(r["out"].view(np.uint32), g["out"].view(np.uint32))
    assert np.array_equal(r["tri"], g["tri"]) and np.array_equal(r["node"], g["node"]) and np.array_equal(r["rays"], g["rays"])
    # accumulation alpha is never written by the shader; the draw alpha copies it while accumulating
    assert np.all(r["accum"][..., 3] == 0) and np.all(r["out"][..., 3] == 0)


def test_oracle_matches_golden_crop(default_scene_arrays):
    g = np.load(os.path.join(GOLDEN, "default_1280x720_crop.npz"))
    x0, y0, x1, y1 = [int(v) for v in g["window"]]
    pc = O.default_pc(8)
    pb = O.pc_bytes(pc)
    O.camera_push_constants(pb, W=1280, H=720)
    r = O.render(default_scene_arrays, pb, 1280, 720, window=(x0, y0, x1, y1), counters=True)
    assert np.array_equal(r["accum"][y0:y1, x0:x1].view(np.uint32), g["accum"].view(np.uint32))
    assert np.array_equal(r["tri"][y0:y1, x0:x1], g["tri"])
    assert np.array_equal(r["node"][y0:y1, x0:x1], g["node"])


def test_oracle_properties(default_scene_arrays):
    """Size-independent facts of the algorithm."""
    W, H = 64, 36
    pc = O.default_pc(8)
    pb = O.pc_bytes(pc)
    O.camera_push_constants(pb, W=W, H=H)
    a = O.render(default_scene_arrays, pb, W, H, counters=True)
    # at most maxBounces+1 segments, each with at most one shadow ray
    assert a["rays"].max() <= 2 * (int(pc["maxBounces"][0]) + 1) and a["rays"].min() >= 1
    # every GetHit tests all 8 parent boxes at least
    assert np.all(a["node"] >= 8 * a["rays"])
    # two rays per pixel in one frame = sum of sample 0 and sample 1, halved (raytrace.comp:779-789)
    pc["raysPerPixel"] = 2
    b2 = O.render(default_scene_arrays, pb, W, H)
    assert np.all(np.isfinite(b2["accum"]))
    # not accumulating: draw alpha is 1 and the image is the frame itself
    pc["raysPerPixel"] = 1
    pc["accumulateFrames"] = 0
    c = O.render(default_scene_arrays, pb, W, H)
    assert np.all(c["out"][..., 3] == 1)
    np.testing.assert_array_equal(c["out"][..., :3], np.sqrt(np.maximum(a["accum"][..., :3], 0), dtype=np.float32))


def test_rejection_loop_that_never_accepts_is_cut_off():
    """Random.glsl:68-78 loops until a vector in the unit ball is drawn; the generator is a fixed
    permutation of the 32-bit states, and the state the Lambert branch starts from at 3840x2160, frame
    1010, pixel (3563, 2062), sample 0 sits on a cycle whose draws are all rejected (found by enumerating
    frames 0..1023 of that size; an independent Python model of the generator confirms the cycle here).
    The reference shader would spin for ever; the oracle and the kernels return the zero vector after
    65 536 rounds."""
    def wang(s):
        s = ((s ^ 61) ^ (s >> 16)) & M
        s = (s * 9) & M
        s = (s ^ (s >> 4)) & M
        s = (s * 0x27d4eb2d) & M
        return (s ^ (s >> 15)) & M
    M = 0xFFFFFFFF
    s0 = O.seed_from_coords(3563, 2062, 1010, 0)
    state = wang((s0 + 1) & M)                     # after MaterialScatter's scatterOrReflect draw
    assert O.rng_draw(s0)[0] == state
    # the python model: RandomUnitVector's loop from that state revisits a state within a few rounds without accepting
    s = (state + 1) & M
    seen = set()
    for _ in range(64):
        s = (s + 1) & M
        start = s
        v = []
        for _ in range(3):
            s = wang((s + 1) & M)
            r = np.float32(s % 10000) / np.float32(10000.0)
            v.append(np.float32(-1.0) * (np.float32(1.0) - r) + np.float32(1.0) * r)
        length = np.sqrt(np.float32(v[0] * v[0] + v[1] * v[1]) + v[2] * v[2], dtype=np.float32)
        lensq = np.float32(length * length)
        assert not (0.0 < lensq <= 1.0)
        if start in seen:
            break
        seen.add(start)
    else:
        raise AssertionError("no cycle found")
    assert O.random_unit_vector(state) == (0.0, 0.0, 0.0)
    # an ordinary state gives a unit vector
    x, y, z = O.random_unit_vector(O.seed_from_coords(10, 20, 3, 0))
    assert abs(x * x + y * y + z * z - 1.0) < 1e-6
