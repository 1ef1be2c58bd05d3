"""GPU parity tests: the HIP path (through the C ABI of librt_amd.so) against the CPU oracle on
the same scene arrays, seeds and push constants.

Bars, stated per test:
  * strict flavour (no FMA contraction), scenes without transcendental shading (the default
    scene): accumulation image, draw image and the integer counter images are BIT-EXACT;
  * scenes that reach exp/pow/acos (glass, sky): >= 99.5 % of pixels within
    |d| <= 1e-4 + 1e-4*|ref| per channel (device libm differs from glibc in the last bits, and
    a last-bit difference can flip a discrete branch of a path), counters equal on >= 99.5 %;
  * fast flavour (FMA contraction allowed): same 99.5 % bar at 1 spp, PSNR >= 45 dB on the
    gamma image after 64 accumulated frames.
"""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib as O
import scenes

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(O.ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def R():
    import raytracer_amd
    return raytracer_amd


@pytest.fixture(scope="module", params=[0, 1, 2, 3, 4], ids=["lockstep", "scheduled", "wavefront", "two-samples", "team"])
def rdr(R, default_scene_arrays, request):
    """One renderer per kernel choice: every test that takes `rdr` runs on all pipelines."""
    r = R.Renderer(0)
    r.set_option(R.RT_OPT_PIPELINE, request.param)
    r.upload_scene(default_scene_arrays)
    yield r
    r.close()


def to_pc(R, opc):
    return R.PushConstants.from_buffer_copy(O.pc_bytes(opc).tobytes())


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def within(a, ref, tol_abs=1e-4, tol_rel=1e-4):
    with np.errstate(invalid="ignore"):
        ok = np.abs(a - ref) <= tol_abs + tol_rel * np.abs(ref)
    ok |= np.isnan(a) & np.isnan(ref)
    return float(np.mean(np.all(ok, axis=-1)))


def gpu_frames(R, r, opc, frames, W, H, clear=True):
    r.resize(W, H) if (r.width, r.height) != (W, H) else None
    if clear:
        r.clear_accumulation()
    frames = list(frames)
    steps = {b - a for a, b in zip(frames, frames[1:])}
    if r.get_option(R.RT_OPT_PIPELINE) == 3 and len(frames) >= 2 and len(steps) == 1 and min(steps) >= 1:
        # the two-samples-per-lane kernel only runs when a launch holds at least two samples of a pixel:
        # hand the whole frame set to rt_render_frames (single-frame dispatches would run the lock-step kernel)
        opc["frame"] = frames[0]
        r.render_frames(to_pc(R, opc), frames[0], steps.pop(), len(frames))
        opc["frame"] = frames[-1]
    else:
        for f in frames:
            opc["frame"] = f
            r.render_frame(to_pc(R, opc))
    return r.read_image(R.RT_IMAGE_ACCUM), r.read_image(R.RT_IMAGE_DRAW)


def oracle_frames(arrays, opc, frames, W, H, counters=False):
    acc = np.zeros((H, W, 4), np.float32)
    res = None
    for f in frames:
        opc["frame"] = f
        res = O.render(arrays, O.pc_bytes(opc), W, H, accum=acc, counters=counters)
        acc = res["accum"]
    return res


def test_default_scene_bit_exact_vs_golden_and_oracle(R, rdr, default_scene_arrays):
    g = np.load(os.path.join(GOLDEN, "default_128x72_f0f1.npz"))
    W, H = 128, 72
    rdr.set_option(R.RT_OPT_FAST_MATH, 0)
    opc = scenes.make_pc(default_scene_arrays, W, H)
    acc, out = gpu_frames(R, rdr, opc, [0, 1], W, H)
    assert np.array_equal(bits(acc), bits(g["accum"]))
    assert np.array_equal(bits(out), bits(g["out"]))
    o = oracle_frames(default_scene_arrays, opc, [0, 1], W, H)
    assert np.array_equal(bits(acc), bits(o["accum"])) and np.array_equal(bits(out), bits(o["out"]))


def test_config1_1280x720_full_frame_bit_exact(R, rdr, default_scene_arrays):
    """Benchmark configuration 0: default scene, 1280x720, 1 spp, shipped bounce count, frame 0."""
    W, H = 1280, 720
    rdr.set_option(R.RT_OPT_FAST_MATH, 0)
    opc = scenes.make_pc(default_scene_arrays, W, H)
    acc, out = gpu_frames(R, rdr, opc, [0], W, H)
    o = oracle_frames(default_scene_arrays, opc, [0], W, H, counters=True)
    assert np.array_equal(bits(acc), bits(o["accum"]))
    assert np.array_equal(bits(out), bits(o["out"]))
    g = np.load(os.path.join(GOLDEN, "default_1280x720_crop.npz"))
    x0, y0, x1, y1 = [int(v) for v in g["window"]]
    assert np.array_equal(bits(acc[y0:y1, x0:x1]), bits(g["accum"]))
    # work counters summed over the frame, from the counting build of the kernel
    opc["frame"] = 0
    st = rdr.frame_stats(to_pc(R, opc))
    assert (st.rays, st.bvhNodeTests, st.triangleTests) == (int(o["rays"].sum()), int(o["node"].sum()), int(o["tri"].sum()))
    assert st.samples == W * H


@pytest.mark.parametrize("mode", [1, 2, 3, 4])
def test_debug_modes_bit_exact(R, rdr, default_scene_arrays, mode):
    W, H = 160, 90
    rdr.set_option(R.RT_OPT_FAST_MATH, 0)
    opc = scenes.make_pc(default_scene_arrays, W, H, renderMode=mode)
    rdr.resize(W, H)
    rdr.render_frame(to_pc(R, opc))
    got = rdr.read_image(R.RT_IMAGE_DRAW)
    want = O.render(default_scene_arrays, O.pc_bytes(opc), W, H)["out"]
    assert np.array_equal(bits(got), bits(want))
    if mode == 2:
        g = np.load(os.path.join(GOLDEN, "default_128x72_depth.npz"))
        opc2 = scenes.make_pc(default_scene_arrays, 128, 72, renderMode=2)
        rdr.resize(128, 72)
        rdr.render_frame(to_pc(R, opc2))
        assert np.array_equal(bits(rdr.read_image(R.RT_IMAGE_DRAW)), bits(g["out"]))


def test_counter_images_equal(R, rdr, default_scene_arrays):
    """Per-pixel triangleTests / bvhNodeTests (reference traversal order) through modes 3/4 with
    thresholds that keep the grey ramp fine-grained (grey = int(tests / threshold * 255))."""
    W, H = 160, 90
    rdr.set_option(R.RT_OPT_FAST_MATH, 0)
    rdr.resize(W, H)
    o = O.render(default_scene_arrays, O.pc_bytes(scenes.make_pc(default_scene_arrays, W, H)), W, H, counters=True)
    assert o["tri"].max() > 100 and o["node"].max() > 100
    for mode in (3, 4):
        for thr in (255, 1000, 4000):
            opc = scenes.make_pc(default_scene_arrays, W, H, renderMode=mode, triangleTestThreshold=thr, bvhNodeTestThreshold=thr)
            rdr.render_frame(to_pc(R, opc))
            got = rdr.read_image(R.RT_IMAGE_DRAW)
            want = O.render(default_scene_arrays, O.pc_bytes(opc), W, H)["out"]
            assert np.array_equal(bits(got), bits(want))
            assert len(np.unique(want[..., 1])) > 8      # not a trivially flat image


@pytest.mark.parametrize("fields", [
    dict(raysPerPixel=3), dict(maxBounces=0), dict(maxBounces=7), dict(accumulateFrames=0),
    dict(raysPerPixel=2, maxBounces=1, accumulateFrames=0),
    dict(sunDirection=(-0.3, -0.9, 0.2), sunIntensity=2.0)])
def test_settings_surface_bit_exact(R, rdr, default_scene_arrays, fields):
    W, H = 96, 54
    rdr.set_option(R.RT_OPT_FAST_MATH, 0)
    opc = scenes.make_pc(default_scene_arrays, W, H, **fields)
    acc, out = gpu_frames(R, rdr, opc, [0, 1, 2], W, H)
    o = oracle_frames(default_scene_arrays, opc, [0, 1, 2], W, H)
    assert np.array_equal(bits(acc), bits(o["accum"])) and np.array_equal(bits(out), bits(o["out"]))


def test_sun_moving_between_dispatches(R, rdr, default_scene_arrays):
    """The order in which shadow rays visit the objects is re-sorted when the sun direction changes (towards
    the sun first); it must change no result: a sun that moves from dispatch to dispatch on ONE context -
    from above, from the side, from below, switched off, a zero vector, back again - against the oracle.
    (A zero direction normalises to NaN in the shader as well: the same pixels turn NaN on both sides, and
    only there the bits - sign and payload of a NaN - are not compared.)"""
    W, H = 96, 54
    rdr.set_option(R.RT_OPT_FAST_MATH, 0)
    for sun, intensity in (((-0.5, -1.0, -0.3), 1.5), ((1.0, -0.1, 0.2), 1.5), ((0.2, 1.0, 0.1), 1.5), ((1.0, -0.1, 0.2), 0.0),
                           ((0.0, 0.0, 0.0), 1.5), ((-0.5, -1.0, -0.3), 1.5), ((0.0, -1.0, 0.0), 3.0)):
        opc = scenes.make_pc(default_scene_arrays, W, H, sunDirection=sun, sunIntensity=intensity)
        acc, out = gpu_frames(R, rdr, opc, [0, 1], W, H)
        o = oracle_frames(default_scene_arrays, opc, [0, 1], W, H)
        for got, want in ((acc, o["accum"]), (out, o["out"])):
            nan = np.isnan(want)
            assert np.array_equal(np.isnan(got), nan), sun
            assert np.array_equal(bits(got)[~nan], bits(want)[~nan]), sun
            assert not nan.any() or sun == (0.0, 0.0, 0.0)


@pytest.mark.parametrize("W,H", [(37, 23), (8, 8), (1, 1), (65, 9), (200, 3)])
def test_ragged_sizes_bit_exact(R, rdr, default_scene_arrays, W, H):
    rdr.set_option(R.RT_OPT_FAST_MATH, 0)
    opc = scenes.make_pc(default_scene_arrays, W, H)
    acc, out = gpu_frames(R, rdr, opc, [5], W, H)
    o = oracle_frames(default_scene_arrays, opc, [5], W, H)
    assert np.array_equal(bits(acc), bits(o["accum"])) and np.array_equal(bits(out), bits(o["out"]))


def test_camera_and_defocus_bit_exact(R, rdr, default_scene_arrays):
    W, H = 120, 80
    rdr.set_option(R.RT_OPT_FAST_MATH, 0)
    opc = scenes.make_pc(default_scene_arrays, W, H, cam_pos=(1.5, 3.0, 4.0), look=(-0.3, -0.25, -1.0), fov=55.0, focus=6.0, defocus=3.0)
    acc, out = gpu_frames(R, rdr, opc, [0, 1], W, H)
    o = oracle_frames(default_scene_arrays, opc, [0, 1], W, H)
    assert np.array_equal(bits(acc), bits(o["accum"])) and np.array_equal(bits(out), bits(o["out"]))


def test_fast_flavour_within_tolerance(R, rdr, default_scene_arrays):
    W, H = 320, 180
    opc = scenes.make_pc(default_scene_arrays, W, H)
    rdr.set_option(R.RT_OPT_FAST_MATH, 1)
    try:
        acc, out = gpu_frames(R, rdr, opc, [0], W, H)
    finally:
        rdr.set_option(R.RT_OPT_FAST_MATH, 0)
    o = oracle_frames(default_scene_arrays, opc, [0], W, H)
    assert within(acc[..., :3], o["accum"][..., :3]) >= 0.995


def test_converged_psnr(R, rdr, default_scene_arrays):
    """64 accumulated frames at 256x144: strict is bit-exact, fast has PSNR >= 45 dB on the gamma image."""
    W, H = 256, 144
    frames = list(range(64))
    opc = scenes.make_pc(default_scene_arrays, W, H)
    o = oracle_frames(default_scene_arrays, opc, frames, W, H)
    rdr.set_option(R.RT_OPT_FAST_MATH, 0)
    acc, out = gpu_frames(R, rdr, opc, frames, W, H)
    assert np.array_equal(bits(acc), bits(o["accum"])) and np.array_equal(bits(out), bits(o["out"]))
    rdr.set_option(R.RT_OPT_FAST_MATH, 1)
    try:
        acc, out = gpu_frames(R, rdr, opc, frames, W, H)
    finally:
        rdr.set_option(R.RT_OPT_FAST_MATH, 0)
    a = np.clip(out[..., :3], 0, 1).astype(np.float64)
    b = np.clip(o["out"][..., :3], 0, 1).astype(np.float64)
    mse = np.mean((a - b) ** 2)
    psnr = 99.0 if mse == 0 else 10 * np.log10(1.0 / mse)
    assert psnr >= 45.0, psnr


@pytest.mark.parametrize("pipeline", [0, 1, 2, 4], ids=["lockstep", "scheduled", "wavefront", "team"])
def test_mixed_material_scene(R, pipeline):
    """Glass (exp/pow), mirror, fuzzy metal, emitter, open sky (acos/exp/pow), sub-unit and
    non-uniform scales, a model without normals and one with a 287-triangle leaf: bit-exact."""
    materials, objects = scenes.mixed_recipe()
    arrays = scenes.build_oracle(materials, objects).bindings()
    r = R.Renderer(0)
    try:
        r.set_option(R.RT_OPT_PIPELINE, pipeline)
        r.upload_scene(arrays)
        W, H = 256, 144
        opc = scenes.make_pc(arrays, W, H, cam_pos=(0.0, 2.5, 8.0), look=(0.0, -0.15, -1.0), fov=60.0, maxBounces=5)
        acc, out = gpu_frames(R, r, opc, [0, 1], W, H)
        o = oracle_frames(arrays, opc, [0, 1], W, H, counters=True)
        # exp / pow / acos come from include/rt_math.h on both sides, so glass and sky are exact too
        # (NaNs included: Reflectance takes pow of a negative base on glass lit by the un-normalised sun)
        assert np.array_equal(bits(acc), bits(o["accum"]))
        assert np.array_equal(bits(out), bits(o["out"]))
        opc["frame"] = 1
        st = r.frame_stats(to_pc(R, opc))
        assert (st.rays, st.bvhNodeTests, st.triangleTests) == (int(o["rays"].sum()), int(o["node"].sum()), int(o["tri"].sum()))
        # depth view has no transcendental in it: bit-exact, including the objects scaled below 1
        opc2 = scenes.make_pc(arrays, W, H, cam_pos=(0.0, 2.5, 8.0), look=(0.0, -0.15, -1.0), fov=60.0, renderMode=2)
        r.render_frame(to_pc(R, opc2))
        want = O.render(arrays, O.pc_bytes(opc2), W, H)["out"]
        assert np.array_equal(bits(r.read_image(R.RT_IMAGE_DRAW)), bits(want))
    finally:
        r.close()


@pytest.mark.parametrize("fused", [8, 3], ids=["launch8", "launch3"])
def test_two_samples_per_lane_bit_exact(R, default_scene_arrays, fused):
    """Pipeline 3 traces two samples of a pixel per lane (a lane walks its second ray while other lanes
    are still on their first). Against the oracle: default scene with 3 samples per pixel over a strided,
    ragged frame set (pairs straddle frame boundaries; the last pair is half empty), accumulation on and
    off, and the mixed-material scene (glass, sky, scaled objects, a 287-triangle leaf, deeper bounces)."""
    r = R.Renderer(0)
    try:
        r.set_option(R.RT_OPT_PIPELINE, 3)
        r.set_option(R.RT_OPT_FRAMES_PER_LAUNCH, fused)
        W, H = 136, 88
        r.upload_scene(default_scene_arrays); r.resize(W, H)
        for accumulate in (1, 0):
            opc = scenes.make_pc(default_scene_arrays, W, H, raysPerPixel=3, accumulateFrames=accumulate)
            frames = [1 + 2 * k for k in range(5)]
            want = oracle_frames(default_scene_arrays, opc, frames, W, H)
            r.clear_accumulation()
            r.render_frames(to_pc(R, opc), 1, 2, 5)
            assert np.array_equal(bits(r.read_image(R.RT_IMAGE_ACCUM)), bits(want["accum"])), accumulate
            assert np.array_equal(bits(r.read_image(R.RT_IMAGE_DRAW)), bits(want["out"])), accumulate
        materials, objects = scenes.mixed_recipe()
        arrays = scenes.build_oracle(materials, objects).bindings()
        r.upload_scene(arrays)
        W, H = 192, 108
        r.resize(W, H)
        opc = scenes.make_pc(arrays, W, H, cam_pos=(0.0, 2.5, 8.0), look=(0.0, -0.15, -1.0), fov=60.0, maxBounces=5, raysPerPixel=2)
        want = oracle_frames(arrays, opc, [0, 1, 2], W, H)
        r.clear_accumulation()
        r.render_frames(to_pc(R, opc), 0, 1, 3)
        assert np.array_equal(bits(r.read_image(R.RT_IMAGE_ACCUM)), bits(want["accum"]))
        assert np.array_equal(bits(r.read_image(R.RT_IMAGE_DRAW)), bits(want["out"]))
        # the fast flavour stays within its tolerance of the strict one
        r.set_option(R.RT_OPT_FAST_MATH, 1)
        r.clear_accumulation()
        r.render_frames(to_pc(R, opc), 0, 1, 3)
        # (a contracted multiply-add can flip a refraction or a scatter decision: a few pixels take another path)
        assert within(r.read_image(R.RT_IMAGE_ACCUM)[..., :3], want["accum"][..., :3], 1e-3, 1e-3) > 0.98
    finally:
        r.close()


def _zeno_obj(path, n):
    """n triangles, each half as far from the origin and half as large as the one before: the midpoint builder
    splits one triangle off per level, so the tree is n - 2 levels deep."""
    with open(path, "w") as f:
        for k in range(n):
            x = 2.0 ** (1 - k); s = x / 4
            f.write("v %.9g 0 0\nv %.9g %.9g 0\nv %.9g 0 %.9g\n" % (x - s / 3, x - s / 3, s, x + 2 * s / 3, s))
        for k in range(n):
            f.write("f %d %d %d\n" % (3 * k + 1, 3 * k + 2, 3 * k + 3))


def test_deepest_tree_the_stack_allows(R, tmp_path):
    """A model whose tree is 31 levels deep - all the shader's `int stack[32]` (raytrace.comp:523) can take -
    on every pipeline, bit-exact against the oracle (pipeline 4, whose walkers' stacks no longer fit a
    workgroup's LDS at this depth, hands the launch to the lock-step kernels); the near-first flavour within
    its bars; one level more is refused at upload."""
    import shutil
    for name in ("flat_quad.obj",):
        shutil.copy(os.path.join(O.MODELS_DIR, name), tmp_path / name)
    _zeno_obj(str(tmp_path / "zeno33.obj"), 33)
    _zeno_obj(str(tmp_path / "zeno34.obj"), 34)
    materials = [scenes.mat((0.5, 0.5, 0.5)), scenes.mat((0.8, 0.3, 0.2)), scenes.mat((0.9, 0.9, 0.9), smoothness=1.0)]
    objects = [("flat_quad.obj", (0, -0.5, 0), (0, 0, 0), (12, 12, 12), 0),
               ("zeno33.obj", (-1.5, 0.0, -1.0), (0, 20, 0), (1.5, 1.5, 1.5), 1),
               ("zeno33.obj", (0.5, 0.2, 0.5), (30, -70, 10), (1.0, 1.0, 1.0), 2)]
    arrays = scenes.build_oracle(materials, objects, models_dir=str(tmp_path)).bindings()
    W, H = 192, 112
    opc = scenes.make_pc(arrays, W, H, cam_pos=(0.0, 1.5, 4.0), look=(0.0, -0.3, -1.0), fov=60.0)
    want = oracle_frames(arrays, opc, [0, 1, 2], W, H)
    assert np.any(want["accum"][..., :3] > 0)
    r = R.Renderer(0)
    try:
        r.upload_scene(arrays)
        assert r.get_option(R.RT_INFO_STACK_DEPTH) == 31
        for pipeline in (0, 1, 2, 3, 4):
            r.set_option(R.RT_OPT_PIPELINE, pipeline)
            acc, out = gpu_frames(R, r, opc, [0, 1, 2], W, H)
            assert np.array_equal(bits(acc), bits(want["accum"])) and np.array_equal(bits(out), bits(want["out"])), pipeline
        for pipeline in (3, 0, 4):
            r.set_option(R.RT_OPT_PIPELINE, pipeline)
            r.set_option(R.RT_OPT_TRAVERSAL, 1)
            acc, out = gpu_frames(R, r, opc, [0, 1, 2], W, H)
            assert within(acc[..., :3], want["accum"][..., :3]) >= 0.995, pipeline
            r.set_option(R.RT_OPT_TRAVERSAL, 0)
        deep = scenes.build_oracle(materials, [("zeno34.obj", (0, 0, 0), (0, 0, 0), (1, 1, 1), 1)], models_dir=str(tmp_path)).bindings()
        with pytest.raises(R.RtError) as e:
            r.upload_scene(deep)
        assert e.value.code == -6 and "31 levels" in str(e.value)
    finally:
        r.close()


def test_reference_order_on_folded_records(R, default_scene_arrays):
    """Pipeline 3 walks the reference's trees folded into four-child records whose slots stand in the reference's
    visiting order (RT_OPT_WIDE_RECORDS 1, the default) or the two-child records themselves (0): the same image, the
    oracle's, bit for bit - on the default scene and on the scene with every material class, shrunk and stretched
    objects, a model without normals and one with a 287-triangle leaf; multi-frame launches and single-frame dispatches."""
    materials, objects = scenes.mixed_recipe()
    mixed = scenes.build_oracle(materials, objects).bindings()
    for arrays, cam in ((default_scene_arrays, {}), (mixed, dict(cam_pos=(0.0, 2.5, 8.0), look=(0.0, -0.15, -1.0), fov=60.0, maxBounces=5))):
        W, H = 208, 120
        opc = scenes.make_pc(arrays, W, H, **cam)
        want = oracle_frames(arrays, opc, [0, 1, 2, 3, 4], W, H)
        r = R.Renderer(0)
        try:
            r.upload_scene(arrays)
            r.set_option(R.RT_OPT_PIPELINE, 3)
            for wide in (1, 0):
                r.set_option(R.RT_OPT_WIDE_RECORDS, wide)
                acc, out = gpu_frames(R, r, opc, [0, 1, 2, 3, 4], W, H)
                assert np.array_equal(bits(acc), bits(want["accum"])) and np.array_equal(bits(out), bits(want["out"])), wide
                r.clear_accumulation()
                for f in range(5):
                    opc["frame"] = f
                    r.render_frame(to_pc(R, opc))
                assert np.array_equal(bits(r.read_image(R.RT_IMAGE_ACCUM)), bits(want["accum"])), wide
        finally:
            r.close()


def test_empty_scene_is_all_sky(R):
    arrays = [np.zeros(0, d) for d in O.BINDING_DTYPES]
    r = R.Renderer(0)
    try:
        r.upload_scene(arrays)
        W, H = 64, 40
        opc = scenes.make_pc(arrays, W, H)
        acc, out = gpu_frames(R, r, opc, [0], W, H)
        o = oracle_frames(arrays, opc, [0], W, H)
        assert within(acc[..., :3], o["accum"][..., :3], 1e-5, 1e-5) == 1.0
        assert acc[..., :3].min() > 0.4     # day sky everywhere
    finally:
        r.close()


def test_tile_partition_is_bit_identical(R, rdr, default_scene_arrays):
    """Tile split across N ranks: every rank renders only its tiles; the union equals the single-GPU image."""
    W, H = 200, 120
    rdr.set_option(R.RT_OPT_FAST_MATH, 0)
    opc = scenes.make_pc(default_scene_arrays, W, H)
    _tile_partition(R, rdr, default_scene_arrays, opc, W, H, [0, 1])


@pytest.mark.parametrize("traversal", [0, 1], ids=["reference-order", "near-first"])
def test_tile_partition_launches_in_pieces(R, default_scene_arrays, traversal):
    """The same with launches of seven frames rendered in pieces of two frames (the launch shape of a tile split's
    ranks: RT_OPT_LAUNCH_PIECES 1): every rank's tiles equal the single-GPU image, which equals the oracle's in
    reference order, and the image of launches in one piece in both orders."""
    W, H = 200, 120
    r = R.Renderer(0)
    try:
        r.upload_scene(default_scene_arrays)
        r.set_option(R.RT_OPT_PIPELINE, 3)
        r.set_option(R.RT_OPT_TRAVERSAL, traversal)
        opc = scenes.make_pc(default_scene_arrays, W, H)
        frames = list(range(7))
        r.set_option(R.RT_OPT_LAUNCH_PIECES, 0)
        one_acc, one_out = gpu_frames(R, r, opc, frames, W, H)
        r.set_option(R.RT_OPT_LAUNCH_PIECES, 1)
        full_acc, full_out = _tile_partition(R, r, default_scene_arrays, opc, W, H, frames)
        assert np.array_equal(bits(full_acc), bits(one_acc)) and np.array_equal(bits(full_out), bits(one_out))
        if traversal == 0:
            want = oracle_frames(default_scene_arrays, opc, frames, W, H)
            assert np.array_equal(bits(full_acc), bits(want["accum"])) and np.array_equal(bits(full_out), bits(want["out"]))
    finally:
        r.close()


def _tile_partition(R, rdr, default_scene_arrays, opc, W, H, frames):
    full_acc, full_out = gpu_frames(R, rdr, opc, frames, W, H)
    world = 3
    rdr.set_option(R.RT_OPT_TILE_SIZE, 32)
    rdr.set_option(R.RT_OPT_TILE_WORLD, world)
    try:
        sum_acc = np.zeros_like(full_acc)
        owner = np.zeros((H, W), np.int32)
        for rank in range(world):
            rdr.set_option(R.RT_OPT_TILE_RANK, rank)
            rdr.resize(W, H)                       # fresh, zeroed draw + accumulation images
            acc, out = gpu_frames(R, rdr, opc, frames, W, H)
            touched = np.any(bits(acc) != 0, axis=2) | np.any(bits(out) != 0, axis=2)
            owner += touched
            sum_acc += acc
            from raytracer_amd.distributed import tile_owner_map
            mine = tile_owner_map(world, W, H, 32) == rank      # tiles are dealt along a serpentine path
            assert not np.any(touched & ~mine)
            assert np.array_equal(bits(acc[mine]), bits(full_acc[mine]))
            assert np.array_equal(bits(out[mine]), bits(full_out[mine]))
        assert np.array_equal(bits(sum_acc), bits(full_acc))
    finally:
        rdr.set_option(R.RT_OPT_TILE_RANK, 0)
        rdr.set_option(R.RT_OPT_TILE_WORLD, 1)
        rdr.set_option(R.RT_OPT_TILE_SIZE, 64)
    return full_acc, full_out


def test_sample_split_merge_and_resolve(R, rdr, default_scene_arrays):
    """Sample split: rank g renders frames g, g+N, ...; the summed accumulation images, resolved with
    1/total, match the sequential render up to f32 re-association of the per-pixel sum."""
    W, H = 160, 90
    rdr.set_option(R.RT_OPT_FAST_MATH, 0)
    total, world = 8, 2
    opc = scenes.make_pc(default_scene_arrays, W, H)
    seq_acc, seq_out = gpu_frames(R, rdr, opc, list(range(total)), W, H)
    parts = []
    for rank in range(world):
        rdr.clear_accumulation()
        rdr.render_frames(to_pc(R, opc), rank, world, total // world)
        parts.append(rdr.read_image(R.RT_IMAGE_ACCUM))
    merged = parts[0] + parts[1]
    rdr.write_accum(merged)
    rdr.resolve(total)
    out = rdr.read_image(R.RT_IMAGE_DRAW)
    assert np.array_equal(bits(rdr.read_image(R.RT_IMAGE_ACCUM)), bits(merged))
    np.testing.assert_allclose(merged[..., :3], seq_acc[..., :3], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(out[..., :3], seq_out[..., :3], rtol=1e-5, atol=1e-6)
    want = np.sqrt(np.maximum(merged[..., :3] * np.float32(1.0 / total), 0), dtype=np.float32)
    assert np.array_equal(bits(out[..., :3]), bits(want))


def test_full_size_properties(R, rdr, default_scene_arrays):
    """Benchmark configuration 1 size (2560x1369), too big for the oracle in a test: size-independent properties."""
    W, H = 2560, 1369
    rdr.set_option(R.RT_OPT_FAST_MATH, 0)
    opc = scenes.make_pc(default_scene_arrays, W, H, accumulateFrames=0)
    rdr.resize(W, H)
    singles = []
    for f in range(3):
        rdr.clear_accumulation()
        opc["frame"] = f
        rdr.render_frame(to_pc(R, opc))
        singles.append(rdr.read_image(R.RT_IMAGE_ACCUM))
    # determinism: the same frame twice gives the same bits
    rdr.clear_accumulation()
    opc["frame"] = 2
    rdr.render_frame(to_pc(R, opc))
    assert np.array_equal(bits(rdr.read_image(R.RT_IMAGE_ACCUM)), bits(singles[2]))
    # progressive accumulation is the in-order f32 sum of the single frames
    opc2 = scenes.make_pc(default_scene_arrays, W, H)
    acc, out = gpu_frames(R, rdr, opc2, [0, 1, 2], W, H)
    want = (singles[0] + singles[1]) + singles[2]
    assert np.array_equal(bits(acc), bits(want))
    assert np.array_equal(bits(out[..., :3]), bits(np.sqrt(np.maximum(want[..., :3] * np.float32(1.0 / 3.0), 0), dtype=np.float32)))
    assert np.all(out[..., 3] == 0)
    # a window of the big frame against the oracle (pixels do not depend on their neighbours)
    o = O.render(default_scene_arrays, O.pc_bytes(scenes.make_pc(default_scene_arrays, W, H)), W, H, window=(1200, 600, 1328, 664))
    assert np.array_equal(bits(singles[0][600:664, 1200:1328]), bits(o["accum"][600:664, 1200:1328]))
    # ray budget: 1..8 GetHit calls per sample
    opc2["frame"] = 0
    st = rdr.frame_stats(to_pc(R, opc2))
    assert st.samples == W * H and W * H <= st.rays <= 8 * W * H and st.bvhNodeTests >= 8 * st.rays


def test_write_accum_resume(R, rdr, default_scene_arrays):
    W, H = 64, 48
    rdr.set_option(R.RT_OPT_FAST_MATH, 0)
    opc = scenes.make_pc(default_scene_arrays, W, H)
    acc3, out3 = gpu_frames(R, rdr, opc, [0, 1, 2], W, H)
    acc2, _ = gpu_frames(R, rdr, opc, [0, 1], W, H)
    rdr.clear_accumulation()
    rdr.write_accum(acc2)
    a, o = gpu_frames(R, rdr, opc, [2], W, H, clear=False)
    assert np.array_equal(bits(a), bits(acc3)) and np.array_equal(bits(o), bits(out3))


def test_bound_images(R, rdr, default_scene_arrays):
    """Rendering into caller-owned device memory (torch tensors) on the caller's stream."""
    import torch
    W, H = 96, 64
    rdr.set_option(R.RT_OPT_FAST_MATH, 0)
    opc = scenes.make_pc(default_scene_arrays, W, H)
    want_acc, want_out = gpu_frames(R, rdr, opc, [0, 1], W, H)
    draw = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0")
    accum = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()
    rdr.bind_images(draw.data_ptr(), accum.data_ptr(), W, H)
    rdr.set_stream(torch.cuda.current_stream().cuda_stream)
    try:
        for f in (0, 1):
            opc["frame"] = f
            rdr.render_frame(to_pc(R, opc))
        torch.cuda.synchronize()
        assert np.array_equal(bits(accum.cpu().numpy()), bits(want_acc))
        assert np.array_equal(bits(draw.cpu().numpy()), bits(want_out))
    finally:
        rdr.set_stream(None)
        rdr.bind_images(None, None, 0, 0)


def test_distributed_driver_single_rank(R, default_scene_arrays):
    """raytracer_amd.distributed on one GPU (world size 1): the GpuTarget plumbing (torch-owned
    images, torch's stream, tile options) and both drivers against the oracle."""
    import torch
    from raytracer_amd import distributed as D
    W, H = 112, 64
    r = R.Renderer(0)
    try:
        r.upload_scene(default_scene_arrays)
        opc = scenes.make_pc(default_scene_arrays, W, H)
        tgt = D.GpuTarget(r, to_pc(R, opc), W, H, "cuda:0")
        assert D.render_sample_split(tgt, 3) == 3
        torch.cuda.synchronize()
        o = oracle_frames(default_scene_arrays, opc, [0, 1, 2], W, H)
        assert np.array_equal(bits(tgt.accum.cpu().numpy()), bits(o["accum"]))
        assert np.array_equal(bits(tgt.draw.cpu().numpy()), bits(o["out"]))
        D.render_tile_split(tgt, 2, tile_size=32)
        torch.cuda.synchronize()
        o = oracle_frames(default_scene_arrays, opc, [0, 1], W, H)
        assert np.array_equal(bits(tgt.accum.cpu().numpy()), bits(o["accum"]))
        assert np.array_equal(bits(tgt.draw.cpu().numpy()), bits(o["out"]))
        tgt.close()
    finally:
        r.close()


def test_offline_render_and_png(R, rdr, default_scene_arrays, tmp_path):
    """ProduceRender / WriteDrawImageToFile: frames pc.frame..N-1 accumulate, the frame counter
    advances, and the PNG holds clamp(x)*255+0.5 of the draw image with alpha 255."""
    from test_host import decode_png
    W, H = 96, 54
    opc = scenes.make_pc(default_scene_arrays, W, H)
    ref_acc, ref_out = gpu_frames(R, rdr, opc, [0, 1, 2], W, H)
    rdr.clear_accumulation()
    pc = to_pc(R, opc); pc.frame = 0
    path = str(tmp_path / "render.png")
    seconds = rdr.produce_render(pc, 3, path)
    assert pc.frame == 3 and seconds > 0.0
    assert np.array_equal(bits(rdr.read_image(R.RT_IMAGE_DRAW)), bits(ref_out))
    png = decode_png(path)
    assert np.array_equal(png, R.rgba32f_to_rgba8(ref_out)) and (png[..., 3] == 255).all()
    # already at the last frame: nothing is dispatched, the image is written again
    assert rdr.produce_render(pc, 3, path) >= 0.0 and pc.frame == 3
    assert np.array_equal(decode_png(path), png)
    rdr.save_png(str(tmp_path / "again.png"))
    assert np.array_equal(decode_png(str(tmp_path / "again.png")), png)


def test_wavefront_knobs_do_not_change_results(R, default_scene_arrays):
    """The number of suspend/walk/resume rounds, the refill threshold and the grid size of the
    wavefront pipeline only move work between kernels: every setting gives the lock-step image,
    also with several samples per pixel, a partial parent count and the counting build."""
    W, H = 200, 120
    opc = scenes.make_pc(default_scene_arrays, W, H, raysPerPixel=3, maxBounces=5)
    r = R.Renderer(0)
    try:
        r.upload_scene(default_scene_arrays)
        r.set_option(R.RT_OPT_PIPELINE, 0)
        ref_acc, ref_out = gpu_frames(R, r, opc, [0, 1], W, H)
        ref_stats = r.frame_stats(to_pc(R, opc))
        r.set_option(R.RT_OPT_PIPELINE, 2)
        for rounds, refill, waves, walk, parts in [(0, 16, 12, 12, 1), (1, 64, 12, 24, 2), (2, 1, 3, 1, 3), (5, 16, 12, 12, 3),
                                                   (12, 32, 16, 32, 3), (30, 48, 1, 5, 2)]:
            r.set_option(R.RT_OPT_WF_ROUNDS, rounds); r.set_option(R.RT_OPT_WF_REFILL, refill)
            r.set_option(R.RT_OPT_WAVES_PER_CU, waves); r.set_option(R.RT_OPT_WF_WALK_WAVES, walk)
            r.set_option(R.RT_OPT_WF_PARTS, parts)
            acc, out = gpu_frames(R, r, opc, [0, 1], W, H)
            assert np.array_equal(bits(acc), bits(ref_acc)), (rounds, refill, waves, walk, parts)
            assert np.array_equal(bits(out), bits(ref_out)), (rounds, refill, waves, walk, parts)
            st = r.frame_stats(to_pc(R, opc))
            assert (st.rays, st.bvhNodeTests, st.triangleTests, st.samples) == \
                   (ref_stats.rays, ref_stats.bvhNodeTests, ref_stats.triangleTests, ref_stats.samples)
        # more partitions than the process has hardware queues to give them are refused
        with pytest.raises(R.RtError):
            r.set_option(R.RT_OPT_WF_PARTS, 4)
        # fewer parents than uploaded: shadow rays keep the scene order, walks start at parent 0
        opc["parentBVHCount"] = 5; opc["raysPerPixel"] = 1
        r.set_option(R.RT_OPT_WF_ROUNDS, 5); r.set_option(R.RT_OPT_WF_REFILL, 16); r.set_option(R.RT_OPT_WAVES_PER_CU, 12)
        r.set_option(R.RT_OPT_WF_WALK_WAVES, 12); r.set_option(R.RT_OPT_WF_PARTS, 3)
        acc, out = gpu_frames(R, r, opc, [0], W, H)
        r.set_option(R.RT_OPT_PIPELINE, 0)
        acc0, out0 = gpu_frames(R, r, opc, [0], W, H)
        assert np.array_equal(bits(acc), bits(acc0)) and np.array_equal(bits(out), bits(out0))
    finally:
        r.close()


def test_frames_per_launch_does_not_change_results(R, default_scene_arrays):
    """rt_render_frames may run several consecutive frames of the call in one launch
    (RT_OPT_FRAMES_PER_LAUNCH): accumulation and draw image stay what single-frame dispatches - and the
    oracle - give, for a strided frame set with a ragged last group, both flavours, accumulation on and off."""
    W, H = 136, 88
    r = R.Renderer(0)
    try:
        r.upload_scene(default_scene_arrays); r.resize(W, H)
        for accumulate in (1, 0):
            opc = scenes.make_pc(default_scene_arrays, W, H, raysPerPixel=2, accumulateFrames=accumulate)
            frames = [2 + 3 * k for k in range(7)]
            want = oracle_frames(default_scene_arrays, opc, frames, W, H)
            for fast in (0, 1):
                r.set_option(R.RT_OPT_FAST_MATH, fast)
                for pipeline in (0, 3):   # 3: two samples per lane, pairs formed inside a launch
                    r.set_option(R.RT_OPT_PIPELINE, pipeline)
                    r.set_option(R.RT_OPT_FRAMES_PER_LAUNCH, 1)
                    r.clear_accumulation()
                    r.render_frames(to_pc(R, opc), 2, 3, 7)
                    one = (r.read_image(R.RT_IMAGE_ACCUM), r.read_image(R.RT_IMAGE_DRAW))
                    if fast == 0:   # (the fast flavour is only the same from launch shape to launch shape of one kernel)
                        assert np.array_equal(bits(one[0]), bits(want["accum"])) and np.array_equal(bits(one[1]), bits(want["out"]))
                    # pipeline 3 renders a launch of more than two frames in pieces of two frames, summed in frame order
                    # afterwards (RT_OPT_LAUNCH_PIECES 1, the default), or in one piece (0)
                    for group, pieces in ((2, 1), (3, 1), (7, 1), (8, 1), (64, 1), (3, 0), (7, 0), (64, 0), (5, 2)):
                        r.set_option(R.RT_OPT_FRAMES_PER_LAUNCH, group)
                        r.set_option(R.RT_OPT_LAUNCH_PIECES, pieces)
                        r.clear_accumulation()
                        r.render_frames(to_pc(R, opc), 2, 3, 7)
                        got = (r.read_image(R.RT_IMAGE_ACCUM), r.read_image(R.RT_IMAGE_DRAW))
                        assert np.array_equal(bits(got[0]), bits(one[0])), (accumulate, fast, pipeline, group, pieces)
                        assert np.array_equal(bits(got[1]), bits(one[1])), (accumulate, fast, pipeline, group, pieces)
                    r.set_option(R.RT_OPT_LAUNCH_PIECES, 1)
                r.set_option(R.RT_OPT_PIPELINE, 0)
        with pytest.raises(R.RtError):
            r.set_option(R.RT_OPT_FRAMES_PER_LAUNCH, 0)
    finally:
        r.close()


def test_device_math_equals_host_math(R):
    """exp2 / exp / log2 / acos of include/rt_math.h: the strict kernels give the bits the host gives,
    over random inputs of every range the renderer can feed in and the special values."""
    rng = np.random.default_rng(11)
    r = R.Renderer(0)
    try:
        special = np.array([0.0, -0.0, 1.0, -1.0, 0.5, 2.0, np.inf, -np.inf, np.nan, 1e-40, 3e38, 88.8, -104.0, 128.0, -150.0], np.float32)
        cases = {
            "exp2": np.concatenate([rng.uniform(-160, 140, 300000), rng.normal(0, 1, 100000), special]),
            "exp": np.concatenate([rng.uniform(-110, 95, 300000), rng.normal(0, 1, 100000), special]),
            "log2": np.concatenate([np.exp(rng.uniform(-100, 88, 300000)), rng.uniform(0, 2, 100000), special]),
            "acos": np.concatenate([rng.uniform(-1.01, 1.01, 300000), 1.0 - np.exp(rng.uniform(-20, 0, 100000)), special]),
        }
        for name, x in cases.items():
            x = x.astype(np.float32)
            assert np.array_equal(bits(r.math_apply(name, x)), bits(O.math_apply(name, x))), name
        # the generator's k / 10000: all 10 000 dividends, against the division itself
        k = np.arange(10000, dtype=np.float32)
        assert np.array_equal(bits(r.math_apply("div10000", k)), bits(k / np.float32(10000.0)))
    finally:
        r.close()


def test_device_math_accuracy_against_double(R):
    """The accuracy bars of tests/test_rt_math.py applied to what the DEVICE computes (exp2, exp within 2 ulp,
    log2 and acos within 4 ulp of double precision), so that a GPU-only run bounds the functions that stand in
    for the GLSL built-ins and not only their agreement with the host's copy of the same header."""
    def ulps(y, ref):
        return np.abs(y.astype(np.float64) - ref) / np.spacing(np.abs(ref.astype(np.float32))).astype(np.float64)
    rng = np.random.default_rng(7)
    r = R.Renderer(0)
    try:
        x = rng.uniform(-125, 127, 400000).astype(np.float32)
        assert ulps(r.math_apply("exp2", x), np.exp2(x.astype(np.float64))).max() < 2.0
        x = rng.uniform(-86, 88, 400000).astype(np.float32)
        assert ulps(r.math_apply("exp", x), np.exp(x.astype(np.float64))).max() < 2.0
        x = np.exp(rng.uniform(-80, 80, 400000)).astype(np.float32)
        ref = np.log2(x.astype(np.float64))
        far = np.abs(ref) > 0.01
        assert ulps(r.math_apply("log2", x)[far], ref[far]).max() < 4.0
        x = rng.uniform(0.5, 2.0, 400000).astype(np.float32)
        assert np.abs(r.math_apply("log2", x) - np.log2(x.astype(np.float64))).max() < 2e-7
        x = rng.uniform(-1, 1, 400000).astype(np.float32)
        assert ulps(r.math_apply("acos", x), np.arccos(x.astype(np.float64))).max() < 4.0
    finally:
        r.close()


def test_error_paths(R, default_scene_arrays):
    r = R.Renderer(0)
    try:
        pc = R.default_push_constants()
        with pytest.raises(R.RtError) as e:
            r.render_frame(pc)
        assert e.value.code == -4                      # no scene
        r.upload_scene(default_scene_arrays)
        with pytest.raises(R.RtError) as e:
            r.render_frame(pc)
        assert e.value.code == -5                      # no images
        r.resize(16, 16)
        pc.parentBVHCount = 9
        with pytest.raises(R.RtError) as e:
            r.render_frame(pc)
        assert "parentBVHCount" in str(e.value)
        buf = np.zeros(10, np.float32)
        assert r.lib.rt_read_image(r.h, 0, buf.ctypes.data, buf.nbytes) == -1
        with pytest.raises(R.RtError):
            r.resize(0, 5)
        with pytest.raises(R.RtError):
            r.set_option(R.RT_OPT_PIPELINE, 7)
        with pytest.raises(R.RtError):
            r.set_option(R.RT_OPT_TEAM_WALKERS, 0)
        # invalid scenes are rejected at upload instead of faulting the GPU
        bad = [a.copy() for a in default_scene_arrays]
        bad[14][5] = 10 ** 6                           # child index out of range
        with pytest.raises(R.RtError) as e:
            r.upload_scene(bad)
        assert e.value.code == -6
        bad = [a.copy() for a in default_scene_arrays]
        l0 = int(bad[14][98])                          # make the dragon's first inner node its own grandchild
        bad[14][l0] = 98
        with pytest.raises(R.RtError) as e:
            r.upload_scene(bad)
        assert e.value.code == -6 and ("cycle" in str(e.value) or "deeper" in str(e.value))
        bad = [a.copy() for a in default_scene_arrays]
        bad[16][200] = 10 ** 7                         # leaf triangle range outside the arrays
        with pytest.raises(R.RtError):
            r.upload_scene(bad)
        bad = [a.copy() for a in default_scene_arrays]
        bad[1].view(np.int32).reshape(8, 20)[3, 2] = -1   # GetMaterialIndex's "not found"
        with pytest.raises(R.RtError) as e:
            r.upload_scene(bad)
        assert "material" in str(e.value)
        bad = [a.copy() for a in default_scene_arrays]
        bad[4] = bad[4][:-4]
        with pytest.raises(R.RtError):
            r.upload_scene(bad)
        # the context is still usable afterwards
        r.upload_scene(default_scene_arrays)
        pc = to_pc(R, scenes.make_pc(default_scene_arrays, 16, 16))
        r.render_frame(pc)
        r.sync()
    finally:
        r.close()
