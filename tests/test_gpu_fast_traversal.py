"""Near-first traversal (RT_OPT_TRAVERSAL = 1: trees rebuilt with a binned surface-area heuristic, nearer
child first, tight world boxes) against the CPU oracle, which walks the reference's trees in the
reference's order.

GetHit's answer is the triangle with the smallest world-space distance whatever the visiting order, for
every object that is not scaled below 1 (objects that are keep the reference's tree and order), so the
only pixels that may differ are those where two triangles tie exactly or where a ray grazes the padded
boxes. Bars, as stated for the renderer as a whole:
  * 1 sample per pixel, linear accumulation value: >= 99.5 % of pixels within |d| <= 1e-4 + 1e-4*|ref| per
    channel (measured: > 99.99 % are bit-identical); the outliers are counted and bounded;
  * converged image (64 accumulated frames), gamma image: PSNR >= 45 dB against the oracle.
Arithmetic stays strict binary32 (no FMA contraction): the traversal order is the only difference."""
import os

import numpy as np
import pytest

import oracle_lib as O
import scenes

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def to_pc(R, opc):
    return R.PushConstants.from_buffer_copy(O.pc_bytes(opc).tobytes())


def shares(a, ref, tol_abs=1e-4, tol_rel=1e-4):
    """(share of pixels bit-identical, share within tolerance) over the rgb channels."""
    same = np.all(bits(a[..., :3]) == bits(ref[..., :3]), axis=-1)
    with np.errstate(invalid="ignore"):
        ok = np.abs(a[..., :3] - ref[..., :3]) <= tol_abs + tol_rel * np.abs(ref[..., :3])
    ok |= np.isnan(a[..., :3]) & np.isnan(ref[..., :3])
    return float(same.mean()), float((np.all(ok, axis=-1) | same).mean())


def psnr(out, ref):
    a = np.clip(np.nan_to_num(out[..., :3]), 0, 1).astype(np.float64)
    b = np.clip(np.nan_to_num(ref[..., :3]), 0, 1).astype(np.float64)
    mse = np.mean((a - b) ** 2)
    return 99.0 if mse == 0 else 10 * np.log10(1.0 / mse)


@pytest.fixture(scope="module")
def R():
    import raytracer_amd
    return raytracer_amd


def render(R, r, opc, frames, W, H, traversal, pipeline):
    r.set_option(R.RT_OPT_PIPELINE, pipeline)
    r.set_option(R.RT_OPT_TRAVERSAL, traversal)
    if (r.width, r.height) != (W, H):
        r.resize(W, H)
    r.clear_accumulation()
    opc["frame"] = frames[0]
    if len(frames) == 1:
        r.render_frame(to_pc(R, opc))
    else:
        r.render_frames(to_pc(R, opc), frames[0], frames[1] - frames[0], len(frames))
    return r.read_image(R.RT_IMAGE_ACCUM), r.read_image(R.RT_IMAGE_DRAW)


def oracle(arrays, opc, frames, W, H, window=None):
    acc = np.zeros((H, W, 4), np.float32)
    res = None
    for f in frames:
        opc["frame"] = f
        res = O.render(arrays, O.pc_bytes(opc), W, H, accum=acc, window=window)
        acc = res["accum"]
    return res["accum"], res["out"]


@pytest.mark.parametrize("pipeline", [3, 0, 4], ids=["two-samples", "lockstep", "team"])
def test_default_scene_1spp_and_converged(R, default_scene_arrays, pipeline):
    r = R.Renderer(0)
    try:
        r.upload_scene(default_scene_arrays)
        assert r.get_option(R.RT_INFO_FAST_OBJECTS) == 2          # icosphere and dragon walk rebuilt trees
        assert r.get_option(R.RT_INFO_FAST_STACK_DEPTH) <= 31
        W, H = 640, 360
        opc = scenes.make_pc(default_scene_arrays, W, H)
        want_acc, want_out = oracle(default_scene_arrays, opc, [0], W, H)
        acc, out = render(R, r, opc, [0], W, H, 1, pipeline)
        same, ok = shares(acc, want_acc)
        assert ok >= 0.995 and same >= 0.999, (same, ok)
        assert np.all(acc[..., 3] == 0)
        # two frames through rt_render_frames (pipeline 3: the two-samples-per-lane kernel)
        want_acc, want_out = oracle(default_scene_arrays, opc, [0, 1], W, H)
        acc, out = render(R, r, opc, [0, 1], W, H, 1, pipeline)
        same, ok = shares(acc, want_acc)
        assert ok >= 0.995 and same >= 0.999, (same, ok)
        # converged: 64 frames
        W, H = 256, 144
        opc = scenes.make_pc(default_scene_arrays, W, H)
        frames = list(range(64))
        want_acc, want_out = oracle(default_scene_arrays, opc, frames, W, H)
        acc, out = render(R, r, opc, frames, W, H, 1, pipeline)
        assert psnr(out, want_out) >= 45.0
        same, ok = shares(acc, want_acc)
        assert ok >= 0.995, (same, ok)
    finally:
        r.close()


def test_settings_and_partial_parent_count(R, default_scene_arrays):
    """Samples per pixel, bounce count, defocus, sun settings: same bars. A dispatch with fewer parents
    than were uploaded falls back to the reference order (bit-exact)."""
    r = R.Renderer(0)
    try:
        r.upload_scene(default_scene_arrays)
        W, H = 200, 120
        for fields in (dict(raysPerPixel=3), dict(maxBounces=7), dict(accumulateFrames=0, raysPerPixel=2),
                       dict(sunDirection=(-0.3, -0.9, 0.2), sunIntensity=2.0)):
            opc = scenes.make_pc(default_scene_arrays, W, H, **fields)
            want_acc, want_out = oracle(default_scene_arrays, opc, [0, 1, 2], W, H)
            acc, out = render(R, r, opc, [0, 1, 2], W, H, 1, 3)
            same, ok = shares(acc, want_acc)
            assert ok >= 0.995 and same >= 0.995, (fields, same, ok)
            same, ok = shares(out, want_out)
            assert ok >= 0.995, (fields, same, ok)
        opc = scenes.make_pc(default_scene_arrays, W, H, cam_pos=(1.5, 3.0, 4.0), look=(-0.3, -0.25, -1.0), fov=55.0, focus=6.0, defocus=3.0)
        want_acc, _ = oracle(default_scene_arrays, opc, [0, 1], W, H)
        acc, _ = render(R, r, opc, [0, 1], W, H, 1, 3)
        assert shares(acc, want_acc)[1] >= 0.995
        opc = scenes.make_pc(default_scene_arrays, W, H)
        opc["parentBVHCount"] = 5
        want_acc, want_out = oracle(default_scene_arrays, opc, [0, 1], W, H)
        acc, out = render(R, r, opc, [0, 1], W, H, 1, 3)
        assert np.array_equal(bits(acc), bits(want_acc)) and np.array_equal(bits(out), bits(want_out))
        # debug views and the counting build do not change with the option
        for mode in (1, 2, 3, 4):
            opc = scenes.make_pc(default_scene_arrays, W, H, renderMode=mode)
            r.set_option(R.RT_OPT_TRAVERSAL, 1)
            r.render_frame(to_pc(R, opc))
            want = O.render(default_scene_arrays, O.pc_bytes(opc), W, H)["out"]
            assert np.array_equal(bits(r.read_image(R.RT_IMAGE_DRAW)), bits(want)), mode
        o = O.render(default_scene_arrays, O.pc_bytes(scenes.make_pc(default_scene_arrays, W, H)), W, H, counters=True)
        st = r.frame_stats(to_pc(R, scenes.make_pc(default_scene_arrays, W, H)))
        assert (st.rays, st.bvhNodeTests, st.triangleTests) == (int(o["rays"].sum()), int(o["node"].sum()), int(o["tri"].sum()))
    finally:
        r.close()


def test_mixed_material_scene(R):
    """Glass, mirror, fuzzy metal, emitter, open sky; objects scaled below 1 (teapot, icosphere, cottage,
    dragon at 0.5) keep the reference's trees, the unit-scale cube and sphere walk rebuilt ones."""
    materials, objects = scenes.mixed_recipe()
    arrays = scenes.build_oracle(materials, objects).bindings()
    r = R.Renderer(0)
    try:
        r.upload_scene(arrays)
        assert r.get_option(R.RT_INFO_FAST_OBJECTS) == 2
        W, H = 320, 180
        opc = scenes.make_pc(arrays, W, H, cam_pos=(0.0, 2.5, 8.0), look=(0.0, -0.15, -1.0), fov=60.0, maxBounces=5)
        want_acc, want_out = oracle(arrays, opc, [0, 1], W, H)
        for pipeline in (3, 0, 4):
            acc, out = render(R, r, opc, [0, 1], W, H, 1, pipeline)
            same, ok = shares(acc, want_acc)
            assert ok >= 0.995 and same >= 0.99, (pipeline, same, ok)
        frames = list(range(64))
        W, H = 192, 108
        opc = scenes.make_pc(arrays, W, H, cam_pos=(0.0, 2.5, 8.0), look=(0.0, -0.15, -1.0), fov=60.0, maxBounces=5)
        want_acc, want_out = oracle(arrays, opc, frames, W, H)
        acc, out = render(R, r, opc, frames, W, H, 1, 3)
        assert psnr(out, want_out) >= 45.0
    finally:
        r.close()


def test_only_shrunk_objects_is_bit_exact(R):
    """A scene in which every BVH object is scaled below 1: nothing is rebuilt, only the world boxes are
    tightened, and the image is the reference-order image bit for bit."""
    materials = [scenes.mat((0.5, 0.5, 0.5)), scenes.mat((0.8, 0.3, 0.2)), scenes.mat((1, 1, 1), smoothness=1.0)]
    objects = [("flat_quad.obj", (0, 0, 0), (0, 0, 0), (12, 12, 12), 0),
               ("teapot.obj", (0.0, 0.0, -2.5), (0, -40, 0), (0.6, 0.6, 0.6), 1),
               ("dragon_lowres.obj", (3.0, 1.2, -3.0), (0, 120, 0), (0.5, 0.7, 0.5), 2)]
    arrays = scenes.build_oracle(materials, objects).bindings()
    r = R.Renderer(0)
    try:
        r.upload_scene(arrays)
        assert r.get_option(R.RT_INFO_FAST_OBJECTS) == 0
        W, H = 256, 144
        opc = scenes.make_pc(arrays, W, H, cam_pos=(0.0, 2.5, 8.0), look=(0.0, -0.15, -1.0), fov=60.0)
        want_acc, want_out = oracle(arrays, opc, [0, 1], W, H)
        acc, out = render(R, r, opc, [0, 1], W, H, 1, 3)
        same, ok = shares(acc, want_acc)
        assert same >= 0.9999, same        # tight world boxes only: a grazing ray at most
    finally:
        r.close()


@pytest.mark.parametrize("name", ["heightfield", "blob_grid"])
def test_generated_configs_full_size(R, name, tmp_path_factory):
    """The 100 352-triangle height field and the 999 424-triangle mixed-material scene at 1920x1080:
    the whole image against the reference-order kernel's (itself bit-exact with the oracle), oracle
    windows, and the rebuilt trees' depth."""
    from raytracer_amd import scenefile as SF
    tmp = str(tmp_path_factory.mktemp("gen_" + name))
    desc = SF.config_heightfield(tmp) if name == "heightfield" else SF.config_blob_grid(tmp)
    scene, cam, pc = SF.build_scene(desc, tmp)
    arrays = scene.buffers().to_arrays()
    W, H = desc["width"], desc["height"]
    opc = np.frombuffer(bytes(pc.as_bytes()), dtype=O.PC_DTYPE).copy()
    r = R.Renderer(0)
    try:
        r.upload_scene(scene.buffers())
        assert r.get_option(R.RT_INFO_FAST_OBJECTS) == (1 if name == "heightfield" else 64)
        assert r.get_option(R.RT_INFO_FAST_STACK_DEPTH) <= 31
        ref_acc, ref_out = render(R, r, opc, [0, 1], W, H, 0, 3)
        acc, out = render(R, r, opc, [0, 1], W, H, 1, 3)
        same, ok = shares(acc, ref_acc)
        assert ok >= 0.995 and same >= 0.999, (same, ok)
        acc0, out0 = render(R, r, opc, [0, 1], W, H, 1, 0)
        assert shares(acc0, ref_acc)[1] >= 0.995
        # path waves + walker waves: the near-first image of the lock-step kernel, and (reference order) the oracle's
        acc4, out4 = render(R, r, opc, [0, 1], W, H, 1, 4)
        assert shares(acc4, acc0)[0] >= 0.9999
        acc4, out4 = render(R, r, opc, [0, 1], W, H, 0, 4)
        assert np.array_equal(bits(acc4), bits(ref_acc)) and np.array_equal(bits(out4), bits(ref_out))
        for win in ([(900, 520, 1028, 568), (300, 700, 396, 748)] if name == "heightfield" else [(900, 500, 996, 548), (600, 640, 664, 688)]):
            x0, y0, x1, y1 = win
            wa, wo = oracle(arrays, opc, [0, 1], W, H, window=win)
            same, ok = shares(acc[y0:y1, x0:x1], wa[y0:y1, x0:x1])
            assert ok >= 0.995, (win, same, ok)
    finally:
        r.close()


def test_more_than_64_bvh_parents_in_a_row(R):
    """70 BVH parents in a row (the lock-step kernels go through runs of BVH parents lane by lane, 64 parents
    at a time), a wall behind them and a floor: reference order bit-exact on pipelines 3 and 0, near-first
    within the bars. Unit-scale cubes and spheres walk rebuilt trees, the shrunk icospheres keep theirs."""
    materials = [scenes.mat((0.6, 0.6, 0.6)), scenes.mat((0.8, 0.3, 0.2)), scenes.mat((0.9, 0.9, 0.9), smoothness=1.0),
                 scenes.mat((1, 1, 1), ior=1.5)]
    objects = []
    for k in range(70):
        gx, gy = k % 10, k // 10
        model, scale = (("cube.obj", 1.0), ("smooth_sphere.obj", 1.0), ("icosphere.obj", 0.5))[k % 3]
        objects.append((model, (-9.0 + 2.0 * gx, 0.6 + 1.6 * gy, -6.0 - 0.7 * (k % 4)), (0, 17 * k, 0), (scale * 0.6,) * 3 if scale != 1.0 else (1.0, 1.0, 1.0), 1 + k % 3))
    objects.append(("flat_quad.obj", (0, 0, 0), (0, 0, 0), (40, 40, 40), 0))
    objects.append(("flat_quad.obj", (0, 0, -12), (90, 0, 0), (40, 40, 40), 0))
    arrays = scenes.build_oracle(materials, objects).bindings()
    r = R.Renderer(0)
    try:
        r.upload_scene(arrays)
        assert 0 < r.get_option(R.RT_INFO_FAST_OBJECTS) < 70
        W, H = 240, 136
        opc = scenes.make_pc(arrays, W, H, cam_pos=(0.0, 6.0, 9.0), look=(0.0, -0.1, -1.0), fov=70.0, maxBounces=4)
        want_acc, want_out = oracle(arrays, opc, [0, 1], W, H)
        for pipeline in (3, 0, 4):
            acc, out = render(R, r, opc, [0, 1], W, H, 0, pipeline)
            assert np.array_equal(bits(acc), bits(want_acc)) and np.array_equal(bits(out), bits(want_out)), pipeline
            acc, out = render(R, r, opc, [0, 1], W, H, 1, pipeline)
            same, ok = shares(acc, want_acc)
            assert ok >= 0.995 and same >= 0.99, (pipeline, same, ok)
    finally:
        r.close()


def test_hierarchy_over_the_parents(R):
    """23 objects that own a tree, none scaled below 1 (so that hits do not depend on the visiting order), a floor
    and a back wall: pipeline 3 walks them from the hierarchy over their world boxes (RT_OPT_WIDE_RECORDS 1).
    Against the oracle within the bars, in launches of several frames and in single-frame dispatches, and
    against the same four-child records without the hierarchy (RT_OPT_WIDE_RECORDS 2) and two-child records (0)."""
    materials = [scenes.mat((0.6, 0.6, 0.6)), scenes.mat((0.8, 0.3, 0.2)), scenes.mat((0.9, 0.9, 0.9), smoothness=1.0, fuzziness=0.2),
                 scenes.mat((1, 1, 1), ior=1.5), scenes.mat((0.2, 0.5, 0.8))]
    objects = []
    for k in range(23):
        gx, gy, gz = k % 4, (k // 4) % 3, k // 12
        model = ("cube.obj", "smooth_sphere.obj", "icosphere.obj", "teapot.obj")[k % 4]
        s = (1.0, 1.3, 1.0, 1.0)[k % 4]
        objects.append((model, (-6.0 + 4.0 * gx + 0.7 * gz, 1.2 + 3.0 * gy, -4.0 - 5.0 * gz), (0, 31 * k, 7 * k), (s, s, s), 1 + k % 4))
    objects.append(("flat_quad.obj", (0, 0, 0), (0, 0, 0), (40, 40, 40), 0))
    objects.append(("flat_quad.obj", (0, 0, -16), (90, 0, 0), (40, 40, 40), 0))
    arrays = scenes.build_oracle(materials, objects).bindings()
    r = R.Renderer(0)
    try:
        r.upload_scene(arrays)
        assert r.get_option(R.RT_INFO_FAST_OBJECTS) == 23
        assert r.get_option(R.RT_INFO_TLAS_RECORDS) > 0
        W, H = 256, 144
        opc = scenes.make_pc(arrays, W, H, cam_pos=(0.0, 5.0, 11.0), look=(0.0, -0.12, -1.0), fov=65.0, maxBounces=4)
        want_acc, want_out = oracle(arrays, opc, [0, 1, 2], W, H)
        images = {}
        for wide in (1, 2, 0):
            r.set_option(R.RT_OPT_WIDE_RECORDS, wide)
            assert (r.get_option(R.RT_INFO_TLAS_RECORDS) > 0) == (wide != 2)      # (0 only stops the kernels from using the wide scene)
            acc, out = render(R, r, opc, [0, 1, 2], W, H, 1, 3)
            same, ok = shares(acc, want_acc)
            assert ok >= 0.995 and same >= 0.99, (wide, same, ok)
            images[wide] = acc
        assert shares(images[1], images[2])[0] >= 0.999 and shares(images[1], images[0])[0] >= 0.999
        # the reference's interactive dispatch: one frame per launch, two pixels per lane
        r.set_option(R.RT_OPT_WIDE_RECORDS, 1)
        r.set_option(R.RT_OPT_PIPELINE, 3)
        r.set_option(R.RT_OPT_TRAVERSAL, 1)
        r.clear_accumulation()
        for f in (0, 1, 2):
            opc["frame"] = f
            r.render_frame(to_pc(R, opc))
        same, ok = shares(r.read_image(R.RT_IMAGE_ACCUM), want_acc)
        assert ok >= 0.995 and same >= 0.99, (same, ok)
        # the sun from another side: shadow rays take the same walk
        opc = scenes.make_pc(arrays, W, H, cam_pos=(0.0, 5.0, 11.0), look=(0.0, -0.12, -1.0), fov=65.0, maxBounces=2, sunDirection=(0.6, -0.5, -0.4))
        want_acc, want_out = oracle(arrays, opc, [0, 1], W, H)
        acc, out = render(R, r, opc, [0, 1], W, H, 1, 3)
        same, ok = shares(acc, want_acc)
        assert ok >= 0.995 and same >= 0.99, (same, ok)
    finally:
        r.close()


def test_many_leaf_parents_every_lane_for_itself(R):
    """40 quads (objects that are one leaf) standing one behind the other, turned so that their boxes are much larger
    than they are - a ray enters many boxes and misses most quads - plus two objects with trees: the wide kernels let
    every lane test its own rays' leaf parents nearest box first, 32 at a time, and measure the boxes again when a
    ray is through with its two nearest. Against the oracle within the bars, on pipelines 3 and 0, in launches of
    several frames and in single-frame dispatches; reference order bit-exact."""
    materials = [scenes.mat((0.6, 0.6, 0.6)), scenes.mat((0.8, 0.3, 0.2)), scenes.mat((0.2, 0.6, 0.8)), scenes.mat((0.9, 0.9, 0.9), smoothness=1.0)]
    objects = [("flat_quad.obj", (0, 0, 0), (0, 0, 0), (40, 40, 40), 0)]
    for k in range(40):
        objects.append(("flat_quad.obj", (-3.0 + 0.15 * k + 1.7 * (k % 3), 1.0 + 0.9 * (k % 5), -1.0 - 0.45 * k), (35 + 11 * (k % 7), 20 * k, 45), (1.0 + 0.1 * (k % 4),) * 3, 1 + k % 3))
    objects.append(("smooth_sphere.obj", (2.5, 1.2, -4.0), (0, 0, 0), (1.2, 1.2, 1.2), 3))
    objects.append(("cube.obj", (-2.5, 0.8, -6.0), (0, 30, 0), (1.5, 1.5, 1.5), 1))
    arrays = scenes.build_oracle(materials, objects).bindings()
    r = R.Renderer(0)
    try:
        r.upload_scene(arrays)
        assert r.get_option(R.RT_INFO_FAST_OBJECTS) == 2
        W, H = 256, 144
        opc = scenes.make_pc(arrays, W, H, cam_pos=(0.0, 3.0, 6.0), look=(0.0, -0.12, -1.0), fov=65.0, maxBounces=4, sunDirection=(0.3, -0.8, -0.5))
        want_acc, want_out = oracle(arrays, opc, [0, 1, 2], W, H)
        for pipeline in (3, 0):
            acc, out = render(R, r, opc, [0, 1, 2], W, H, 0, pipeline)
            assert np.array_equal(bits(acc), bits(want_acc)) and np.array_equal(bits(out), bits(want_out)), pipeline
            acc, out = render(R, r, opc, [0, 1, 2], W, H, 1, pipeline)
            same, ok = shares(acc, want_acc)
            assert ok >= 0.995 and same >= 0.99, (pipeline, same, ok)
        r.set_option(R.RT_OPT_PIPELINE, 3)
        r.set_option(R.RT_OPT_TRAVERSAL, 1)
        r.clear_accumulation()
        for f in (0, 1, 2):
            opc["frame"] = f
            r.render_frame(to_pc(R, opc))
        same, ok = shares(r.read_image(R.RT_IMAGE_ACCUM), want_acc)
        assert ok >= 0.995 and same >= 0.99, (same, ok)
    finally:
        r.close()


def test_single_frame_dispatch_with_a_tile_size_that_is_no_multiple_of_16(R, default_scene_arrays):
    """rt_render_frame pairs two rows of pixels per lane, which needs tiles of whole 16-row bands; with another tile size
    the single-frame dispatch is rendered by kernel 0, which walks two-child records - in both flavours (the four-child
    scenes are for the pair kernels only)."""
    W, H = 200, 120
    opc = scenes.make_pc(default_scene_arrays, W, H)
    want_acc, want_out = oracle(default_scene_arrays, opc, [0, 1], W, H)
    r = R.Renderer(0)
    try:
        r.upload_scene(default_scene_arrays)
        r.resize(W, H)
        r.set_option(R.RT_OPT_TILE_SIZE, 24)
        for traversal in (0, 1):
            r.set_option(R.RT_OPT_TRAVERSAL, traversal)
            r.clear_accumulation()
            for f in (0, 1):
                opc["frame"] = f
                r.render_frame(to_pc(R, opc))
            acc = r.read_image(R.RT_IMAGE_ACCUM)
            if traversal == 0:
                assert np.array_equal(bits(acc), bits(want_acc))
            else:
                same, ok = shares(acc, want_acc)
                assert ok >= 0.995 and same >= 0.999, (same, ok)
    finally:
        r.close()


def test_launch_shapes_do_not_change_the_near_first_image(R, default_scene_arrays):
    """The near-first flavour is not bit-exact against the oracle (exact ties), but it is one function of the scene and the
    settings: single-frame dispatches, launches of several frames in one piece and launches in pieces give the same
    accumulation and draw images bit for bit, at a ragged size, with 2 rays per pixel, for a strided frame set whose last
    group is ragged, accumulation on and off."""
    W, H = 131, 85
    r = R.Renderer(0)
    try:
        r.upload_scene(default_scene_arrays); r.resize(W, H)
        r.set_option(R.RT_OPT_TRAVERSAL, 1)
        r.set_option(R.RT_OPT_PIPELINE, 3)
        for accumulate in (1, 0):
            opc = scenes.make_pc(default_scene_arrays, W, H, raysPerPixel=2, accumulateFrames=accumulate)
            r.set_option(R.RT_OPT_FRAMES_PER_LAUNCH, 1)
            r.clear_accumulation()
            r.render_frames(to_pc(R, opc), 2, 3, 7)
            one = (r.read_image(R.RT_IMAGE_ACCUM), r.read_image(R.RT_IMAGE_DRAW))
            for group, pieces in ((2, 1), (3, 1), (7, 1), (64, 1), (3, 0), (64, 0), (5, 2)):
                r.set_option(R.RT_OPT_FRAMES_PER_LAUNCH, group)
                r.set_option(R.RT_OPT_LAUNCH_PIECES, pieces)
                r.clear_accumulation()
                r.render_frames(to_pc(R, opc), 2, 3, 7)
                got = (r.read_image(R.RT_IMAGE_ACCUM), r.read_image(R.RT_IMAGE_DRAW))
                assert np.array_equal(bits(got[0]), bits(one[0])), (accumulate, group, pieces)
                assert np.array_equal(bits(got[1]), bits(one[1])), (accumulate, group, pieces)
            r.set_option(R.RT_OPT_LAUNCH_PIECES, 1)
            if accumulate:
                # and the tolerance against the oracle holds at this shape as well
                frames = [2 + 3 * k for k in range(7)]
                want_acc, _ = oracle(default_scene_arrays, opc, frames, W, H)
                same, ok = shares(one[0], want_acc)
                assert ok >= 0.995 and same >= 0.99, (same, ok)
    finally:
        r.close()
