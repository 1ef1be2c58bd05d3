"""Scene descriptions and OBJ generators: deterministic output, product scene builder equal to the
oracle's on generated models, JSON round trip; on the GPU, parity of every pipeline with the oracle
on small instances of the two generated benchmark scenes."""
import hashlib
import os

import numpy as np
import pytest

import oracle_lib as O
import scenes


def desc_to_recipe(desc):
    mats = [scenes.mat(tuple(m["albedo"]), m["smoothness"], m["fuzziness"], m["emission"], m["refractionIndex"], tuple(m["absorbtion"]))
            for m in desc["materials"]]
    objs = [(o["model"], tuple(o["position"]), tuple(o["rotation"]), tuple(o["scale"]), o["material"]) for o in desc["objects"]]
    return mats, objs


def oracle_pc(desc, W, H, parent_count):
    c = desc["camera"]
    pc = O.default_pc(parent_count)
    O.camera_push_constants(O.pc_bytes(pc), tuple(c["position"]), tuple(c["lookDirection"]), c["fov"], c["focusDistance"], c["defocusAngle"], W, H)
    for k, v in desc.get("settings", {}).items():
        pc[k] = v
    return pc


def closeness(a, ref, tol=1e-5):
    """Share of pixels whose channels all agree within tol (absolute + relative), NaN matching NaN."""
    with np.errstate(invalid="ignore"):
        ok = np.abs(a - ref) <= tol + tol * np.abs(ref)
    ok |= np.isnan(a) & np.isnan(ref)
    return float(np.all(ok, axis=2).mean())


def small_configs(tmp):
    from raytracer_amd import scenefile as SF
    hf = SF.config_heightfield(tmp, cells=24, width=160, height=90, frames=2)
    bg = SF.config_blob_grid(tmp, side=2, segments=16, rings=10, width=160, height=90, frames=2)
    return hf, bg


def test_generators_are_deterministic(tmp_path):
    from raytracer_amd import scenefile as SF
    a, b = str(tmp_path / "a"), str(tmp_path / "b")
    for d in (a, b):
        assert SF.write_heightfield_obj(os.path.join(d, "h.obj"), cells=16, seed=7) == 2 * 16 * 16
        assert SF.write_blob_obj(os.path.join(d, "s.obj"), seed=9, segments=12, rings=8) == 2 * 12 * 7
    for name in ("h.obj", "s.obj"):
        ha = hashlib.sha256(open(os.path.join(a, name), "rb").read()).hexdigest()
        hb = hashlib.sha256(open(os.path.join(b, name), "rb").read()).hexdigest()
        assert ha == hb
    # a different seed gives a different file
    SF.write_heightfield_obj(os.path.join(a, "h2.obj"), cells=16, seed=8)
    assert open(os.path.join(a, "h2.obj"), "rb").read() != open(os.path.join(a, "h.obj"), "rb").read()
    # the sequence itself (known answers of x <- 1664525 x + 1013904223 mod 2^32)
    g = SF.Lcg(0)
    assert [g.next() for _ in range(3)] == [1013904223, 1196435762, 3519870697]


def test_generated_scenes_build_equal_to_oracle(tmp_path):
    """Product OBJ reader + BVH builder + object conversion against the oracle's on generated
    models: all 18 arrays byte-identical, structure facts equal, BVH depth within the 32-entry stack."""
    from raytracer_amd import scenefile as SF
    tmp = str(tmp_path)
    for desc in small_configs(tmp):
        scene, cam, pc = SF.build_scene(desc, tmp)
        mats, objs = desc_to_recipe(desc)
        osc = scenes.build_oracle(mats, objs, models_dir=tmp)
        arrays = scene.buffers().to_arrays()
        for b in range(18):
            assert np.array_equal(np.asarray(arrays[b]).view(np.uint8).ravel(), np.asarray(osc.binding(b)).view(np.uint8).ravel()), (desc["name"], b)
        for m in sorted({o["model"] for o in desc["objects"]}):
            info = scene.model_info(os.path.join(tmp, m))
            assert info == tuple(osc.model_info(os.path.join(tmp, m)))[:4] and info[2] <= 31
        opc = oracle_pc(desc, desc["width"], desc["height"], len(desc["objects"]))
        assert bytes(pc.as_bytes()) == O.pc_bytes(opc).tobytes()


def test_scene_file_round_trip(tmp_path):
    from raytracer_amd import scenefile as SF
    tmp = str(tmp_path)
    desc = small_configs(tmp)[1]
    path = os.path.join(tmp, "scene.json")
    SF.save_scene(desc, path)
    back = SF.load_scene(path)
    assert back == desc
    s1, _, pc1 = SF.build_scene(desc, tmp)
    s2, _, pc2 = SF.build_scene(back, tmp)
    for x, y in zip(s1.buffers().to_arrays(), s2.buffers().to_arrays()):
        assert np.array_equal(np.asarray(x).view(np.uint8).ravel(), np.asarray(y).view(np.uint8).ravel())
    assert bytes(pc1.as_bytes()) == bytes(pc2.as_bytes())
    bad = dict(desc); bad["settings"] = {"noSuchSetting": 1}
    with pytest.raises(ValueError):
        SF.build_scene(bad, tmp)
    bad = dict(desc); bad["objects"] = [dict(desc["objects"][0], material=99)]
    with pytest.raises(ValueError):
        SF.build_scene(bad, tmp)
    with pytest.raises(ValueError):
        open(os.path.join(tmp, "empty.json"), "w").write("{}")
        SF.load_scene(os.path.join(tmp, "empty.json"))


@pytest.mark.gpu
@pytest.mark.parametrize("pipeline", [0, 1, 2], ids=["lockstep", "scheduled", "wavefront"])
def test_generated_scenes_bit_exact_on_gpu(tmp_path, pipeline):
    import raytracer_amd as R
    from raytracer_amd import scenefile as SF
    tmp = str(tmp_path)
    for desc in small_configs(tmp):
        W, H = desc["width"], desc["height"]
        scene, cam, pc = SF.build_scene(desc, tmp)
        arrays = scene.buffers().to_arrays()
        r = R.Renderer(0)
        try:
            r.set_option(R.RT_OPT_PIPELINE, pipeline)
            r.upload_scene(scene.buffers()); r.resize(W, H)
            acc_o = np.zeros((H, W, 4), np.float32)
            for f in range(desc["frames"]):
                pc.frame = f
                r.render_frame(pc)
                o = O.render(arrays, pc.as_bytes(), W, H, accum=acc_o, counters=True); acc_o = o["accum"]
            acc = r.read_image(R.RT_IMAGE_ACCUM); out = r.read_image(R.RT_IMAGE_DRAW)
            assert np.array_equal(acc.view(np.uint32), acc_o.view(np.uint32)), desc["name"]
            assert np.array_equal(out.view(np.uint32), o["out"].view(np.uint32)), desc["name"]
            st = r.frame_stats(pc)
            assert (st.rays, st.bvhNodeTests, st.triangleTests) == (int(o["rays"].sum()), int(o["node"].sum()), int(o["tri"].sum()))
            # the depth view has no transcendental in it: bit-exact
            pc.renderMode = 2; pc.frame = 0
            r.render_frame(pc)
            want = O.render(arrays, pc.as_bytes(), W, H)["out"]
            assert np.array_equal(r.read_image(R.RT_IMAGE_DRAW).view(np.uint32), want.view(np.uint32)), desc["name"]
            pc.renderMode = 0
        finally:
            r.close()
