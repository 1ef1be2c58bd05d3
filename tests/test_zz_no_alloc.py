"""rt_frame_stats_noalloc: the entry point of the C ABI that allocates nothing (per-wave counter slots in
a file-scope device array, summed by a second kernel into memory the caller passes). Checked against
rt_frame_stats_compute - whose counters the other GPU tests compare with the oracle's - for exact
equality of all four counters, on two calls in a row of which the second has the smaller input (stale
slots of the first call must not leak into the second), with the result in device memory and in pinned
host memory, and against the oracle itself on the small case."""
import numpy as np
import pytest

import oracle_lib as O
import scenes

pytestmark = pytest.mark.gpu


def to_pc(R, opc):
    return R.PushConstants.from_buffer_copy(O.pc_bytes(opc).tobytes())


def test_frame_stats_noalloc_matches_allocating_entry_point(default_scene_arrays):
    import torch
    import raytracer_amd as R
    r = R.Renderer(0)
    try:
        r.upload_scene(default_scene_arrays)
        dev_out = torch.full((4,), -1, dtype=torch.int64, device="cuda:0")
        pin_out = torch.full((4,), -1, dtype=torch.int64).pin_memory()
        torch.cuda.synchronize()
        sizes = [(1000, 600), (200, 120)]          # second call: smaller input, fewer slots
        for k, (W, H) in enumerate(sizes):
            r.resize(W, H)
            opc = scenes.make_pc(default_scene_arrays, W, H, raysPerPixel=2 - k)
            opc["frame"] = 3 + k
            pc = to_pc(R, opc)
            want = r.frame_stats(pc)
            want = (want.rays, want.bvhNodeTests, want.triangleTests, want.samples)
            out = dev_out if k == 0 else pin_out
            r.frame_stats_noalloc(pc, out.data_ptr())
            r.sync()
            got = tuple(int(v) for v in out.cpu().numpy())
            assert got == want, (W, H, got, want)
            assert got[3] == W * H * (2 - k) and got[0] >= got[3]
        # the small case against the oracle's own counters
        W, H = sizes[1]
        o = O.render(default_scene_arrays, O.pc_bytes(opc), W, H, counters=True)
        assert got[:3] == (int(o["rays"].sum()), int(o["node"].sum()), int(o["tri"].sum()))
        # both outputs once more, back to back on one stream, no sync in between, other pipeline selected
        r.set_option(R.RT_OPT_PIPELINE, 2)
        dev_out.fill_(-1); pin_out.fill_(-1)
        torch.cuda.synchronize()
        r.frame_stats_noalloc(pc, dev_out.data_ptr())
        r.frame_stats_noalloc(pc, pin_out.data_ptr())
        r.sync()
        assert tuple(int(v) for v in dev_out.cpu().numpy()) == want
        assert tuple(int(v) for v in pin_out.numpy()) == want
        # tile partition: the two halves add up
        r.set_option(R.RT_OPT_TILE_WORLD, 2); r.set_option(R.RT_OPT_TILE_SIZE, 32)
        parts = []
        for rank in range(2):
            r.set_option(R.RT_OPT_TILE_RANK, rank)
            r.frame_stats_noalloc(pc, dev_out.data_ptr())
            r.sync()
            parts.append(dev_out.cpu().numpy().copy())
        assert tuple(int(v) for v in parts[0] + parts[1]) == want
        r.set_option(R.RT_OPT_TILE_RANK, 0); r.set_option(R.RT_OPT_TILE_WORLD, 1); r.set_option(R.RT_OPT_TILE_SIZE, 64)
        # a host buffer registered after the fact (hipHostRegister): the kernels get its device address
        rt = torch.cuda.cudart()
        if hasattr(rt, "cudaHostRegister"):
            import mmap
            pages = mmap.mmap(-1, 4 * mmap.PAGESIZE)                # page-aligned, whole pages
            raw = np.frombuffer(pages, dtype=np.int64)
            raw[:] = -1
            err = rt.cudaHostRegister(raw.ctypes.data, raw.nbytes, 0)
            if int(err) == 0:                                       # (a runtime that cannot register this mapping: nothing to check)
                try:
                    r.frame_stats_noalloc(pc, raw.ctypes.data)
                    r.sync()
                    assert tuple(int(v) for v in raw[:4]) == want
                finally:
                    rt.cudaHostUnregister(raw.ctypes.data)
            del raw
        # refusals: pageable host memory, a dispatch with more waves than the array has slots
        plain = np.zeros(4, np.int64)
        with pytest.raises(R.RtError) as e:
            r.frame_stats_noalloc(pc, plain.ctypes.data)
        assert e.value.code == -1
        r.resize(4096, 2304)                       # 147 456 waves > 131 072 slots
        big = to_pc(R, scenes.make_pc(default_scene_arrays, 4096, 2304))
        with pytest.raises(R.RtError) as e:
            r.frame_stats_noalloc(big, dev_out.data_ptr())
        assert e.value.code == -8 and "131072" in str(e.value)
        assert r.frame_stats(big).samples == 4096 * 2304      # the allocating entry point still takes it
    finally:
        r.close()
