"""CPU rehearsal of the multi-GPU drivers: two processes over gloo. The renderer behind the
`target` protocol is the CPU oracle here (a test stand-in: the product has no CPU renderer); what
is under test is the split / merge logic of raytracer_amd.distributed, which on the GPU box runs
unchanged over RCCL."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_lib as O
import scenes
from raytracer_amd import distributed as D

W, H = 48, 32


class OracleTarget:
    def __init__(self, arrays, opc):
        self.arrays, self.opc = arrays, opc
        self.accum = torch.zeros((H, W, 4), dtype=torch.float32)
        self.draw = torch.zeros((H, W, 4), dtype=torch.float32)
        self.tiles = (0, 1, 64)

    def clear(self):
        self.accum.zero_(); self.draw.zero_()

    def set_tiles(self, rank, world, tile_size):
        self.tiles = (rank, world, tile_size)

    def render_frames(self, first, stride, count):
        rank, world, ts = self.tiles
        for k in range(count):
            self.opc["frame"] = first + k * stride
            acc = self.accum.numpy()
            for tile_x, tile_y in D.tiles_of_rank(rank, world, W, H, ts):
                x0, y0 = tile_x * ts, tile_y * ts
                r = O.render(self.arrays, O.pc_bytes(self.opc), W, H, accum=acc, window=(x0, y0, x0 + ts, y0 + ts), nthreads=2)
                acc = r["accum"]
                self.draw.numpy()[y0:y0 + ts, x0:x0 + ts] = r["out"][y0:y0 + ts, x0:x0 + ts]
            self.accum.copy_(torch.from_numpy(acc))

    def resolve(self, total):
        a = self.accum.numpy()
        d = self.draw.numpy()
        d[..., :3] = np.sqrt(np.maximum(a[..., :3] * np.float32(1.0 / total), 0), dtype=np.float32)
        d[..., 3] = a[..., 3]


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    s = O.OracleScene()
    s.default_scene()
    arrays = s.bindings()
    opc = scenes.make_pc(arrays, W, H)
    tgt = OracleTarget(arrays, opc)
    total = D.render_sample_split(tgt, frames_per_rank=3)
    np.save(os.path.join(out_dir, "ss_acc_%d.npy" % rank), tgt.accum.numpy())
    np.save(os.path.join(out_dir, "ss_out_%d.npy" % rank), tgt.draw.numpy())
    assert total == 3 * world
    D.render_tile_split(tgt, frames=2, tile_size=16)
    np.save(os.path.join(out_dir, "ts_acc_%d.npy" % rank), tgt.accum.numpy())
    np.save(os.path.join(out_dir, "ts_out_%d.npy" % rank), tgt.draw.numpy())
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_partitions_cover_everything_once():
    for world in (1, 2, 3, 8):
        frames = sorted(f for r in range(world) for f in D.frames_of_rank(r, world, 5))
        assert frames == list(range(5 * world))
        tiles = sorted(t for r in range(world) for t in D.tiles_of_rank(r, world, 200, 120, 32))
        assert tiles == sorted((x, y) for x in range(7) for y in range(4))
        owner = D.tile_owner_map(world, 200, 120, 32)
        for r in range(world):
            for x, y in D.tiles_of_rank(r, world, 200, 120, 32):
                assert (owner[y * 32:(y + 1) * 32, x * 32:(x + 1) * 32] == r).all()
    # serpentine numbering: consecutive numbers are neighbouring tiles
    path = [D.tile_of_position(p, 7) for p in range(28)]
    assert path[:8] == [(0, 0), (1, 0), (2, 0), (3, 0), (4, 0), (5, 0), (6, 0), (6, 1)]
    assert all(abs(a[0] - b[0]) + abs(a[1] - b[1]) == 1 for a, b in zip(path, path[1:]))


def test_two_rank_gloo_sample_and_tile_split(tmp_path, default_scene_arrays):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    # sequential single-process render of the same frames
    opc = scenes.make_pc(default_scene_arrays, W, H)
    acc = np.zeros((H, W, 4), np.float32)
    for f in range(3 * world):
        opc["frame"] = f
        r = O.render(default_scene_arrays, O.pc_bytes(opc), W, H, accum=acc, nthreads=4)
        acc = r["accum"]
    for rank in range(world):
        got = np.load(tmp_path / ("ss_acc_%d.npy" % rank))
        out = np.load(tmp_path / ("ss_out_%d.npy" % rank))
        # same samples, summed in a different order: f32 re-association only
        np.testing.assert_allclose(got[..., :3], acc[..., :3], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(out[..., :3], r["out"][..., :3], rtol=1e-5, atol=1e-6)
    a0, a1 = np.load(tmp_path / "ss_acc_0.npy"), np.load(tmp_path / "ss_acc_1.npy")
    assert np.array_equal(a0.view(np.uint32), a1.view(np.uint32))      # every rank holds the same merged image
    # tile split is bit-identical to the single-process render
    acc = np.zeros((H, W, 4), np.float32)
    for f in range(2):
        opc["frame"] = f
        r = O.render(default_scene_arrays, O.pc_bytes(opc), W, H, accum=acc, nthreads=4)
        acc = r["accum"]
    for rank in range(world):
        got = np.load(tmp_path / ("ts_acc_%d.npy" % rank))
        out = np.load(tmp_path / ("ts_out_%d.npy" % rank))
        assert np.array_equal(got.view(np.uint32), acc.view(np.uint32))
        assert np.array_equal(out.view(np.uint32), r["out"].view(np.uint32))


def _gpu_worker(rank, world, port, out_dir):
    """Two processes that share cuda:0 (a rehearsal: RCCL wants one GPU per rank, so the collective
    goes over gloo here, through a pinned host copy; the drivers and the kernels are the ones an 8-GPU
    run uses)."""
    import faulthandler
    faulthandler.dump_traceback_later(90, exit=True)     # a stuck collective ends the worker with a traceback
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import raytracer_amd as R
    Wg, Hg = 160, 96
    s = R.Scene(); s.init_default()
    r = R.Renderer(0)
    r.upload_scene(s.buffers())
    pc = R.default_push_constants(); R.camera_to_push_constants(R.default_camera(), Wg, Hg, pc); pc.parentBVHCount = 8
    for pipeline in (0, 2):
        r.set_option(R.RT_OPT_PIPELINE, pipeline)
        tgt = D.GpuTarget(r, pc, Wg, Hg, "cuda:0")
        total = D.render_sample_split(tgt, frames_per_rank=2)
        assert total == 2 * world
        torch.cuda.synchronize()
        np.save(os.path.join(out_dir, "g_ss_acc_%d_%d.npy" % (pipeline, rank)), tgt.accum.cpu().numpy())
        np.save(os.path.join(out_dir, "g_ss_out_%d_%d.npy" % (pipeline, rank)), tgt.draw.cpu().numpy())
        D.render_tile_split(tgt, frames=2, tile_size=32)
        torch.cuda.synchronize()
        np.save(os.path.join(out_dir, "g_ts_acc_%d_%d.npy" % (pipeline, rank)), tgt.accum.cpu().numpy())
        np.save(os.path.join(out_dir, "g_ts_out_%d_%d.npy" % (pipeline, rank)), tgt.draw.cpu().numpy())
        tgt.close()
    r.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_sample_and_tile_split(tmp_path):
    import raytracer_amd as R
    world, Wg, Hg = 2, 160, 96
    mp.spawn(_gpu_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    s = R.Scene(); s.init_default()
    r = R.Renderer(0)
    try:
        r.upload_scene(s.buffers()); r.resize(Wg, Hg)
        pc = R.default_push_constants(); R.camera_to_push_constants(R.default_camera(), Wg, Hg, pc); pc.parentBVHCount = 8
        # what each rank accumulates on its own: frames r, r+2
        parts = []
        for rank in range(world):
            r.clear_accumulation()
            r.render_frames(pc, rank, world, 2)
            parts.append(r.read_image(R.RT_IMAGE_ACCUM))
        merged = parts[0] + parts[1]
        r.clear_accumulation()
        r.render_frames(pc, 0, 1, 2)
        seq_acc, seq_out = r.read_image(R.RT_IMAGE_ACCUM), r.read_image(R.RT_IMAGE_DRAW)
    finally:
        r.close()
    for pipeline in (0, 2):
        for rank in range(world):
            got = np.load(tmp_path / ("g_ss_acc_%d_%d.npy" % (pipeline, rank)))
            assert np.array_equal(got[..., :3].view(np.uint32), merged[..., :3].view(np.uint32))   # a + b, the same on every rank
            out = np.load(tmp_path / ("g_ss_out_%d_%d.npy" % (pipeline, rank)))
            want = np.sqrt(np.maximum(got[..., :3] * np.float32(1.0 / 4), 0), dtype=np.float32)
            assert np.array_equal(out[..., :3].view(np.uint32), want.view(np.uint32))
            assert np.array_equal(np.load(tmp_path / ("g_ts_acc_%d_%d.npy" % (pipeline, rank))).view(np.uint32), seq_acc.view(np.uint32))
            assert np.array_equal(np.load(tmp_path / ("g_ts_out_%d_%d.npy" % (pipeline, rank))).view(np.uint32), seq_out.view(np.uint32))


def _rccl_worker(rank, world, port, out_dir):
    """One rank, backend nccl (= RCCL): the all-reduce of a group of one changes no data, but it goes
    through RCCL on the target's stream like the all-reduce of an 8-GPU run."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    import raytracer_amd as R
    Wg, Hg = 160, 96
    s = R.Scene(); s.init_default()
    r = R.Renderer(0)
    r.upload_scene(s.buffers())
    pc = R.default_push_constants(); R.camera_to_push_constants(R.default_camera(), Wg, Hg, pc); pc.parentBVHCount = 8
    tgt = D.GpuTarget(r, pc, Wg, Hg, "cuda:0")
    for k in range(3):   # back to back without host synchronisation: clear / kernels / collective / resolve must stay in order
        D.render_sample_split(tgt, frames_per_rank=3, always_reduce=True)
    torch.cuda.synchronize()
    np.save(os.path.join(out_dir, "n_ss_acc.npy"), tgt.accum.cpu().numpy())
    np.save(os.path.join(out_dir, "n_ss_out.npy"), tgt.draw.cpu().numpy())
    D.render_tile_split(tgt, frames=3, tile_size=32, always_reduce=True)
    torch.cuda.synchronize()
    np.save(os.path.join(out_dir, "n_ts_acc.npy"), tgt.accum.cpu().numpy())
    # the same through the library's own communicator (rt_comm_init / rt_merge_accum / rt_gather_tiles)
    tgt.init_comm()
    assert tgt.native_comm and r.comm_info() == (0, 1)
    for k in range(2):
        D.render_sample_split(tgt, frames_per_rank=3, always_reduce=True)
    torch.cuda.synchronize()
    np.save(os.path.join(out_dir, "c_ss_acc.npy"), tgt.accum.cpu().numpy())
    np.save(os.path.join(out_dir, "c_ss_out.npy"), tgt.draw.cpu().numpy())
    D.render_tile_split(tgt, frames=3, tile_size=32, always_reduce=True)
    torch.cuda.synchronize()
    np.save(os.path.join(out_dir, "c_ts_acc.npy"), tgt.accum.cpu().numpy())
    np.save(os.path.join(out_dir, "c_ts_out.npy"), tgt.draw.cpu().numpy())
    tgt.close()
    assert r.comm_info() == (0, 0)
    r.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_rccl_all_reduce_on_the_target_stream(tmp_path):
    import raytracer_amd as R
    Wg, Hg = 160, 96
    mp.spawn(_rccl_worker, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)
    s = R.Scene(); s.init_default()
    r = R.Renderer(0)
    try:
        r.upload_scene(s.buffers()); r.resize(Wg, Hg)
        pc = R.default_push_constants(); R.camera_to_push_constants(R.default_camera(), Wg, Hg, pc); pc.parentBVHCount = 8
        r.render_frames(pc, 0, 1, 3)
        acc, out = r.read_image(R.RT_IMAGE_ACCUM), r.read_image(R.RT_IMAGE_DRAW)
    finally:
        r.close()
    assert np.array_equal(np.load(tmp_path / "n_ss_acc.npy").view(np.uint32), acc.view(np.uint32))
    assert np.array_equal(np.load(tmp_path / "n_ss_out.npy")[..., :3].view(np.uint32), out[..., :3].view(np.uint32))
    assert np.array_equal(np.load(tmp_path / "n_ts_acc.npy").view(np.uint32), acc.view(np.uint32))
    assert np.array_equal(np.load(tmp_path / "c_ss_acc.npy").view(np.uint32), acc.view(np.uint32))
    assert np.array_equal(np.load(tmp_path / "c_ss_out.npy")[..., :3].view(np.uint32), out[..., :3].view(np.uint32))
    assert np.array_equal(np.load(tmp_path / "c_ts_acc.npy").view(np.uint32), acc.view(np.uint32))
    assert np.array_equal(np.load(tmp_path / "c_ts_out.npy").view(np.uint32), out.view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("world,size,tile", [(2, (200, 120), 32), (3, (333, 187), 64), (8, (520, 200), 16)])
def test_local_merges_between_contexts(world, size, tile):
    """rt_gather_tiles_local / rt_merge_accum_local: `world` contexts on one GPU play the ranks. The tile
    exchange (pack kernel, the copies an all-gather performs, unpack kernel - the kernels rt_gather_tiles
    uses) must complete every context's images to the single-context render bit for bit; the sample merge
    must give the rank-ordered sum on every context."""
    import raytracer_amd as R
    Wg, Hg = size
    s = R.Scene(); s.init_default()
    pc = R.default_push_constants(); R.camera_to_push_constants(R.default_camera(), Wg, Hg, pc); pc.parentBVHCount = 8
    rs = [R.Renderer(0) for _ in range(world)]
    try:
        for r in rs:
            r.upload_scene(s.buffers()); r.resize(Wg, Hg)
        ref = R.Renderer(0)
        ref.upload_scene(s.buffers()); ref.resize(Wg, Hg)
        ref.render_frames(pc, 0, 1, 3)
        want_acc, want_out = ref.read_image(R.RT_IMAGE_ACCUM), ref.read_image(R.RT_IMAGE_DRAW)
        for rank, r in enumerate(rs):
            r.set_option(R.RT_OPT_TILE_SIZE, tile); r.set_option(R.RT_OPT_TILE_WORLD, world); r.set_option(R.RT_OPT_TILE_RANK, rank)
            r.render_frames(pc, 0, 1, 3)
        owner = D.tile_owner_map(world, Wg, Hg, tile)
        before = rs[1].read_image(R.RT_IMAGE_ACCUM)
        assert np.array_equal(before[owner == 1].view(np.uint32), want_acc[owner == 1].view(np.uint32))
        assert not before[owner != 1].any()
        R.gather_tiles_local(rs)
        for r in rs:
            assert np.array_equal(r.read_image(R.RT_IMAGE_ACCUM).view(np.uint32), want_acc.view(np.uint32))
            assert np.array_equal(r.read_image(R.RT_IMAGE_DRAW).view(np.uint32), want_out.view(np.uint32))
        # sample split: rank r renders frames r, r + world, ...; merged = ((a0 + a1) + a2) + ...
        parts = []
        for rank, r in enumerate(rs):
            r.set_option(R.RT_OPT_TILE_WORLD, 1); r.set_option(R.RT_OPT_TILE_RANK, 0)
            r.clear_accumulation()
            r.render_frames(pc, rank, world, 2)
            parts.append(r.read_image(R.RT_IMAGE_ACCUM))
        R.merge_accum_local(rs)
        total = parts[0]
        for p in parts[1:]:
            total = total + p
        for r in rs:
            assert np.array_equal(r.read_image(R.RT_IMAGE_ACCUM).view(np.uint32), total.view(np.uint32))
        ref.clear_accumulation()
        ref.render_frames(pc, 0, 1, 2 * world)
        np.testing.assert_allclose(total[..., :3], ref.read_image(R.RT_IMAGE_ACCUM)[..., :3], rtol=1e-5, atol=1e-6)
        # mismatched partition options are refused
        rs[0].set_option(R.RT_OPT_TILE_WORLD, 5)
        with pytest.raises(R.RtError):
            R.gather_tiles_local(rs)
        ref.close()
    finally:
        for r in rs:
            r.close()
