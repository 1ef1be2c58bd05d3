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
        tx = (W + ts - 1) // ts
        for k in range(count):
            self.opc["frame"] = first + k * stride
            acc = self.accum.numpy()
            for t in D.tiles_of_rank(rank, world, W, H, ts):
                x0, y0 = (t % tx) * ts, (t // tx) * ts
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
        assert tiles == list(range(7 * 4))


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
