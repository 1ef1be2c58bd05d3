"""Multi-GPU drivers: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on
ROCm; "gloo" for CPU rehearsals). The reference renderer is single-GPU; these two splits are how
its progressive accumulation (HardwareRenderer::ProduceRender, HardwareRenderer.cpp:1030-1061:
frames 0..N-1 added into one accumulation image) is spread over the GPUs of a node.

sample split  rank r renders frames r, r+W, r+2W, ... of the full image into its own
              accumulation image; ONE all-reduce(sum) of that W*H*4-float image merges them; the
              draw image is then resolved from the merged sum. Differs from the sequential
              render only by f32 re-association of the per-pixel sum of frames.
tile split    the image is cut into square tiles dealt round-robin to the ranks; every rank
              renders all frames for its tiles only (other pixels stay zero); the images are
              merged by an all-reduce(sum) with zeros, which is bit-identical to single-GPU.

Both take a `target`: any object with
    accum, draw             torch tensors (H, W, 4) float32 the renderer writes into
    clear()                 zero both images
    render_frames(first_frame, stride, count)
    set_tiles(rank, world, tile_size)
    resolve(total_frames)   draw = gamma(accum / total_frames)
and optionally stream_context(): a context manager under which torch work (the clears, the
collectives) is ordered with the target's kernels. GpuTarget below is the real one (HIP kernels
through the C ABI); tests substitute a CPU one.
"""
import contextlib

import torch
import torch.distributed as dist

from . import api


def frames_of_rank(rank, world, frames_per_rank):
    """Frame indices rank `rank` renders in a sample split."""
    return [rank + k * world for k in range(frames_per_rank)]


def tiles_of_rank(rank, world, width, height, tile_size):
    """Row-major tile numbers owned by `rank` in a tile split."""
    tx = (width + tile_size - 1) // tile_size
    ty = (height + tile_size - 1) // tile_size
    return [t for t in range(tx * ty) if t % world == rank]


def _world(group):
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def _ordered(target):
    return getattr(target, "stream_context", contextlib.nullcontext)()


def render_sample_split(target, frames_per_rank, group=None, always_reduce=False):
    """Every rank ends with the merged accumulation image (frames_per_rank * world samples per
    pixel) and the resolved draw image. Returns the total number of frames. always_reduce: issue
    the all-reduce even in a group of one (a no-op on the data; lets one GPU exercise the RCCL path)."""
    rank, world = _world(group)
    with _ordered(target):
        target.set_tiles(0, 1, 64)
        target.clear()
        target.render_frames(rank, world, frames_per_rank)
        if world > 1 or always_reduce:
            dist.all_reduce(target.accum, op=dist.ReduceOp.SUM, group=group)
        total = frames_per_rank * world
        target.resolve(total)
    return total


def render_tile_split(target, frames, tile_size=64, group=None, always_reduce=False):
    """Every rank renders `frames` accumulated frames of its own tiles; afterwards every rank
    holds the complete accumulation and draw images."""
    rank, world = _world(group)
    with _ordered(target):
        target.set_tiles(rank, world, tile_size)
        target.clear()
        target.render_frames(0, 1, frames)
        if world > 1 or always_reduce:
            dist.all_reduce(target.accum, op=dist.ReduceOp.SUM, group=group)
            dist.all_reduce(target.draw, op=dist.ReduceOp.SUM, group=group)
        target.set_tiles(0, 1, tile_size)
    return frames


class GpuTarget:
    """A Renderer drawing into torch tensors on its GPU. The kernels run on a torch stream the
    target owns; torch work issued under stream_context() (the drivers above do that) goes to the
    same stream, so clears, kernels and collectives are ordered without host synchronisation.
    (rt_set_stream(NULL) means "the context's own stream", so torch's default stream, whose handle
    is 0, cannot be handed to the library: hence a stream of the target's own.)"""

    def __init__(self, renderer, push_constants, width, height, device):
        self.r = renderer
        self.pc = push_constants
        self.device = torch.device(device)
        self.stream = torch.cuda.Stream(self.device)
        self.draw = torch.zeros((height, width, 4), dtype=torch.float32, device=self.device)
        self.accum = torch.zeros((height, width, 4), dtype=torch.float32, device=self.device)
        torch.cuda.synchronize(self.device)
        renderer.bind_images(self.draw.data_ptr(), self.accum.data_ptr(), width, height)
        renderer.set_stream(self.stream.cuda_stream)

    def stream_context(self):
        return torch.cuda.stream(self.stream)

    def clear(self):
        with torch.cuda.stream(self.stream):
            self.draw.zero_()
            self.accum.zero_()

    def render_frames(self, first_frame, stride, count):
        self.r.render_frames(self.pc, first_frame, stride, count)

    def set_tiles(self, rank, world, tile_size):
        self.r.set_option(api.RT_OPT_TILE_SIZE, tile_size)
        self.r.set_option(api.RT_OPT_TILE_WORLD, world)
        self.r.set_option(api.RT_OPT_TILE_RANK, rank)

    def resolve(self, total_frames):
        self.r.resolve(total_frames)

    def close(self):
        torch.cuda.synchronize(self.device)
        self.r.set_stream(None)
        self.r.bind_images(None, None, 0, 0)
