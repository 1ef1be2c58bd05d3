"""Multi-GPU drivers: one process per GPU. The reference renderer is single-GPU; these two splits are
how its progressive accumulation (HardwareRenderer::ProduceRender, HardwareRenderer.cpp:1030-1061:
frames 0..N-1 added into one accumulation image) is spread over the GPUs of a node.

sample split  rank r renders frames r, r+W, r+2W, ... of the full image into its own accumulation
              image; ONE all-reduce(sum) of that W*H*4-float image merges them; the draw image is then
              resolved from the merged sum. Differs from the sequential render only by f32
              re-association of the per-pixel sum of frames.
tile split    the image is cut into square tiles, numbered along a serpentine path and dealt
              round-robin to the ranks; every rank renders all frames for its tiles only; ONE all-gather
              in which each rank sends its own tiles of both images completes the images on every rank:
              bit-identical to single-GPU.

Where the merge runs:
  * a GpuTarget whose library communicator is set up (GpuTarget.init_comm, the default of bench.py on
    RCCL): inside librt_amd.so - rt_merge_accum / rt_gather_tiles call RCCL on the stream the kernels
    run on; torch.distributed only carried the 128-byte communicator id;
  * otherwise torch.distributed: with backend "nccl" (= RCCL) on the device tensors; with any other
    backend (gloo rehearsals) through a pinned host copy - the target's stream is synchronised, the
    host buffer is reduced / gathered, the result is copied back.

Both drivers take a `target`: any object with
    accum, draw             torch tensors (H, W, 4) float32 the renderer writes into
    clear()                 zero both images
    render_frames(first_frame, stride, count)
    set_tiles(rank, world, tile_size)
    resolve(total_frames)   draw = gamma(accum / total_frames)
and optionally stream_context() (a context manager under which torch work is ordered with the target's
kernels), sync() (wait for the target's device work), merge_accum() / gather_tiles() (library-side
merges, used when `native_comm` is true). GpuTarget below is the real one (HIP kernels through the
C ABI); tests substitute a CPU one.
"""
import contextlib

import torch
import torch.distributed as dist

from . import api


def frames_of_rank(rank, world, frames_per_rank):
    """Frame indices rank `rank` renders in a sample split."""
    return [rank + k * world for k in range(frames_per_rank)]


def tile_of_position(p, tiles_x):
    """(tx, ty) of the tile with serpentine number p: row 0 runs left to right, row 1 right to left, ..."""
    row, col = divmod(p, tiles_x)
    return (tiles_x - 1 - col if row & 1 else col), row


def tiles_of_rank(rank, world, width, height, tile_size):
    """(tx, ty) of the tiles owned by `rank` in a tile split: serpentine numbers p with p % world == rank."""
    tx = (width + tile_size - 1) // tile_size
    ty = (height + tile_size - 1) // tile_size
    return [tile_of_position(p, tx) for p in range(tx * ty) if p % world == rank]


def tile_owner_map(world, width, height, tile_size):
    """(H, W) int array: the rank that owns each pixel in a tile split."""
    import numpy as np
    tx = (width + tile_size - 1) // tile_size
    ty = (height + tile_size - 1) // tile_size
    owner = np.zeros((ty, tx), np.int32)
    for p in range(tx * ty):
        x, y = tile_of_position(p, tx)
        owner[y, x] = p % world
    return np.repeat(np.repeat(owner, tile_size, axis=0), tile_size, axis=1)[:height, :width]


def _world(group):
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def _ordered(target):
    return getattr(target, "stream_context", contextlib.nullcontext)()


def _sync(target):
    fn = getattr(target, "sync", None)
    if fn is not None:
        fn()


def _device_collectives(tensor, group):
    """True when torch.distributed may be handed the tensor as it is: CPU tensors always, device tensors
    only on the RCCL backend. Other backends get a pinned host copy (see the module text)."""
    return (not tensor.is_cuda) or dist.get_backend(group) == "nccl"


def _all_reduce_sum(target, tensor, group):
    if _device_collectives(tensor, group):
        dist.all_reduce(tensor, op=dist.ReduceOp.SUM, group=group)
        return
    _sync(target)
    host = torch.empty(tensor.shape, dtype=tensor.dtype, pin_memory=True)
    host.copy_(tensor)
    torch.cuda.synchronize(tensor.device)
    dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
    tensor.copy_(host)
    torch.cuda.synchronize(tensor.device)


def _all_gather(target, send, world, group):
    """-> list of `world` tensors like `send` (on send's device)."""
    if _device_collectives(send, group):
        parts = [torch.empty_like(send) for _ in range(world)]
        dist.all_gather(parts, send, group=group)
        return parts
    _sync(target)
    host = torch.empty(send.shape, dtype=send.dtype, pin_memory=True)
    host.copy_(send)
    torch.cuda.synchronize(send.device)
    parts = [torch.empty_like(host) for _ in range(world)]
    dist.all_gather(parts, host, group=group)
    return [p.to(send.device) for p in parts]


def _gather_tiles_torch(target, rank, world, tile_size, group):
    """The tile exchange of rt_gather_tiles through torch.distributed: each rank sends the tiles it owns
    (both images, padded to whole tiles and to an equal tile count per rank) in ONE all-gather."""
    accum, draw = target.accum, target.draw
    H, W = accum.shape[0], accum.shape[1]
    tiles_x = (W + tile_size - 1) // tile_size
    tiles = tiles_x * ((H + tile_size - 1) // tile_size)
    per_rank = (tiles + world - 1) // world
    send = torch.zeros((per_rank, 2, tile_size, tile_size, 4), dtype=accum.dtype, device=accum.device)

    def span(p):
        tx, ty = tile_of_position(p, tiles_x)
        x0, y0 = tx * tile_size, ty * tile_size
        return x0, y0, min(W, x0 + tile_size) - x0, min(H, y0 + tile_size) - y0

    for k in range(per_rank):
        p = rank + k * world
        if p < tiles:
            x0, y0, w, h = span(p)
            send[k, 0, :h, :w] = accum[y0:y0 + h, x0:x0 + w]
            send[k, 1, :h, :w] = draw[y0:y0 + h, x0:x0 + w]
    parts = _all_gather(target, send, world, group)
    for r in range(world):
        if r == rank:
            continue
        for k in range(per_rank):
            p = r + k * world
            if p < tiles:
                x0, y0, w, h = span(p)
                accum[y0:y0 + h, x0:x0 + w] = parts[r][k, 0, :h, :w]
                draw[y0:y0 + h, x0:x0 + w] = parts[r][k, 1, :h, :w]


def exchange_sample_split(target, group=None):
    """The exchange step of a sample split alone: the accumulation images of all ranks summed in place."""
    if getattr(target, "native_comm", False):
        target.merge_accum()
    else:
        _all_reduce_sum(target, target.accum, group)


def exchange_tile_split(target, tile_size=64, group=None):
    """The exchange step of a tile split alone: every rank's own tiles of both images to every other rank.
    The target's tile options must name this rank's share (render_tile_split sets them)."""
    rank, world = _world(group)
    if getattr(target, "native_comm", False):
        target.gather_tiles()
    else:
        _gather_tiles_torch(target, rank, world, tile_size, group)


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
            exchange_sample_split(target, group)
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
            exchange_tile_split(target, tile_size, group)
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
        self.native_comm = False
        # allocated on the current stream, used on the target's own from here on
        self.draw.record_stream(self.stream)
        self.accum.record_stream(self.stream)
        torch.cuda.synchronize(self.device)
        renderer.bind_images(self.draw.data_ptr(), self.accum.data_ptr(), width, height)
        renderer.set_stream(self.stream.cuda_stream)

    def init_comm(self, group=None):
        """Set up the library's own RCCL communicator for this rank: rank 0 creates the id,
        torch.distributed broadcasts its 128 bytes, every rank calls rt_comm_init. One GPU per rank."""
        rank, world = _world(group)
        on_device = dist.is_available() and dist.is_initialized() and dist.get_backend(group) == "nccl"
        ident = torch.zeros(128, dtype=torch.uint8)
        if rank == 0:
            ident.copy_(torch.frombuffer(bytearray(api.comm_unique_id()), dtype=torch.uint8))
        if world > 1:
            if on_device:
                ident = ident.to(self.device)
            dist.broadcast(ident, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        self.r.comm_init(ident.cpu().numpy().tobytes(), rank, world)
        self.native_comm = True

    def stream_context(self):
        return torch.cuda.stream(self.stream)

    def sync(self):
        self.r.sync()

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

    def merge_accum(self):
        self.r.merge_accum()

    def gather_tiles(self):
        self.r.gather_tiles()

    def close(self):
        torch.cuda.synchronize(self.device)
        if self.native_comm:
            self.r.comm_destroy()
            self.native_comm = False
        self.r.set_stream(None)
        self.r.bind_images(None, None, 0, 0)
