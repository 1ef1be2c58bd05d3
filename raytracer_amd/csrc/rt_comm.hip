// rt_comm.hip - merging the images of several GPUs behind the C ABI.
//
// The reference renders on one GPU; its progressive accumulation (ProduceRender,
// HardwareRenderer.cpp:1030-1061: frames 0..N-1 added into one accumulation image) is spread over the
// GPUs of a node in two ways:
//   sample split  rank r renders frames r, r+W, r+2W, ... of the whole image; rt_merge_accum sums the
//                 accumulation images with ONE all-reduce of W*H*4 floats (RCCL over xGMI), after which
//                 rt_resolve produces the draw image;
//   tile split    rank r renders every frame of the tiles with serpentine number p % W == r
//                 (RT_OPT_TILE_*); rt_gather_tiles packs the owned tiles of both images into the rank's
//                 slice of one exchange buffer, all-gathers the slices in place and unpacks every other
//                 rank's tiles: each rank sends its own 1/W of the two images once, bit for bit.
// One process per GPU: rank 0 obtains an id with rt_comm_unique_id, hands it to the other ranks by any
// channel (a file, MPI, torch.distributed) and every rank calls rt_comm_init. RCCL is loaded with
// dlopen at that moment, so the library itself does not depend on it. All collectives are enqueued on
// the context's stream, behind the render kernels, without host synchronisation.
//
// rt_*_local are the same merges among contexts of ONE process (any mix of devices) by plain device
// copies: a host that drives several GPUs from one thread needs no communicator, and the pack /
// unpack / sum logic can be exercised with several contexts on a single GPU.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <algorithm>
#include <cstring>
#include <mutex>
#include <string>

#include "rt_ctx.h"

namespace {

// ---- the few RCCL entry points used, resolved at run time (rccl.h: ncclGetUniqueId :187,
// ncclCommInitRank :220, ncclCommDestroy :260, ncclAllReduce :611, ncclAllGather :678) ----
struct RcclId { char internal[128]; };
typedef int (*GetUniqueIdFn)(RcclId*);
typedef int (*CommInitRankFn)(void**, int, RcclId, int);
typedef int (*CommDestroyFn)(void*);
typedef int (*AllReduceFn)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*AllGatherFn)(const void*, void*, size_t, int, void*, hipStream_t);
typedef const char* (*GetErrorStringFn)(int);
const int kRcclFloat = 7, kRcclChar = 0, kRcclSum = 0;
// Where the RCCL header is installed, the values and the size of the id are checked against it at compile time (the
// library is still only opened at run time, with dlopen: a single-GPU user never loads it).
#if defined(__has_include)
#if __has_include(<rccl/rccl.h>)
} // namespace
#include <rccl/rccl.h>
namespace {
static_assert(ncclFloat == kRcclFloat && ncclChar == kRcclChar && ncclSum == kRcclSum, "RCCL enum values differ from the ones this file passes");
static_assert(sizeof(ncclUniqueId) == sizeof(RcclId) && NCCL_UNIQUE_ID_BYTES == RT_COMM_ID_BYTES, "RCCL unique id size differs");
#endif
#endif

struct Rccl {
    void* handle = nullptr;
    GetUniqueIdFn getUniqueId = nullptr;
    CommInitRankFn commInitRank = nullptr;
    CommDestroyFn commDestroy = nullptr;
    AllReduceFn allReduce = nullptr;
    AllGatherFn allGather = nullptr;
    GetErrorStringFn getErrorString = nullptr;
    std::string error;
} g_rccl;
std::mutex g_rcclLock;

bool loadRccl()
{
    std::lock_guard<std::mutex> g(g_rcclLock);
    if (g_rccl.handle) return true;
    void* h = nullptr;
    for (const char* name : { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" }) {
        h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    if (!h) { g_rccl.error = std::string("cannot load librccl: ") + dlerror(); return false; }
    g_rccl.getUniqueId = (GetUniqueIdFn)dlsym(h, "ncclGetUniqueId");
    g_rccl.commInitRank = (CommInitRankFn)dlsym(h, "ncclCommInitRank");
    g_rccl.commDestroy = (CommDestroyFn)dlsym(h, "ncclCommDestroy");
    g_rccl.allReduce = (AllReduceFn)dlsym(h, "ncclAllReduce");
    g_rccl.allGather = (AllGatherFn)dlsym(h, "ncclAllGather");
    g_rccl.getErrorString = (GetErrorStringFn)dlsym(h, "ncclGetErrorString");
    if (!g_rccl.getUniqueId || !g_rccl.commInitRank || !g_rccl.commDestroy || !g_rccl.allReduce || !g_rccl.allGather) {
        g_rccl.error = "librccl lacks a required entry point";
        dlclose(h);
        return false;
    }
    g_rccl.handle = h;
    return true;
}

int failRccl(rt_ctx* ctx, int code, const char* what)
{
    const char* text = g_rccl.getErrorString ? g_rccl.getErrorString(code) : "?";
    return rt_fail(ctx, RT_ERR_HIP, std::string(what) + ": RCCL error " + std::to_string(code) + " (" + text + ")");
}

// ---- tile exchange ---------------------------------------------------------------------------
// Exchange buffer: world slices of perRank slots; slot k of slice r holds the tile with serpentine
// number r + k * world: tileSize^2 accumulation pixels followed by tileSize^2 draw pixels (pixels
// outside the image, and slots past the last tile, are never read back).
struct TileGeometry {
    int width, height, tileSize, tilesX, tiles, world, perRank;
    size_t slotPixels;     // tileSize^2
};

TileGeometry geometry(const rt_ctx* c, int world)
{
    TileGeometry g{};
    g.width = c->width; g.height = c->height; g.tileSize = c->tileSize;
    g.tilesX = (c->width + c->tileSize - 1) / c->tileSize;
    g.tiles = g.tilesX * ((c->height + c->tileSize - 1) / c->tileSize);
    g.world = world;
    g.perRank = (g.tiles + world - 1) / world;
    g.slotPixels = (size_t)c->tileSize * (size_t)c->tileSize;
    return g;
}

// One block per (slot, rank) pair: pack copies image -> buffer for rank == own, unpack copies
// buffer -> image for every rank but `skip`.
__global__ __launch_bounds__(256) void rt_tiles_copy(TileGeometry g, float4* __restrict__ accum, float4* __restrict__ draw,
                                                    float4* __restrict__ buffer, int firstRank, int skip, int toBuffer)
{
    const int rank = firstRank + (int)blockIdx.y;
    if (rank == skip) return;
    const int k = (int)blockIdx.x;
    const int p = rank + k * g.world;
    if (p >= g.tiles) return;
    int tx, ty;
    rt_tile_of_position(p, g.tilesX, &tx, &ty);
    float4* slot = buffer + ((size_t)rank * (size_t)g.perRank + (size_t)k) * 2 * g.slotPixels;
    for (int i = (int)threadIdx.x; i < (int)g.slotPixels; i += 256) {
        const int y = ty * g.tileSize + i / g.tileSize, x = tx * g.tileSize + i % g.tileSize;
        if (x >= g.width || y >= g.height) continue;
        const size_t idx = (size_t)y * (size_t)g.width + (size_t)x;
        if (toBuffer) { slot[i] = accum[idx]; slot[g.slotPixels + i] = draw[idx]; }
        else { accum[idx] = slot[i]; draw[idx] = slot[g.slotPixels + i]; }
    }
}

__global__ __launch_bounds__(256) void rt_add_images(float4* __restrict__ dst, const float4* __restrict__ src, size_t pixels)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < pixels; i += stride) {
        float4 a = dst[i];
        const float4 b = src[i];
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        dst[i] = a;
    }
}

int ensureGatherBuffer(rt_ctx* ctx, const TileGeometry& g)
{
    const size_t bytes = (size_t)g.world * (size_t)g.perRank * 2 * g.slotPixels * sizeof(float4);
    if (ctx->dGather && ctx->gatherBytes >= bytes) return RT_OK;
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->dGather) (void)hipFree(ctx->dGather);
    ctx->dGather = nullptr;
    ctx->gatherBytes = 0;
    RT_HIP(ctx, hipMalloc(&ctx->dGather, bytes));
    ctx->gatherBytes = bytes;
    return RT_OK;
}

int checkImages(rt_ctx* ctx, const char* who)
{
    if (!ctx) return rt_fail(nullptr, RT_ERR_INVALID_ARG, std::string(who) + ": null context");
    if (!ctx->draw || !ctx->accum) return rt_fail(ctx, RT_ERR_NO_IMAGES, "no images (call rt_resize or rt_bind_images first)");
    return RT_OK;
}

int launchTiles(rt_ctx* ctx, const TileGeometry& g, int firstRank, int ranks, int skip, int toBuffer)
{
    if (g.perRank <= 0 || ranks <= 0) return RT_OK;
    hipLaunchKernelGGL(rt_tiles_copy, dim3((unsigned)g.perRank, (unsigned)ranks), dim3(256), 0, ctx->stream, g,
                       static_cast<float4*>(ctx->accum), static_cast<float4*>(ctx->draw), static_cast<float4*>(ctx->dGather), firstRank,
                       skip, toBuffer);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return rt_fail_hip(ctx, e, "tile copy launch");
    return RT_OK;
}

} // namespace

extern "C" {

int rt_comm_unique_id(unsigned char id[RT_COMM_ID_BYTES]) try
{
    if (!id) return rt_fail(nullptr, RT_ERR_INVALID_ARG, "rt_comm_unique_id: null pointer");
    if (!loadRccl()) return rt_fail(nullptr, RT_ERR_UNSUPPORTED, g_rccl.error);
    RcclId u;
    std::memset(&u, 0, sizeof(u));
    const int rc = g_rccl.getUniqueId(&u);
    if (rc != 0) return failRccl(nullptr, rc, "ncclGetUniqueId");
    static_assert(sizeof(u) == RT_COMM_ID_BYTES, "id size");
    std::memcpy(id, &u, sizeof(u));
    return RT_OK;
} catch (...) { return rt_caught(nullptr); }

int rt_comm_init(rt_ctx* ctx, const unsigned char id[RT_COMM_ID_BYTES], int rank, int world) try
{
    if (!ctx || !id) return rt_fail(ctx, RT_ERR_INVALID_ARG, "rt_comm_init: null argument");
    if (world < 1 || rank < 0 || rank >= world) return rt_fail(ctx, RT_ERR_INVALID_ARG, "rt_comm_init: rank must be in [0, world)");
    if (ctx->comm) return rt_fail(ctx, RT_ERR_INVALID_ARG, "rt_comm_init: this context already has a communicator (rt_comm_destroy first)");
    if (!loadRccl()) return rt_fail(ctx, RT_ERR_UNSUPPORTED, g_rccl.error);
    RT_HIP(ctx, hipSetDevice(ctx->device));
    RcclId u;
    std::memcpy(&u, id, sizeof(u));
    void* comm = nullptr;
    const int rc = g_rccl.commInitRank(&comm, world, u, rank);
    if (rc != 0) return failRccl(ctx, rc, "ncclCommInitRank");
    ctx->comm = comm;
    ctx->commRank = rank;
    ctx->commWorld = world;
    return RT_OK;
} catch (...) { return rt_caught(ctx); }

int rt_comm_destroy(rt_ctx* ctx) try
{
    if (!ctx) return rt_fail(nullptr, RT_ERR_INVALID_ARG, "null context");
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->comm && g_rccl.commDestroy) (void)g_rccl.commDestroy(ctx->comm);
    ctx->comm = nullptr;
    ctx->commRank = 0;
    ctx->commWorld = 1;
    if (ctx->dGather) (void)hipFree(ctx->dGather);
    ctx->dGather = nullptr;
    ctx->gatherBytes = 0;
    return RT_OK;
} catch (...) { return rt_caught(ctx); }

int rt_comm_info(rt_ctx* ctx, int* rank, int* world) try
{
    if (!ctx) return rt_fail(nullptr, RT_ERR_INVALID_ARG, "null context");
    if (rank) *rank = ctx->comm ? ctx->commRank : 0;
    if (world) *world = ctx->comm ? ctx->commWorld : 0;
    return RT_OK;
} catch (...) { return rt_caught(ctx); }

int rt_merge_accum(rt_ctx* ctx) try
{
    int rc = checkImages(ctx, "rt_merge_accum");
    if (rc != RT_OK) return rc;
    if (!ctx->comm) return rt_fail(ctx, RT_ERR_INVALID_ARG, "rt_merge_accum: no communicator (call rt_comm_init first)");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    const size_t floats = (size_t)ctx->width * (size_t)ctx->height * 4;
    const int r = g_rccl.allReduce(ctx->accum, ctx->accum, floats, kRcclFloat, kRcclSum, ctx->comm, ctx->stream);
    if (r != 0) return failRccl(ctx, r, "ncclAllReduce");
    return RT_OK;
} catch (...) { return rt_caught(ctx); }

int rt_gather_tiles(rt_ctx* ctx) try
{
    int rc = checkImages(ctx, "rt_gather_tiles");
    if (rc != RT_OK) return rc;
    if (!ctx->comm) return rt_fail(ctx, RT_ERR_INVALID_ARG, "rt_gather_tiles: no communicator (call rt_comm_init first)");
    if (ctx->tileWorld != ctx->commWorld || ctx->tileRank != ctx->commRank)
        return rt_fail(ctx, RT_ERR_INVALID_ARG, "rt_gather_tiles: RT_OPT_TILE_RANK / RT_OPT_TILE_WORLD must equal the communicator's rank and size");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    const TileGeometry g = geometry(ctx, ctx->commWorld);
    if ((rc = ensureGatherBuffer(ctx, g)) != RT_OK) return rc;
    if ((rc = launchTiles(ctx, g, ctx->commRank, 1, -1, 1)) != RT_OK) return rc;
    const size_t sliceBytes = (size_t)g.perRank * 2 * g.slotPixels * sizeof(float4);
    char* base = static_cast<char*>(ctx->dGather);
    // in place: this rank's slice already sits at its position in the receive buffer
    const int r = g_rccl.allGather(base + (size_t)ctx->commRank * sliceBytes, base, sliceBytes, kRcclChar, ctx->comm, ctx->stream);
    if (r != 0) return failRccl(ctx, r, "ncclAllGather");
    return launchTiles(ctx, g, 0, ctx->commWorld, ctx->commRank, 0);
} catch (...) { return rt_caught(ctx); }

int rt_merge_accum_local(rt_ctx** ctxs, int n) try
{
    if (!ctxs || n < 1) return rt_fail(nullptr, RT_ERR_INVALID_ARG, "rt_merge_accum_local: need at least one context");
    for (int i = 0; i < n; ++i) {
        int rc = checkImages(ctxs[i], "rt_merge_accum_local");
        if (rc != RT_OK) return rc;
        if (ctxs[i]->width != ctxs[0]->width || ctxs[i]->height != ctxs[0]->height)
            return rt_fail(ctxs[i], RT_ERR_INVALID_ARG, "rt_merge_accum_local: the contexts' images differ in size");
    }
    if (n == 1) return RT_OK;
    rt_ctx* root = ctxs[0];
    const size_t pixels = (size_t)root->width * (size_t)root->height, bytes = pixels * sizeof(float4);
    RT_HIP(root, hipSetDevice(root->device));
    void* staging = nullptr;
    RT_HIP(root, hipMalloc(&staging, bytes));
    int rc = RT_OK;
    // sum into context 0 in context order (the order an in-order reduction over ranks 0..n-1 uses), then copy back
    for (int i = 1; i < n && rc == RT_OK; ++i) {
        hipError_t e = hipSetDevice(ctxs[i]->device);
        if (e == hipSuccess) e = hipStreamSynchronize(ctxs[i]->stream);
        if (e == hipSuccess) e = hipSetDevice(root->device);
        if (e == hipSuccess) e = hipMemcpyAsync(staging, ctxs[i]->accum, bytes, hipMemcpyDefault, root->stream);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(rt_add_images, dim3(2048), dim3(256), 0, root->stream, static_cast<float4*>(root->accum),
                               static_cast<const float4*>(staging), pixels);
            e = hipGetLastError();
        }
        if (e != hipSuccess) rc = rt_fail_hip(root, e, "rt_merge_accum_local");
    }
    hipError_t e = hipStreamSynchronize(root->stream);
    if (rc == RT_OK && e != hipSuccess) rc = rt_fail_hip(root, e, "rt_merge_accum_local");
    for (int i = 1; i < n && rc == RT_OK; ++i) {
        e = hipMemcpy(ctxs[i]->accum, root->accum, bytes, hipMemcpyDefault);
        if (e != hipSuccess) rc = rt_fail_hip(ctxs[i], e, "rt_merge_accum_local copy back");
    }
    (void)hipFree(staging);
    return rc;
} catch (...) { return rt_caught(nullptr); }

int rt_gather_tiles_local(rt_ctx** ctxs, int n) try
{
    if (!ctxs || n < 1) return rt_fail(nullptr, RT_ERR_INVALID_ARG, "rt_gather_tiles_local: need at least one context");
    for (int i = 0; i < n; ++i) {
        int rc = checkImages(ctxs[i], "rt_gather_tiles_local");
        if (rc != RT_OK) return rc;
        rt_ctx* c = ctxs[i];
        if (c->width != ctxs[0]->width || c->height != ctxs[0]->height || c->tileSize != ctxs[0]->tileSize)
            return rt_fail(c, RT_ERR_INVALID_ARG, "rt_gather_tiles_local: the contexts differ in image or tile size");
        if (c->tileWorld != n || c->tileRank != i)
            return rt_fail(c, RT_ERR_INVALID_ARG, "rt_gather_tiles_local: context i must have RT_OPT_TILE_RANK = i and RT_OPT_TILE_WORLD = n");
    }
    const TileGeometry g = geometry(ctxs[0], n);
    const size_t sliceBytes = (size_t)g.perRank * 2 * g.slotPixels * sizeof(float4);
    int rc;
    for (int i = 0; i < n; ++i) {
        rt_ctx* c = ctxs[i];
        RT_HIP(c, hipSetDevice(c->device));
        if ((rc = ensureGatherBuffer(c, g)) != RT_OK) return rc;
        if ((rc = launchTiles(c, g, i, 1, -1, 1)) != RT_OK) return rc;
        RT_HIP(c, hipStreamSynchronize(c->stream));
    }
    // what the all-gather does: every context receives every other context's slice at the same offset
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            if (i == j) continue;
            RT_HIP(ctxs[i], hipMemcpy(static_cast<char*>(ctxs[i]->dGather) + (size_t)j * sliceBytes,
                                      static_cast<char*>(ctxs[j]->dGather) + (size_t)j * sliceBytes, sliceBytes, hipMemcpyDefault));
        }
    for (int i = 0; i < n; ++i) {
        rt_ctx* c = ctxs[i];
        RT_HIP(c, hipSetDevice(c->device));
        if ((rc = launchTiles(c, g, 0, n, i, 0)) != RT_OK) return rc;
        RT_HIP(c, hipStreamSynchronize(c->stream));
    }
    return RT_OK;
} catch (...) { return rt_caught(nullptr); }

} // extern "C"
