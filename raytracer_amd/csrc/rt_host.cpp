// rt_host.cpp - the C-ABI device layer of librt_amd.so: what replaces the
// reference's Vulkan plumbing around its ray-tracing dispatch
// (src/Renderers/Hardware/HardwareRenderer.cpp: BufferSceneData :651-771,
// swapchain image creation :131-158, RefreshAccumulation :840-848,
// DispatchRayTracingCommands :793-838, WriteDrawImageToFile :1063-1104).
//
// It validates the 18 reference-layout arrays, re-lays them out for the GPU
// (rt_device.h), keeps them resident in HBM, and launches the gfx950 kernels of
// rt_kernels.hip on one HIP stream per context. There is no CPU rendering path:
// without a HIP device every entry point that needs one fails with
// RT_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <new>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/rt_abi.h"
#include "rt_device.h"
#include "rt_ctx.h"

void rt_mat4_inverse(const float in[16], float out[16]);   // rt_scene.cpp
int32_t rt_build_sah(const DevTri* src, size_t count, int maxLeaf, std::vector<DevNode>& nodes, std::vector<DevTri>& tris,
                     float boundsLo[3], float boundsHi[3], int& depth);   // rt_bvh.cpp
void rt_triangle_bounds(const DevTri* src, size_t count, float boundsLo[3], float boundsHi[3]);   // rt_bvh.cpp
int32_t rt_build_tlas(const float* boxes, size_t count, bool balanced, std::vector<DevNode4>& out, int& need);   // rt_bvh.cpp
int32_t rt_collapse4(const std::vector<DevNode>& nodes2, int32_t root, std::vector<DevNode4>& out, int& need);   // rt_bvh.cpp
int32_t rt_collapse4_ordered(const std::vector<DevNode>& nodes2, int32_t root, std::vector<DevNode4>& out, int& need);   // rt_bvh.cpp

extern "C" {
hipError_t rt_launch_megakernel_strict(const DevScene*, const rt_push_constants*, const DevLaunch*, void*, void*,
                                       DevCounters*, int, int, hipStream_t);
hipError_t rt_launch_megakernel_fast(const DevScene*, const rt_push_constants*, const DevLaunch*, void*, void*,
                                     DevCounters*, int, int, hipStream_t);
hipError_t rt_launch_pairkernel_strict(const DevScene*, const rt_push_constants*, const DevLaunch*, void*, void*, int, void*, hipStream_t);
hipError_t rt_launch_pairkernel_fast(const DevScene*, const rt_push_constants*, const DevLaunch*, void*, void*, int, void*, hipStream_t);
hipError_t rt_launch_schedkernel_strict(const DevScene*, const rt_push_constants*, const DevLaunch*, void*, void*,
                                        DevCounters*, int, int, unsigned long long*, int, hipStream_t, unsigned long long*);
hipError_t rt_launch_schedkernel_fast(const DevScene*, const rt_push_constants*, const DevLaunch*, void*, void*,
                                      DevCounters*, int, int, unsigned long long*, int, hipStream_t, unsigned long long*);
hipError_t rt_launch_wf_path_strict(const DevScene*, const rt_push_constants*, const DevLaunch*, const DevWavefront*, void*, void*,
                                    DevCounters*, int, int, int, int, int, hipStream_t);
hipError_t rt_launch_wf_path_fast(const DevScene*, const rt_push_constants*, const DevLaunch*, const DevWavefront*, void*, void*,
                                  DevCounters*, int, int, int, int, int, hipStream_t);
hipError_t rt_launch_wf_walk_strict(const DevScene*, const rt_push_constants*, const DevWavefront*, int, int, int, hipStream_t);
hipError_t rt_launch_wf_walk_fast(const DevScene*, const rt_push_constants*, const DevWavefront*, int, int, int, hipStream_t);
hipError_t rt_launch_teamkernel_strict(const DevScene*, const rt_push_constants*, const DevLaunch*, void*, void*, unsigned long long*, int, int, int, hipStream_t);
hipError_t rt_launch_teamkernel_fast(const DevScene*, const rt_push_constants*, const DevLaunch*, void*, void*, unsigned long long*, int, int, int, hipStream_t);
size_t rt_team_lds_bytes(int stackDepth, int walkers);
hipError_t rt_launch_math_strict(int, const float*, float*, size_t, hipStream_t);
hipError_t rt_launch_math_fast(int, const float*, float*, size_t, hipStream_t);
hipError_t rt_launch_resolve_strict(const void*, void*, size_t, float, hipStream_t);
hipError_t rt_launch_resolve_fast(const void*, void*, size_t, float, hipStream_t);
// rt_noalloc.hip: per-wave counter slots in a file-scope device array
hipError_t rt_noalloc_setup(int device);
void* rt_noalloc_scratch(int device);
long long rt_noalloc_max_waves(void);
hipError_t rt_noalloc_sum(const void* partials, int n, void* out, hipStream_t stream);
}


thread_local std::string g_rtError;

int rt_fail(rt_ctx* ctx, int code, const std::string& msg)
{
    if (ctx) ctx->error = msg;
    g_rtError = msg;
    return code;
}

int rt_fail_hip(rt_ctx* ctx, hipError_t e, const char* what)
{
    return rt_fail(ctx, RT_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}

// Handler of every exported function's function-try-block: no C++ exception crosses the C ABI. The
// messages are short enough for the string's in-place buffer, so recording them cannot throw again.
int rt_caught(rt_ctx* ctx) noexcept
{
    int code = RT_ERR_INVALID_ARG;
    const char* text = "internal error";
    try { throw; }
    catch (const std::bad_alloc&) { code = RT_ERR_NOMEM; text = "out of memory"; }
    catch (const std::length_error&) { code = RT_ERR_NOMEM; text = "too large"; }
    catch (...) { }
    try { if (ctx) ctx->error.assign(text); g_rtError.assign(text); } catch (...) { }
    return code;
}

const char* rt_global_error() { return g_rtError.c_str(); }

namespace {


void freeFastScene(rt_ctx* c)
{
    c->hFastObjects.clear();
    for (void** p : { &c->dFastObjects, &c->dFastNodes, &c->dFastTris, &c->dFastAnyOrder, &c->dWideObjects, &c->dWideNodes, &c->dWideAnyOrder }) {
        if (*p) (void)hipFree(*p);
        *p = nullptr;
    }
    c->haveFast = false;
    c->haveWide = false;
    c->wideNodeCount = 0;
    c->hWideObjects.clear();
    c->fastObjects = 0;
    c->fastNodeCount = c->fastTriCount = 0;
}

void freeScene(rt_ctx* c)
{
    freeFastScene(c);
    c->haveRefWide = false;
    c->refWideNodeCount = 0;
    for (void** p : { &c->dObjects, &c->dNodes, &c->dTris, &c->dNrms, &c->dMaterials, &c->dAnyOrder, &c->dRefWideObjects, &c->dRefWideNodes }) {
        if (*p) (void)hipFree(*p);
        *p = nullptr;
    }
    c->haveScene = false;
    c->sceneBytes = 0;
}

void freeWavefront(rt_ctx* c)
{
    for (int k = 0; k < rt_ctx::kMaxParts; ++k) {
        if (c->wfMem[k]) (void)hipFree(c->wfMem[k]);
        c->wfMem[k] = nullptr;
        c->wf[k] = DevWavefront{};
    }
    for (size_t& x : c->wfKey) x = 0;
}

void freeImages(rt_ctx* c)
{
    if (c->ownDraw) (void)hipFree(c->ownDraw);
    if (c->ownAccum) (void)hipFree(c->ownAccum);
    c->ownDraw = c->ownAccum = nullptr;
    if (c->dPartial) (void)hipFree(c->dPartial);
    c->dPartial = nullptr;
    c->partialBytes = 0;
}

struct Float4 { float x, y, z, w; };

// Host-side builder of the device layout.
struct Relayout {
    const rt_scene_buffers& in;
    const rt_parent_bvh_node* parents;
    const rt_gpu_object* objects;
    const Float4 *v0, *v1, *v2;
    const Float4 *bmin, *bmax;
    const int32_t *left, *right, *triStart, *triCount;
    size_t nodeCount, triCountTotal;

    std::vector<DevObject> dObjects;
    std::vector<DevNode> dNodes;
    std::vector<DevTri> dTris;
    std::map<std::pair<int, int>, int32_t> leafMemo;    // (triStart, count) -> ref
    std::map<std::pair<int, int>, int32_t> nodeMemo;    // (first child, second child) -> DevNode index
    std::vector<int> nodeHeight;                        // per DevNode; -1 while being built
    int maxDepth = 0;
    std::string error;

    explicit Relayout(const rt_scene_buffers& b) : in(b)
    {
        parents = static_cast<const rt_parent_bvh_node*>(b.parentBVH.data);
        objects = static_cast<const rt_gpu_object*>(b.objects.data);
        v0 = static_cast<const Float4*>(b.triangleV0s.data);
        v1 = static_cast<const Float4*>(b.triangleV1s.data);
        v2 = static_cast<const Float4*>(b.triangleV2s.data);
        bmin = static_cast<const Float4*>(b.aabbMins.data);
        bmax = static_cast<const Float4*>(b.aabbMaxs.data);
        left = static_cast<const int32_t*>(b.bvhNodeLeftChildren.data);
        right = static_cast<const int32_t*>(b.bvhNodeRightChildren.data);
        triStart = static_cast<const int32_t*>(b.bvhNodeTriangleStartIndices.data);
        triCount = static_cast<const int32_t*>(b.bvhNodeTriangleCounts.data);
        nodeCount = b.aabbMins.count;
        triCountTotal = b.triangleV0s.count;
    }

    bool leafRef(int start, int count, int32_t& ref)
    {
        if (count <= 0) { ref = RT_REF_EMPTY; return true; }
        if (start < 0 || (size_t)start + (size_t)count > triCountTotal) {
            error = "leaf triangle range [" + std::to_string(start) + ", +" + std::to_string(count) + ") is outside the triangle arrays";
            return false;
        }
        auto key = std::make_pair(start, count);
        auto it = leafMemo.find(key);
        if (it != leafMemo.end()) { ref = it->second; return true; }
        const size_t base = dTris.size();
        if (base + (size_t)count >= 0x7ffffff0u) { error = "too many triangles"; return false; }
        for (int j = 0; j < count; ++j) {
            const Float4 a = v0[start + j], b = v1[start + j], c = v2[start + j];
            DevTri t{};
            t.v0[0] = a.x; t.v0[1] = a.y; t.v0[2] = a.z;
            t.e1[0] = b.x - a.x; t.e1[1] = b.y - a.y; t.e1[2] = b.z - a.z;   // raytrace.comp:288
            t.e2[0] = c.x - a.x; t.e2[1] = c.y - a.y; t.e2[2] = c.z - a.z;   // raytrace.comp:289
            t.remaining = count - j;
            t.orig = start + j;
            dTris.push_back(t);
        }
        ref = ~(int32_t)base;
        leafMemo[key] = ref;
        return true;
    }

    // Reference for BVH node `n` as seen from its parent record. `height` receives
    // the number of two-child records on the longest path below (0 for a leaf).
    bool childRef(int n, int depth, int32_t& ref, int& height)
    {
        height = 0;
        if (n == -1) { ref = RT_REF_ABSENT; return true; }
        if (n < 0 || (size_t)n >= nodeCount) { error = "child index " + std::to_string(n) + " is outside the node arrays"; return false; }
        const int l = left[n], r = right[n];
        if (l == -1 && r == -1) return leafRef(triStart[n], triCount[n], ref);
        return innerRef(l, r, depth, ref, height);   // the shader pushes right then left: left is popped first
    }

    // Two-child record whose `first` child is visited first. `depth` counts the
    // records on the path from the object root down to and including this one;
    // it bounds the traversal stack a lane can need.
    bool innerRef(int first, int second, int depth, int32_t& ref, int& height)
    {
        const char* tooDeep = "BVH deeper than 31 levels: the shader's int stack[32] would overflow";
        if (depth > RT_MAX_STACK - 1) { error = tooDeep; return false; }
        auto key = std::make_pair(first, second);
        auto it = nodeMemo.find(key);
        if (it != nodeMemo.end()) {
            if (nodeHeight[it->second] < 0) { error = "the BVH child indices form a cycle"; return false; }
            ref = it->second;
            height = nodeHeight[ref];
            if (depth - 1 + height > RT_MAX_STACK - 1) { error = tooDeep; return false; }
            maxDepth = std::max(maxDepth, depth - 1 + height);
            return true;
        }
        const int32_t idx = (int32_t)dNodes.size();
        dNodes.push_back(DevNode{});
        nodeHeight.push_back(-1);   // in progress
        nodeMemo[key] = idx;
        maxDepth = std::max(maxDepth, depth);
        int32_t fr, sr;
        int hf, hs;
        if (!childRef(first, depth + 1, fr, hf)) return false;
        if (!childRef(second, depth + 1, sr, hs)) return false;
        DevNode n{};
        auto box = [&](int node, int c) {
            if (node == -1) return;   // stays all zero
            n.lo[0][c] = bmin[node].x; n.lo[1][c] = bmin[node].y; n.lo[2][c] = bmin[node].z;
            n.hi[0][c] = bmax[node].x; n.hi[1][c] = bmax[node].y; n.hi[2][c] = bmax[node].z;
        };
        box(first, 0);
        box(second, 1);
        // a box with min > max on an axis can never be entered (raytrace.comp:221,239): keep it
        // counted but dead, so the kernels may take the slab interval by min/max of the planes
        auto inverted = [&](int c) { return n.lo[0][c] > n.hi[0][c] || n.lo[1][c] > n.hi[1][c] || n.lo[2][c] > n.hi[2][c]; };
        if (fr != RT_REF_ABSENT && inverted(0)) fr = RT_REF_DEAD;
        if (sr != RT_REF_ABSENT && inverted(1)) sr = RT_REF_DEAD;
        n.first = fr;
        n.second = sr;
        dNodes[idx] = n;
        height = 1 + std::max(hf, hs);
        nodeHeight[idx] = height;
        ref = idx;
        return true;
    }

    bool run()
    {
        const size_t P = in.parentBVH.count;
        dObjects.resize(P);
        for (size_t i = 0; i < P; ++i) {
            const rt_parent_bvh_node& p = parents[i];
            if (p.objectIndex < 0 || (size_t)p.objectIndex >= in.objects.count) {
                error = "parentBVH[" + std::to_string(i) + "].objectIndex out of range";
                return false;
            }
            const rt_gpu_object& o = objects[p.objectIndex];
            if (o.materialIndex < 0 || (size_t)o.materialIndex >= in.materials.count) {
                error = "object " + std::to_string(p.objectIndex) + " has material index " + std::to_string(o.materialIndex) +
                        " outside the material array";
                return false;
            }
            DevObject d{};
            for (int a = 0; a < 3; ++a) { d.bmin[a] = p.node.aabb.min[a]; d.bmax[a] = p.node.aabb.max[a]; }
            d.materialIndex = o.materialIndex;
            float fwd[16];
            rt_mat4_inverse(o.inverseTransform, fwd);   // raytrace.comp:485, hoisted
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 4; ++c) {
                    d.inv[4 * r + c] = o.inverseTransform[4 * c + r];
                    d.fwd[4 * r + c] = fwd[4 * c + r];
                }
            if (p.node.leftChild == -1 && p.node.rightChild == -1) {
                if (!leafRef(p.node.triangleStartIndex, p.node.triangleCount, d.root)) return false;
            } else {
                // raytrace.comp:525-526 pushes left then right, so the RIGHT child is popped first
                int h;
                if (!innerRef(p.node.rightChild, p.node.leftChild, 1, d.root, h)) return false;
            }
            if (d.bmin[0] > d.bmax[0] || d.bmin[1] > d.bmax[1] || d.bmin[2] > d.bmax[2]) d.root = RT_REF_DEAD;
            dObjects[i] = d;
        }
        return true;
    }
};

int uploadArray(rt_ctx* ctx, void** dst, const void* src, size_t bytes)
{
    *dst = nullptr;
    const size_t alloc = std::max<size_t>(bytes, 256) + 64;   // the walk kernel fetches 64 bytes at every record's address
    RT_HIP(ctx, hipMalloc(dst, alloc));
    RT_HIP(ctx, hipMemsetAsync(*dst, 0, alloc, ctx->stream));
    if (bytes) RT_HIP(ctx, hipMemcpyAsync(*dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    return RT_OK;
}

// Order in which "any hit" rays (shadow rays) visit the objects: leaf parents (a handful of triangles, no
// walk) before BVH parents; inside each group the parents whose box centre lies furthest towards the sun
// come first (towardsSun = -sunDirection; null or a zero vector: scene order) - what blocks a shadow ray lies
// between its origin and the sun, and a wave is done with its shadow rays as soon as every one of them is
// blocked. The order changes no result (a shadow ray only reports whether anything is hit).
// Layout: anyOrder[P], pad, nextAny[P + 1], runLen[P], runLenAny[P] (see DevScene). sortBvhGroup = false keeps the
// BVH parents in their stored order (the wide scene stores them in the order its box hierarchy needs).
std::vector<int32_t> orderTables(const std::vector<DevObject>& objects, int32_t& anyFirst, int32_t& leafCount, const float* towardsSun = nullptr,
                                 bool sortBvhGroup = true)
{
    const int32_t P = (int32_t)objects.size();
    std::vector<int32_t> anyOrder, tables(4 * (size_t)P + 2, P);
    for (int pass = 0; pass < 2; ++pass)
        for (int32_t i = 0; i < P; ++i)
            if ((objects[i].root >= 0) == (pass == 1)) anyOrder.push_back(i);
    leafCount = 0;
    for (int32_t i = 0; i < P; ++i) leafCount += objects[i].root < 0 ? 1 : 0;
    if (towardsSun && (towardsSun[0] != 0.0f || towardsSun[1] != 0.0f || towardsSun[2] != 0.0f) &&
        std::isfinite(towardsSun[0]) && std::isfinite(towardsSun[1]) && std::isfinite(towardsSun[2])) {
        std::vector<float> score((size_t)P);
        for (int32_t i = 0; i < P; ++i) {
            const DevObject& o = objects[i];
            const float v = 0.5f * ((o.bmin[0] + o.bmax[0]) * towardsSun[0] + (o.bmin[1] + o.bmax[1]) * towardsSun[1] +
                                    (o.bmin[2] + o.bmax[2]) * towardsSun[2]);
            score[(size_t)i] = std::isfinite(v) ? v : -INFINITY;
        }
        auto nearerTheSun = [&](int32_t x, int32_t y) { return score[(size_t)x] > score[(size_t)y]; };
        std::stable_sort(anyOrder.begin(), anyOrder.begin() + leafCount, nearerTheSun);
        if (sortBvhGroup) std::stable_sort(anyOrder.begin() + leafCount, anyOrder.end(), nearerTheSun);
    }
    for (int32_t p = 0; p < P; ++p) {
        tables[p] = anyOrder[p];                                         // anyOrder
        tables[P + 1 + anyOrder[p]] = p + 1 < P ? anyOrder[p + 1] : P;   // nextAny
    }
    // runs of BVH parents, counted from the back: in scene order and in the any-hit order
    for (int pass = 0; pass < 2; ++pass) {
        int32_t* run = tables.data() + 2 * (size_t)P + 2 + (size_t)pass * (size_t)P;
        int32_t following = 0;
        for (int32_t p = P - 1; p >= 0; --p) {
            const int32_t obj = pass ? anyOrder[p] : p;
            following = objects[(size_t)obj].root >= 0 ? following + 1 : 0;
            run[p] = std::min(following, 64);
        }
    }
    anyFirst = P > 0 ? anyOrder[0] : 0;
    return tables;
}

// Re-sorts the any-hit order of the uploaded scene (and of its near-first twin, if built) for a new sun
// direction. The tables keep their size, so they are overwritten in place, on the context's stream, behind
// the launches that still read the old order.
int refreshAnyOrder(rt_ctx* ctx, const float sunDirection[3])
{
    if (!ctx->haveScene) return RT_OK;
    if (std::memcmp(ctx->anyOrderSun, sunDirection, sizeof(ctx->anyOrderSun)) == 0) return RT_OK;
    const float towards[3] = { -sunDirection[0], -sunDirection[1], -sunDirection[2] };
    for (int which = 0; which < 3; ++which) {
        const std::vector<DevObject>& objects = which == 2 ? ctx->hWideObjects : which ? ctx->hFastObjects : ctx->hObjects;
        void* dst = which == 2 ? ctx->dWideAnyOrder : which ? ctx->dFastAnyOrder : ctx->dAnyOrder;
        DevScene& scene = which == 2 ? ctx->wideScene : which ? ctx->fastScene : ctx->scene;
        if (!dst || objects.empty()) continue;
        int32_t anyFirst = 0, leafCount = 0;
        const std::vector<int32_t> tables = orderTables(objects, anyFirst, leafCount, towards, which != 2);
        RT_HIP(ctx, hipMemcpyAsync(dst, tables.data(), tables.size() * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
        RT_HIP(ctx, hipStreamSynchronize(ctx->stream));     // the staging vector dies here
        scene.anyFirst = anyFirst;
    }
    std::memcpy(ctx->anyOrderSun, sunDirection, sizeof(ctx->anyOrderSun));
    return RT_OK;
}

// Largest singular value of the 3x3 linear part of a row-major 3x4 matrix: the square root of the
// largest eigenvalue of A^T A (closed form for a symmetric 3x3 matrix).
double largestSingularValue(const float m[12])
{
    double a[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double v = 0.0;
            for (int k = 0; k < 3; ++k) v += (double)m[4 * k + i] * (double)m[4 * k + j];
            a[i][j] = v;
        }
    const double p1 = a[0][1] * a[0][1] + a[0][2] * a[0][2] + a[1][2] * a[1][2];
    const double q = (a[0][0] + a[1][1] + a[2][2]) / 3.0;
    if (p1 <= 1e-300) return std::sqrt(std::max(0.0, std::max(a[0][0], std::max(a[1][1], a[2][2]))));
    const double p2 = (a[0][0] - q) * (a[0][0] - q) + (a[1][1] - q) * (a[1][1] - q) + (a[2][2] - q) * (a[2][2] - q) + 2.0 * p1;
    const double p = std::sqrt(p2 / 6.0);
    double b[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) b[i][j] = (a[i][j] - (i == j ? q : 0.0)) / p;
    const double det = b[0][0] * (b[1][1] * b[2][2] - b[1][2] * b[2][1]) - b[0][1] * (b[1][0] * b[2][2] - b[1][2] * b[2][0]) +
                       b[0][2] * (b[1][0] * b[2][1] - b[1][1] * b[2][0]);
    const double r = std::max(-1.0, std::min(1.0, det / 2.0));
    const double eig = q + 2.0 * p * std::cos(std::acos(r) / 3.0);
    return std::sqrt(std::max(0.0, eig));
}

// Triangle records an object's reference tree can reach, in the reference's visiting order. A child
// whose box is inverted is marked dead at upload; one whose box has no thickness on some axis can
// not be entered either (IntersectAABB fails on `t.max <= t.min`, raytrace.comp:239), so neither
// contributes triangles.
void gatherTriangles(const rt_ctx* c, int32_t root, std::vector<DevTri>& out)
{
    std::vector<int32_t> stack{ root };
    while (!stack.empty()) {
        const int32_t ref = stack.back();
        stack.pop_back();
        if (!(ref > RT_REF_LIVE_MIN)) continue;
        if (ref < 0) {
            for (size_t k = (size_t)~ref; k < c->hTris.size(); ++k) {
                out.push_back(c->hTris[k]);
                if (c->hTris[k].remaining <= 1) break;
            }
            continue;
        }
        const DevNode& n = c->hNodes[(size_t)ref];
        auto enterable = [&](int ch) { return n.lo[0][ch] < n.hi[0][ch] && n.lo[1][ch] < n.hi[1][ch] && n.lo[2][ch] < n.hi[2][ch]; };
        if (enterable(1)) stack.push_back(n.second);
        if (enterable(0)) stack.push_back(n.first);
    }
}

// Builds and uploads the near-first scene (see rt_ctx) from the host copies of the uploaded scene.
int ensureFastScene(rt_ctx* ctx)
{
    if (ctx->haveFast) return RT_OK;
    const size_t P = ctx->hObjects.size();
    std::vector<DevObject> objects = ctx->hObjects;
    // an object may walk a rebuilt tree when its transform shrinks nothing: every singular value of the
    // model matrix >= 1, i.e. the largest singular value of inverseTransform <= 1 (rt_bvh.cpp)
    std::vector<char> eligible(P, 0);
    bool keepReference = false;
    for (size_t i = 0; i < P; ++i) {
        eligible[i] = largestSingularValue(objects[i].inv) <= 1.0 + 1e-5;
        if (!eligible[i] && objects[i].root >= 0) keepReference = true;
    }
    std::vector<DevNode> nodes;
    std::vector<DevTri> tris;
    // every object keeps its leaf runs in the reference arrays unless it gets a rebuilt tree, so the
    // reference triangle records always come along; the reference two-child records only when needed
    tris = ctx->hTris;
    if (keepReference) nodes = ctx->hNodes;
    const int32_t nearFirstBase = (int32_t)nodes.size();
    struct Built { int32_t root; float lo[3], hi[3]; int depth; };
    // (reference root, rebuilt?) -> tree and tight bounds; instances of one model share an entry
    std::map<std::pair<int32_t, bool>, Built> memo;
    int depth = keepReference ? ctx->hStackDepth : 1;
    int rebuilt = 0;
    for (size_t i = 0; i < P; ++i) {
        DevObject& o = objects[i];
        if (!(o.root > RT_REF_LIVE_MIN)) continue;
        const bool rebuild = o.root >= 0 && eligible[i];
        auto key = std::make_pair(o.root, rebuild);
        auto it = memo.find(key);
        if (it == memo.end()) {
            std::vector<DevTri> src;
            gatherTriangles(ctx, o.root, src);
            Built b{};
            b.root = o.root;      // keeps its reference tree (or is a single leaf): only the tight bounds are taken
            if (src.empty()) b.root = RT_REF_EMPTY;
            else if (rebuild) b.root = rt_build_sah(src.data(), src.size(), ctx->sahMaxLeaf, nodes, tris, b.lo, b.hi, b.depth);
            else rt_triangle_bounds(src.data(), src.size(), b.lo, b.hi);
            it = memo.emplace(key, b).first;
        }
        const Built& b = it->second;
        if (rebuild && b.root != RT_REF_EMPTY) { rebuilt++; depth = std::max(depth, b.depth); }
        o.root = b.root;
        if (b.root == RT_REF_EMPTY) { o.root = RT_REF_EMPTY; continue; }
        // tight world box: the 8 corners of the model's tight bounds through the model matrix, padded
        float lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY };
        for (int corner = 0; corner < 8; ++corner) {
            const float x = (corner & 1) ? b.hi[0] : b.lo[0], y = (corner & 2) ? b.hi[1] : b.lo[1], z = (corner & 4) ? b.hi[2] : b.lo[2];
            for (int r = 0; r < 3; ++r) {
                const float w = (o.fwd[4 * r + 0] * x + o.fwd[4 * r + 1] * y) + (o.fwd[4 * r + 2] * z + o.fwd[4 * r + 3]);
                lo[r] = std::min(lo[r], w); hi[r] = std::max(hi[r], w);
            }
        }
        for (int r = 0; r < 3; ++r) {
            const float pad = std::max(std::max(std::fabs(lo[r]), std::fabs(hi[r])) * 1e-5f, 1e-6f);
            // never larger than the reference's own box
            o.bmin[r] = std::max(o.bmin[r], lo[r] - pad);
            o.bmax[r] = std::min(o.bmax[r], hi[r] + pad);
        }
        if (o.bmin[0] > o.bmax[0] || o.bmin[1] > o.bmax[1] || o.bmin[2] > o.bmax[2]) o.root = RT_REF_DEAD;
    }
    if (depth > RT_MAX_STACK) return rt_fail(ctx, RT_ERR_UNSUPPORTED, "rebuilt tree deeper than the traversal stack");

    RT_HIP(ctx, hipSetDevice(ctx->device));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    freeFastScene(ctx);
    int32_t anyFirst = 0, leafCount = 0;
    const std::vector<int32_t> tables = orderTables(objects, anyFirst, leafCount);
    int rc;
    if ((rc = uploadArray(ctx, &ctx->dFastObjects, objects.data(), objects.size() * sizeof(DevObject))) != RT_OK) return rc;
    if ((rc = uploadArray(ctx, &ctx->dFastNodes, nodes.data(), nodes.size() * sizeof(DevNode))) != RT_OK) return rc;
    if ((rc = uploadArray(ctx, &ctx->dFastTris, tris.data(), tris.size() * sizeof(DevTri))) != RT_OK) return rc;
    if ((rc = uploadArray(ctx, &ctx->dFastAnyOrder, tables.data(), tables.size() * sizeof(int32_t))) != RT_OK) return rc;
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));   // host staging vectors die at return
    ctx->hFastObjects = objects;
    std::memset(ctx->anyOrderSun, 0xff, sizeof(ctx->anyOrderSun));     // both tables are sorted again at the next launch
    DevScene f = ctx->scene;
    f.objects = static_cast<const DevObject*>(ctx->dFastObjects);
    f.nodes = static_cast<const DevNode*>(ctx->dFastNodes);
    f.tris = static_cast<const DevTri*>(ctx->dFastTris);
    f.anyOrder = static_cast<const int32_t*>(ctx->dFastAnyOrder);
    f.nextAny = static_cast<const int32_t*>(ctx->dFastAnyOrder) + P + 1;
    f.runLen = static_cast<const int32_t*>(ctx->dFastAnyOrder) + 2 * P + 2;
    f.runLenAny = static_cast<const int32_t*>(ctx->dFastAnyOrder) + 3 * P + 2;
    f.anyFirst = anyFirst;
    f.anyLeafCount = leafCount;
    f.stackDepth = std::max(1, std::min(RT_MAX_STACK, depth));
    f.nearFirstBase = nearFirstBase;
    f.orderFree = 1;
    for (size_t i = 0; i < P; ++i)
        if (objects[i].root > RT_REF_LIVE_MIN && !eligible[i]) f.orderFree = 0;
    ctx->fastScene = f;
    // the wide twin: every rebuilt tree folded into four-child records (only when no object keeps the
    // reference's order, so that the kernel that walks them needs no second kind of record)
    if (f.orderFree && rebuilt > 0) {
        std::vector<DevNode4> nodes4;
        std::map<int32_t, std::pair<int32_t, int>> folded;      // two-child root -> (four-child root, stack need)
        int need = 1;
        std::vector<int32_t> leafParents, bvhParents;
        for (size_t i = 0; i < P; ++i) (objects[i].root >= 0 ? bvhParents : leafParents).push_back((int32_t)i);
        // BVH parents along a space-filling curve (Morton code of the world box centre, 10 bits per axis), so that parents
        // that follow one another are neighbours in space
        {
            float lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY };
            for (int32_t i : bvhParents)
                for (int a = 0; a < 3; ++a) {
                    const float c = 0.5f * (objects[(size_t)i].bmin[a] + objects[(size_t)i].bmax[a]);
                    lo[a] = std::min(lo[a], c); hi[a] = std::max(hi[a], c);
                }
            auto morton = [&](int32_t i) {
                uint32_t code = 0;
                for (int a = 0; a < 3; ++a) {
                    const float c = 0.5f * (objects[(size_t)i].bmin[a] + objects[(size_t)i].bmax[a]);
                    const float ext = hi[a] - lo[a];
                    uint32_t q = ext > 0.0f ? (uint32_t)std::min(1023.0f, std::max(0.0f, (c - lo[a]) / ext * 1023.0f)) : 0u;
                    for (int bit = 0; bit < 10; ++bit) code |= ((q >> bit) & 1u) << (3 * bit + a);
                }
                return code;
            };
            std::stable_sort(bvhParents.begin(), bvhParents.end(), [&](int32_t x, int32_t y) { return morton(x) < morton(y); });
        }
        std::vector<DevObject> wide;
        for (int32_t i : leafParents) wide.push_back(objects[(size_t)i]);
        for (int32_t i : bvhParents) wide.push_back(objects[(size_t)i]);
        for (size_t i = leafParents.size(); i < P; ++i) {
            auto it = folded.find(wide[i].root);
            if (it == folded.end()) {
                int n = 0;
                const int32_t r4 = rt_collapse4(nodes, wide[i].root, nodes4, n);
                it = folded.emplace(wide[i].root, std::make_pair(r4, n)).first;
            }
            wide[i].root = it->second.first;
            need = std::max(need, it->second.second);
        }
        const size_t B = bvhParents.size();
        // the hierarchy over the BVH parents' world boxes, behind the objects' own records
        const int32_t tlasBase = (int32_t)nodes4.size();
        int32_t tlasRoot = RT_REF_EMPTY;
        if (B >= RT_TLAS_MIN_PARENTS && B < (size_t)RT_REF_INSTANCE - P && ctx->optTlas) {
            std::vector<float> boxes(6 * B);
            for (size_t k = 0; k < B; ++k)
                for (int a = 0; a < 3; ++a) {
                    boxes[6 * k + a] = wide[leafParents.size() + k].bmin[a];
                    boxes[6 * k + 3 + a] = wide[leafParents.size() + k].bmax[a];
                }
            for (int attempt = B <= 4096 ? 0 : 1; attempt < 2; ++attempt) {
                nodes4.resize((size_t)tlasBase);
                int above = 0;
                tlasRoot = rt_build_tlas(boxes.data(), B, attempt == 1, nodes4, above);
                if (need + above <= RT_WIDE_LDS_LEVELS + RT_WIDE_DEEP_LEVELS) { need += above; break; }
                tlasRoot = RT_REF_EMPTY;
                nodes4.resize((size_t)tlasBase);
            }
            // instance references count the BVH parents; the objects are stored behind the leaf parents
            if (tlasRoot >= RT_REF_INSTANCE) tlasRoot += (int32_t)leafParents.size();
            for (size_t k = (size_t)tlasBase; k < nodes4.size(); ++k)
                for (int c = 0; c < 4; ++c)
                    if (nodes4[k].ref[c] >= RT_REF_INSTANCE) nodes4[k].ref[c] += (int32_t)leafParents.size();
        }
        if (nodes4.size() < (size_t)RT_REF_INSTANCE && need <= RT_WIDE_LDS_LEVELS + RT_WIDE_DEEP_LEVELS) {
            int32_t wideAnyFirst = 0, wideLeafCount = 0;
            const std::vector<int32_t> wideTables = orderTables(wide, wideAnyFirst, wideLeafCount, nullptr, false);
            if ((rc = uploadArray(ctx, &ctx->dWideObjects, wide.data(), wide.size() * sizeof(DevObject))) != RT_OK) return rc;
            if ((rc = uploadArray(ctx, &ctx->dWideNodes, nodes4.data(), nodes4.size() * sizeof(DevNode4))) != RT_OK) return rc;
            if ((rc = uploadArray(ctx, &ctx->dWideAnyOrder, wideTables.data(), wideTables.size() * sizeof(int32_t))) != RT_OK) return rc;
            RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
            DevScene w = f;
            w.objects = static_cast<const DevObject*>(ctx->dWideObjects);
            w.nodes = nullptr;
            w.nodes4 = static_cast<const DevNode4*>(ctx->dWideNodes);
            w.anyOrder = static_cast<const int32_t*>(ctx->dWideAnyOrder);
            w.nextAny = w.anyOrder + P + 1;
            w.runLen = w.anyOrder + 2 * P + 2;
            w.runLenAny = w.anyOrder + 3 * P + 2;
            w.anyFirst = wideAnyFirst;
            w.anyLeafCount = wideLeafCount;
            w.tlasRoot = tlasRoot;
            w.tlasBase = tlasBase;
            w.nearFirstBase = 0;
            w.wideStackNeed = need;
            // entries a lane can need, one more because the branch-free push writes one slot ahead; how many of them live in
            // LDS is the kernels' business (the launcher clamps to the level count the kernels were compiled with)
            w.stackDepth = need + 1;
            w.wideLdsLevels = 0;
            ctx->wideScene = w;
            ctx->wideNodeCount = nodes4.size();
            ctx->hWideObjects = wide;
            ctx->haveWide = true;
        }
    }
    ctx->fastObjects = rebuilt;
    ctx->fastNodeCount = nodes.size();
    ctx->fastTriCount = tris.size();
    ctx->haveFast = true;
    return RT_OK;
}

DevLaunch makeLaunch(const rt_ctx* c)
{
    DevLaunch L{};
    L.width = c->width;
    L.height = c->height;
    L.tileSize = c->tileSize;
    L.tilesX = (c->width + c->tileSize - 1) / c->tileSize;
    L.tilesY = (c->height + c->tileSize - 1) / c->tileSize;
    L.tileRank = c->tileRank;
    L.tileWorld = c->tileWorld;
    const int total = L.tilesX * L.tilesY;
    L.ownedTiles = total > c->tileRank ? (total - c->tileRank + c->tileWorld - 1) / c->tileWorld : 0;
    L.biasObject = c->policy[0]; L.biasTrav = c->policy[1]; L.biasShade = c->policy[2];
    L.quorumMin = std::max(1, c->policy[3]); L.quorumDiv = std::max(1, c->policy[4]);
    L.frames = 1;
    L.frameStride = 1;
    return L;
}

int checkRenderable(rt_ctx* ctx, const rt_push_constants* pc)
{
    if (!ctx) return rt_fail(nullptr, RT_ERR_INVALID_ARG, "null context");
    if (!pc) return rt_fail(ctx, RT_ERR_INVALID_ARG, "null push constants");
    if (!ctx->haveScene) return rt_fail(ctx, RT_ERR_NO_SCENE, "no scene uploaded (call rt_upload_scene first)");
    if (!ctx->draw || !ctx->accum) return rt_fail(ctx, RT_ERR_NO_IMAGES, "no images (call rt_resize or rt_bind_images first)");
    if (pc->parentBVHCount < 0 || pc->parentBVHCount > ctx->scene.objectCount)
        return rt_fail(ctx, RT_ERR_INVALID_ARG, "parentBVHCount " + std::to_string(pc->parentBVHCount) + " exceeds the " +
                                                 std::to_string(ctx->scene.objectCount) + " uploaded parent nodes");
    if (pc->renderMode == 0 || pc->renderMode == 3 || pc->renderMode == 4) {
        if (pc->raysPerPixel < 0) return rt_fail(ctx, RT_ERR_INVALID_ARG, "raysPerPixel is negative");
        // a launch holds up to RT_OPT_FRAMES_PER_LAUNCH frames of raysPerPixel samples, counted in an int
        if ((long long)pc->raysPerPixel * (long long)std::max(1, ctx->framesPerLaunch) > 0x7fffffffLL)
            return rt_fail(ctx, RT_ERR_INVALID_ARG, "raysPerPixel x frames per launch exceeds 2^31 - 1");
    }
    return RT_OK;
}

DevLaunch partLaunch(const DevLaunch& L, int part, int parts)
{
    DevLaunch P = L;
    P.tileRank = L.tileRank + part * L.tileWorld;
    P.tileWorld = L.tileWorld * parts;
    const int total = L.tilesX * L.tilesY;
    P.ownedTiles = total > P.tileRank ? (total - P.tileRank + P.tileWorld - 1) / P.tileWorld : 0;
    return P;
}

// Per partition one allocation: control words, two job buffers, the deep stack levels of the walk kernel.
int ensureWavefront(rt_ctx* ctx, const DevLaunch& L, int parts)
{
    const size_t key[6] = { (size_t)ctx->width, (size_t)ctx->height, (size_t)parts, (size_t)L.tileSize,
                            (size_t)L.tileWorld * 65536 + (size_t)L.tileRank, (size_t)ctx->wavesPerCU * 64 + (size_t)ctx->wfWalkWaves };
    if (ctx->wfMem[0] && std::memcmp(key, ctx->wfKey, sizeof(key)) == 0) return RT_OK;
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int k = 0; k < rt_ctx::kMaxParts; ++k)
        if (ctx->wfStream[k]) RT_HIP(ctx, hipStreamSynchronize(ctx->wfStream[k]));
    freeWavefront(ctx);
    const size_t cus = (size_t)std::max(1, ctx->computeUnits);
    for (int k = 0; k < parts; ++k) {
        const DevLaunch P = partLaunch(L, k, parts);
        // Upper bound of a round's list. A path is one job at most, so a round has at most `pixels` live
        // entries. A wave fills 256-entry blocks and closes a block when the (at most 64) entries of one
        // append do not fit, leaving fewer than 64 of its 256 entries unused: live entries take at most
        // 256/193 < 4/3 of their number in list space. At the end of the kernel every wave of the grid leaves
        // one partly used block. Unused entries are skipped by the next round and do not carry over.
        const size_t pixels = (size_t)P.ownedTiles * (size_t)L.tileSize * (size_t)L.tileSize;
        const size_t listCap = ((pixels + pixels / 3 + 64 + cus * (size_t)ctx->wavesPerCU * 256 + 63) / 64) * 64 + 64;
        if (listCap > 0x7fffffffULL) return rt_fail(ctx, RT_ERR_UNSUPPORTED, "image too large for the wavefront pipeline");
        const size_t ctlBytes = 4096;
        static_assert((RT_WF_MAX_ROUNDS + 2) * RT_WF_WORDS_PER_ROUND * sizeof(uint32_t) <= 4096, "control block");
        const size_t jobBytes = (size_t)RT_WF_JOB_ARRAYS * listCap * 16;
        // deep traversal-stack levels of the walk kernel: (32 - 12) levels x 64 lanes x 8 bytes per wave
        const size_t deepBytes = cus * (size_t)std::max(1, std::min(32, ctx->wfWalkWaves)) * (RT_MAX_STACK - 12) * 64 * 8;
        RT_HIP(ctx, hipMalloc(&ctx->wfMem[k], ctlBytes + 2 * jobBytes + deepBytes));
        char* base = static_cast<char*>(ctx->wfMem[k]);
        ctx->wf[k].counts = reinterpret_cast<uint32_t*>(base);
        ctx->wf[k].jobs[0] = base + ctlBytes;
        ctx->wf[k].jobs[1] = base + ctlBytes + jobBytes;
        ctx->wf[k].deepStack = base + ctlBytes + 2 * jobBytes;
        ctx->wf[k].listCap = (int32_t)listCap;
        if (parts > 1 && !ctx->wfStream[k]) {
            RT_HIP(ctx, hipStreamCreateWithFlags(&ctx->wfStream[k], hipStreamNonBlocking));
            RT_HIP(ctx, hipEventCreateWithFlags(&ctx->wfJoin[k], hipEventDisableTiming));
        }
    }
    if (parts > 1 && !ctx->wfFork) RT_HIP(ctx, hipEventCreateWithFlags(&ctx->wfFork, hipEventDisableTiming));
    std::memcpy(ctx->wfKey, key, sizeof(key));
    return RT_OK;
}

// fork / join: the partitions' streams wait for the context's stream before the first frame of a
// call and the context's stream waits for them after the last, so that between the two the chains of
// different partitions (and of consecutive frames) overlap freely; they touch disjoint pixels.
int launchWavefront(rt_ctx* ctx, const rt_push_constants& pc, const DevLaunch& L, bool statsOnly, bool countBuild, bool fork, bool join)
{
    const int parts = std::max(1, std::min(rt_ctx::kMaxParts, ctx->wfParts));
    int rc = ensureWavefront(ctx, L, parts);
    if (rc != RT_OK) return rc;
    const int rounds = std::max(0, std::min(RT_WF_MAX_ROUNDS, ctx->wfRounds));
    auto path = ctx->optFast ? rt_launch_wf_path_fast : rt_launch_wf_path_strict;
    auto walk = ctx->optFast ? rt_launch_wf_walk_fast : rt_launch_wf_walk_strict;
    const int resident = std::max(1, ctx->computeUnits) * ctx->wavesPerCU;
    const int walkers = std::max(1, ctx->computeUnits) * std::max(1, std::min(32, ctx->wfWalkWaves));
    DevCounters* cnt = statsOnly ? ctx->dCounters : nullptr;
    if (parts > 1 && fork) RT_HIP(ctx, hipEventRecord(ctx->wfFork, ctx->stream));
    for (int k = 0; k < parts; ++k) {
        const DevLaunch P = partLaunch(L, k, parts);
        hipStream_t stream = parts > 1 ? ctx->wfStream[k] : ctx->stream;
        if (parts > 1 && fork) RT_HIP(ctx, hipStreamWaitEvent(stream, ctx->wfFork, 0));
        DevWavefront W = ctx->wf[k];
        W.refillMin = std::max(1, std::min(64, ctx->wfRefill));
        RT_HIP(ctx, hipMemsetAsync(W.counts, 0, 4096, stream));
        const long long tileWaves = (long long)P.ownedTiles * (P.tileSize / 8) * (P.tileSize / 8);
        if (tileWaves > 0x7fffffffLL) return rt_fail(ctx, RT_ERR_UNSUPPORTED, "too many tiles");
        hipError_t e = path(&ctx->scene, &pc, &P, &W, ctx->accum, ctx->draw, cnt, countBuild ? 1 : 0, statsOnly ? 1 : 0, 0,
                            rounds == 0 ? 1 : 0, (int)std::min<long long>(tileWaves, resident), stream);
        for (int r = 0; r < rounds && e == hipSuccess; ++r) {
            e = walk(&ctx->scene, &pc, &W, countBuild ? 1 : 0, r, walkers, stream);
            if (e != hipSuccess) break;
            e = path(&ctx->scene, &pc, &P, &W, ctx->accum, ctx->draw, cnt, countBuild ? 1 : 0, statsOnly ? 1 : 0, r + 1,
                     r + 1 == rounds ? 1 : 0, resident, stream);
        }
        if (e != hipSuccess) return rt_fail_hip(ctx, e, "kernel launch");
        if (parts > 1 && join) {
            RT_HIP(ctx, hipEventRecord(ctx->wfJoin[k], stream));
            RT_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->wfJoin[k], 0));
        }
    }
    return RT_OK;
}

int launchFrame(rt_ctx* ctx, const rt_push_constants& pc, bool statsOnly, unsigned long long* diag = nullptr, bool first = true, bool last = true,
                int frames = 1, int frameStride = 1)
{
    DevLaunch L = makeLaunch(ctx);
    L.frames = frames;
    L.frameStride = frameStride;
    const bool countBuild = statsOnly || pc.renderMode == 3 || pc.renderMode == 4;
    const bool pathTrace = pc.renderMode == 0 || pc.renderMode == 3 || pc.renderMode == 4;
    hipError_t e;
    // near-first traversal: path tracing on the lock-step kernels only; the counting builds and the debug
    // views always walk the reference's trees in the reference's order
    const DevScene* scene = &ctx->scene;
    // the two-samples-per-lane kernels render this launch (they are the only ones that walk four-child records)
    const bool pairKernels = ctx->optPipeline == 3 && pc.renderMode == 0 && !statsOnly && pc.raysPerPixel >= 1 &&
                             ((long long)frames * pc.raysPerPixel >= 2 || (L.tileSize & 15) == 0);
    if (ctx->optTraversal == 1 && (ctx->optPipeline == 0 || ctx->optPipeline == 3 || ctx->optPipeline == 4) && pc.renderMode == 0 && !statsOnly &&
        pc.parentBVHCount == ctx->scene.objectCount) {
        int rc = ensureFastScene(ctx);
        if (rc != RT_OK) return rc;
        scene = &ctx->fastScene;
        if (ctx->optWide && ctx->haveWide && pairKernels) scene = &ctx->wideScene;
    }
    if (scene == &ctx->scene && ctx->optWide && ctx->haveRefWide && pairKernels) {
        // reference order on pipeline 3: the reference's trees folded into four-child records in its visiting order
        scene = &ctx->refWideScene;
    }
    if (!statsOnly) {
        // shadow rays visit the parents that lie towards the sun first (re-sorted when the sun has moved)
        int rc = refreshAnyOrder(ctx, pc.sunDirection);
        if (rc != RT_OK) return rc;
        ctx->refWideScene.anyFirst = ctx->scene.anyFirst;       // (the any-hit tables are shared with the reference scene and re-sorted with it)
    }
    if (ctx->optPipeline == 2 && pathTrace && pc.raysPerPixel >= 1 && ctx->scene.objectCount < 65535)   // object indices are packed in 16 bits
        return launchWavefront(ctx, pc, L, statsOnly, countBuild, first, last);
    if (ctx->optPipeline == 1 && pathTrace && pc.raysPerPixel >= 1) {
        // persistent grid: enough 64-lane blocks to fill every CU; lanes pull pixels from a counter
        const long long wavesNeeded = ((long long)L.ownedTiles * L.tileSize * L.tileSize + 63) / 64;
        const long long resident = (long long)std::max(1, ctx->computeUnits) * ctx->wavesPerCU;
        const int blocks = (int)std::max<long long>(1, std::min(wavesNeeded, resident));
        auto fn = ctx->optFast ? rt_launch_schedkernel_fast : rt_launch_schedkernel_strict;
        e = fn(&ctx->scene, &pc, &L, ctx->accum, ctx->draw, statsOnly ? ctx->dCounters : nullptr, countBuild ? 1 : 0,
               statsOnly ? 1 : 0, ctx->dQueue, blocks, ctx->stream, diag);
    } else if (ctx->optPipeline == 4 && pc.renderMode == 0 && !statsOnly && pc.raysPerPixel >= 1 &&
               rt_team_lds_bytes(scene->stackDepth, ctx->teamWalkers) <= (size_t)160 * 1024) {
        // (a tree so deep that the walkers' stacks do not fit the workgroup's LDS beside the job slots is rendered
        // by the lock-step kernels below: same images)
        // one workgroup of 16 waves per compute unit; its path waves pull tiles from a counter
        auto fn = ctx->optFast ? rt_launch_teamkernel_fast : rt_launch_teamkernel_strict;
        e = fn(scene, &pc, &L, ctx->accum, ctx->draw, ctx->dQueue, std::max(1, ctx->computeUnits), ctx->teamWalkers, ctx->wfRefill, ctx->stream);
    } else if (pairKernels) {
        // two samples of a pixel per lane; a launch that holds one sample per pixel pairs two pixels instead
        auto fn = ctx->optFast ? rt_launch_pairkernel_fast : rt_launch_pairkernel_strict;
        // launches in pieces (RT_OPT_LAUNCH_PIECES): pieces of two frames, their contributions summed in frame order afterwards
        void* partial = nullptr;
        const long long waves = (long long)L.ownedTiles * (L.tileSize / 8) * (L.tileSize / 8);
        const long long resident = (long long)std::max(1, ctx->computeUnits) * 16;
        const bool inPieces = frames > 2 && (ctx->optPieces == 1 || (ctx->optPieces == 2 && waves < 3 * resident));
        if (inPieces) {
            const size_t need = (size_t)frames * (size_t)ctx->width * (size_t)ctx->height * 16;
            if (need > ctx->partialBytes) {
                RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
                if (ctx->dPartial) (void)hipFree(ctx->dPartial);
                ctx->dPartial = nullptr; ctx->partialBytes = 0;
                RT_HIP(ctx, hipMalloc(&ctx->dPartial, need));
                ctx->partialBytes = need;
            }
            partial = ctx->dPartial;
            L.pieceFrames = 2;
            L.pieces = (frames + 1) / 2;
        }
        e = fn(scene, &pc, &L, ctx->accum, ctx->draw, (long long)frames * pc.raysPerPixel < 2 ? 1 : 0, partial, ctx->stream);
    } else {
        // the depth / box views and the degenerate raysPerPixel <= 0 case stay on the lock-step kernel
        auto fn = ctx->optFast ? rt_launch_megakernel_fast : rt_launch_megakernel_strict;
        e = fn(scene, &pc, &L, ctx->accum, ctx->draw, statsOnly ? ctx->dCounters : nullptr, countBuild ? 1 : 0,
               statsOnly ? 1 : 0, ctx->stream);
    }
    if (e != hipSuccess) return rt_fail_hip(ctx, e, "kernel launch");
    return RT_OK;
}

} // namespace

extern "C" {

const char* rt_version(void) { return "rt_amd 0.1 (gfx950)"; }

const char* rt_last_error(rt_ctx* ctx) { return ctx ? ctx->error.c_str() : rt_global_error(); }

int rt_create(int device_id, rt_ctx** out_ctx) try
{
    if (!out_ctx) return rt_fail(nullptr, RT_ERR_INVALID_ARG, "rt_create: null out pointer");
    *out_ctx = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return rt_fail(nullptr, RT_ERR_NO_DEVICE, "no HIP device available: this library has no CPU path");
    if (device_id < 0 || device_id >= n) return rt_fail(nullptr, RT_ERR_INVALID_ARG, "device ordinal out of range");
    rt_ctx* c = new (std::nothrow) rt_ctx();
    if (!c) return rt_fail(nullptr, RT_ERR_HIP, "out of host memory");
    c->device = device_id;
    if ((e = hipSetDevice(device_id)) != hipSuccess || (e = hipStreamCreateWithFlags(&c->ownStream, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipEventCreate(&c->evStart)) != hipSuccess || (e = hipEventCreate(&c->evStop)) != hipSuccess ||
        (e = hipMalloc((void**)&c->dCounters, sizeof(DevCounters))) != hipSuccess ||
        (e = hipMalloc((void**)&c->dQueue, 256)) != hipSuccess ||
        (e = hipDeviceGetAttribute(&c->computeUnits, hipDeviceAttributeMultiprocessorCount, device_id)) != hipSuccess ||
        (e = rt_noalloc_setup(device_id)) != hipSuccess) {
        int rc = rt_fail_hip(nullptr, e, "rt_create");
        rt_destroy(c);
        return rc;
    }
    c->stream = c->ownStream;
    *out_ctx = c;
    return RT_OK;
} catch (...) { return rt_caught(nullptr); }

void rt_destroy(rt_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    (void)rt_comm_destroy(c);
    freeScene(c);
    freeImages(c);
    for (int k = 0; k < rt_ctx::kMaxParts; ++k) {
        if (c->wfStream[k]) { (void)hipStreamSynchronize(c->wfStream[k]); (void)hipStreamDestroy(c->wfStream[k]); }
        if (c->wfJoin[k]) (void)hipEventDestroy(c->wfJoin[k]);
    }
    if (c->wfFork) (void)hipEventDestroy(c->wfFork);
    freeWavefront(c);
    if (c->dCounters) (void)hipFree(c->dCounters);
    if (c->dQueue) (void)hipFree(c->dQueue);
    if (c->evStart) (void)hipEventDestroy(c->evStart);
    if (c->evStop) (void)hipEventDestroy(c->evStop);
    if (c->ownStream) (void)hipStreamDestroy(c->ownStream);
    delete c;
}

int rt_set_stream(rt_ctx* ctx, void* hip_stream) try
{
    if (!ctx) return rt_fail(nullptr, RT_ERR_INVALID_ARG, "null context");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : ctx->ownStream;
    ctx->timed = false;
    return RT_OK;
} catch (...) { return rt_caught(ctx); }

int rt_set_option(rt_ctx* ctx, int option, int value) try
{
    if (!ctx) return rt_fail(nullptr, RT_ERR_INVALID_ARG, "null context");
    switch (option) {
    case RT_OPT_FAST_MATH: ctx->optFast = value ? 1 : 0; return RT_OK;
    case RT_OPT_PIPELINE:
        if (value < 0 || value > 4) return rt_fail(ctx, RT_ERR_UNSUPPORTED, "pipeline must be 0 (lock-step), 1 (wave-scheduled), 2 (wavefront), 3 (lock-step, two samples per lane) or 4 (path waves + walker waves)");
        ctx->optPipeline = value;
        return RT_OK;
    case RT_OPT_WAVES_PER_CU:
        if (value < 1 || value > 32) return rt_fail(ctx, RT_ERR_INVALID_ARG, "waves per CU must be in [1,32]");
        ctx->wavesPerCU = value;
        return RT_OK;
    case RT_OPT_SCHED_BIAS_OBJECT: case RT_OPT_SCHED_BIAS_TRAV: case RT_OPT_SCHED_BIAS_SHADE:
    case RT_OPT_SCHED_QUORUM_MIN: case RT_OPT_SCHED_QUORUM_DIV:
        if (value < -64 || value > 64) return rt_fail(ctx, RT_ERR_INVALID_ARG, "scheduling parameter out of range");
        ctx->policy[option - RT_OPT_SCHED_BIAS_OBJECT] = value;
        return RT_OK;
    case RT_OPT_WF_ROUNDS:
        if (value < 0 || value > RT_WF_MAX_ROUNDS) return rt_fail(ctx, RT_ERR_INVALID_ARG, "wavefront rounds must be in [0,30]");
        ctx->wfRounds = value;
        return RT_OK;
    case RT_OPT_WF_WALK_WAVES:
        if (value < 1 || value > 32) return rt_fail(ctx, RT_ERR_INVALID_ARG, "walk waves per CU must be in [1,32]");
        ctx->wfWalkWaves = value;
        return RT_OK;
    case RT_OPT_WF_PARTS:
        // a partition's chain of kernels runs on a stream of its own: with the context's stream that is one hardware queue each,
        // and a process has four (GPU_MAX_HW_QUEUES); chains that share a queue wait for one another behind fork / join events
        if (value < 1 || value > 3) return rt_fail(ctx, RT_ERR_INVALID_ARG, "wavefront partitions must be in [1,3] (one hardware queue each beside the context's own; a process has four)");
        ctx->wfParts = value;
        return RT_OK;
    case RT_OPT_FRAMES_PER_LAUNCH:
        if (value < 1 || value > 4096) return rt_fail(ctx, RT_ERR_INVALID_ARG, "frames per launch must be in [1,4096]");
        ctx->framesPerLaunch = value;
        return RT_OK;
    case RT_OPT_LAUNCH_PIECES:
        if (value < 0 || value > 2) return rt_fail(ctx, RT_ERR_INVALID_ARG, "launch pieces must be 0 (never), 1 (always) or 2 (when a launch holds few waves)");
        ctx->optPieces = value;
        return RT_OK;
    case RT_OPT_TEAM_WALKERS:
        if (value < 1 || value > 8) return rt_fail(ctx, RT_ERR_INVALID_ARG, "walker waves per workgroup must be in [1,8]");
        ctx->teamWalkers = value;
        return RT_OK;
    case RT_OPT_TRAVERSAL:
        if (value != 0 && value != 1) return rt_fail(ctx, RT_ERR_INVALID_ARG, "traversal must be 0 (reference order) or 1 (near-first on rebuilt trees)");
        ctx->optTraversal = value;
        return RT_OK;
    case RT_OPT_WIDE_RECORDS: {
        if (value < 0 || value > 2) return rt_fail(ctx, RT_ERR_INVALID_ARG, "wide records must be 0, 1 or 2");
        const int tlas = value == 2 ? 0 : 1;
        if (tlas != ctx->optTlas) {        // the wide scene is laid out again at its next use
            RT_HIP(ctx, hipSetDevice(ctx->device));
            RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
            freeFastScene(ctx);
        }
        ctx->optTlas = tlas;
        ctx->optWide = value ? 1 : 0;
        return RT_OK;
    }
    case RT_OPT_SAH_MAX_LEAF:
        if (value < 1 || value > 64) return rt_fail(ctx, RT_ERR_INVALID_ARG, "triangles per rebuilt leaf must be in [1,64]");
        if (value != ctx->sahMaxLeaf) {
            RT_HIP(ctx, hipSetDevice(ctx->device));
            RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
            freeFastScene(ctx);
        }
        ctx->sahMaxLeaf = value;
        return RT_OK;
    case RT_OPT_WF_REFILL:
        if (value < 1 || value > 64) return rt_fail(ctx, RT_ERR_INVALID_ARG, "wavefront refill threshold must be in [1,64]");
        ctx->wfRefill = value;
        return RT_OK;
    case RT_OPT_TILE_RANK:
        if (value < 0) return rt_fail(ctx, RT_ERR_INVALID_ARG, "tile rank must be >= 0");
        ctx->tileRank = value;
        return RT_OK;
    case RT_OPT_TILE_WORLD:
        if (value < 1) return rt_fail(ctx, RT_ERR_INVALID_ARG, "tile world must be >= 1");
        ctx->tileWorld = value;
        return RT_OK;
    case RT_OPT_TILE_SIZE:
        if (value < 8 || value > 1024 || (value & 7)) return rt_fail(ctx, RT_ERR_INVALID_ARG, "tile size must be a multiple of 8 in [8,1024]");
        ctx->tileSize = value;
        return RT_OK;
    default: return rt_fail(ctx, RT_ERR_INVALID_ARG, "unknown option");
    }
} catch (...) { return rt_caught(ctx); }

int rt_get_option(rt_ctx* ctx, int option, int* value) try
{
    if (!ctx || !value) return rt_fail(ctx, RT_ERR_INVALID_ARG, "null argument");
    switch (option) {
    case RT_OPT_FAST_MATH: *value = ctx->optFast; return RT_OK;
    case RT_OPT_PIPELINE: *value = ctx->optPipeline; return RT_OK;
    case RT_OPT_WAVES_PER_CU: *value = ctx->wavesPerCU; return RT_OK;
    case RT_OPT_WF_ROUNDS: *value = ctx->wfRounds; return RT_OK;
    case RT_OPT_WF_REFILL: *value = ctx->wfRefill; return RT_OK;
    case RT_OPT_WF_WALK_WAVES: *value = ctx->wfWalkWaves; return RT_OK;
    case RT_OPT_WF_PARTS: *value = ctx->wfParts; return RT_OK;
    case RT_OPT_FRAMES_PER_LAUNCH: *value = ctx->framesPerLaunch; return RT_OK;
    case RT_OPT_TEAM_WALKERS: *value = ctx->teamWalkers; return RT_OK;
    case RT_OPT_LAUNCH_PIECES: *value = ctx->optPieces; return RT_OK;
    case RT_OPT_TRAVERSAL: *value = ctx->optTraversal; return RT_OK;
    case RT_OPT_SAH_MAX_LEAF: *value = ctx->sahMaxLeaf; return RT_OK;
    case RT_OPT_WIDE_RECORDS: *value = !ctx->optWide ? 0 : ctx->optTlas ? 1 : 2; return RT_OK;
    case RT_INFO_WIDE_RECORDS: case RT_INFO_WIDE_STACK_NEED: case RT_INFO_TLAS_RECORDS: {
        if (!ctx->haveScene) { *value = 0; return RT_OK; }
        int rc = ensureFastScene(ctx);
        if (rc != RT_OK) return rc;
        *value = !ctx->haveWide ? 0 : option == RT_INFO_WIDE_RECORDS ? (int)std::min<size_t>(ctx->wideNodeCount, 0x7fffffff)
               : option == RT_INFO_TLAS_RECORDS ? (ctx->wideScene.tlasRoot == RT_REF_EMPTY ? 0 : (int)ctx->wideNodeCount - ctx->wideScene.tlasBase)
               : ctx->wideScene.wideStackNeed;
        return RT_OK;
    }
    case RT_INFO_FAST_OBJECTS: case RT_INFO_FAST_RECORDS: case RT_INFO_FAST_TRIANGLES: case RT_INFO_FAST_STACK_DEPTH: {
        if (!ctx->haveScene) { *value = 0; return RT_OK; }
        int rc = ensureFastScene(ctx);
        if (rc != RT_OK) return rc;
        *value = option == RT_INFO_FAST_OBJECTS ? ctx->fastObjects
               : option == RT_INFO_FAST_RECORDS ? (int)std::min<size_t>(ctx->fastNodeCount, 0x7fffffff)
               : option == RT_INFO_FAST_TRIANGLES ? (int)std::min<size_t>(ctx->fastTriCount, 0x7fffffff) : ctx->fastScene.stackDepth;
        return RT_OK;
    }
    case RT_INFO_STACK_DEPTH: *value = ctx->haveScene ? ctx->scene.stackDepth : 0; return RT_OK;
    case RT_INFO_RECORDS: *value = ctx->haveScene ? (int)std::min<size_t>(ctx->nodeCount, 0x7fffffff) : 0; return RT_OK;
    case RT_INFO_TRIANGLES: *value = ctx->haveScene ? (int)std::min<size_t>(ctx->triCount, 0x7fffffff) : 0; return RT_OK;
    case RT_OPT_TILE_RANK: *value = ctx->tileRank; return RT_OK;
    case RT_OPT_TILE_WORLD: *value = ctx->tileWorld; return RT_OK;
    case RT_OPT_TILE_SIZE: *value = ctx->tileSize; return RT_OK;
    default: return rt_fail(ctx, RT_ERR_INVALID_ARG, "unknown option");
    }
} catch (...) { return rt_caught(ctx); }

int rt_upload_scene(rt_ctx* ctx, const rt_scene_buffers* b) try
{
    if (!ctx || !b) return rt_fail(ctx, RT_ERR_INVALID_ARG, "rt_upload_scene: null argument");
    // shape checks that must hold before any array is indexed
    const size_t T = b->triangleV0s.count;
    for (const rt_buffer_view* v : { &b->triangleV1s, &b->triangleV2s, &b->triangleN0s, &b->triangleN1s, &b->triangleN2s })
        if (v->count != T) return rt_fail(ctx, RT_ERR_SCENE_INVALID, "the six triangle arrays must have the same length");
    const size_t N = b->aabbMins.count;
    for (const rt_buffer_view* v : { &b->aabbMaxs, &b->bvhNodeLeftChildren, &b->bvhNodeRightChildren,
                                     &b->bvhNodeTriangleStartIndices, &b->bvhNodeTriangleCounts })
        if (v->count != N) return rt_fail(ctx, RT_ERR_SCENE_INVALID, "the six BVH node arrays must have the same length");
    for (const rt_buffer_view* v : { &b->parentBVH, &b->objects, &b->materials, &b->triangleV0s, &b->triangleV1s, &b->triangleV2s,
                                     &b->triangleN0s, &b->triangleN1s, &b->triangleN2s, &b->aabbMins, &b->aabbMaxs,
                                     &b->bvhNodeLeftChildren, &b->bvhNodeRightChildren, &b->bvhNodeTriangleStartIndices,
                                     &b->bvhNodeTriangleCounts })
        if (v->count != 0 && v->data == nullptr) return rt_fail(ctx, RT_ERR_INVALID_ARG, "a non-empty buffer view has a null pointer");
    if (T >= 0x7ffffff0u || N >= 0x7ffffff0u) return rt_fail(ctx, RT_ERR_SCENE_INVALID, "scene too large for 32-bit indices");

    {
        auto finite = [](const void* p, size_t vec4s) {
            const float* f = static_cast<const float*>(p);
            for (size_t i = 0; i < vec4s; ++i)
                for (int k = 0; k < 3; ++k)
                    if (!std::isfinite(f[4 * i + k])) return false;
            return true;
        };
        if (!finite(b->aabbMins.data, N) || !finite(b->aabbMaxs.data, N))
            return rt_fail(ctx, RT_ERR_SCENE_INVALID, "a BVH node box has a non-finite coordinate");
        const rt_parent_bvh_node* pp = static_cast<const rt_parent_bvh_node*>(b->parentBVH.data);
        for (size_t i = 0; i < b->parentBVH.count; ++i)
            for (int k = 0; k < 3; ++k)
                if (!std::isfinite(pp[i].node.aabb.min[k]) || !std::isfinite(pp[i].node.aabb.max[k]))
                    return rt_fail(ctx, RT_ERR_SCENE_INVALID, "a parent box has a non-finite coordinate");
    }
    Relayout rl(*b);
    if (!rl.run()) return rt_fail(ctx, RT_ERR_SCENE_INVALID, rl.error);

    // normals, verbatim, indexed by original triangle
    std::vector<DevNrm> nrms(T);
    const Float4* n0 = static_cast<const Float4*>(b->triangleN0s.data);
    const Float4* n1 = static_cast<const Float4*>(b->triangleN1s.data);
    const Float4* n2 = static_cast<const Float4*>(b->triangleN2s.data);
    for (size_t i = 0; i < T; ++i) {
        std::memcpy(nrms[i].n0, &n0[i], 16);
        std::memcpy(nrms[i].n1, &n1[i], 16);
        std::memcpy(nrms[i].n2, &n2[i], 16);
    }

    RT_HIP(ctx, hipSetDevice(ctx->device));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    freeScene(ctx);
    int rc;
    if ((rc = uploadArray(ctx, &ctx->dObjects, rl.dObjects.data(), rl.dObjects.size() * sizeof(DevObject))) != RT_OK) return rc;
    if ((rc = uploadArray(ctx, &ctx->dNodes, rl.dNodes.data(), rl.dNodes.size() * sizeof(DevNode))) != RT_OK) return rc;
    if ((rc = uploadArray(ctx, &ctx->dTris, rl.dTris.data(), rl.dTris.size() * sizeof(DevTri))) != RT_OK) return rc;
    if ((rc = uploadArray(ctx, &ctx->dNrms, nrms.data(), nrms.size() * sizeof(DevNrm))) != RT_OK) return rc;
    if ((rc = uploadArray(ctx, &ctx->dMaterials, b->materials.data, b->materials.count * sizeof(DevMaterial))) != RT_OK) return rc;
    int32_t anyFirst = 0, anyLeafCount = 0;
    const int32_t P = (int32_t)rl.dObjects.size();
    const std::vector<int32_t> tables = orderTables(rl.dObjects, anyFirst, anyLeafCount);
    if ((rc = uploadArray(ctx, &ctx->dAnyOrder, tables.data(), tables.size() * sizeof(int32_t))) != RT_OK) return rc;
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));   // host staging vectors die at return

    ctx->scene.objects = static_cast<const DevObject*>(ctx->dObjects);
    ctx->scene.nodes = static_cast<const DevNode*>(ctx->dNodes);
    ctx->scene.tris = static_cast<const DevTri*>(ctx->dTris);
    ctx->scene.nrms = static_cast<const DevNrm*>(ctx->dNrms);
    ctx->scene.materials = static_cast<const DevMaterial*>(ctx->dMaterials);
    ctx->scene.anyOrder = static_cast<const int32_t*>(ctx->dAnyOrder);
    ctx->scene.nextAny = static_cast<const int32_t*>(ctx->dAnyOrder) + P + 1;
    ctx->scene.runLen = static_cast<const int32_t*>(ctx->dAnyOrder) + 2 * (size_t)P + 2;
    ctx->scene.runLenAny = static_cast<const int32_t*>(ctx->dAnyOrder) + 3 * (size_t)P + 2;
    ctx->scene.anyFirst = anyFirst;
    ctx->scene.anyLeafCount = anyLeafCount;
    ctx->scene.objectCount = (int32_t)rl.dObjects.size();
    ctx->scene.stackDepth = std::max(1, std::min(RT_MAX_STACK, rl.maxDepth));
    ctx->scene.nearFirstBase = 0x7fffffff;
    ctx->scene.orderFree = 0;
    ctx->scene.tlasRoot = RT_REF_EMPTY;
    ctx->scene.tlasBase = 0x7fffffff;
    ctx->hStackDepth = ctx->scene.stackDepth;
    ctx->nodeCount = rl.dNodes.size();
    ctx->triCount = rl.dTris.size();
    ctx->sceneBytes = rl.dObjects.size() * sizeof(DevObject) + rl.dNodes.size() * sizeof(DevNode) + rl.dTris.size() * sizeof(DevTri) +
                      nrms.size() * sizeof(DevNrm) + b->materials.count * sizeof(DevMaterial);
    // the four-child twin of the reference-order scene (see rt_ctx): every tree folded once, instances of a model share it
    {
        std::vector<DevNode4> nodes4;
        std::vector<DevObject> objects4 = rl.dObjects;
        std::map<int32_t, std::pair<int32_t, int>> folded;
        int need = 1;
        for (DevObject& o : objects4) {
            if (o.root < 0) continue;
            auto it = folded.find(o.root);
            if (it == folded.end()) {
                int n = 0;
                const int32_t r4 = rt_collapse4_ordered(rl.dNodes, o.root, nodes4, n);
                it = folded.emplace(o.root, std::make_pair(r4, n)).first;
            }
            o.root = it->second.first;
            need = std::max(need, it->second.second);
        }
        if (!nodes4.empty() && nodes4.size() < (size_t)RT_REF_INSTANCE && need < RT_WIDE_LDS_LEVELS + RT_WIDE_DEEP_LEVELS) {
            if ((rc = uploadArray(ctx, &ctx->dRefWideObjects, objects4.data(), objects4.size() * sizeof(DevObject))) != RT_OK) return rc;
            if ((rc = uploadArray(ctx, &ctx->dRefWideNodes, nodes4.data(), nodes4.size() * sizeof(DevNode4))) != RT_OK) return rc;
            RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
            DevScene w = ctx->scene;
            w.objects = static_cast<const DevObject*>(ctx->dRefWideObjects);
            w.nodes = nullptr;
            w.nodes4 = static_cast<const DevNode4*>(ctx->dRefWideNodes);
            w.stackDepth = need + 1;
            w.wideStackNeed = need;
            ctx->refWideScene = w;
            ctx->refWideNodeCount = nodes4.size();
            ctx->haveRefWide = true;
        }
    }
    ctx->hObjects = std::move(rl.dObjects);
    ctx->hFastObjects.clear();
    std::memset(ctx->anyOrderSun, 0xff, sizeof(ctx->anyOrderSun));
    ctx->hNodes = std::move(rl.dNodes);
    ctx->hTris = std::move(rl.dTris);
    ctx->haveScene = true;
    return RT_OK;
} catch (...) { return rt_caught(ctx); }

int rt_resize(rt_ctx* ctx, int width, int height) try
{
    if (!ctx) return rt_fail(nullptr, RT_ERR_INVALID_ARG, "null context");
    if (width <= 0 || height <= 0 || width > 65536 || height > 65536) return rt_fail(ctx, RT_ERR_INVALID_ARG, "image size out of range");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    freeImages(ctx);
    const size_t bytes = (size_t)width * (size_t)height * 16;
    RT_HIP(ctx, hipMalloc(&ctx->ownDraw, bytes));
    RT_HIP(ctx, hipMalloc(&ctx->ownAccum, bytes));
    ctx->draw = ctx->ownDraw;
    ctx->accum = ctx->ownAccum;
    ctx->width = width;
    ctx->height = height;
    RT_HIP(ctx, hipMemsetAsync(ctx->draw, 0, bytes, ctx->stream));
    RT_HIP(ctx, hipMemsetAsync(ctx->accum, 0, bytes, ctx->stream));
    return RT_OK;
} catch (...) { return rt_caught(ctx); }

int rt_bind_images(rt_ctx* ctx, void* draw_dev, void* accum_dev, int width, int height) try
{
    if (!ctx) return rt_fail(nullptr, RT_ERR_INVALID_ARG, "null context");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (!draw_dev && !accum_dev) {
        ctx->draw = ctx->ownDraw;
        ctx->accum = ctx->ownAccum;
        if (!ctx->ownDraw) ctx->width = ctx->height = 0;
        return RT_OK;
    }
    if (!draw_dev || !accum_dev || width <= 0 || height <= 0) return rt_fail(ctx, RT_ERR_INVALID_ARG, "rt_bind_images: need both images and a size");
    if (width > 65536 || height > 65536) return rt_fail(ctx, RT_ERR_INVALID_ARG, "image size out of range");
    if (((uintptr_t)draw_dev & 15) || ((uintptr_t)accum_dev & 15)) return rt_fail(ctx, RT_ERR_INVALID_ARG, "images must be 16-byte aligned");
    freeImages(ctx);
    ctx->draw = draw_dev;
    ctx->accum = accum_dev;
    ctx->width = width;
    ctx->height = height;
    return RT_OK;
} catch (...) { return rt_caught(ctx); }

int rt_clear_accumulation(rt_ctx* ctx) try
{
    if (!ctx) return rt_fail(nullptr, RT_ERR_INVALID_ARG, "null context");
    if (!ctx->accum) return rt_fail(ctx, RT_ERR_NO_IMAGES, "no images (call rt_resize or rt_bind_images first)");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    RT_HIP(ctx, hipMemsetAsync(ctx->accum, 0, (size_t)ctx->width * ctx->height * 16, ctx->stream));
    return RT_OK;
} catch (...) { return rt_caught(ctx); }

int rt_render_frames(rt_ctx* ctx, const rt_push_constants* pc, int first_frame, int stride, int count) try
{
    int rc = checkRenderable(ctx, pc);
    if (rc != RT_OK) return rc;
    if (count < 0 || stride < 1) return rt_fail(ctx, RT_ERR_INVALID_ARG, "rt_render_frames: count must be >= 0 and stride >= 1");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    RT_HIP(ctx, hipEventRecord(ctx->evStart, ctx->stream));
    rt_push_constants p = *pc;
    // the lock-step kernel can run several consecutive frames of a pixel in one launch (same
    // arithmetic in the same order, one read and one write of the accumulation pixel)
    const bool fuse = (ctx->optPipeline == 0 || ctx->optPipeline == 3 || ctx->optPipeline == 4) && pc->renderMode == 0 && pc->raysPerPixel >= 1 && ctx->framesPerLaunch > 1;
    int group = fuse ? ctx->framesPerLaunch : 1;
    // launches in pieces keep 16 bytes per pixel and frame: at most 4 GiB of it
    if (fuse && ctx->optPipeline == 3 && ctx->optPieces != 0) {
        const size_t perFrame = (size_t)ctx->width * (size_t)ctx->height * 16;
        group = (int)std::max<size_t>(2, std::min<size_t>((size_t)group, ((size_t)4 << 30) / std::max<size_t>(perFrame, 1)));
    }
    for (int k = 0; k < count; k += group) {
        const int n = std::min(group, count - k);
        p.frame = first_frame + k * stride;
        if ((rc = launchFrame(ctx, p, false, nullptr, k == 0, k + n >= count, n, stride)) != RT_OK) return rc;
    }
    RT_HIP(ctx, hipEventRecord(ctx->evStop, ctx->stream));
    ctx->timed = true;
    ctx->lastFrames = count;
    return RT_OK;
} catch (...) { return rt_caught(ctx); }

int rt_render_frame(rt_ctx* ctx, const rt_push_constants* pc) try
{
    if (!pc) return rt_fail(ctx, RT_ERR_INVALID_ARG, "null push constants");
    return rt_render_frames(ctx, pc, pc->frame, 1, 1);
} catch (...) { return rt_caught(ctx); }

int rt_sync(rt_ctx* ctx) try
{
    if (!ctx) return rt_fail(nullptr, RT_ERR_INVALID_ARG, "null context");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RT_OK;
} catch (...) { return rt_caught(ctx); }

int rt_read_image(rt_ctx* ctx, int which, float* dst, size_t bytes) try
{
    if (!ctx || !dst) return rt_fail(ctx, RT_ERR_INVALID_ARG, "rt_read_image: null argument");
    if (!ctx->draw || !ctx->accum) return rt_fail(ctx, RT_ERR_NO_IMAGES, "no images");
    if (which != RT_IMAGE_DRAW && which != RT_IMAGE_ACCUM) return rt_fail(ctx, RT_ERR_INVALID_ARG, "which must be 0 (draw) or 1 (accum)");
    const size_t need = (size_t)ctx->width * ctx->height * 16;
    if (bytes != need) return rt_fail(ctx, RT_ERR_INVALID_ARG, "rt_read_image: bytes must be exactly width*height*16 = " + std::to_string(need));
    RT_HIP(ctx, hipSetDevice(ctx->device));
    RT_HIP(ctx, hipMemcpyAsync(dst, which == RT_IMAGE_DRAW ? ctx->draw : ctx->accum, need, hipMemcpyDeviceToHost, ctx->stream));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RT_OK;
} catch (...) { return rt_caught(ctx); }

int rt_write_accum(rt_ctx* ctx, const float* src, size_t bytes) try
{
    if (!ctx || !src) return rt_fail(ctx, RT_ERR_INVALID_ARG, "rt_write_accum: null argument");
    if (!ctx->accum) return rt_fail(ctx, RT_ERR_NO_IMAGES, "no images");
    const size_t need = (size_t)ctx->width * ctx->height * 16;
    if (bytes != need) return rt_fail(ctx, RT_ERR_INVALID_ARG, "rt_write_accum: bytes must be exactly width*height*16");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    RT_HIP(ctx, hipMemcpyAsync(ctx->accum, src, need, hipMemcpyHostToDevice, ctx->stream));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RT_OK;
} catch (...) { return rt_caught(ctx); }

int rt_resolve(rt_ctx* ctx, int total_frames) try
{
    if (!ctx) return rt_fail(nullptr, RT_ERR_INVALID_ARG, "null context");
    if (!ctx->draw || !ctx->accum) return rt_fail(ctx, RT_ERR_NO_IMAGES, "no images");
    if (total_frames < 1) return rt_fail(ctx, RT_ERR_INVALID_ARG, "total_frames must be >= 1");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    auto fn = ctx->optFast ? rt_launch_resolve_fast : rt_launch_resolve_strict;
    hipError_t e = fn(ctx->accum, ctx->draw, (size_t)ctx->width * ctx->height, 1.0f / (float)total_frames, ctx->stream);
    if (e != hipSuccess) return rt_fail_hip(ctx, e, "resolve launch");
    return RT_OK;
} catch (...) { return rt_caught(ctx); }

int rt_frame_stats_compute(rt_ctx* ctx, const rt_push_constants* pc, rt_frame_stats* out) try
{
    int rc = checkRenderable(ctx, pc);
    if (rc != RT_OK) return rc;
    if (!out) return rt_fail(ctx, RT_ERR_INVALID_ARG, "null stats pointer");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    RT_HIP(ctx, hipMemsetAsync(ctx->dCounters, 0, sizeof(DevCounters), ctx->stream));
    if ((rc = launchFrame(ctx, *pc, true)) != RT_OK) return rc;
    DevCounters h{};
    RT_HIP(ctx, hipMemcpyAsync(&h, ctx->dCounters, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    out->rays = h.rays;
    out->bvhNodeTests = h.nodeTests;
    out->triangleTests = h.triTests;
    out->samples = h.samples;
    return RT_OK;
} catch (...) { return rt_caught(ctx); }

int rt_frame_stats_noalloc(rt_ctx* ctx, const rt_push_constants* pc, rt_frame_stats* out_dev) try
{
    int rc = checkRenderable(ctx, pc);
    if (rc != RT_OK) return rc;
    if (!out_dev) return rt_fail(ctx, RT_ERR_INVALID_ARG, "null stats pointer");
    static_assert(sizeof(rt_frame_stats) == sizeof(DevCounters), "rt_frame_stats and DevCounters share one layout");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    {
        // the kernels store the result themselves: it has to be memory the GPU can write
        hipPointerAttribute_t at{};
        hipError_t e = hipPointerGetAttributes(&at, out_dev);
        if (e != hipSuccess) (void)hipGetLastError();
        if (e != hipSuccess || !(at.type == hipMemoryTypeDevice || at.type == hipMemoryTypeHost || at.type == hipMemoryTypeManaged))
            return rt_fail(ctx, RT_ERR_INVALID_ARG, "rt_frame_stats_noalloc: out_dev must be device memory or pinned host memory");
        // memory of another GPU is only reachable with peer access, which this entry point does not set up
        if (at.type == hipMemoryTypeDevice && at.device != ctx->device)
            return rt_fail(ctx, RT_ERR_INVALID_ARG, "rt_frame_stats_noalloc: out_dev is memory of device " + std::to_string(at.device) +
                                                     ", the context renders on device " + std::to_string(ctx->device));
        // a registered host buffer may be mapped at a device address that differs from its host address
        if (at.devicePointer) out_dev = static_cast<rt_frame_stats*>(at.devicePointer);
        else if (at.type != hipMemoryTypeDevice)
            return rt_fail(ctx, RT_ERR_INVALID_ARG, "rt_frame_stats_noalloc: out_dev has no device address");
    }
    const DevLaunch L = makeLaunch(ctx);
    const long long waves = (long long)L.ownedTiles * (L.tileSize / 8) * (L.tileSize / 8);
    if (waves > rt_noalloc_max_waves())
        return rt_fail(ctx, RT_ERR_UNSUPPORTED, "rt_frame_stats_noalloc: this dispatch needs " + std::to_string(waves) +
                                                 " per-wave counter slots, the static scratch array holds " +
                                                 std::to_string(rt_noalloc_max_waves()) + " (use rt_frame_stats_compute)");
    void* slots = rt_noalloc_scratch(ctx->device);
    if (!slots) return rt_fail(ctx, RT_ERR_HIP, "rt_frame_stats_noalloc: the scratch array was not set up on this device");
    // always the lock-step counting kernel: the other pipelines' counting passes use allocated queues
    auto fn = ctx->optFast ? rt_launch_megakernel_fast : rt_launch_megakernel_strict;
    hipError_t e = fn(&ctx->scene, pc, &L, ctx->accum, ctx->draw, static_cast<DevCounters*>(slots), 1, 2, ctx->stream);
    if (e == hipSuccess) e = rt_noalloc_sum(slots, (int)waves, out_dev, ctx->stream);
    if (e != hipSuccess) return rt_fail_hip(ctx, e, "rt_frame_stats_noalloc launch");
    return RT_OK;
} catch (...) { return rt_caught(ctx); }

int rt_schedule_stats(rt_ctx* ctx, const rt_push_constants* pc, uint64_t out[16]) try
{
    int rc = checkRenderable(ctx, pc);
    if (rc != RT_OK) return rc;
    if (!out) return rt_fail(ctx, RT_ERR_INVALID_ARG, "null output pointer");
    if (ctx->optPipeline != 1 || !(pc->renderMode == 0) || pc->raysPerPixel < 1)
        return rt_fail(ctx, RT_ERR_UNSUPPORTED, "schedule statistics exist for pipeline 1, render mode 0 only");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    unsigned long long* d = ctx->dQueue + 8;   // the queue allocation is 256 bytes: counter + 16 statistics words
    RT_HIP(ctx, hipMemsetAsync(d, 0, 16 * sizeof(unsigned long long), ctx->stream));
    if ((rc = launchFrame(ctx, *pc, true, d)) != RT_OK) return rc;
    RT_HIP(ctx, hipMemcpyAsync(out, d, 16 * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RT_OK;
} catch (...) { return rt_caught(ctx); }

int rt_last_render_ms(rt_ctx* ctx, float* ms_total, float* ms_per_frame) try
{
    if (!ctx) return rt_fail(nullptr, RT_ERR_INVALID_ARG, "null context");
    if (!ctx->timed) return rt_fail(ctx, RT_ERR_INVALID_ARG, "nothing has been rendered on this stream yet");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    RT_HIP(ctx, hipEventSynchronize(ctx->evStop));
    float ms = 0.0f;
    RT_HIP(ctx, hipEventElapsedTime(&ms, ctx->evStart, ctx->evStop));
    if (ms_total) *ms_total = ms;
    if (ms_per_frame) *ms_per_frame = ctx->lastFrames > 0 ? ms / (float)ctx->lastFrames : 0.0f;
    return RT_OK;
} catch (...) { return rt_caught(ctx); }

int rt_image_size(rt_ctx* ctx, int* width, int* height) try
{
    if (!ctx) return rt_fail(nullptr, RT_ERR_INVALID_ARG, "null context");
    if (width) *width = ctx->width;
    if (height) *height = ctx->height;
    return RT_OK;
} catch (...) { return rt_caught(ctx); }

int rt_save_png(rt_ctx* ctx, const char* path) try
{
    if (!ctx || !path) return rt_fail(ctx, RT_ERR_INVALID_ARG, "rt_save_png: null argument");
    if (!ctx->draw) return rt_fail(ctx, RT_ERR_NO_IMAGES, "no images (call rt_resize or rt_bind_images first)");
    const size_t pixels = (size_t)ctx->width * (size_t)ctx->height;
    std::vector<float> f(pixels * 4);
    int rc = rt_read_image(ctx, RT_IMAGE_DRAW, f.data(), f.size() * sizeof(float));
    if (rc != RT_OK) return rc;
    std::vector<uint8_t> b(pixels * 4);
    rt_rgba32f_to_rgba8(f.data(), b.data(), pixels);
    rc = rt_write_png_rgba8(path, b.data(), ctx->width, ctx->height);
    if (rc != RT_OK) return rt_fail(ctx, rc, std::string("cannot write ") + path);
    return RT_OK;
} catch (...) { return rt_caught(ctx); }

int rt_produce_render(rt_ctx* ctx, rt_push_constants* pc, int render_frames, const char* png_path, double* seconds) try
{
    if (!ctx || !pc) return rt_fail(ctx, RT_ERR_INVALID_ARG, "rt_produce_render: null argument");
    const auto t0 = std::chrono::system_clock::now();
    const int first = pc->frame;
    if (render_frames > first) {
        int rc = rt_render_frames(ctx, pc, first, 1, render_frames - first);
        if (rc != RT_OK) return rc;
        pc->frame = render_frames;
    }
    int rc = png_path ? rt_save_png(ctx, png_path) : rt_sync(ctx);
    if (rc != RT_OK) return rc;
    if (seconds) *seconds = std::chrono::duration<double>(std::chrono::system_clock::now() - t0).count();
    return RT_OK;
} catch (...) { return rt_caught(ctx); }

int rt_math_apply(rt_ctx* ctx, int which, const float* x, float* y, size_t n) try
{
    if (!ctx || (n && (!x || !y)) || which < 0 || which > 4) return rt_fail(ctx, RT_ERR_INVALID_ARG, "rt_math_apply: bad argument");
    if (n == 0) return RT_OK;
    RT_HIP(ctx, hipSetDevice(ctx->device));
    float *dx = nullptr, *dy = nullptr;
    RT_HIP(ctx, hipMalloc((void**)&dx, n * sizeof(float)));
    hipError_t e = hipMalloc((void**)&dy, n * sizeof(float));
    if (e == hipSuccess) e = hipMemcpyAsync(dx, x, n * sizeof(float), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = (ctx->optFast ? rt_launch_math_fast : rt_launch_math_strict)(which, dx, dy, n, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(y, dy, n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(dx);
    if (dy) (void)hipFree(dy);
    if (e != hipSuccess) return rt_fail_hip(ctx, e, "rt_math_apply");
    return RT_OK;
} catch (...) { return rt_caught(ctx); }

} // extern "C"
