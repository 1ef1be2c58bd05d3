// rt_bvh.cpp - a second bounding-volume hierarchy per model, built at upload time for the
// "near-first" traversal (RT_OPT_TRAVERSAL = 1).
//
// The reference walks its midpoint-split tree in one fixed order (left child first,
// raytrace.comp:603-604) and carries a TODO for visiting the nearer child first
// (raytrace.comp:578-601); its surface-area-heuristic builder (ModelLoader.cpp:124-260) is unfinished
// and never called. This file is the build's own take on both: a binned SAH builder over the
// triangle records of one model, producing the same two-child 64-byte records and leaf runs the
// kernels already walk, with tight boxes. The kernels then enter the nearer child first and drop
// every child whose entry distance is not inside the closest hit so far.
//
// Which objects may use it. GetHit accepts a triangle when its object-space hitT AND its world-space
// distance lie inside (tMin, closestSoFar), closestSoFar being a world-space distance
// (raytrace.comp:553-572). For an object whose transform does not shrink anything (every singular
// value of the model matrix >= 1) the object-space distance is never larger than the world-space one,
// the first test is implied by the second, and the accepted hit is the triangle with the smallest
// world-space distance whatever the visiting order - only exact ties can fall the other way. For an
// object scaled below 1 the first test does bite and the result depends on the order, so such
// objects keep the reference's tree and order (rt_host.cpp decides per object).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "rt_device.h"

namespace {

struct Box {
    float lo[3], hi[3];
    void clear() { for (int a = 0; a < 3; ++a) { lo[a] = INFINITY; hi[a] = -INFINITY; } }
    void grow(const float p[3]) { for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], p[a]); hi[a] = std::max(hi[a], p[a]); } }
    void grow(const Box& b) { for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], b.lo[a]); hi[a] = std::max(hi[a], b.hi[a]); } }
    float area() const
    {
        const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return (dx < 0.0f || dy < 0.0f || dz < 0.0f) ? 0.0f : 2.0f * (dx * dy + dy * dz + dz * dx);
    }
};

struct Prim { Box b; float c[3]; uint32_t src; };

struct Builder {
    const DevTri* src;
    std::vector<Prim> prims;
    std::vector<DevNode>* nodes;
    std::vector<DevTri>* tris;
    int maxLeaf;
    int maxDepth = 0;
    static const int kBins = 16;

    // The boxes of the records are what the slab test sees; the triangle test works on v0/e1/e2. Pad
    // every triangle's box by a few units in the last place of its largest coordinate so that a hit
    // the triangle test accepts right at a vertex or edge is not cut off by a box that is exactly tight.
    static Box triBox(const DevTri& t)
    {
        Box b; b.clear();
        float v1[3], v2[3];
        for (int a = 0; a < 3; ++a) { v1[a] = t.v0[a] + t.e1[a]; v2[a] = t.v0[a] + t.e2[a]; }
        b.grow(t.v0); b.grow(v1); b.grow(v2);
        for (int a = 0; a < 3; ++a) {
            const float m = std::max(std::fabs(b.lo[a]), std::fabs(b.hi[a]));
            const float pad = std::max(m * 4e-6f, 1e-7f);
            b.lo[a] -= pad; b.hi[a] += pad;
        }
        return b;
    }

    int32_t makeLeaf(size_t begin, size_t end)
    {
        const size_t base = tris->size();
        for (size_t i = begin; i < end; ++i) {
            DevTri t = src[prims[i].src];
            t.remaining = (int32_t)(end - i);
            tris->push_back(t);
        }
        return ~(int32_t)base;
    }

    static int log2ceil(size_t n) { int k = 0; while (((size_t)1 << k) < n) ++k; return k; }

    // Builds [begin, end) and returns its reference; `depth` = number of records on the path from the
    // root down to and including the record this call may create.
    int32_t build(size_t begin, size_t end, int depth, Box& bounds)
    {
        bounds.clear();
        Box cb; cb.clear();
        for (size_t i = begin; i < end; ++i) { bounds.grow(prims[i].b); cb.grow(prims[i].c); }
        const size_t n = end - begin;
        if (n <= 2 && n <= (size_t)maxLeaf) return makeLeaf(begin, end);
        if (n == 1) return makeLeaf(begin, end);

        // no room left for an unbalanced tree below this point: split by count (depth stays within the
        // kernels' 32-entry traversal stack)
        const bool forceMedian = depth + log2ceil(n) + 1 >= RT_MAX_STACK - 1;
        size_t mid = begin;
        bool haveSplit = false;
        if (!forceMedian) {
            float bestCost = INFINITY;
            int bestAxis = -1, bestBin = -1;
            for (int a = 0; a < 3; ++a) {
                const float ext = cb.hi[a] - cb.lo[a];
                if (!(ext > 0.0f)) continue;
                Box bb[kBins]; size_t cnt[kBins];
                for (int k = 0; k < kBins; ++k) { bb[k].clear(); cnt[k] = 0; }
                const float scale = (float)kBins / ext;
                for (size_t i = begin; i < end; ++i) {
                    int k = (int)((prims[i].c[a] - cb.lo[a]) * scale);
                    k = k < 0 ? 0 : (k >= kBins ? kBins - 1 : k);
                    bb[k].grow(prims[i].b); cnt[k]++;
                }
                float rightArea[kBins]; size_t rightCnt[kBins];
                Box acc; acc.clear(); size_t c = 0;
                for (int k = kBins - 1; k > 0; --k) { acc.grow(bb[k]); c += cnt[k]; rightArea[k] = acc.area(); rightCnt[k] = c; }
                acc.clear(); c = 0;
                for (int k = 0; k < kBins - 1; ++k) {
                    acc.grow(bb[k]); c += cnt[k];
                    if (c == 0 || rightCnt[k + 1] == 0) continue;
                    const float cost = acc.area() * (float)c + rightArea[k + 1] * (float)rightCnt[k + 1];
                    if (cost < bestCost) { bestCost = cost; bestAxis = a; bestBin = k; }
                }
            }
            if (bestAxis >= 0) {
                const float parentArea = std::max(bounds.area(), 1e-30f);
                // a record visit weighs three triangle tests: measured on the lock-step kernels (1, 2, 3, 4, 6 tried:
                // 2.70, 2.64, 2.65, 2.65, 2.65 ms per frame on the default scene, 4.24, 4.13, 4.07, 4.10, 4.10 on 64 x 15 616 triangles)
                const float splitCost = 3.0f + bestCost / parentArea;     // one record visit + expected triangle tests below
                if (n <= (size_t)maxLeaf && (float)n <= splitCost) return makeLeaf(begin, end);
                const float ext = cb.hi[bestAxis] - cb.lo[bestAxis];
                const float scale = (float)kBins / ext;
                const float lo = cb.lo[bestAxis];
                auto it = std::partition(prims.begin() + begin, prims.begin() + end, [&](const Prim& p) {
                    int k = (int)((p.c[bestAxis] - lo) * scale);
                    k = k < 0 ? 0 : (k >= kBins ? kBins - 1 : k);
                    return k <= bestBin;
                });
                mid = (size_t)(it - prims.begin());
                haveSplit = mid > begin && mid < end;
            }
        }
        if (!haveSplit) {
            // all centroids in one place, or the depth guard: halve by count along the widest axis
            if (!forceMedian && n <= (size_t)std::max(maxLeaf, 8)) return makeLeaf(begin, end);
            int a = 0;
            for (int k = 1; k < 3; ++k) if (cb.hi[k] - cb.lo[k] > cb.hi[a] - cb.lo[a]) a = k;
            mid = begin + n / 2;
            std::nth_element(prims.begin() + begin, prims.begin() + mid, prims.begin() + end,
                             [a](const Prim& x, const Prim& y) { return x.c[a] < y.c[a]; });
        }

        const int32_t idx = (int32_t)nodes->size();
        nodes->push_back(DevNode{});
        maxDepth = std::max(maxDepth, depth);
        Box bl, br;
        const int32_t rl = build(begin, mid, depth + 1, bl);
        const int32_t rr = build(mid, end, depth + 1, br);
        DevNode nd{};
        for (int a = 0; a < 3; ++a) {
            nd.lo[a][0] = bl.lo[a]; nd.hi[a][0] = bl.hi[a];
            nd.lo[a][1] = br.lo[a]; nd.hi[a][1] = br.hi[a];
        }
        nd.first = rl;
        nd.second = rr;
        (*nodes)[idx] = nd;
        return idx;
    }
};

} // namespace

// Builds a SAH tree over `count` triangle records (object space). New two-child records are appended to
// `nodes`, the leaves' triangle records (copies of the input records, regrouped) to `tris`; references
// are indices into those vectors (record index, or ~first triangle for a leaf). Returns the root
// reference, the tight bounds of the model and the number of traversal-stack entries a lane can need.
int32_t rt_build_sah(const DevTri* src, size_t count, int maxLeaf, std::vector<DevNode>& nodes, std::vector<DevTri>& tris,
                     float boundsLo[3], float boundsHi[3], int& depth)
{
    Builder b;
    b.src = src;
    b.nodes = &nodes;
    b.tris = &tris;
    b.maxLeaf = std::max(1, maxLeaf);
    b.prims.resize(count);
    for (size_t i = 0; i < count; ++i) {
        Prim& p = b.prims[i];
        p.b = Builder::triBox(src[i]);
        for (int a = 0; a < 3; ++a) p.c[a] = 0.5f * (p.b.lo[a] + p.b.hi[a]);
        p.src = (uint32_t)i;
    }
    Box bounds; bounds.clear();
    int32_t root = RT_REF_EMPTY;
    if (count > 0) root = b.build(0, count, 1, bounds);
    for (int a = 0; a < 3; ++a) { boundsLo[a] = bounds.lo[a]; boundsHi[a] = bounds.hi[a]; }
    depth = b.maxDepth;
    return root;
}

// Bounds of `count` triangle records, padded like the boxes of the rebuilt trees.
void rt_triangle_bounds(const DevTri* src, size_t count, float boundsLo[3], float boundsHi[3])
{
    Box bounds; bounds.clear();
    for (size_t i = 0; i < count; ++i) bounds.grow(Builder::triBox(src[i]));
    for (int a = 0; a < 3; ++a) { boundsLo[a] = bounds.lo[a]; boundsHi[a] = bounds.hi[a]; }
}

// Folds the two-child tree below record `root` (an index into `nodes2`) into four-child records appended
// to `out`: starting from a record's two children, the inner child with the largest box is replaced by its
// own two children until four slots are used or only leaves are left. Same boxes, same leaves; a
// root-to-leaf path becomes about half as many records. Returns the index of the new root record; `need`
// receives the number of traversal-stack entries a lane can need below it (up to three per level).
namespace {
struct Slot4 { float lo[3], hi[3]; int32_t ref; };

int32_t collapse4(const std::vector<DevNode>& nodes2, int32_t root, std::vector<DevNode4>& out, int& need)
{
    Slot4 s[4];
    int n = 0;
    auto take = [&](const DevNode& nd, int c, Slot4& dst) {
        for (int a = 0; a < 3; ++a) { dst.lo[a] = nd.lo[a][c]; dst.hi[a] = nd.hi[a][c]; }
        dst.ref = c == 0 ? nd.first : nd.second;
    };
    take(nodes2[(size_t)root], 0, s[n++]);
    take(nodes2[(size_t)root], 1, s[n++]);
    while (n < 4) {
        int pick = -1;
        float bestArea = -1.0f;
        for (int i = 0; i < n; ++i) {
            if (s[i].ref < 0) continue;
            const float dx = s[i].hi[0] - s[i].lo[0], dy = s[i].hi[1] - s[i].lo[1], dz = s[i].hi[2] - s[i].lo[2];
            const float area = dx * dy + dy * dz + dz * dx;
            if (area > bestArea) { bestArea = area; pick = i; }
        }
        if (pick < 0) break;
        const DevNode& nd = nodes2[(size_t)s[pick].ref];
        take(nd, 1, s[n++]);
        take(nd, 0, s[pick]);
    }
    const int32_t idx = (int32_t)out.size();
    out.push_back(DevNode4{});
    DevNode4 rec{};
    int below = 0;
    for (int c = 0; c < 4; ++c) {
        if (c >= n || !(s[c].ref > RT_REF_LIVE_MIN)) {      // unused slot, or a child nothing can enter
            for (int a = 0; a < 3; ++a) { rec.lo[a][c] = RT_FAR_PLANE; rec.hi[a][c] = RT_FAR_PLANE; }
            rec.ref[c] = RT_REF_EMPTY;
            continue;
        }
        for (int a = 0; a < 3; ++a) { rec.lo[a][c] = s[c].lo[a]; rec.hi[a][c] = s[c].hi[a]; }
        if (s[c].ref >= 0) {
            int sub = 0;
            rec.ref[c] = collapse4(nodes2, s[c].ref, out, sub);
            below = std::max(below, sub);
        } else {
            rec.ref[c] = s[c].ref;
        }
    }
    out[(size_t)idx] = rec;
    need = (n - 1) + below;
    return idx;
}
} // namespace

int32_t rt_collapse4(const std::vector<DevNode>& nodes2, int32_t root, std::vector<DevNode4>& out, int& need)
{
    need = 0;
    if (root < 0) return root;
    return collapse4(nodes2, root, out, need);
}

// Hierarchy over the world boxes of the objects that own a tree (boxes: `count` x {lo xyz, hi xyz}), as
// four-child records appended to `out`. A child reference of these records is either another such record or
// RT_REF_INSTANCE + k for object k of the list; returns the root reference (an instance reference when there is
// a single object) and in `need` the traversal-stack entries a lane can need above the objects' own trees.
// Two-child tree first - surface-area heuristic over the sorted centroids of each axis, or halving by count
// when `balanced` (shallowest tree, for scenes whose heuristic tree would be too deep for the stack) - then
// folded like the objects' trees.
namespace {
struct TlasBuilder {
    std::vector<Prim> prims;
    std::vector<DevNode> nodes;
    bool balanced;

    int32_t build(size_t begin, size_t end, Box& bounds)
    {
        bounds.clear();
        Box cb; cb.clear();
        for (size_t i = begin; i < end; ++i) { bounds.grow(prims[i].b); cb.grow(prims[i].c); }
        const size_t n = end - begin;
        if (n == 1) return ~(int32_t)prims[begin].src;
        size_t mid = begin + n / 2;
        int axis = 0;
        for (int k = 1; k < 3; ++k) if (cb.hi[k] - cb.lo[k] > cb.hi[axis] - cb.lo[axis]) axis = k;
        if (!balanced) {
            float bestCost = INFINITY;
            int bestAxis = -1; size_t bestMid = mid;
            std::vector<float> rightArea(n);
            for (int a = 0; a < 3; ++a) {
                if (!(cb.hi[a] - cb.lo[a] > 0.0f)) continue;
                std::sort(prims.begin() + begin, prims.begin() + end, [a](const Prim& x, const Prim& y) { return x.c[a] < y.c[a]; });
                Box acc; acc.clear();
                for (size_t i = n - 1; i > 0; --i) { acc.grow(prims[begin + i].b); rightArea[i] = acc.area(); }
                acc.clear();
                for (size_t i = 1; i < n; ++i) {
                    acc.grow(prims[begin + i - 1].b);
                    const float cost = acc.area() * (float)i + rightArea[i] * (float)(n - i);
                    if (cost < bestCost) { bestCost = cost; bestAxis = a; bestMid = begin + i; }
                }
            }
            if (bestAxis >= 0) { axis = bestAxis; mid = bestMid; }
        }
        {
            // (re)order along the chosen axis; the heuristic's split position counts from the sorted order
            std::sort(prims.begin() + begin, prims.begin() + end, [axis](const Prim& x, const Prim& y) { return x.c[axis] < y.c[axis]; });
        }
        const int32_t idx = (int32_t)nodes.size();
        nodes.push_back(DevNode{});
        Box bl, br;
        const int32_t rl = build(begin, mid, bl);
        const int32_t rr = build(mid, end, br);
        DevNode nd{};
        for (int a = 0; a < 3; ++a) {
            nd.lo[a][0] = bl.lo[a]; nd.hi[a][0] = bl.hi[a];
            nd.lo[a][1] = br.lo[a]; nd.hi[a][1] = br.hi[a];
        }
        nd.first = rl;
        nd.second = rr;
        nodes[(size_t)idx] = nd;
        return idx;
    }
};
} // namespace

int32_t rt_build_tlas(const float* boxes, size_t count, bool balanced, std::vector<DevNode4>& out, int& need)
{
    need = 0;
    if (count == 0) return RT_REF_EMPTY;
    if (count == 1) return RT_REF_INSTANCE;
    TlasBuilder b;
    b.balanced = balanced;
    b.prims.resize(count);
    for (size_t i = 0; i < count; ++i) {
        Prim& p = b.prims[i];
        for (int a = 0; a < 3; ++a) { p.b.lo[a] = boxes[6 * i + a]; p.b.hi[a] = boxes[6 * i + 3 + a]; p.c[a] = 0.5f * (p.b.lo[a] + p.b.hi[a]); }
        p.src = (uint32_t)i;
    }
    Box bounds;
    const int32_t root2 = b.build(0, count, bounds);
    const size_t base = out.size();
    const int32_t root = collapse4(b.nodes, root2, out, need);
    for (size_t k = base; k < out.size(); ++k)
        for (int c = 0; c < 4; ++c)
            if (out[k].ref[c] < 0 && out[k].ref[c] > RT_REF_LIVE_MIN) out[k].ref[c] = RT_REF_INSTANCE + ~out[k].ref[c];
    return root;
}

// The reference's own two-child records below `root` folded into four-child records WITHOUT changing what a ray hits:
// the slots of a record stand in the reference's visiting order (first child before second, at every level that is
// folded away), children that no ray can enter (absent, empty, inverted box) are left out, and an inner child is
// replaced by its own children in place. A walk that tests a record's slots, descends into the first one hit and stacks
// the later ones visits the reference's leaves in the reference's order; the box tests it skips (of the children folded
// away) are implied by the tests of the boxes inside them - a slab interval only grows with the box. Returns the index
// of the new root record in `out`; `need` = traversal-stack entries a lane can need below it.
namespace {
int32_t collapse4_ordered(const std::vector<DevNode>& nodes2, int32_t root, std::vector<DevNode4>& out, int& need)
{
    std::vector<Slot4> s;
    auto take = [&](const DevNode& nd, int c, std::vector<Slot4>& dst, size_t at) {
        const int32_t ref = c == 0 ? nd.first : nd.second;
        if (!(ref > RT_REF_LIVE_MIN)) return (size_t)0;
        Slot4 v;
        for (int a = 0; a < 3; ++a) { v.lo[a] = nd.lo[a][c]; v.hi[a] = nd.hi[a][c]; }
        v.ref = ref;
        dst.insert(dst.begin() + (long)at, v);
        return (size_t)1;
    };
    {
        const DevNode& nd = nodes2[(size_t)root];
        size_t at = 0;
        at += take(nd, 0, s, at);
        at += take(nd, 1, s, at);
    }
    for (;;) {
        if (s.size() >= 4) break;
        int pick = -1;
        float bestArea = -1.0f;
        for (size_t i = 0; i < s.size(); ++i) {
            if (s[i].ref < 0) continue;
            const float dx = s[i].hi[0] - s[i].lo[0], dy = s[i].hi[1] - s[i].lo[1], dz = s[i].hi[2] - s[i].lo[2];
            const float area = dx * dy + dy * dz + dz * dx;
            if (area > bestArea || pick < 0) { bestArea = area; pick = (int)i; }
        }
        if (pick < 0) break;
        const DevNode nd = nodes2[(size_t)s[(size_t)pick].ref];
        s.erase(s.begin() + pick);
        size_t at = (size_t)pick;
        at += take(nd, 0, s, at);
        at += take(nd, 1, s, at);
    }
    const int32_t idx = (int32_t)out.size();
    out.push_back(DevNode4{});
    DevNode4 rec{};
    int below = 0;
    const int n = (int)s.size();
    for (int c = 0; c < 4; ++c) {
        if (c >= n) {
            for (int a = 0; a < 3; ++a) { rec.lo[a][c] = RT_FAR_PLANE; rec.hi[a][c] = RT_FAR_PLANE; }
            rec.ref[c] = RT_REF_EMPTY;
            continue;
        }
        for (int a = 0; a < 3; ++a) { rec.lo[a][c] = s[(size_t)c].lo[a]; rec.hi[a][c] = s[(size_t)c].hi[a]; }
        if (s[(size_t)c].ref >= 0) {
            int sub = 0;
            rec.ref[c] = collapse4_ordered(nodes2, s[(size_t)c].ref, out, sub);
            below = std::max(below, sub);
        } else {
            rec.ref[c] = s[(size_t)c].ref;
        }
    }
    out[(size_t)idx] = rec;
    need = std::max(0, n - 1) + below;
    return idx;
}
} // namespace

int32_t rt_collapse4_ordered(const std::vector<DevNode>& nodes2, int32_t root, std::vector<DevNode4>& out, int& need)
{
    need = 0;
    if (root < 0) return root;
    return collapse4_ordered(nodes2, root, out, need);
}
