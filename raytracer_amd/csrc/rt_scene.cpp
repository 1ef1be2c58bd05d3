// rt_scene.cpp - host-side scene builder: the CPU half of the reference's
// HardwareRenderer that feeds its ray-tracing shader (no GPU code here).
//
//   rt_scene_load_model      <- HardwareRenderer::LoadModel          (HardwareRenderer.cpp:1157-1314)
//                               LoadObjFile / BuildBVH / SplitBVHNode (ModelLoader.cpp:10-120,262-330)
//   rt_scene_add_object      <- AddSceneObject / GetMaterialIndex    (HardwareRenderer.cpp:479-500)
//   rt_scene_buffer_data     <- ConvertSceneObjectToGPUObject        (HardwareRenderer.cpp:502-551)
//   rt_scene_init_default    <- InitializeScene                      (HardwareRenderer.cpp:574-649)
//   rt_camera_to_push_constants <- DispatchRayTracingCommands        (HardwareRenderer.cpp:795-824)
//
// The arrays it produces are meant to be bit-identical to what the reference
// uploads, because triangle order, node numbering and box values decide every
// tie-break of the traversal. All arithmetic is plain binary32 in the operand
// order of the reference (glm's operators are restated where the reference
// calls glm); build with -ffp-contract=off.
#include <algorithm>
#include <climits>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <memory>
#include <sstream>
#include <new>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/rt_abi.h"

namespace {

struct Vec3 {
    float x = 0, y = 0, z = 0;
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
inline Vec3 operator+(Vec3 a, Vec3 b) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }
inline Vec3 operator-(Vec3 a, Vec3 b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
inline Vec3 operator*(Vec3 a, float s) { return { a.x * s, a.y * s, a.z * s }; }
inline Vec3 operator/(Vec3 a, float s) { return { a.x / s, a.y / s, a.z / s }; }
inline Vec3 operator-(Vec3 a) { return { -a.x, -a.y, -a.z }; }
inline float dot(Vec3 a, Vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline Vec3 cross(Vec3 a, Vec3 b) { return { a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y }; }
// glm::normalize = v * inversesqrt(dot(v,v)), inversesqrt(x) = 1/sqrt(x)
inline Vec3 normalize(Vec3 a) { return a * (1.0f / std::sqrt(dot(a, a))); }
inline bool isZero(Vec3 a) { return a.x == 0.0f && a.y == 0.0f && a.z == 0.0f; }

// Column-major 4x4 (glm::mat4): col[c][r].
struct Mat4 {
    float col[4][4];
    static Mat4 identity()
    {
        Mat4 m{};
        for (int i = 0; i < 4; ++i) m.col[i][i] = 1.0f;
        return m;
    }
};

// glm::translate(m, v): m[3] = m[0]*v.x + m[1]*v.y + m[2]*v.z + m[3]
Mat4 translate(const Mat4& m, Vec3 v)
{
    Mat4 out = m;
    for (int r = 0; r < 4; ++r) out.col[3][r] = m.col[0][r] * v.x + m.col[1][r] * v.y + m.col[2][r] * v.z + m.col[3][r];
    return out;
}

// glm::rotate(m, angle, axis) (ext/matrix_transform.inl)
Mat4 rotate(const Mat4& m, float angle, Vec3 axisIn)
{
    const float c = std::cos(angle), s = std::sin(angle);
    const Vec3 axis = normalize(axisIn);
    const Vec3 temp = axis * (1.0f - c);
    float R[3][3];
    R[0][0] = c + temp.x * axis.x;
    R[0][1] = temp.x * axis.y + s * axis.z;
    R[0][2] = temp.x * axis.z - s * axis.y;
    R[1][0] = temp.y * axis.x - s * axis.z;
    R[1][1] = c + temp.y * axis.y;
    R[1][2] = temp.y * axis.z + s * axis.x;
    R[2][0] = temp.z * axis.x + s * axis.y;
    R[2][1] = temp.z * axis.y - s * axis.x;
    R[2][2] = c + temp.z * axis.z;
    Mat4 out;
    for (int j = 0; j < 3; ++j)
        for (int r = 0; r < 4; ++r) out.col[j][r] = m.col[0][r] * R[j][0] + m.col[1][r] * R[j][1] + m.col[2][r] * R[j][2];
    for (int r = 0; r < 4; ++r) out.col[3][r] = m.col[3][r];
    return out;
}

Mat4 scale(const Mat4& m, Vec3 v)
{
    Mat4 out;
    for (int r = 0; r < 4; ++r) {
        out.col[0][r] = m.col[0][r] * v.x;
        out.col[1][r] = m.col[1][r] * v.y;
        out.col[2][r] = m.col[2][r] * v.z;
        out.col[3][r] = m.col[3][r];
    }
    return out;
}

// glm mat4 * vec4: (m[0]*x + m[1]*y) + (m[2]*z + m[3]*w)
void mulPoint(const Mat4& m, float x, float y, float z, float w, float out[4])
{
    for (int r = 0; r < 4; ++r) out[r] = (m.col[0][r] * x + m.col[1][r] * y) + (m.col[2][r] * z + m.col[3][r] * w);
}

inline float radiansf(float deg) { return deg * 0.01745329251994329576923690768489f; }

} // namespace

// glm::inverse(mat4) (detail/func_matrix.inl, compute_inverse<4,4>). Exported
// (C++ linkage, library-internal) so the device upload uses the very same code
// to hoist the shader's per-ray inverse() to upload time.
void rt_mat4_inverse(const float in[16], float out[16])
{
    float m[4][4];
    std::memcpy(m, in, sizeof(m));
    const float c00 = m[2][2] * m[3][3] - m[3][2] * m[2][3];
    const float c02 = m[1][2] * m[3][3] - m[3][2] * m[1][3];
    const float c03 = m[1][2] * m[2][3] - m[2][2] * m[1][3];
    const float c04 = m[2][1] * m[3][3] - m[3][1] * m[2][3];
    const float c06 = m[1][1] * m[3][3] - m[3][1] * m[1][3];
    const float c07 = m[1][1] * m[2][3] - m[2][1] * m[1][3];
    const float c08 = m[2][1] * m[3][2] - m[3][1] * m[2][2];
    const float c10 = m[1][1] * m[3][2] - m[3][1] * m[1][2];
    const float c11 = m[1][1] * m[2][2] - m[2][1] * m[1][2];
    const float c12 = m[2][0] * m[3][3] - m[3][0] * m[2][3];
    const float c14 = m[1][0] * m[3][3] - m[3][0] * m[1][3];
    const float c15 = m[1][0] * m[2][3] - m[2][0] * m[1][3];
    const float c16 = m[2][0] * m[3][2] - m[3][0] * m[2][2];
    const float c18 = m[1][0] * m[3][2] - m[3][0] * m[1][2];
    const float c19 = m[1][0] * m[2][2] - m[2][0] * m[1][2];
    const float c20 = m[2][0] * m[3][1] - m[3][0] * m[2][1];
    const float c22 = m[1][0] * m[3][1] - m[3][0] * m[1][1];
    const float c23 = m[1][0] * m[2][1] - m[2][0] * m[1][1];

    const float f0[4] = { c00, c00, c02, c03 }, f1[4] = { c04, c04, c06, c07 }, f2[4] = { c08, c08, c10, c11 };
    const float f3[4] = { c12, c12, c14, c15 }, f4[4] = { c16, c16, c18, c19 }, f5[4] = { c20, c20, c22, c23 };
    const float v0[4] = { m[1][0], m[0][0], m[0][0], m[0][0] }, v1[4] = { m[1][1], m[0][1], m[0][1], m[0][1] };
    const float v2[4] = { m[1][2], m[0][2], m[0][2], m[0][2] }, v3[4] = { m[1][3], m[0][3], m[0][3], m[0][3] };
    float inv[4][4];
    for (int i = 0; i < 4; ++i) {
        const float sa = (i & 1) ? -1.0f : 1.0f, sb = -sa;
        inv[0][i] = (v1[i] * f0[i] - v2[i] * f1[i] + v3[i] * f2[i]) * sa;
        inv[1][i] = (v0[i] * f0[i] - v2[i] * f3[i] + v3[i] * f4[i]) * sb;
        inv[2][i] = (v0[i] * f1[i] - v1[i] * f3[i] + v3[i] * f5[i]) * sa;
        inv[3][i] = (v0[i] * f2[i] - v1[i] * f4[i] + v2[i] * f5[i]) * sb;
    }
    const float det = (m[0][0] * inv[0][0] + m[0][1] * inv[1][0]) + (m[0][2] * inv[2][0] + m[0][3] * inv[3][0]);
    const float oneOverDet = 1.0f / det;
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 4; ++r) out[4 * c + r] = inv[c][r] * oneOverDet;
}

namespace {

// ---- OBJ reading (tiny_obj_loader.h v2.0 semantics for v / vn / f) -------------

// tryParseDouble, tiny_obj_loader.h:891-1021: tinyobj's own decimal reader, not
// strtod - its rounding decides the float the reference ends up with.
//
// This number reader and the face triangulation further down re-express tinyobjloader v2.0.0's, step for step
// where the byte-exact vertex stream depends on it; tinyobjloader is
//   Copyright (c) 2012-Present, Syoyo Fujita and many contributors, released under the MIT License:
//   "Permission is hereby granted, free of charge, to any person obtaining a copy of this software and
//   associated documentation files (the "Software"), to deal in the Software without restriction,
//   including without limitation the rights to use, copy, modify, merge, publish, distribute,
//   sublicense, and/or sell copies of the Software, and to permit persons to whom the Software is
//   furnished to do so, subject to the following conditions: The above copyright notice and this
//   permission notice shall be included in all copies or substantial portions of the Software.
//   THE SOFTWARE IS PROVIDED "AS IS", WITHOUT WARRANTY OF ANY KIND, EXPRESS OR IMPLIED, INCLUDING BUT
//   NOT LIMITED TO THE WARRANTIES OF MERCHANTABILITY, FITNESS FOR A PARTICULAR PURPOSE AND
//   NONINFRINGEMENT. IN NO EVENT SHALL THE AUTHORS OR COPYRIGHT HOLDERS BE LIABLE FOR ANY CLAIM, DAMAGES
//   OR OTHER LIABILITY, WHETHER IN AN ACTION OF CONTRACT, TORT OR OTHERWISE, ARISING FROM, OUT OF OR IN
//   CONNECTION WITH THE SOFTWARE OR THE USE OR OTHER DEALINGS IN THE SOFTWARE."
bool parseDecimal(const char* p, const char* end, double& value)
{
    if (p >= end) return false;
    double mant = 0.0;
    int expo = 0;
    bool negative = false, expNegative = false, leadingDot = false;
    const char* q = p;
    auto digit = [](char ch) { return ch >= '0' && ch <= '9'; };

    if (*q == '+' || *q == '-') {
        negative = (*q == '-');
        ++q;
        if (q != end && *q == '.') leadingDot = true;
    } else if (*q == '.') {
        leadingDot = true;
    } else if (!digit(*q)) {
        return false;
    }
    int count = 0;
    if (!leadingDot) {
        for (; q != end && digit(*q); ++q, ++count) {
            mant *= 10;
            mant += static_cast<int>(*q - '0');
        }
        if (count == 0) return false;
    }
    auto finish = [&]() {
        value = (negative ? -1 : 1) * (expo ? std::ldexp(mant * std::pow(5.0, expo), expo) : mant);
        return true;
    };
    if (q == end) return finish();
    if (*q == '.') {
        static const double lut[8] = { 1.0, 0.1, 0.01, 0.001, 0.0001, 0.00001, 0.000001, 0.0000001 };
        ++q;
        for (count = 1; q != end && digit(*q); ++q, ++count)
            mant += static_cast<int>(*q - '0') * (count < 8 ? lut[count] : std::pow(10.0, -count));
    } else if (*q != 'e' && *q != 'E') {
        return finish();
    }
    if (q == end) return finish();
    if (*q == 'e' || *q == 'E') {
        ++q;
        if (q != end && (*q == '+' || *q == '-')) {
            expNegative = (*q == '-');
            ++q;
        } else if (!digit(*q)) {
            return false;
        }
        for (count = 0; q != end && digit(*q); ++q, ++count) {
            if (expo > INT_MAX / 10) return false;
            expo = expo * 10 + static_cast<int>(*q - '0');
        }
        if (expNegative) expo = -expo;
        if (count == 0) return false;
    }
    return finish();
}

struct Cursor {
    const char* p;
    void skipBlanks() { p += std::strspn(p, " \t"); }
    bool atEol() const { return *p == '\0' || *p == '\r' || *p == '\n'; }
    float real()   // parseReal, tiny_obj_loader.h:1023-1031 (missing / malformed -> 0)
    {
        skipBlanks();
        const char* end = p + std::strcspn(p, " \t\r");
        double v = 0.0;
        parseDecimal(p, end, v);
        p = end;
        return static_cast<float>(v);
    }
};

// fixIndex, tiny_obj_loader.h:816-846
bool resolveIndex(int raw, int count, bool zeroAllowed, int& out)
{
    if (raw > 0) { out = raw - 1; return true; }
    if (raw == 0) { out = -1; return zeroAllowed; }
    out = count + raw;
    return out >= 0;
}

struct Corner { int v = -1, vn = -1; };

// parseTriple, tiny_obj_loader.h:1161-1212 (i, i/j, i//k, i/j/k)
bool parseCorner(Cursor& c, int nv, int nvn, int nvt, Corner& out)
{
    int vt = -1;
    if (!resolveIndex(std::atoi(c.p), nv, false, out.v)) return false;
    c.p += std::strcspn(c.p, "/ \t\r");
    if (*c.p != '/') return true;
    ++c.p;
    if (*c.p == '/') {
        ++c.p;
        if (!resolveIndex(std::atoi(c.p), nvn, true, out.vn)) return false;
        c.p += std::strcspn(c.p, "/ \t\r");
        return true;
    }
    if (!resolveIndex(std::atoi(c.p), nvt, true, vt)) return false;
    c.p += std::strcspn(c.p, "/ \t\r");
    if (*c.p != '/') return true;
    ++c.p;
    if (!resolveIndex(std::atoi(c.p), nvn, true, out.vn)) return false;
    c.p += std::strcspn(c.p, "/ \t\r");
    return true;
}

struct ObjVertex { Vec3 position, normal; };   // Vertex, Vulkan/VulkanTypes.h:38-43 (uv unused)

// LoadObjFile, ModelLoader.cpp:292-330, on top of tinyobj::LoadObj with
// triangulate = true: corners in file order, quads split along the shorter
// diagonal (tiny_obj_loader.h:1488-1584). Polygons with more than four corners
// would need tinyobj's ear clipping and are rejected.
bool readObj(const std::string& path, std::vector<ObjVertex>& corners, std::string& error)
{
    std::ifstream in(path, std::ios::binary);
    if (!in) { error = path + " does not exist."; return false; }
    std::vector<float> pos, nrm;
    size_t texcoords = 0;
    std::vector<Corner> flat;
    std::string line;
    size_t lineNo = 0;
    while (std::getline(in, line)) {
        ++lineNo;
        while (!line.empty() && (line.back() == '\n' || line.back() == '\r')) line.pop_back();
        Cursor c{ line.c_str() };
        c.skipBlanks();
        if (*c.p == '\0' || *c.p == '#') continue;
        const char k0 = c.p[0], k1 = c.p[1];
        const bool sp1 = (k1 == ' ' || k1 == '\t');
        if (k0 == 'v' && sp1) {
            c.p += 2;
            float x = c.real(), y = c.real(), z = c.real();
            pos.insert(pos.end(), { x, y, z });
        } else if (k0 == 'v' && k1 == 'n' && (c.p[2] == ' ' || c.p[2] == '\t')) {
            c.p += 3;
            float x = c.real(), y = c.real(), z = c.real();
            nrm.insert(nrm.end(), { x, y, z });
        } else if (k0 == 'v' && k1 == 't' && (c.p[2] == ' ' || c.p[2] == '\t')) {
            ++texcoords;
        } else if (k0 == 'f' && sp1) {
            c.p += 2;
            c.skipBlanks();
            std::vector<Corner> face;
            while (!c.atEol()) {
                Corner cr;
                if (!parseCorner(c, (int)(pos.size() / 3), (int)(nrm.size() / 3), (int)texcoords, cr)) {
                    error = "Failed to parse `f' line " + std::to_string(lineNo) + " of " + path;
                    return false;
                }
                face.push_back(cr);
                c.p += std::strspn(c.p, " \t\r");
            }
            if (face.size() < 3) continue;
            if (face.size() == 3) {
                flat.insert(flat.end(), face.begin(), face.end());
            } else if (face.size() == 4) {
                bool valid = true;
                for (const Corner& cr : face) valid = valid && (size_t)(3 * (size_t)cr.v + 2) < pos.size();
                if (!valid) continue;
                auto P = [&](int i) { const float* p = &pos[3 * (size_t)face[i].v]; return Vec3{ p[0], p[1], p[2] }; };
                const Vec3 d02 = P(2) - P(0), d13 = P(3) - P(1);
                const float sq02 = d02.x * d02.x + d02.y * d02.y + d02.z * d02.z;
                const float sq13 = d13.x * d13.x + d13.y * d13.y + d13.z * d13.z;
                if (sq02 < sq13) {
                    for (int i : { 0, 1, 2, 0, 2, 3 }) flat.push_back(face[i]);
                } else {
                    for (int i : { 0, 1, 3, 1, 2, 3 }) flat.push_back(face[i]);
                }
            } else {
                error = "polygon with " + std::to_string(face.size()) + " corners at line " + std::to_string(lineNo) +
                        " of " + path + " (only triangles and quads are supported)";
                return false;
            }
        }
        // o / g / s / usemtl / mtllib do not change the order corners are emitted in
    }
    corners.reserve(flat.size());
    for (const Corner& cr : flat) {
        if (cr.v < 0 || (size_t)(3 * (size_t)cr.v + 2) >= pos.size() ||
            (cr.vn >= 0 && (size_t)(3 * (size_t)cr.vn + 2) >= nrm.size())) {
            error = "face references a vertex that does not exist in " + path;
            return false;
        }
        ObjVertex v;
        v.position = { pos[3 * (size_t)cr.v], pos[3 * (size_t)cr.v + 1], pos[3 * (size_t)cr.v + 2] };
        if (!nrm.empty() && cr.vn >= 0) v.normal = { nrm[3 * (size_t)cr.vn], nrm[3 * (size_t)cr.vn + 1], nrm[3 * (size_t)cr.vn + 2] };
        corners.push_back(v);
    }
    return true;
}

// ---- model + BVH ----------------------------------------------------------------

struct Tri { Vec3 v0, v1, v2, n0, n1, n2, centroid; };   // Triangle, RaytracerTypes.h:100-115

struct Box {
    Vec3 lo, hi;
    // AABB::GetLongestAxis, RaytracerTypes.h:61-70
    int longestAxis() const
    {
        const Vec3 e = hi - lo;
        if (e.x > e.y && e.x > e.z) return 0;
        return e.y > e.z ? 1 : 2;
    }
};

struct BuildNode { Box box; int left = -1, right = -1, first = 0, count = 0; };

// ExpandNodeAABB, ModelLoader.cpp:262-290
Box tightBox(const std::vector<Tri>& tris, int first, int count)
{
    Box b{ { FLT_MAX, FLT_MAX, FLT_MAX }, { -FLT_MAX, -FLT_MAX, -FLT_MAX } };
    auto grow = [&b](const Vec3& p) {
        if (p.x < b.lo.x) b.lo.x = p.x;
        if (p.y < b.lo.y) b.lo.y = p.y;
        if (p.z < b.lo.z) b.lo.z = p.z;
        if (p.x > b.hi.x) b.hi.x = p.x;
        if (p.y > b.hi.y) b.hi.y = p.y;
        if (p.z > b.hi.z) b.hi.z = p.z;
    };
    for (int i = first; i < first + count; ++i) { grow(tris[i].v0); grow(tris[i].v1); grow(tris[i].v2); }
    return b;
}

// The in-place partition of ModelLoader.cpp:19-32 / 78-91: walk i up while the
// centroid is left of the split, otherwise swap with the back. The swap order
// decides the final triangle order, so it is kept exactly. Returns the first
// index of the right part.
int partitionAtMidpoint(std::vector<Tri>& tris, int first, int count, const Box& box)
{
    const int axis = box.longestAxis();
    const float split = (box.lo[axis] + box.hi[axis]) * 0.5f;
    int i = first, j = first + count - 1;
    while (i <= j) {
        if (tris[i].centroid[axis] < split) ++i;
        else std::swap(tris[i], tris[j--]);
    }
    return i;
}

// BuildBVH + SplitBVHNode, ModelLoader.cpp:10-120, without recursion: a node's
// two children are appended when it is split, then the left subtree is
// completed before the right one, which is the order the recursive original
// appends in. Node 0/1 are the parent's children. Returns false when the parent
// stays a leaf.
bool buildMidpointBvh(std::vector<Tri>& tris, const Box& parentBox, std::vector<BuildNode>& nodes)
{
    const int total = (int)tris.size();
    if (total <= 2) return false;
    const int mid = partitionAtMidpoint(tris, 0, total, parentBox);
    if (mid == 0 || mid == total) return false;
    auto makeChild = [&](int first, int count) {
        BuildNode n;
        n.first = first; n.count = count;
        n.box = tightBox(tris, first, count);
        return n;
    };
    nodes.push_back(makeChild(0, mid));
    nodes.push_back(makeChild(mid, total - mid));
    std::vector<int> todo = { 1, 0 };
    while (!todo.empty()) {
        const int idx = todo.back();
        todo.pop_back();
        const BuildNode node = nodes[idx];
        if (node.count <= 2) continue;
        const int cut = partitionAtMidpoint(tris, node.first, node.count, node.box);
        const int leftCount = cut - node.first;
        if (leftCount == 0 || leftCount == node.count) continue;
        const int l = (int)nodes.size(), r = l + 1;
        nodes[idx].left = l;
        nodes[idx].right = r;
        nodes.push_back(makeChild(node.first, leftCount));
        nodes.push_back(makeChild(cut, node.count - leftCount));
        todo.push_back(r);
        todo.push_back(l);
    }
    return true;
}

struct Model {   // Model, HardwareRenderer.h:77-83
    int triangleStartIndex = -1, triangleCount = 0;
    rt_parent_bvh_node parentBVH{};
    int nodeStart = 0, nodeCount = 0;
};

struct SceneObject {   // SceneObject, HardwareRenderer.h:85-92
    std::string modelName;
    Vec3 position{ 0, 0, 0 }, rotation{ 0, 0, 0 }, scale{ 1, 1, 1 };
    int materialIndex = -1;
};

bool sameMaterial(const rt_gpu_material& a, const rt_gpu_material& b)   // GPUMaterial::operator==
{
    return a.albedo[0] == b.albedo[0] && a.albedo[1] == b.albedo[1] && a.albedo[2] == b.albedo[2] &&
           a.smoothness == b.smoothness && a.fuzziness == b.fuzziness && a.emission == b.emission &&
           a.refractiveIndex == b.refractiveIndex && a.absorbtion[0] == b.absorbtion[0] &&
           a.absorbtion[1] == b.absorbtion[1] && a.absorbtion[2] == b.absorbtion[2];
}

struct Float4 { float x, y, z, w; };
inline Float4 f4(Vec3 v) { return { v.x, v.y, v.z, 0.0f }; }

} // namespace

struct rt_scene {
    // the reference's member vectors (HardwareRenderer.h:174-226)
    std::vector<rt_parent_bvh_node> parentBVH;
    std::vector<rt_gpu_object> gpuObjects;
    std::vector<rt_gpu_material> materials;
    std::vector<Float4> triV0, triV1, triV2, triN0, triN1, triN2, triUV;
    std::vector<Float4> aabbMins, aabbMaxs;
    std::vector<int32_t> left, right, triStart, triCount;
    std::unordered_map<std::string, Model> models;
    std::vector<SceneObject> objects;
    std::string error;
};

namespace {
thread_local std::string g_sceneError;
int fail(rt_scene* s, int code, const std::string& msg)
{
    if (s) s->error = msg;
    g_sceneError = msg;
    return code;
}

// Handler of the exported functions' function-try-blocks: no C++ exception crosses the C ABI.
int scene_caught(rt_scene* s) noexcept
{
    int code = RT_ERR_INVALID_ARG;
    const char* text = "internal error";
    try { throw; }
    catch (const std::bad_alloc&) { code = RT_ERR_NOMEM; text = "out of memory"; }
    catch (const std::length_error&) { code = RT_ERR_NOMEM; text = "too large"; }
    catch (...) { }
    try { if (s) s->error.assign(text); g_sceneError.assign(text); } catch (...) { }
    return code;
}

int nodeDepth(const rt_scene& s, int node, int& maxLeaf, int guard)
{
    if (guard > 4096) return guard;
    const int l = s.left[node], r = s.right[node];
    if (l == -1 && r == -1) { maxLeaf = std::max(maxLeaf, (int)s.triCount[node]); return 1; }
    const int dl = l != -1 ? nodeDepth(s, l, maxLeaf, guard + 1) : 0;
    const int dr = r != -1 ? nodeDepth(s, r, maxLeaf, guard + 1) : 0;
    return 1 + std::max(dl, dr);
}
} // namespace

extern "C" {

rt_scene* rt_scene_create(void) { return new (std::nothrow) rt_scene(); }
void rt_scene_destroy(rt_scene* s) { delete s; }
const char* rt_scene_last_error(rt_scene* s) { return s ? s->error.c_str() : g_sceneError.c_str(); }

void rt_default_material(rt_gpu_material* m)
{
    std::memset(m, 0, sizeof(*m));
    m->albedo[0] = m->albedo[1] = m->albedo[2] = 1.0f;
}

void rt_default_push_constants(rt_push_constants* pc)
{
    std::memset(pc, 0, sizeof(*pc));
    pc->raysPerPixel = 1;
    pc->maxBounces = 3;
    pc->accumulateFrames = 1;
    pc->frame = 0;
    pc->sunDirection[0] = -0.5f; pc->sunDirection[1] = -1.0f; pc->sunDirection[2] = -0.3f;
    pc->sunIntensity = 5.0f;
    pc->sunColour[0] = 1.0f; pc->sunColour[1] = 0.95f; pc->sunColour[2] = 0.85f;
    pc->renderMode = 0;
    pc->triangleTestThreshold = 150;
    pc->bvhNodeTestThreshold = 120;
    pc->depthDebugScale = 50.0f;
}

void rt_default_camera(rt_camera_settings* cam)
{
    cam->cameraPosition[0] = 0.0f; cam->cameraPosition[1] = 1.0f; cam->cameraPosition[2] = 5.0f;
    cam->cameraLookDirection[0] = 0.0f; cam->cameraLookDirection[1] = 0.0f; cam->cameraLookDirection[2] = -1.0f;
    cam->cameraFov = 90.0f;
    cam->focusDistance = 10.0f;
    cam->defocusAngle = 0.0f;
}

int rt_scene_add_material(rt_scene* s, const rt_gpu_material* m) try
{
    if (!s || !m) return fail(s, RT_ERR_INVALID_ARG, "rt_scene_add_material: null argument");
    s->materials.push_back(*m);
    return (int)s->materials.size() - 1;
} catch (...) { return scene_caught(s); }

int rt_scene_material_index(rt_scene* s, const rt_gpu_material* m) try
{
    if (!s || !m) return -1;
    for (size_t i = 0; i < s->materials.size(); ++i)
        if (sameMaterial(s->materials[i], *m)) return (int)i;
    return -1;
} catch (...) { return scene_caught(s); }

int rt_scene_set_material(rt_scene* s, int index, const rt_gpu_material* m) try
{
    if (!s || !m || index < 0 || (size_t)index >= s->materials.size())
        return fail(s, RT_ERR_INVALID_ARG, "rt_scene_set_material: bad index");
    s->materials[index] = *m;
    return RT_OK;
} catch (...) { return scene_caught(s); }

int rt_scene_load_model(rt_scene* s, const char* obj_path) try
{
    if (!s || !obj_path) return fail(s, RT_ERR_INVALID_ARG, "rt_scene_load_model: null argument");
    const std::string path(obj_path);
    const size_t dotPos = path.find_last_of('.');
    if (dotPos == std::string::npos || path.substr(dotPos + 1) != "obj")
        return fail(s, RT_ERR_UNSUPPORTED, "Only .obj files are supported.");
    if (s->models.count(path)) return RT_OK;

    std::vector<ObjVertex> vertices;
    std::string err;
    if (!readObj(path, vertices, err)) return fail(s, RT_ERR_IO, err);

    // per-triangle set-up and the integer-snapped model box, HardwareRenderer.cpp:1184-1253
    std::vector<Tri> tris;
    tris.reserve(vertices.size() / 3);
    float lo[3] = { (float)INT_MAX, (float)INT_MAX, (float)INT_MAX };
    float hi[3] = { (float)INT_MIN, (float)INT_MIN, (float)INT_MIN };
    for (size_t i = 0; i + 2 < vertices.size(); i += 3) {
        Tri t;
        t.v0 = vertices[i].position; t.v1 = vertices[i + 1].position; t.v2 = vertices[i + 2].position;
        t.centroid = (t.v0 + t.v1 + t.v2) * 0.33f;
        if (!isZero(vertices[i].normal) && !isZero(vertices[i + 1].normal) && !isZero(vertices[i + 2].normal)) {
            t.n0 = normalize(vertices[i].normal);
            t.n1 = normalize(vertices[i + 1].normal);
            t.n2 = normalize(vertices[i + 2].normal);
        } else {
            const Vec3 face = normalize(cross(t.v1 - t.v0, t.v2 - t.v0));
            t.n0 = t.n1 = t.n2 = face;
        }
        for (const Vec3* p : { &t.v0, &t.v1, &t.v2 }) {
            for (int a = 0; a < 3; ++a) {
                const float c = (*p)[a];
                if (c < lo[a]) lo[a] = (float)static_cast<int>(std::floor(c));
            }
            for (int a = 0; a < 3; ++a) {
                const float c = (*p)[a];
                if (c > hi[a]) hi[a] = (float)static_cast<int>(std::ceil(c));
            }
        }
        tris.push_back(t);
    }
    for (int a = 0; a < 3; ++a)
        if (hi[a] - lo[a] == 0) { hi[a] += 1; lo[a] -= 1; }

    Model model;
    rt_parent_bvh_node parent{};
    parent.node.aabb.min[0] = lo[0]; parent.node.aabb.min[1] = lo[1]; parent.node.aabb.min[2] = lo[2];
    parent.node.aabb.max[0] = hi[0]; parent.node.aabb.max[1] = hi[1]; parent.node.aabb.max[2] = hi[2];
    parent.objectIndex = -1;
    parent.node.leftChild = -1;
    parent.node.rightChild = -1;
    parent.node.triangleStartIndex = 0;
    parent.node.triangleCount = (int)tris.size();

    const int nodeOffset = (int)s->aabbMins.size();
    const int triOffset = (int)s->triV0.size();
    std::vector<BuildNode> nodes;
    const Box parentBox{ { lo[0], lo[1], lo[2] }, { hi[0], hi[1], hi[2] } };
    if (!tris.empty() && buildMidpointBvh(tris, parentBox, nodes)) {
        parent.node.leftChild = nodeOffset;
        parent.node.rightChild = nodeOffset + 1;
    }

    // rebase to global indices and append (HardwareRenderer.cpp:1272-1309)
    for (const BuildNode& n : nodes) {
        s->aabbMins.push_back(f4(n.box.lo));
        s->aabbMaxs.push_back(f4(n.box.hi));
        s->left.push_back(n.left != -1 ? n.left + nodeOffset : -1);
        s->right.push_back(n.right != -1 ? n.right + nodeOffset : -1);
        s->triStart.push_back(n.first + triOffset);
        s->triCount.push_back(n.count);
    }
    parent.node.triangleStartIndex += triOffset;
    for (const Tri& t : tris) {
        s->triV0.push_back(f4(t.v0)); s->triV1.push_back(f4(t.v1)); s->triV2.push_back(f4(t.v2));
        s->triN0.push_back(f4(t.n0)); s->triN1.push_back(f4(t.n1)); s->triN2.push_back(f4(t.n2));
        s->triUV.push_back({ 0.0f, 0.0f, 0.0f, 0.0f });
    }
    model.triangleStartIndex = triOffset;
    model.triangleCount = (int)tris.size();
    model.parentBVH = parent;
    model.nodeStart = nodeOffset;
    model.nodeCount = (int)nodes.size();
    s->models[path] = model;
    return RT_OK;
} catch (...) { return scene_caught(s); }

int rt_scene_add_object(rt_scene* s, const char* model_path, const float position[3], const float rotation_deg[3],
                        const float scale_[3], const rt_gpu_material* material) try
{
    if (!s || !model_path || !position || !rotation_deg || !scale_)
        return fail(s, RT_ERR_INVALID_ARG, "rt_scene_add_object: null argument");
    SceneObject o;
    o.modelName = model_path;
    o.position = { position[0], position[1], position[2] };
    o.rotation = { rotation_deg[0], rotation_deg[1], rotation_deg[2] };
    o.scale = { scale_[0], scale_[1], scale_[2] };
    o.materialIndex = material ? rt_scene_material_index(s, material) : -1;
    s->objects.push_back(o);
    return (int)s->objects.size() - 1;
} catch (...) { return scene_caught(s); }

int rt_scene_set_object(rt_scene* s, int index, const float position[3], const float rotation_deg[3],
                        const float scale_[3], int material_index) try
{
    if (!s || index < 0 || (size_t)index >= s->objects.size())
        return fail(s, RT_ERR_INVALID_ARG, "rt_scene_set_object: bad index");
    SceneObject& o = s->objects[index];
    if (position) o.position = { position[0], position[1], position[2] };
    if (rotation_deg) o.rotation = { rotation_deg[0], rotation_deg[1], rotation_deg[2] };
    if (scale_) o.scale = { scale_[0], scale_[1], scale_[2] };
    o.materialIndex = material_index;
    return RT_OK;
} catch (...) { return scene_caught(s); }

// The list half of ClearSceneData (HardwareRenderer.cpp:773-785): the loaded models stay cached.
int rt_scene_clear_objects(rt_scene* s)
{
    if (!s) return fail(s, RT_ERR_INVALID_ARG, "rt_scene_clear_objects: null scene");
    s->objects.clear();
    return RT_OK;
}

int rt_scene_clear_materials(rt_scene* s)
{
    if (!s) return fail(s, RT_ERR_INVALID_ARG, "rt_scene_clear_materials: null scene");
    s->materials.clear();
    return RT_OK;
}

int rt_scene_object_count(rt_scene* s) { return s ? (int)s->objects.size() : 0; }
int rt_scene_material_count(rt_scene* s) { return s ? (int)s->materials.size() : 0; }

int rt_scene_get_object(rt_scene* s, int index, char* model_path, size_t model_path_cap, float position[3],
                        float rotation_deg[3], float scale_[3], int* material_index) try
{
    if (!s || index < 0 || (size_t)index >= s->objects.size())
        return fail(s, RT_ERR_INVALID_ARG, "rt_scene_get_object: bad index");
    const SceneObject& o = s->objects[index];
    if (model_path) {
        if (o.modelName.size() + 1 > model_path_cap)
            return fail(s, RT_ERR_INVALID_ARG, "rt_scene_get_object: model path needs " + std::to_string(o.modelName.size() + 1) + " bytes");
        std::memcpy(model_path, o.modelName.c_str(), o.modelName.size() + 1);
    }
    if (position) { position[0] = o.position.x; position[1] = o.position.y; position[2] = o.position.z; }
    if (rotation_deg) { rotation_deg[0] = o.rotation.x; rotation_deg[1] = o.rotation.y; rotation_deg[2] = o.rotation.z; }
    if (scale_) { scale_[0] = o.scale.x; scale_[1] = o.scale.y; scale_[2] = o.scale.z; }
    if (material_index) *material_index = o.materialIndex;
    return RT_OK;
} catch (...) { return scene_caught(s); }

int rt_scene_get_material(rt_scene* s, int index, rt_gpu_material* out) try
{
    if (!s || !out || index < 0 || (size_t)index >= s->materials.size())
        return fail(s, RT_ERR_INVALID_ARG, "rt_scene_get_material: bad index");
    *out = s->materials[index];
    return RT_OK;
} catch (...) { return scene_caught(s); }

int rt_scene_buffer_data(rt_scene* s) try
{
    if (!s) return fail(s, RT_ERR_INVALID_ARG, "rt_scene_buffer_data: null scene");
    s->gpuObjects.clear();
    s->parentBVH.clear();
    for (const SceneObject& obj : s->objects) {
        auto it = s->models.find(obj.modelName);
        if (it == s->models.end()) return fail(s, RT_ERR_SCENE_INVALID, "object references a model that was not loaded: " + obj.modelName);
        const Model& model = it->second;

        // ConvertSceneObjectToGPUObject, HardwareRenderer.cpp:502-551
        Mat4 m = Mat4::identity();
        m = translate(m, obj.position);
        m = rotate(m, radiansf(obj.rotation.x), { 1, 0, 0 });
        m = rotate(m, radiansf(obj.rotation.y), { 0, 1, 0 });
        m = rotate(m, radiansf(obj.rotation.z), { 0, 0, 1 });
        m = scale(m, obj.scale);

        const float* bl = model.parentBVH.node.aabb.min;
        const float* bh = model.parentBVH.node.aabb.max;
        float newLo[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, newHi[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
        for (int corner = 0; corner < 8; ++corner) {
            const float cx = (corner & 1) ? bh[0] : bl[0];
            const float cy = (corner & 2) ? bh[1] : bl[1];
            const float cz = (corner & 4) ? bh[2] : bl[2];
            float w[4];
            mulPoint(m, cx, cy, cz, 1.0f, w);
            for (int a = 0; a < 3; ++a) {
                newLo[a] = w[a] < newLo[a] ? w[a] : newLo[a];   // glm::min
                newHi[a] = newHi[a] < w[a] ? w[a] : newHi[a];   // glm::max
            }
        }

        rt_gpu_object g{};
        rt_mat4_inverse(&m.col[0][0], g.inverseTransform);
        g.materialIndex = obj.materialIndex;
        g.triangleStartIndex = model.triangleStartIndex;
        g.triangleCount = model.triangleCount;

        rt_parent_bvh_node p = model.parentBVH;
        for (int a = 0; a < 3; ++a) { p.node.aabb.min[a] = newLo[a]; p.node.aabb.max[a] = newHi[a]; }
        p.objectIndex = (int)s->gpuObjects.size();
        s->gpuObjects.push_back(g);
        s->parentBVH.push_back(p);
    }
    return RT_OK;
} catch (...) { return scene_caught(s); }

int rt_scene_get_buffers(rt_scene* s, rt_scene_buffers* out) try
{
    if (!s || !out) return fail(s, RT_ERR_INVALID_ARG, "rt_scene_get_buffers: null argument");
    auto view = [](const auto& v) { return rt_buffer_view{ v.empty() ? nullptr : (const void*)v.data(), v.size() }; };
    out->parentBVH = view(s->parentBVH);
    out->objects = view(s->gpuObjects);
    out->materials = view(s->materials);
    out->triangleV0s = view(s->triV0); out->triangleV1s = view(s->triV1); out->triangleV2s = view(s->triV2);
    out->triangleN0s = view(s->triN0); out->triangleN1s = view(s->triN1); out->triangleN2s = view(s->triN2);
    out->triangleUV0s = view(s->triUV); out->triangleUV1s = view(s->triUV); out->triangleUV2s = view(s->triUV);
    out->aabbMins = view(s->aabbMins); out->aabbMaxs = view(s->aabbMaxs);
    out->bvhNodeLeftChildren = view(s->left); out->bvhNodeRightChildren = view(s->right);
    out->bvhNodeTriangleStartIndices = view(s->triStart); out->bvhNodeTriangleCounts = view(s->triCount);
    return RT_OK;
} catch (...) { return scene_caught(s); }

int rt_scene_model_info(rt_scene* s, const char* model_path, int* triangles, int* nodes, int* max_depth, int* max_leaf) try
{
    if (!s || !model_path) return fail(s, RT_ERR_INVALID_ARG, "rt_scene_model_info: null argument");
    auto it = s->models.find(model_path);
    if (it == s->models.end()) return fail(s, RT_ERR_INVALID_ARG, std::string("model not loaded: ") + model_path);
    const Model& m = it->second;
    int depth = 0, leaf = 0;
    if (m.parentBVH.node.leftChild == -1 && m.parentBVH.node.rightChild == -1) {
        leaf = m.triangleCount;
    } else {
        depth = std::max(nodeDepth(*s, m.parentBVH.node.leftChild, leaf, 0), nodeDepth(*s, m.parentBVH.node.rightChild, leaf, 0));
    }
    if (triangles) *triangles = m.triangleCount;
    if (nodes) *nodes = m.nodeCount;
    if (max_depth) *max_depth = depth;
    if (max_leaf) *max_leaf = leaf;
    return RT_OK;
} catch (...) { return scene_caught(s); }

int rt_scene_init_default(rt_scene* s, const char* models_dir, const char* dragon_file) try
{
    if (!s || !models_dir) return fail(s, RT_ERR_INVALID_ARG, "rt_scene_init_default: null argument");
    auto material = [](float r, float g, float b) {
        rt_gpu_material m;
        rt_default_material(&m);
        m.albedo[0] = r; m.albedo[1] = g; m.albedo[2] = b;
        return m;
    };
    rt_gpu_material ground = material(0.5f, 0.5f, 0.5f);
    rt_gpu_material diffuse = material(0.8f, 0.8f, 0.8f);
    rt_gpu_material glass = material(1.0f, 1.0f, 1.0f);       glass.refractiveIndex = 1.5f;
    rt_gpu_material mirror = material(1.0f, 1.0f, 1.0f);      mirror.smoothness = 1.0f;
    rt_gpu_material emissive = material(1.0f, 1.0f, 1.0f);    emissive.emission = 15.0f;
    rt_gpu_material dullGold = material(1.0f, 0.71f, 0.29f);  dullGold.smoothness = 0.8f; dullGold.fuzziness = 0.4f;
    rt_gpu_material shinyGold = material(1.0f, 0.71f, 0.29f); shinyGold.smoothness = 1.0f;
    rt_gpu_material red = material(0.8f, 0.1f, 0.1f);
    rt_gpu_material green = material(0.1f, 0.8f, 0.1f);
    for (const rt_gpu_material* m : { &ground, &diffuse, &glass, &mirror, &emissive, &dullGold, &shinyGold, &red, &green })
        rt_scene_add_material(s, m);

    const std::string dir(models_dir);
    const std::string quad = dir + "/flat_quad.obj", sphere = dir + "/icosphere.obj";
    const std::string dragon = dir + "/" + (dragon_file ? dragon_file : "dragon.obj");
    for (const std::string* p : { &quad, &sphere, &dragon }) {
        const int rc = rt_scene_load_model(s, p->c_str());
        if (rc != RT_OK) return rc;
    }
    struct Placement { const std::string* model; float p[3], r[3], sc[3]; const rt_gpu_material* mat; };
    const Placement placements[8] = {
        { &sphere, { 0, 10, 0 }, { 0, 0, 0 }, { 2, 2, 2 }, &emissive },
        { &dragon, { 0, 2, 0 }, { 0, 195, 0 }, { 1, 1, 1 }, &dullGold },
        { &quad, { 0, 0, 0 }, { 0, 0, 0 }, { 5, 5, 5 }, &ground },
        { &quad, { 0, 10, 0 }, { 0, 0, 180 }, { 5, 5, 5 }, &ground },
        { &quad, { 0, 5, -5 }, { 90, 0, 0 }, { 5, 5, 5 }, &ground },
        { &quad, { -5, 5, 0 }, { 0, 0, 90 }, { 5, 5, 5 }, &red },
        { &quad, { 5, 5, 0 }, { 0, 0, -90 }, { 5, 5, 5 }, &green },
        { &quad, { 0, 5, 5 }, { -90, 0, 0 }, { 5, 5, 5 }, &ground },
    };
    for (const Placement& pl : placements) rt_scene_add_object(s, pl.model->c_str(), pl.p, pl.r, pl.sc, pl.mat);
    return rt_scene_buffer_data(s);
} catch (...) { return scene_caught(s); }

int rt_camera_to_push_constants(const rt_camera_settings* cam, int width, int height, rt_push_constants* pc) try
{
    if (!cam || !pc || width <= 0 || height <= 0) return fail(nullptr, RT_ERR_INVALID_ARG, "rt_camera_to_push_constants: bad argument");
    const float theta = radiansf(cam->cameraFov);
    const double h = std::tan(theta / 2.0);
    const float viewportHeight = (float)(2 * h * cam->focusDistance);
    const float viewportWidth = (float)(viewportHeight * (double(width) / height));

    const Vec3 look{ cam->cameraLookDirection[0], cam->cameraLookDirection[1], cam->cameraLookDirection[2] };
    const Vec3 position{ cam->cameraPosition[0], cam->cameraPosition[1], cam->cameraPosition[2] };
    const Vec3 up{ 0, 1, 0 };
    const Vec3 w = normalize(-look);
    const Vec3 u = normalize(cross(up, w));
    const Vec3 v = cross(w, u);

    const Vec3 viewportU = u * viewportWidth;
    const Vec3 viewportV = (-v) * viewportHeight;
    const Vec3 pixelDeltaU = viewportU / (float)width;
    const Vec3 pixelDeltaV = viewportV / (float)height;
    const Vec3 upperLeft = position - (w * cam->focusDistance) - viewportU / 2.0f - viewportV / 2.0f;
    const Vec3 pixel00 = upperLeft + (pixelDeltaU + pixelDeltaV) * 0.5f;
    const double defocusRadius = cam->focusDistance * std::tan((cam->defocusAngle / 2.0) * 0.01745329251994329576923690768489);
    const float dr = (float)defocusRadius;

    auto put = [](float dst[3], Vec3 a) { dst[0] = a.x; dst[1] = a.y; dst[2] = a.z; };
    put(pc->pixel00Location, pixel00);
    put(pc->pixelDeltaU, pixelDeltaU);
    put(pc->pixelDeltaV, pixelDeltaV);
    put(pc->cameraPosition, position);
    pc->defocusAngle = cam->defocusAngle;
    put(pc->defocusDiskU, Vec3{ dr * u.x, dr * u.y, dr * u.z });
    put(pc->defocusDiskV, Vec3{ dr * v.x, dr * v.y, dr * v.z });
    return RT_OK;
} catch (...) { return scene_caught(nullptr); }

void rt_rgba32f_to_rgba8(const float* src, uint8_t* dst, size_t pixel_count)
{
    for (size_t i = 0; i < pixel_count; ++i) {
        for (int c = 0; c < 3; ++c) {
            float v = src[4 * i + c];
            v = std::min(std::max(v, 0.0f), 1.0f);
            dst[4 * i + c] = static_cast<uint8_t>(v * 255.0f + 0.5f);
        }
        dst[4 * i + 3] = 255;
    }
}

// PNG file with 8-bit RGBA pixels, the format WriteDrawImageToFile hands to stbi_write_png
// (HardwareRenderer.cpp:1127-1134). The pixel rows go into stored (uncompressed) deflate blocks with
// filter type 0, so every decoder returns exactly the bytes passed in; only the file size differs
// from what a compressing writer produces.
int rt_write_png_rgba8(const char* path, const uint8_t* rgba, int width, int height) try
{
    if (!path || !rgba || width <= 0 || height <= 0) return RT_ERR_INVALID_ARG;
    static uint32_t crcTable[256];
    static bool haveTable = false;
    if (!haveTable) {
        for (uint32_t n = 0; n < 256; ++n) {
            uint32_t c = n;
            for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xedb88320u ^ (c >> 1) : c >> 1;
            crcTable[n] = c;
        }
        haveTable = true;
    }
    auto be32 = [](std::vector<uint8_t>& v, uint32_t x) {
        v.push_back(uint8_t(x >> 24)); v.push_back(uint8_t(x >> 16)); v.push_back(uint8_t(x >> 8)); v.push_back(uint8_t(x));
    };
    std::vector<uint8_t> file = { 0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a };
    auto chunk = [&](const char type[4], const std::vector<uint8_t>& data) {
        be32(file, (uint32_t)data.size());
        const size_t start = file.size();
        file.insert(file.end(), type, type + 4);
        file.insert(file.end(), data.begin(), data.end());
        uint32_t c = 0xffffffffu;
        for (size_t i = start; i < file.size(); ++i) c = crcTable[(c ^ file[i]) & 0xffu] ^ (c >> 8);
        be32(file, c ^ 0xffffffffu);
    };
    std::vector<uint8_t> ihdr;
    be32(ihdr, (uint32_t)width); be32(ihdr, (uint32_t)height);
    ihdr.push_back(8); ihdr.push_back(6); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);   // 8-bit RGBA, no interlace
    chunk("IHDR", ihdr);

    const size_t rowBytes = (size_t)width * 4 + 1;   // filter byte + pixels
    const size_t rawBytes = rowBytes * (size_t)height;
    std::vector<uint8_t> raw(rawBytes);
    for (int y = 0; y < height; ++y) {
        raw[(size_t)y * rowBytes] = 0;
        std::memcpy(&raw[(size_t)y * rowBytes + 1], rgba + (size_t)y * width * 4, (size_t)width * 4);
    }
    uint32_t a = 1, b = 0;   // Adler-32 of the raw stream
    for (size_t i = 0; i < rawBytes;) {
        const size_t n = std::min<size_t>(5552, rawBytes - i);
        for (size_t k = 0; k < n; ++k) { a += raw[i + k]; b += a; }
        a %= 65521u; b %= 65521u;
        i += n;
    }
    std::vector<uint8_t> z;
    z.reserve(rawBytes + rawBytes / 65535 * 5 + 16);
    z.push_back(0x78); z.push_back(0x01);
    for (size_t i = 0; i < rawBytes;) {
        const size_t n = std::min<size_t>(65535, rawBytes - i);
        z.push_back(i + n == rawBytes ? 1 : 0);
        z.push_back(uint8_t(n)); z.push_back(uint8_t(n >> 8));
        z.push_back(uint8_t(~n)); z.push_back(uint8_t((~n) >> 8));
        z.insert(z.end(), raw.begin() + (std::ptrdiff_t)i, raw.begin() + (std::ptrdiff_t)(i + n));
        i += n;
    }
    be32(z, (b << 16) | a);
    chunk("IDAT", z);
    chunk("IEND", {});

    FILE* f = std::fopen(path, "wb");
    if (!f) return RT_ERR_IO;
    const bool ok = std::fwrite(file.data(), 1, file.size(), f) == file.size();
    return (std::fclose(f) == 0 && ok) ? RT_OK : RT_ERR_IO;
} catch (...) { return scene_caught(nullptr); }

} // extern "C"
