// rt_kernels.hip - gfx950 (MI355X) kernels for the per-pixel path-tracing path of
// the reference renderer: everything its compute shader does in main()
// (src/Shaders/Compute/raytrace.comp:857-875 and the functions it calls, plus
// src/Shaders/Compute/Random.glsl), written for 64-wide wavefronts.
//
// This file is compiled twice: -DRT_FLAVOUR_STRICT with -ffp-contract=off (plain
// IEEE binary32, the build that is compared value-for-value with the CPU
// oracle) and -DRT_FLAVOUR_FAST with FMA contraction allowed.
//
// Structure of the megakernel (one lane = one pixel, one wave = one 8x8 tile):
//   * the object loop of GetHit (raytrace.comp:458) is wave-uniform, so the
//     per-object record is fetched with scalar loads into SGPRs;
//   * the per-ray `inverse(mat4)` of raytrace.comp:485 is hoisted to upload time;
//   * the BVH walk visits 64-byte two-child records: both child boxes are
//     tested in one visit; the child the reference would pop first is entered
//     at once, the other is pushed together with its entry distance so that
//     the reference's pop-time interval test (against the *then current*
//     closest hit) is reproduced by one compare, without re-reading the box;
//   * the traversal stack lives in LDS, laid out [depth][lane] (8-byte
//     entries: conflict-free for any mix of depths across lanes);
//   * normals are fetched only for the winning triangle of a GetHit call, and
//     the world-space hit record is rebuilt from (object, triangle, u, v, t)
//     with the same arithmetic the shader applies at every accepted hit;
//   * shadow rays stop at the first accepted hit (ShadowTransmission only asks
//     whether anything was hit, raytrace.comp:612-621). The counting build
//     (render modes 3/4 and the statistics pass) never stops early, so its
//     counters are those of the reference traversal.
// Every floating-point expression keeps the operand order of the GLSL source.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rt_types.h"
#include "../../include/rt_math.h"
#include "rt_device.h"

#if defined(RT_FLAVOUR_FAST)
#define RT_SYM(name) name##_fast
#else
#define RT_SYM(name) name##_strict
#endif

// Diagnostic build only (-DRT_SECTION_TIMERS): wave-level cycle counts per section of the lock-step
// kernel, summed over all waves. No such code exists in the normal build.
// Diagnostic build (-DRT_WORK_COUNTERS): the box tests, triangle tests, GetHit calls and record fetches the two-
// samples-per-lane kernels really perform, whichever trees and order they walk (the counting build of kernel 0
// only knows the reference's). Every lane adds into its own LDS word; a wave adds its sums to g_workCounters
// when it ends; rt_debug_work_counters() reads and clears them.
#ifdef RT_WORK_COUNTERS
// 0 box tests, 1 triangle tests, 2 GetHit calls, 3 record fetches; lane slots offered / used (RT_WORK_LANES: 64 per wave that
// passes, 1 per lane that is active there): 4 / 5 four-child visits, 6 / 7 triangle tests of the walks' leaves, 8 / 9 walls
// tested lane by lane
#define RT_WORK_SLOTS 10
__device__ unsigned long long g_workCounters[RT_WORK_SLOTS];
__device__ __noinline__ unsigned* rt_work_lds() { __shared__ unsigned a[RT_WORK_SLOTS * 64]; return a; }
#define RT_WORK(k, n) do { rt_work_lds()[(k) * 64 + (threadIdx.x & 63)] += (unsigned)(n); } while (0)
#define RT_WORK_LANES(k) do { if ((int)(threadIdx.x & 63) == __ffsll((long long)__ballot(true)) - 1) RT_WORK(k, 64); RT_WORK((k) + 1, 1); } while (0)
#define RT_WORK_BEGIN() do { unsigned* a_ = rt_work_lds(); for (int k_ = 0; k_ < RT_WORK_SLOTS; ++k_) a_[k_ * 64 + (threadIdx.x & 63)] = 0; } while (0)
#define RT_WORK_END() do { const unsigned* a_ = rt_work_lds(); \
        if ((int)threadIdx.x == __ffsll((long long)__ballot(true)) - 1) \
            for (int k_ = 0; k_ < RT_WORK_SLOTS; ++k_) { unsigned long long s_ = 0; for (int l_ = 0; l_ < 64; ++l_) s_ += a_[k_ * 64 + l_]; atomicAdd(&g_workCounters[k_], s_); } } while (0)
#else
#define RT_WORK(k, n) do { } while (0)
#define RT_WORK_LANES(k) do { } while (0)
#define RT_WORK_BEGIN() do { } while (0)
#define RT_WORK_END() do { } while (0)
#endif

#ifdef RT_SECTION_TIMERS
__device__ unsigned long long g_sectionCycles[8];
__device__ __noinline__ unsigned* rt_sec_lds() { __shared__ unsigned a[8]; return a; }
#define RT_SEC_BEGIN(t) const unsigned t = (unsigned)__builtin_amdgcn_s_memtime()
#define RT_SEC_END(k, t) do { unsigned* a_ = rt_sec_lds(); a_[k] += (unsigned)__builtin_amdgcn_s_memtime() - (t); } while (0)
#else
#define RT_SEC_BEGIN(t) do { } while (0)
#define RT_SEC_END(k, t) do { } while (0)
#endif

namespace {

struct V3 { float x, y, z; };

__device__ __forceinline__ V3 mk(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator*(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ V3 operator-(V3 a) { return mk(-a.x, -a.y, -a.z); }
__device__ __forceinline__ float dot3(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross3(V3 a, V3 b)
{
    return mk(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);
}
__device__ __forceinline__ float length3(V3 a) { return sqrtf(dot3(a, a)); }
__device__ __forceinline__ V3 normalize3(V3 a) { float inv = 1.0f / sqrtf(dot3(a, a)); return a * inv; }
// GLSL max/min: the second operand wins only on a strict compare.
__device__ __forceinline__ float gmax(float a, float b) { return a < b ? b : a; }
__device__ __forceinline__ float gmin(float a, float b) { return b < a ? b : a; }
__device__ __forceinline__ float gclamp(float x, float lo, float hi) { return gmin(gmax(x, lo), hi); }
__device__ __forceinline__ float gmix(float a, float b, float t) { return a * (1.0f - t) + b * t; }
__device__ __forceinline__ V3 gmix3(V3 a, V3 b, float t) { return mk(gmix(a.x, b.x, t), gmix(a.y, b.y, t), gmix(a.z, b.z, t)); }
// GLSL pow as exp2(y * log2(x)); the three transcendental functions the shader uses come from
// rt_math.h, whose results do not depend on the machine's math library.
__device__ __forceinline__ float gpow(float x, float y) { return rt_exp2f(y * rt_log2f(x)); }
__device__ __forceinline__ V3 ld3(const float* p) { return mk(p[0], p[1], p[2]); }

// ---- Random.glsl ----------------------------------------------------------

// Random.glsl:4-12
__device__ __forceinline__ float rng_hash(uint32_t& seed)
{
    uint32_t s = seed;
    s = (s ^ 61u) ^ (s >> 16);
    s *= 9u;
    s = s ^ (s >> 4);
    s *= 0x27d4eb2du;
    s = s ^ (s >> 15);
    seed = s;
    return rt_div10000f((float)(s % 10000u));     // (float)(s % 10000u) / 10000.0f, bit for bit
}
// Random.glsl:28-32
__device__ __forceinline__ float rng_range(float lo, float hi, uint32_t& seed)
{
    seed += 1u;
    return gmix(lo, hi, rng_hash(seed));
}
// Random.glsl:34-41
__device__ __forceinline__ uint32_t seed_from_coords(uint32_t x, uint32_t y, uint32_t frame, uint32_t i)
{
    uint32_t s = x * 374761393u + y * 668265263u + frame * 1442695040u + i * 88963407u;
    s = (s ^ (s >> 16)) * 0x85ebca6bu;
    s = (s ^ (s >> 13)) * 0xc2b2ae35u;
    s = s ^ (s >> 16);
    return s;
}
// The shader's two rejection loops end only on acceptance, and its generator is a fixed permutation of
// the 32-bit states: a state on a short cycle whose draws are all rejected never leaves the loop (at
// 3840x2160, frame 1010, pixel (3563, 2062), sample 0 does exactly that in RandomUnitVector; the shader
// would hang the GPU there). The reference defines no value for such a sample: the loops here give up
// after RT_REJECTION_CAP rounds and return the zero vector (no lens offset, no fuzz, a Lambert direction
// equal to the normal through MaterialScatter's NearZero branch).
#define RT_REJECTION_CAP 65536

// Random.glsl:68-78 (with RandomVector :62-66 inlined). 1e-160 reads as 0 in f32.
__device__ __forceinline__ V3 random_unit_vector(uint32_t& seed)
{
    seed += 1u;
    for (int round = 0; round < RT_REJECTION_CAP; ++round) {
        seed += 1u;
        float x = rng_range(-1.0f, 1.0f, seed);
        float y = rng_range(-1.0f, 1.0f, seed);
        float z = rng_range(-1.0f, 1.0f, seed);
        V3 p = mk(x, y, z);
        float len = length3(p);
        float lensq = len * len;
        if (0.0f < lensq && lensq <= 1.0f) {
            float l = sqrtf(lensq);
            return mk(p.x / l, p.y / l, p.z / l);
        }
    }
    return mk(0.0f, 0.0f, 0.0f);
}
// Random.glsl:80-88
__device__ __forceinline__ V3 random_on_hemisphere(V3 normal, uint32_t& seed)
{
    seed += 1u;
    V3 u = random_unit_vector(seed);
    return dot3(u, normal) > 0.0f ? u : -u;
}

// ---- traversal --------------------------------------------------------------

struct Ray { V3 o, d; };

// Per-axis constants of a ray for the slab test of raytrace.comp:205-243:
// reciprocal, "invD < 0" swap flag, "abs(dir) < 1e-8" parallel flag.
struct SlabRay {
    V3 o, inv;
    bool negx, negy, negz, parx, pary, parz;
};

__device__ __forceinline__ SlabRay make_slab_ray(V3 o, V3 d)
{
    SlabRay s;
    s.o = o;
    s.parx = fabsf(d.x) < 1e-8f; s.pary = fabsf(d.y) < 1e-8f; s.parz = fabsf(d.z) < 1e-8f;
    s.inv.x = 1.0f / d.x; s.inv.y = 1.0f / d.y; s.inv.z = 1.0f / d.z;
    s.negx = s.inv.x < 0.0f; s.negy = s.inv.y < 0.0f; s.negz = s.inv.z < 0.0f;
    return s;
}

// One axis of IntersectAABB. For a parallel axis the interval is left alone and
// only containment is checked, exactly as the shader does. PAR = false is the
// build for rays with no parallel axis (the caller checks that wave-wide).
// fmaxf/fminf stand in for the GLSL max/min forms: their first operand (the
// running interval bound) is never NaN, so both forms return the same value up
// to the sign of a zero, and only comparisons consume the result.
template <bool PAR>
__device__ __forceinline__ void slab_axis(float o, float inv, bool neg, bool par, float mn, float mx,
                                          float& tEnter, float& tExit, bool& ok)
{
    float t0 = (mn - o) * inv;
    float t1 = (mx - o) * inv;
    float lo = neg ? t1 : t0;
    float hi = neg ? t0 : t1;
    if (PAR) {
        bool inside = !(o < mn || o > mx);
        lo = par ? -INFINITY : lo;
        hi = par ? INFINITY : hi;
        ok = ok && (!par || inside);
    }
    tEnter = fmaxf(tEnter, lo);
    tExit = fminf(tExit, hi);
}

// IntersectAABB split in two: the part that does not depend on the current
// closest hit (returned: ok, entry distance E with tMin folded in), and the
// part that does (C > E), which the caller applies at the moment the reference
// would run the test. The shader's per-axis early-outs
// `if (t.max <= t.min) return false` are equivalent to the single final test
// min(C,T1) > E because t.min only grows and t.max only shrinks.
template <bool PAR>
__device__ __forceinline__ bool slab_box_t(const SlabRay& r, V3 mn, V3 mx, float tMin, float& E)
{
    float e = tMin, x = INFINITY;
    bool ok = true;
    slab_axis<PAR>(r.o.x, r.inv.x, r.negx, r.parx, mn.x, mx.x, e, x, ok);
    slab_axis<PAR>(r.o.y, r.inv.y, r.negy, r.pary, mn.y, mx.y, e, x, ok);
    slab_axis<PAR>(r.o.z, r.inv.z, r.negz, r.parz, mn.z, mx.z, e, x, ok);
    E = e;
    return ok && (x > e);
}
__device__ __forceinline__ bool slab_box(const SlabRay& r, V3 mn, V3 mx, float tMin, float& E)
{
    return slab_box_t<true>(r, mn, mx, tMin, E);
}

// raytrace.comp:345-355
__device__ __forceinline__ float distance_to_aabb(V3 p, V3 mn, V3 mx)
{
    float sq = 0.0f;
    if (p.x < mn.x) sq += (mn.x - p.x) * (mn.x - p.x);
    if (p.x > mx.x) sq += (p.x - mx.x) * (p.x - mx.x);
    if (p.y < mn.y) sq += (mn.y - p.y) * (mn.y - p.y);
    if (p.y > mx.y) sq += (p.y - mx.y) * (p.y - mx.y);
    if (p.z < mn.z) sq += (mn.z - p.z) * (mn.z - p.z);
    if (p.z > mx.z) sq += (p.z - mx.z) * (p.z - mx.z);
    return sqrtf(sq);
}

// rows 0..2 of a 3x4 matrix times (v,w), paired as (c0*x + c1*y) + (c2*z + c3*w)
__device__ __forceinline__ V3 xform(const float* m, V3 v, float w)
{
    return mk((m[0] * v.x + m[1] * v.y) + (m[2] * v.z + m[3] * w),
              (m[4] * v.x + m[5] * v.y) + (m[6] * v.z + m[7] * w),
              (m[8] * v.x + m[9] * v.y) + (m[10] * v.z + m[11] * w));
}

// The slab test in the form the kernels use. A ray's constants are just origin and
// reciprocal direction. The "invD < 0" swap of raytrace.comp:230-236 is taken by min/max of
// the two plane distances, which is the same thing for a box with min <= max (the upload marks
// inverted boxes, which no ray can enter, as dead), and an axis with abs(dir) < 1e-8 gets a NaN
// reciprocal, which makes both plane distances NaN and leaves the interval untouched in
// fmaxf/fminf exactly as the shader's parallel branch does; its containment test only runs
// in the PAR = true build, which callers select wave-wide when some lane has such an axis.
enum : int { F_HAVE_PATH = 1, F_SHADOW = 2, F_PAR_WORLD = 4, F_PAR_LOCAL = 8 };

struct LeanRay { V3 o, inv; };

// Reciprocal direction with NaN on axes the shader treats as parallel; returns
// whether any axis is.
__device__ __forceinline__ bool lean_inverse(V3 d, V3& inv)
{
    const bool px = fabsf(d.x) < 1e-8f, py = fabsf(d.y) < 1e-8f, pz = fabsf(d.z) < 1e-8f;
    const float nan = __int_as_float(0x7fc00000);
    inv.x = px ? nan : 1.0f / d.x;
    inv.y = py ? nan : 1.0f / d.y;
    inv.z = pz ? nan : 1.0f / d.z;
    return px || py || pz;
}

template <bool PAR>
__device__ __forceinline__ bool lean_box(V3 o, V3 inv, V3 mn, V3 mx, float tMin, float& E)
{
    float e = tMin, x = INFINITY;
    bool ok = true;
    {
        float t0 = (mn.x - o.x) * inv.x, t1 = (mx.x - o.x) * inv.x;
        e = fmaxf(e, fminf(t0, t1)); x = fminf(x, fmaxf(t0, t1));
        if (PAR) ok = ok && !((inv.x != inv.x) && (o.x < mn.x || o.x > mx.x));
    }
    {
        float t0 = (mn.y - o.y) * inv.y, t1 = (mx.y - o.y) * inv.y;
        e = fmaxf(e, fminf(t0, t1)); x = fminf(x, fmaxf(t0, t1));
        if (PAR) ok = ok && !((inv.y != inv.y) && (o.y < mn.y || o.y > mx.y));
    }
    {
        float t0 = (mn.z - o.z) * inv.z, t1 = (mx.z - o.z) * inv.z;
        e = fmaxf(e, fminf(t0, t1)); x = fminf(x, fmaxf(t0, t1));
        if (PAR) ok = ok && !((inv.z != inv.z) && (o.z < mn.z || o.z > mx.z));
    }
    E = e;
    return ok && (x > e);
}

// One visit of a two-child record: both children's slab tests (lean_box) at once. The record keeps
// the two boxes side by side per plane, so the plane distances of both children come out of packed
// f32 subtracts and multiplies (two IEEE operations per instruction, same results as the scalar
// form); the interval logic stays per child.
typedef float f32x2 __attribute__((ext_vector_type(2)));
struct NodeVisit { bool okF, okS; float EF, ES; int2 refs; };

template <bool PAR>
__device__ __forceinline__ void node_boxes_loaded(float4 a, float4 b, float4 c, int2 refs, V3 o, V3 inv, float tMin, NodeVisit& r);

template <bool PAR>
__device__ __forceinline__ void node_boxes(const DevNode* node, V3 o, V3 inv, float tMin, NodeVisit& r)
{
    const float4* np = reinterpret_cast<const float4*>(node);
    const float4 a = np[0], b = np[1], c = np[2];
    node_boxes_loaded<PAR>(a, b, c, *reinterpret_cast<const int2*>(np + 3), o, inv, tMin, r);
}

template <bool PAR>
__device__ __forceinline__ void node_boxes_loaded(float4 a, float4 b, float4 c, int2 refs, V3 o, V3 inv, float tMin, NodeVisit& r)
{
    r.refs = refs;
    const f32x2 lox = { a.x, a.y }, loy = { a.z, a.w }, loz = { b.x, b.y };
    const f32x2 hix = { b.z, b.w }, hiy = { c.x, c.y }, hiz = { c.z, c.w };
    const f32x2 ox = { o.x, o.x }, oy = { o.y, o.y }, oz = { o.z, o.z };
    const f32x2 ix = { inv.x, inv.x }, iy = { inv.y, inv.y }, iz = { inv.z, inv.z };
    const f32x2 t0x = (lox - ox) * ix, t1x = (hix - ox) * ix;
    const f32x2 t0y = (loy - oy) * iy, t1y = (hiy - oy) * iy;
    const f32x2 t0z = (loz - oz) * iz, t1z = (hiz - oz) * iz;
    float eF = tMin, xF = INFINITY, eS = tMin, xS = INFINITY;
    eF = fmaxf(eF, fminf(t0x.x, t1x.x)); xF = fminf(xF, fmaxf(t0x.x, t1x.x));
    eS = fmaxf(eS, fminf(t0x.y, t1x.y)); xS = fminf(xS, fmaxf(t0x.y, t1x.y));
    eF = fmaxf(eF, fminf(t0y.x, t1y.x)); xF = fminf(xF, fmaxf(t0y.x, t1y.x));
    eS = fmaxf(eS, fminf(t0y.y, t1y.y)); xS = fminf(xS, fmaxf(t0y.y, t1y.y));
    eF = fmaxf(eF, fminf(t0z.x, t1z.x)); xF = fminf(xF, fmaxf(t0z.x, t1z.x));
    eS = fmaxf(eS, fminf(t0z.y, t1z.y)); xS = fminf(xS, fmaxf(t0z.y, t1z.y));
    bool okF = true, okS = true;
    if (PAR) {
        // an axis the ray runs parallel to (NaN reciprocal): the origin must lie between the planes
        if (inv.x != inv.x) { okF = okF && !(o.x < lox.x || o.x > hix.x); okS = okS && !(o.x < lox.y || o.x > hix.y); }
        if (inv.y != inv.y) { okF = okF && !(o.y < loy.x || o.y > hiy.x); okS = okS && !(o.y < loy.y || o.y > hiy.y); }
        if (inv.z != inv.z) { okF = okF && !(o.z < loz.x || o.z > hiz.x); okS = okS && !(o.z < loz.y || o.z > hiz.y); }
    }
    r.EF = eF; r.ES = eS;
    r.okF = okF && (xF > eF);
    r.okS = okS && (xS > eS);
}

__device__ __forceinline__ void visit_node(const DevScene& S, int cur, V3 o, V3 inv, bool anyParallel, float tMin, NodeVisit& r)
{
    if (anyParallel) node_boxes<true>(S.nodes + cur, o, inv, tMin, r);
    else node_boxes<false>(S.nodes + cur, o, inv, tMin, r);
    RT_WORK(0, (r.refs.x != RT_REF_ABSENT) + (r.refs.y != RT_REF_ABSENT));
    RT_WORK(3, 1);
}

// One visit of a four-child record (DevNode4): four slab tests, the children that the ray can enter in front of
// its closest hit sorted by entry distance. E[k] / R[k], k = 0 nearest; a slot that is missed carries E = +inf.
// Same plane distances and interval logic as lean_box; the closest hit is folded into the exit distance
// (min(exit, closest) > entry is "exit > entry and closest > entry").
struct Visit4 { float E0, E1, E2, E3; int R0, R1, R2, R3; };

// v_min_f32 / v_max_f32 and their three-operand forms as the hardware has them: a quiet NaN operand is
// ignored (the plane distances of an axis the ray runs parallel to are NaN and must leave the interval alone),
// and no operand here is ever a signalling NaN, so the canonicalising copies the compiler puts in front of
// fminf / fmaxf are not needed.
__device__ __forceinline__ float hw_min(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float hw_max(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float hw_min3(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float hw_max3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

// The rare case of a ray with an axis it runs parallel to (NaN reciprocal): a child is only entered when the
// origin lies between its two planes of that axis (raytrace.comp:213-221). Returns the children to drop, one
// bit each. Kept out of line of the common path.
__device__ __noinline__ unsigned visit4_parallel(const DevNode4* node, V3 o, V3 inv)
{
    const float* f = reinterpret_cast<const float*>(node);
    const float oc[3] = { o.x, o.y, o.z }, ic[3] = { inv.x, inv.y, inv.z };
    unsigned drop = 0;
    for (int a = 0; a < 3; ++a) {
        if (!(ic[a] != ic[a])) continue;
        for (int c = 0; c < 4; ++c)
            if (oc[a] < f[4 * a + c] || oc[a] > f[12 + 4 * a + c]) drop |= 1u << c;
    }
    return drop;
}

// SORT = false: the slots stay in the record's order (E = +inf where the ray does not enter), for records whose slots stand in a visiting order
// that has to be kept.
template <bool SORT = true>
__device__ __forceinline__ void visit4(const DevNode4* node, V3 o, V3 inv, bool anyParallel, float tMin, float closest, Visit4& r)
{
    const float4* np = reinterpret_cast<const float4*>(node);
    const float4 lx = np[0], ly = np[1], lz = np[2], hx = np[3], hy = np[4], hz = np[5];
    const int4 refs = *reinterpret_cast<const int4*>(np + 6);
    RT_WORK(0, (refs.x != RT_REF_EMPTY) + (refs.y != RT_REF_EMPTY) + (refs.z != RT_REF_EMPTY) + (refs.w != RT_REF_EMPTY));
    RT_WORK(3, 1);
    const f32x2 ox = { o.x, o.x }, oy = { o.y, o.y }, oz = { o.z, o.z };
    const f32x2 ix = { inv.x, inv.x }, iy = { inv.y, inv.y }, iz = { inv.z, inv.z };
    const f32x2 ax0 = (f32x2{ lx.x, lx.y } - ox) * ix, ax1 = (f32x2{ hx.x, hx.y } - ox) * ix;
    const f32x2 bx0 = (f32x2{ lx.z, lx.w } - ox) * ix, bx1 = (f32x2{ hx.z, hx.w } - ox) * ix;
    const f32x2 ay0 = (f32x2{ ly.x, ly.y } - oy) * iy, ay1 = (f32x2{ hy.x, hy.y } - oy) * iy;
    const f32x2 by0 = (f32x2{ ly.z, ly.w } - oy) * iy, by1 = (f32x2{ hy.z, hy.w } - oy) * iy;
    const f32x2 az0 = (f32x2{ lz.x, lz.y } - oz) * iz, az1 = (f32x2{ hz.x, hz.y } - oz) * iz;
    const f32x2 bz0 = (f32x2{ lz.z, lz.w } - oz) * iz, bz1 = (f32x2{ hz.z, hz.w } - oz) * iz;
    const float inf = INFINITY;
#define RT_CHILD4(E, x0, x1, y0, y1, z0, z1)                                                  \
    {                                                                                         \
        const float e_ = hw_max3(hw_min(x0, x1), hw_min(y0, y1), hw_max(hw_min(z0, z1), tMin));      \
        const float x_ = hw_min3(hw_max(x0, x1), hw_max(y0, y1), hw_min(hw_max(z0, z1), closest));   \
        E = x_ > e_ ? e_ : inf;                                                               \
    }
    float e0, e1, e2, e3;
    RT_CHILD4(e0, ax0.x, ax1.x, ay0.x, ay1.x, az0.x, az1.x)
    RT_CHILD4(e1, ax0.y, ax1.y, ay0.y, ay1.y, az0.y, az1.y)
    RT_CHILD4(e2, bx0.x, bx1.x, by0.x, by1.x, bz0.x, bz1.x)
    RT_CHILD4(e3, bx0.y, bx1.y, by0.y, by1.y, bz0.y, bz1.y)
#undef RT_CHILD4
    if (anyParallel) {
        const unsigned drop = visit4_parallel(node, o, inv);
        e0 = (drop & 1u) ? inf : e0; e1 = (drop & 2u) ? inf : e1; e2 = (drop & 4u) ? inf : e2; e3 = (drop & 8u) ? inf : e3;
    }
    int r0 = refs.x, r1 = refs.y, r2 = refs.z, r3 = refs.w;
    // sorting network (0,1) (2,3) (0,2) (1,3) (1,2); entry distances are never NaN
#define RT_CE(ea, ra, eb, rb)                                                       \
    {                                                                               \
        const bool sw_ = eb < ea;                                                   \
        const float lo_ = hw_min(ea, eb), hi_ = hw_max(ea, eb);                     \
        const int rl_ = sw_ ? rb : ra, rh_ = sw_ ? ra : rb;                         \
        ea = lo_; eb = hi_; ra = rl_; rb = rh_;                                     \
    }
    if (SORT) {
        RT_CE(e0, r0, e1, r1)
        RT_CE(e2, r2, e3, r3)
        RT_CE(e0, r0, e2, r2)
        RT_CE(e1, r1, e3, r3)
        RT_CE(e1, r1, e2, r2)
    }
#undef RT_CE
    r.E0 = e0; r.E1 = e1; r.E2 = e2; r.E3 = e3;
    r.R0 = r0; r.R1 = r1; r.R2 = r2; r.R3 = r3;
}

// Traversal stack of the four-child walk: the first RT_WIDE_LDS_LEVELS entries of a lane in LDS ([depth][lane],
// as the two-child walks keep theirs), deeper ones - rare - in the lane's private memory.
typedef __attribute__((address_space(3))) uint2 lds_uint2;
struct WideDeep { uint2 e[RT_WIDE_DEEP_LEVELS]; };
__device__ __forceinline__ void wide_push(lds_uint2* lds, WideDeep& deep, int& sp, int ref, float E)
{
    if (sp < RT_WIDE_LDS_LEVELS) { lds[sp * 64].x = (uint32_t)ref; lds[sp * 64].y = __float_as_uint(E); }
    else deep.e[sp - RT_WIDE_LDS_LEVELS] = make_uint2((uint32_t)ref, __float_as_uint(E));
    ++sp;
}
__device__ __forceinline__ uint2 wide_top(lds_uint2* lds, WideDeep& deep, int sp)     // entry sp (already decremented)
{
    uint2 e;
    if (sp < RT_WIDE_LDS_LEVELS) { e.x = lds[sp * 64].x; e.y = lds[sp * 64].y; }
    else e = deep.e[sp - RT_WIDE_LDS_LEVELS];
    return e;
}

struct Counters { int rays, nodeTests, triTests; };

// What a GetHit call keeps while it walks: enough to rebuild the RayHit.
struct Best {
    float closest;   // closestSoFar (world t)
    float tLocal;    // hitT of the winning triangle in object space
    float u, v;
    int   object;    // -1: nothing hit
    int   tri;       // original triangle index
};

struct HitRec { V3 point, normal; int matIndex; float t; bool frontFace; bool hit; };

// IntersectTriangle (raytrace.comp:278-328) on one DevTri record, followed by the acceptance test
// GetHit applies to its result (raytrace.comp:559-572): the hit must lie inside (tMin, closest) both
// as the object-space hitT and as the world-space distance. lo / ld: the ray in the object's space;
// fwd: rows 0-2 of inverse(inverseTransform). Returns true, with `best` updated, when accepted.
__device__ __forceinline__ bool triangle_hit(float4 t0, float4 t1, float4 t2, V3 lo, V3 ld, const float* fwd, V3 org, V3 dirN,
                                             float tMin, int objectIndex, Best& best)
{
    RT_WORK(1, 1);
    V3 v0 = mk(t0.x, t0.y, t0.z), e1 = mk(t1.x, t1.y, t1.z), e2 = mk(t2.x, t2.y, t2.z);
    V3 h = cross3(ld, e2);
    float det = dot3(e1, h);
    const float eps = 1e-5f;
    if (fabsf(det) < eps) return false;
    float invDet = 1.0f / det;
    V3 s = lo - v0;
    float u = invDet * dot3(s, h);
    V3 sc = cross3(s, e1);
    float v = invDet * dot3(ld, sc);
    bool miss = (u < -eps || u > 1.0f + eps) || (v < -eps || u + v > 1.0f + eps);
    float hitT = invDet * dot3(e2, sc);
    if (miss || !(hitT >= tMin && hitT <= best.closest) || !(hitT > eps)) return false;
    V3 p = lo + ld * hitT;
    V3 wp = xform(fwd, p, 1.0f);
    float worldT = dot3(wp - org, dirN);
    if (!(worldT > tMin && worldT < best.closest)) return false;
    best.closest = worldT;
    best.tLocal = hitT;
    best.u = u; best.v = v;
    best.object = objectIndex;
    best.tri = __float_as_int(t1.w);
    return true;
}

#define RT_NONE 0x7fffffff

#ifndef RT_RUN_MIN
#define RT_RUN_MIN 2        /* consecutive BVH parents from which on a run is walked lane by lane */
#endif
// Record visits a lane makes in one round of a run before the wave turns to the leaves that have been
// reached: a lane at a leaf waits for the lanes that are still descending, and the longest descents are
// several times the average, so they are cut into pieces (a lane that is cut short simply goes on in the
// next round: same visits in the same order). Measured 3 ... 10 and none: 6-7 is the flat optimum
// (2.60 -> 2.49 ms per frame on the default scene, 4.11 -> 3.40 with 64 BVH parents).
#ifndef RT_DESCENT_CAP
#define RT_DESCENT_CAP 6
#endif
#ifndef RT_PREFETCH_ALL
#define RT_PREFETCH_ALL 0     /* 1: the record behind a leaf's last triangle is fetched as well (unconditional fetch-ahead) */
#endif
#ifndef RT_DESCENT_CAP4
#define RT_DESCENT_CAP4 3   /* the same cut for walks of four-child records */
#endif

// A run of consecutive BVH parents gone through lane by lane (see walk_run_pair below for the idea):
// every lane visits the parents of the run that its ray can enter, in the run's order, each parent's
// box tested against the ray's closest hit of that moment as the wave-uniform loop of trace() does;
// different lanes are inside different parents at the same time. One ray per lane.
template <bool ANYHIT>
__device__ __forceinline__ void walk_run_single(const DevScene& S, bool permute, int pos0, int count, bool tightScene, V3 org, V3 dirN, V3 winv,
                                                bool anyPar, Best& best, float tMin, uint2* stack)
{
    unsigned long long m = 0;
    for (int k = 0; k < count; ++k) {
        const int j = permute ? S.anyOrder[pos0 + k] : pos0 + k;
        const DevObject* o2 = S.objects + j;     // wave-uniform address
        const V3 bmn = ld3(o2->bmin), bmx = ld3(o2->bmax);
        float E;
        if ((anyPar ? lean_box<true>(org, winv, bmn, bmx, tMin, E) : lean_box<false>(org, winv, bmn, bmx, tMin, E)) && (best.closest > E))
            m |= 1ull << k;
    }
    V3 lo = org, ld = dirN, linv = dirN;
    bool parL = false, nearFirst = false;
    int cur = RT_NONE, sp = 0, obj = 0;
    const DevObject* ob = S.objects;
    for (;;) {
        if (cur == RT_NONE && m != 0) {
            const int k = __ffsll((long long)m) - 1;
            m &= m - 1;
            obj = permute ? S.anyOrder[pos0 + k] : pos0 + k;
            ob = S.objects + obj;
            const V3 bmn = ld3(ob->bmin), bmx = ld3(ob->bmax);
            float E;
            bool ok = (anyPar ? lean_box<true>(org, winv, bmn, bmx, tMin, E) : lean_box<false>(org, winv, bmn, bmx, tMin, E)) && (best.closest > E);
            if (ok && !tightScene) ok = !(distance_to_aabb(org, bmn, bmx) > best.closest);
            if (ok) {
                lo = xform(ob->inv, org, 1.0f);
                ld = normalize3(xform(ob->inv, dirN, 0.0f));
                linv = ld;
                parL = lean_inverse(ld, linv);
                cur = ob->root;
                nearFirst = cur >= S.nearFirstBase;
                sp = 0;
            }
        }
        if (!__any(cur != RT_NONE || m != 0)) break;
        if (cur != RT_NONE) {
            int budget = RT_DESCENT_CAP;
            while ((unsigned)cur < (unsigned)RT_NONE && budget-- > 0) {
                NodeVisit nv;
                visit_node(S, cur, lo, linv, __any(parL), tMin, nv);
                int2 refs = nv.refs;
                const bool goF = refs.x > RT_REF_LIVE_MIN && nv.okF && (best.closest > nv.EF);
                bool canS = refs.y > RT_REF_LIVE_MIN && nv.okS;
                if (nearFirst) {
                    canS = canS && (best.closest > nv.ES);
                    if (goF && canS && nv.ES < nv.EF) {
                        const int r = refs.x; refs.x = refs.y; refs.y = r;
                        const float e = nv.EF; nv.EF = nv.ES; nv.ES = e;
                    }
                }
                if (goF) {
                    if (canS) { stack[sp * 64] = make_uint2((uint32_t)refs.y, __float_as_uint(nv.ES)); ++sp; }
                    cur = refs.x;
                } else if (canS && (best.closest > nv.ES)) {
                    cur = refs.y;
                } else {
                    cur = RT_NONE;
                    while (sp > 0) {
                        --sp;
                        const uint2 e = stack[sp * 64];
                        if (best.closest > __uint_as_float(e.y)) { cur = (int)e.x; break; }
                    }
                }
            }
            if (cur < 0) {      // at a leaf (a lane whose descent was cut short goes on in the next round)
                int k = ~cur;
                bool stop = false;
                for (;;) {
                    const float4* tp = reinterpret_cast<const float4*>(S.tris + k);
                    const float4 t0 = tp[0], t1 = tp[1], t2 = tp[2];
                    const int remaining = __float_as_int(t0.w);
                    if (triangle_hit(t0, t1, t2, lo, ld, ob->fwd, org, dirN, tMin, obj, best) && ANYHIT) { stop = true; break; }
                    if (remaining <= 1) break;
                    ++k;
                }
                cur = RT_NONE;
                if (stop) { sp = 0; m = 0; }
                while (sp > 0) {
                    --sp;
                    const uint2 e = stack[sp * 64];
                    if (best.closest > __uint_as_float(e.y)) { cur = (int)e.x; break; }
                }
            }
        }
    }
}

// GetHit, raytrace.comp:448-610. dirN must already be normalize(r_in.direction).
template <bool COUNT, bool ANYHIT>
__device__ __forceinline__ bool trace(const DevScene& S, int parentCount, V3 org, V3 dirN, float tMin, float tMax,
                                      Best& best, Counters& cnt, uint2* stack)
{
    best.closest = tMax;
    best.object = -1;
    best.tri = 0; best.u = 0.0f; best.v = 0.0f; best.tLocal = 0.0f;
    if (COUNT) cnt.rays++;
    RT_SEC_BEGIN(secTrace);

    V3 winv;
    const bool parW = lean_inverse(dirN, winv);

    // A ray that only asks whether anything is hit keeps closestSoFar at tMax until its first
    // accepted hit, so which triangles it can reach does not depend on the visiting order: it
    // may take the cheap leaf parents first. (Only when the dispatch covers every uploaded parent.)
    const bool permute = ANYHIT && parentCount == S.objectCount;
    const bool tightScene = !COUNT && S.nearFirstBase != 0x7fffffff;   // the near-first scene (wave-uniform)
    for (int pos = 0; pos < parentCount; ++pos) {
        const int i = permute ? S.anyOrder[pos] : pos;
        const DevObject* ob = S.objects + i;   // wave-uniform address
        if (!COUNT && ob->root >= 0) {
            // consecutive BVH parents (up to 64): RT_RUN_MIN or more are gone through lane by lane
            int count = permute ? S.runLenAny[pos] : S.runLen[pos];
            count = count < parentCount - pos ? count : parentCount - pos;
            if (count >= RT_RUN_MIN) {
                walk_run_single<ANYHIT>(S, permute, pos, count, tightScene, org, dirN, winv, __any(parW), best, tMin, stack);
                if (ANYHIT && best.object >= 0) return true;
                pos += count - 1;
                continue;
            }
        }
        if (COUNT) cnt.nodeTests++;
        V3 bmn = ld3(ob->bmin), bmx = ld3(ob->bmax);
        float E;
        bool ok = (__any(parW) ? lean_box<true>(org, winv, bmn, bmx, tMin, E) : lean_box<false>(org, winv, bmn, bmx, tMin, E)) &&
                  (best.closest > E);
        // (the point-to-box distance is never larger than the entry distance E, so this cull of
        // raytrace.comp:469-471 can only fire on a rounding tie: the near-first scene, whose boxes are only
        // culls, leaves it out)
        if (ok && !tightScene) ok = !(distance_to_aabb(org, bmn, bmx) > best.closest);
        const int root = ob->root;   // wave-uniform
        if (root < 0) {
            // A parent that is one leaf (a wall: two triangles): every lane that enters tests the same
            // triangles, so their records come through the scalar cache, once for the wave.
            if (!(root > RT_REF_LIVE_MIN) || !__any(ok)) continue;
            RT_SEC_BEGIN(secLeaf);
            V3 lo = xform(ob->inv, org, 1.0f);
            V3 ld = normalize3(xform(ob->inv, dirN, 0.0f));
            const float4* tp = reinterpret_cast<const float4*>(S.tris + ~root);
            for (;;) {
                const float4 t0 = tp[0], t1 = tp[1], t2 = tp[2];
                if (ok) {
                    if (COUNT) cnt.triTests++;
                    if (triangle_hit(t0, t1, t2, lo, ld, ob->fwd, org, dirN, tMin, i, best)) {
                        if (ANYHIT) return true;
                    }
                }
                if (__float_as_int(t0.w) <= 1) break;
                tp += 3;
            }
            RT_SEC_END(2, secLeaf);
            continue;
        }
        int cur = root;
        if (!ok) continue;
        // a root in the rebuilt part of the record array: this object's tree may be walked nearer
        // child first (wave-uniform)
        const bool nearFirst = !COUNT && root >= S.nearFirstBase;
        RT_SEC_BEGIN(secWalk);

        V3 lo = xform(ob->inv, org, 1.0f);
        V3 ld = normalize3(xform(ob->inv, dirN, 0.0f));
        V3 linv = ld;
        const bool parL = lean_inverse(ld, linv);
        int sp = 0;

        for (;;) {
            // Descent. A lane that reaches a leaf leaves this loop and waits for the others, so
            // that the triangle tests below run once for all the leaves the wave has reached
            // rather than once per lane (each lane still performs its own steps in its own order).
            int budget = COUNT ? 0x7fffffff : RT_DESCENT_CAP;
            while ((unsigned)cur < (unsigned)RT_NONE && budget-- > 0) {
                NodeVisit nv;
                visit_node(S, cur, lo, linv, __any(parL), tMin, nv);
                int2 refs = nv.refs;
                float EF = nv.EF, ES = nv.ES;
                const bool okF = nv.okF, okS = nv.okS;
                if (COUNT) cnt.nodeTests += (refs.x != RT_REF_ABSENT ? 1 : 0) + (refs.y != RT_REF_ABSENT ? 1 : 0);
                bool goF = refs.x > RT_REF_LIVE_MIN && okF && (best.closest > EF);
                bool canS = refs.y > RT_REF_LIVE_MIN && okS;
                if (nearFirst) {
                    // a child behind the closest hit is dropped at once; of two candidates the nearer goes first
                    canS = canS && (best.closest > ES);
                    if (goF && canS && ES < EF) {
                        const int r = refs.x; refs.x = refs.y; refs.y = r;
                        const float e = EF; EF = ES; ES = e;
                    }
                }
                if (goF) {
                    if (canS) { stack[sp * 64] = make_uint2((uint32_t)refs.y, __float_as_uint(ES)); ++sp; }
                    cur = refs.x;
                } else if (canS && (best.closest > ES)) {
                    cur = refs.y;
                } else {
                    // pop: discard entries whose box no longer reaches inside the closest hit
                    cur = RT_NONE;
                    while (sp > 0) {
                        --sp;
                        uint2 e = stack[sp * 64];
                        if (best.closest > __uint_as_float(e.y)) { cur = (int)e.x; break; }
                    }
                }
            }
            if (cur == RT_NONE) break;
            if (cur >= 0) continue;      // descent cut short: the leaves the other lanes have reached first
            {
                int k = ~cur;
                for (;;) {
                    const float4* tp = reinterpret_cast<const float4*>(S.tris + k);
                    float4 t0 = tp[0], t1 = tp[1], t2 = tp[2];
                    const int remaining = __float_as_int(t0.w);
                    if (COUNT) cnt.triTests++;
                    if (triangle_hit(t0, t1, t2, lo, ld, ob->fwd, org, dirN, tMin, i, best)) {
                        if (ANYHIT) return true;
                    }
                    if (remaining <= 1) break;
                    ++k;
                }
            }
            // pop: discard entries whose box no longer reaches inside the closest hit
            cur = RT_NONE;
            while (sp > 0) {
                --sp;
                uint2 e = stack[sp * 64];
                if (best.closest > __uint_as_float(e.y)) { cur = (int)e.x; break; }
            }
            if (cur == RT_NONE) break;
        }
        RT_SEC_END(3, secWalk);
    }
    RT_SEC_END(1, secTrace);
    return best.object >= 0;
}

// Rebuild the RayHit of the winning triangle (raytrace.comp:311-320 + 559-571).
__device__ __forceinline__ void finish_hit(const DevScene& S, V3 org, V3 dirN, const Best& best, HitRec& rec)
{
    rec.hit = best.object >= 0;
    rec.t = best.closest;
    if (!rec.hit) return;
    const DevObject* ob = S.objects + best.object;
    V3 lo = xform(ob->inv, org, 1.0f);
    V3 ld = normalize3(xform(ob->inv, dirN, 0.0f));
    V3 p = lo + ld * best.tLocal;
    rec.point = xform(ob->fwd, p, 1.0f);
    const float4* np = reinterpret_cast<const float4*>(S.nrms + best.tri);
    float4 a = np[0], b = np[1], c = np[2];
    float w = 1.0f - best.u - best.v;
    V3 interp = normalize3((mk(a.x, a.y, a.z) * w + mk(b.x, b.y, b.z) * best.u) + mk(c.x, c.y, c.z) * best.v);
    bool frontLocal = dot3(ld, interp) < 0.0f;
    V3 nl = frontLocal ? interp : -interp;
    // normalMatrix = mat3(transpose(inverseTransform)); row r of the product is
    // column r of inverseTransform dotted with the normal
    const float* m = ob->inv;
    V3 wn = mk(m[0] * nl.x + m[4] * nl.y + m[8] * nl.z,
               m[1] * nl.x + m[5] * nl.y + m[9] * nl.z,
               m[2] * nl.x + m[6] * nl.y + m[10] * nl.z);
    rec.normal = normalize3(wn);
    rec.matIndex = ob->materialIndex;
    rec.frontFace = dot3(dirN, rec.normal) < 0.0f;
}

// ---- shading ------------------------------------------------------------------

// raytrace.comp:363-369
__device__ __forceinline__ float reflectance(float cosine, float refIdx)
{
    float r0 = (1.0f - refIdx) / (1.0f + refIdx);
    r0 = r0 * r0;
    return r0 + (1.0f - r0) * gpow(1.0f - cosine, 5.0f);
}
__device__ __forceinline__ V3 reflect3(V3 I, V3 N) { return I - N * (2.0f * dot3(N, I)); }
__device__ __forceinline__ V3 refract3(V3 I, V3 N, float eta)
{
    float d = dot3(N, I);
    float k = 1.0f - eta * eta * (1.0f - d * d);
    if (k < 0.0f) return mk(0.0f, 0.0f, 0.0f);
    return I * eta - N * (eta * d + sqrtf(k));
}

// raytrace.comp:623-668
// UNIT: rayDir is normalize3() of the ray's direction already (the value this function would compute itself)
template <bool UNIT = false>
__device__ __forceinline__ V3 environment_colour(const rt_push_constants& pc, V3 rayDir)
{
    V3 unitDir = UNIT ? rayDir : normalize3(rayDir);
    V3 sunDir = normalize3(ld3(pc.sunDirection));
    float sunDotUp = dot3(-sunDir, mk(0.0f, 1.0f, 0.0f));
    const V3 dayZenith = mk(0.5f, 0.7f, 1.0f), dayHorizon = mk(1.0f, 1.0f, 1.0f), sunrise = mk(1.0f, 0.4f, 0.2f);
    const V3 night = mk(0.0f, 0.0f, 0.0f);
    float viewT = gclamp(0.5f * (unitDir.y + 1.0f), 0.0f, 1.0f);
    V3 baseSky;
    if (sunDotUp > 0.1f) {
        baseSky = gmix3(dayHorizon, dayZenith, viewT);
    } else if (sunDotUp > -0.1f) {
        float blend = (sunDotUp + 0.1f) / 0.2f;
        V3 daySky = gmix3(dayHorizon, dayZenith, viewT);
        V3 sunSky = gmix3(sunrise, sunrise * 0.5f, viewT);
        baseSky = gmix3(sunSky, daySky, blend);
    } else {
        baseSky = gmix3(night, night, viewT);
    }
    float cosTheta = gclamp(dot3(unitDir, -sunDir), 0.0f, 1.0f);
    const float halfDegree = 0.5f * 0.017453292519943295f;
    float sunDisk = rt_expf(-gpow(rt_acosf(cosTheta) / halfDegree, 2.0f));
    V3 sunGlow = (ld3(pc.sunColour) * pc.sunIntensity) * sunDisk;
    return baseSky + sunGlow;
}

// raytrace.comp:676-690 with SampleSquare (Random.glsl:43-49) and
// DefocusDiskSample / RandomInUnitDisk (raytrace.comp:670-674, Random.glsl:51-60).
// The seed arrives by value: the caller's state is not advanced.
__device__ __forceinline__ Ray get_ray(const rt_push_constants& pc, int px, int py, uint32_t seed)
{
    seed += 1u;
    float sx = rng_range(0.0f, 1.0f, seed);
    float sy = rng_range(0.0f, 1.0f, seed);
    float ox = sx - 0.5f, oy = sy - 0.5f;
    V3 ps = (ld3(pc.pixel00Location) + ld3(pc.pixelDeltaU) * ((float)px + ox)) + ld3(pc.pixelDeltaV) * ((float)py + oy);
    V3 origin;
    if (pc.defocusAngle <= 0.0f) {
        origin = ld3(pc.cameraPosition);
    } else {
        uint32_t s2 = seed + 1u;
        V3 p = mk(0.0f, 0.0f, 0.0f);
        for (int round = 0; round < RT_REJECTION_CAP; ++round) {
            float dx = rng_range(-1.0f, 1.0f, s2);
            float dy = rng_range(-1.0f, 1.0f, s2);
            if (sqrtf(length3(mk(dx, dy, 0.0f))) < 1.0f) { p = mk(dx, dy, 0.0f); break; }
        }
        origin = (ld3(pc.cameraPosition) + ld3(pc.defocusDiskU) * p.x) + ld3(pc.defocusDiskV) * p.y;
    }
    Ray r;
    r.o = origin;
    r.d = normalize3(ps - origin);
    return r;
}

__device__ __forceinline__ DevMaterial load_material(const DevScene& S, int idx)
{
    const float4* mp = reinterpret_cast<const float4*>(S.materials + idx);
    float4 a = mp[0], b = mp[1], c = mp[2];
    DevMaterial m;
    m.colour[0] = a.x; m.colour[1] = a.y; m.colour[2] = a.z; m.smoothness = a.w;
    m.fuzziness = b.x; m.emission = b.y; m.refractionIndex = b.z; m.pad1 = b.w;
    m.absorbtion[0] = c.x; m.absorbtion[1] = c.y; m.absorbtion[2] = c.z; m.pad3 = c.w;
    return m;
}

// The shader hands `seed` to MaterialScatter by value (raytrace.comp:759), so every bounce of a sample starts
// the generator from the same state: the unit vector of the Lambert branch (RandomUnitVector, Random.glsl:68)
// and the one inside the metal branch's RandomOnHemisphere (one `seed += 1` later, Random.glsl:80-82) are
// functions of the sample's seed alone. The kernels that keep a path per lane across its bounces draw both
// once per sample - two rejection loops per sample instead of up to two per bounce at wave level.
struct ScatterDraws { V3 lambert, metal; };

__device__ __forceinline__ ScatterDraws scatter_draws(uint32_t seed)
{
    (void)rng_range(0.0f, 1.0f, seed);      // scatterOrReflect: MaterialScatter's first draw
    ScatterDraws d;
    uint32_t s = seed;
    d.lambert = random_unit_vector(s);
    s = seed + 1u;                          // RandomOnHemisphere's own increment
    d.metal = random_unit_vector(s);
    return d;
}

// MaterialScatter + DielectricScatter, raytrace.comp:371-423. `seed` by value. pre: the sample's two unit
// vectors drawn ahead (scatter_draws), or null to draw them here. UNIT: rayDir is normalize3() of the ray's
// direction already.
template <bool UNIT = false>
__device__ __forceinline__ void material_scatter(const DevMaterial& m, V3 rayDir, const HitRec& rec, uint32_t seed,
                                                 V3& attenuation, Ray& scattered, const ScatterDraws* pre = nullptr)
{
    float scatterOrReflect = rng_range(0.0f, 1.0f, seed);
    if (m.refractionIndex > 0.0f) {
        float refIdx = m.refractionIndex;
        V3 ab = ld3(m.absorbtion);
        V3 e = mk(rt_expf(-ab.x * rec.t), rt_expf(-ab.y * rec.t), rt_expf(-ab.z * rec.t));
        attenuation = ld3(m.colour) * e;
        float etai = rec.frontFace ? (1.0f / refIdx) : refIdx;
        V3 unit = UNIT ? rayDir : normalize3(rayDir);
        float cosTheta = gmin(dot3(-unit, rec.normal), 1.0f);
        float sinTheta = sqrtf(1.0f - cosTheta * cosTheta);
        bool cannotRefract = etai * sinTheta > 1.0f;
        V3 dir;
        if (cannotRefract || reflectance(cosTheta, refIdx) > rng_range(0.0f, 1.0f, seed))
            dir = reflect3(unit, rec.normal);
        else
            dir = refract3(unit, rec.normal, etai);
        scattered.o = rec.point;
        scattered.d = normalize3(dir);
        return;
    }
    const float EPSILON = 1e-4f;
    if (scatterOrReflect < m.smoothness) {
        V3 reflected = reflect3(UNIT ? rayDir : normalize3(rayDir), rec.normal);
        V3 fuzz;
        if (pre) fuzz = dot3(pre->metal, rec.normal) > 0.0f ? pre->metal : -pre->metal;     // Random.glsl:83-87
        else fuzz = random_on_hemisphere(rec.normal, seed);
        reflected = reflected + fuzz * m.fuzziness;
        scattered.o = rec.point + reflected * EPSILON;
        scattered.d = normalize3(reflected);
        attenuation = ld3(m.colour);
    } else {
        V3 sd = rec.normal + (pre ? pre->lambert : random_unit_vector(seed));
        const float s = 1e-8f;
        if (fabsf(sd.x) < s && fabsf(sd.y) < s && fabsf(sd.z) < s) sd = rec.normal;
        scattered.o = rec.point + sd * EPSILON;
        scattered.d = normalize3(sd);
        attenuation = ld3(m.colour);
    }
}

// RayColour, raytrace.comp:692-769. The shader passes `seed` on by value at
// every bounce, so every bounce of one sample draws the same numbers.
template <bool COUNT>
__device__ __forceinline__ V3 ray_colour(const DevScene& S, const rt_push_constants& pc, Ray ray, uint32_t seed,
                                         Counters& cnt, uint2* stack)
{
    V3 result = mk(0.0f, 0.0f, 0.0f);
    V3 throughput = mk(1.0f, 1.0f, 1.0f);
    const int bounces = pc.maxBounces + 1;
    const ScatterDraws draws = scatter_draws(seed);
    for (int bounce = 0; bounce < bounces; ++bounce) {
        V3 dirN = normalize3(ray.d);
        Best best;
        trace<COUNT, false>(S, pc.parentBVHCount, ray.o, dirN, 0.001f, 1e20f, best, cnt, stack);
        HitRec rec;
        RT_SEC_BEGIN(secFinish);
        finish_hit(S, ray.o, dirN, best, rec);
        RT_SEC_END(4, secFinish);
        if (!rec.hit) {
            result = result + throughput * environment_colour(pc, ray.d);
            break;
        }
        if (!rec.frontFace) break;

        DevMaterial m = load_material(S, rec.matIndex);
        V3 emitted = ld3(m.colour) * m.emission;
        result = result + throughput * emitted;
        if (m.emission > 0.0f) break;

        V3 negSun = -ld3(pc.sunDirection);
        float nDotL = gmax(dot3(rec.normal, negSun), 0.0f);
        if (nDotL > 0.0f) {
            V3 so = rec.point + rec.normal * 0.01f;
            Best sb;
            bool blocked = trace<COUNT, !COUNT>(S, pc.parentBVHCount, so, normalize3(negSun), 0.001f, 1e20f, sb, cnt, stack);
            if (!blocked) {
                V3 sun = ld3(pc.sunColour) * pc.sunIntensity;
                const float INV_PI = 0.31830988618f;
                V3 direct = sun * (nDotL * INV_PI);
                if (m.refractionIndex == 0.0f) {
                    result = result + (throughput * ld3(m.colour)) * direct;
                } else {
                    float fresnel = reflectance(nDotL, m.refractionIndex);
                    result = result + (throughput * fresnel) * direct;
                }
            }
        }
        V3 attenuation;
        Ray scattered;
        RT_SEC_BEGIN(secScatter);
        material_scatter(m, ray.d, rec, seed, attenuation, scattered, &draws);
        RT_SEC_END(5, secScatter);
        throughput = throughput * attenuation;
        ray = scattered;
    }
    return result;
}

// ---- two samples per lane ------------------------------------------------------------------
//
// A wave leaves a BVH walk when its slowest lane does, and walks are short on average (11 steps)
// with a long tail (a few hundred), so a wave of 64 rays spends most of a walk waiting for one or two
// lanes. Giving every lane TWO independent samples of its pixel (consecutive entries of the pixel's
// sample sequence: frames and samples per pixel) and letting it walk the second ray as soon as its
// first is done makes the wave wait for the largest SUM of two walks instead of twice for the largest
// walk; sums of two have a much lighter tail. The parents are still visited by the whole wave in
// step (their records stay scalar); inside a BVH parent a lane walks ray A, then ray B. Each ray sees
// the objects, nodes and triangles in the reference's order, and the two samples are added to the
// pixel in sequence order, so the image is the one the one-sample kernel renders.

struct PathState {
    V3 o, d;              // next regular ray; d = normalize(direction as the shader keeps it), the only form it is ever used in
                          // (GetHit raytrace.comp:451, GetEnvironmentColour :626, MaterialScatter :376,:402)
    V3 throughput, result;
    V3 sun;               // sun term of the last hit, added when its shadow ray comes back unblocked
    V3 so;                // origin of that shadow ray
    ScatterDraws draws;   // the sample's two unit vectors (the same at every bounce)
    bool alive;           // wants a regular GetHit
    bool shadow;          // wants a shadow GetHit
};

// Path state that the wide kernels (NF >= 2) keep in LDS rather than in registers: a sample's result, throughput and
// its two scatter vectors are only touched between GetHit calls, whose own loops need every register that four
// waves per SIMD leave a wave - held in registers they are pushed out to private memory around every call, and
// private memory is device memory: the spill areas of an XCD's 512 waves do not fit its L2, so that traffic
// reaches HBM (measured on the default scene at 2560x1369: 6.2 GB written per launch of 8 frames with the state in
// registers, of 7.4 GB before the path kept its direction in one form only). Laid out [dword][lane] behind the
// traversal stack, 12 dwords per path: result, throughput, Lambert vector, metal vector.
#define RT_RES_DWORDS 24
#ifndef RT_RES_SET
#define RT_RES_SET 6      /* which fields live in LDS: 1 result, 2 throughput, 4 scatter vectors (measured, ms per frame / GB written per
                             launch on the default scene: 7: 2.075 / 1.96; 6: 1.985 / 3.08; 5: 2.04 / 3.30; 4: 2.02 / 4.13; none: 2.13 / 6.23) */
#endif
enum { RES_RESULT = 0, RES_THROUGHPUT = 3, RES_LAMBERT = 6, RES_METAL = 9, RES_PER_PATH = 12 };
__device__ __forceinline__ V3 res_ld(const float* res, int k) { return mk(res[k * 64], res[(k + 1) * 64], res[(k + 2) * 64]); }
__device__ __forceinline__ void res_st(float* res, int k, V3 v) { res[k * 64] = v.x; res[(k + 1) * 64] = v.y; res[(k + 2) * 64] = v.z; }

__device__ __forceinline__ void best_reset(Best& b, float tMax)
{
    b.closest = tMax; b.object = -1; b.tri = 0; b.u = 0.0f; b.v = 0.0f; b.tLocal = 0.0f;
}

// ---------------------------------------------------------------------------------
// A run of consecutive BVH parents, lane by lane. The object loop of GetHit is wave-uniform: the
// whole wave enters parent after parent, and inside each one only the lanes whose ray meets that
// parent's box have work (a dozen of 64 on the scenes measured), so a scene with many BVH parents
// is walked as many sparse walks one after the other. A ray, however, only has to see ITS parents
// in order (its closest hit is its own): here every lane goes through the parents of the run that
// its rays can enter, one ray after the other, parent after parent in the run's order, each parent's
// box tested against the ray's closest hit of that moment exactly as the wave-uniform loop does.
// Different lanes are inside different parents at the same time; object records come through
// per-lane (vector) loads instead of scalar ones. Same tests in the same order per ray: same bits.
template <bool ANYHIT, int NF>
__device__ __forceinline__ void walk_run_pair(const DevScene& S, bool permute, int pos0, int count, V3 oA, V3 dA, V3 winvA, bool liveA, Best& bA,
                                              V3 oB, V3 dB, V3 winvB, bool liveB, Best& bB, bool anyPar, float tMin, uint2* stack)
{
    // which parents of the run a ray can enter at all (a parent whose box does not reach inside the
    // closest hit now will not later either: closest hits only come closer)
    unsigned long long mA = 0, mB = 0;
    RT_SEC_BEGIN(secPre);
    for (int k = 0; k < count; ++k) {
        const int j = permute ? S.anyOrder[pos0 + k] : pos0 + k;
        const DevObject* o2 = S.objects + j;     // wave-uniform address
        const V3 bmn = ld3(o2->bmin), bmx = ld3(o2->bmax);
        float E;
        RT_WORK(0, (liveA ? 1 : 0) + (liveB ? 1 : 0));
        if (liveA && (anyPar ? lean_box<true>(oA, winvA, bmn, bmx, tMin, E) : lean_box<false>(oA, winvA, bmn, bmx, tMin, E)) && (bA.closest > E))
            mA |= 1ull << k;
        if (liveB && (anyPar ? lean_box<true>(oB, winvB, bmn, bmx, tMin, E) : lean_box<false>(oB, winvB, bmn, bmx, tMin, E)) && (bB.closest > E))
            mB |= 1ull << k;
    }
    RT_SEC_END(5, secPre);
    if (!__any((mA | mB) != 0)) return;

    bool active = (mA | mB) != 0;
    bool second = false;
    unsigned long long m = mA;
    V3 org = oA, dirN = dA, winv = winvA;
    Best bw = bA;
    V3 lo = oA, ld = dA, linv = dA;
    bool parL = false, nearFirst = false;
    int cur = RT_NONE, sp = 0, obj = 0;
    const DevObject* ob = S.objects;
    lds_uint2* const wlds = (lds_uint2*)stack;
    WideDeep wdeep;
    for (;;) {
        if (active && cur == RT_NONE) {
            // the ray's next parent; after its last one the lane's other ray; after that the lane is done
            if (ANYHIT && bw.object >= 0) m = 0;
            if (m == 0 && !second) {
                bA = bw;
                second = true;
                m = mB;
                org = oB; dirN = dB; winv = winvB;
                bw = bB;
            }
            if (m == 0) {
                bB = bw;
                active = false;
            } else {
                const int k = __ffsll((long long)m) - 1;
                m &= m - 1;
                obj = permute ? S.anyOrder[pos0 + k] : pos0 + k;
                ob = S.objects + obj;
                const V3 bmn = ld3(ob->bmin), bmx = ld3(ob->bmax);
                float E;
                bool ok = (anyPar ? lean_box<true>(org, winv, bmn, bmx, tMin, E) : lean_box<false>(org, winv, bmn, bmx, tMin, E)) && (bw.closest > E);
                RT_WORK(0, 1);
                if ((NF == 0 || NF == 4) && ok) ok = !(distance_to_aabb(org, bmn, bmx) > bw.closest);
                if (ok) {
                    lo = xform(ob->inv, org, 1.0f);
                    ld = normalize3(xform(ob->inv, dirN, 0.0f));
                    linv = ld;
                    parL = lean_inverse(ld, linv);
                    cur = ob->root;
                    nearFirst = NF == 1 && cur >= S.nearFirstBase;
                    sp = 0;
                }
            }
        }
        if (!__any(active)) break;
        if (NF == 4) {
            // the reference's trees as four-child records whose slots stand in the reference's visiting order: into the first
            // slot the ray enters, the later ones stacked so that they come back in order
            if (cur != RT_NONE) {
                int budget = RT_DESCENT_CAP4;
                while ((unsigned)cur < (unsigned)RT_NONE && budget-- > 0) {
                    Visit4 v;
                    visit4<false>(S.nodes4 + cur, lo, linv, __any(parL), tMin, bw.closest, v);
                    const bool h0 = v.E0 < INFINITY, h1 = v.E1 < INFINITY, h2 = v.E2 < INFINITY, h3 = v.E3 < INFINITY;
                    const bool p1 = h1 && h0, p2 = h2 && (h0 || h1), p3 = h3 && (h0 || h1 || h2);
                    if (!__any(sp > RT_WIDE_LDS_LEVELS - 3)) {
                        lds_uint2* slot = wlds + sp * 64;
                        slot->x = (uint32_t)v.R3; slot->y = __float_as_uint(v.E3);
                        if (p3) { ++sp; slot += 64; }
                        slot->x = (uint32_t)v.R2; slot->y = __float_as_uint(v.E2);
                        if (p2) { ++sp; slot += 64; }
                        slot->x = (uint32_t)v.R1; slot->y = __float_as_uint(v.E1);
                        if (p1) ++sp;
                    } else {
                        if (p3) wide_push(wlds, wdeep, sp, v.R3, v.E3);
                        if (p2) wide_push(wlds, wdeep, sp, v.R2, v.E2);
                        if (p1) wide_push(wlds, wdeep, sp, v.R1, v.E1);
                    }
                    cur = h0 ? v.R0 : h1 ? v.R1 : h2 ? v.R2 : v.R3;
                    if (!(h0 || h1 || h2 || h3)) {
                        cur = RT_NONE;
                        while (sp > 0) {
                            --sp;
                            const uint2 e = wide_top(wlds, wdeep, sp);
                            if (bw.closest > __uint_as_float(e.y)) { cur = (int)e.x; break; }
                        }
                    }
                }
                if (cur < 0) {      // at a leaf (a lane whose descent was cut short goes on in the next round)
                    bool stop = false;
                    const float4* tp = reinterpret_cast<const float4*>(S.tris + ~cur);
                    float4 t0 = tp[0], t1 = tp[1], t2 = tp[2];
                    for (;;) {
                        // (only lanes whose leaf goes on fetch ahead: the address units are the busiest part of the memory path)
                        const int remaining = __float_as_int(t0.w);
                        float4 n0 = t0, n1 = t1, n2 = t2;
                        if (RT_PREFETCH_ALL || remaining > 1) { tp += 3; n0 = tp[0]; n1 = tp[1]; n2 = tp[2]; }
                        if (triangle_hit(t0, t1, t2, lo, ld, ob->fwd, org, dirN, tMin, obj, bw) && ANYHIT) { stop = true; break; }
                        if (remaining <= 1) break;
                        t0 = n0; t1 = n1; t2 = n2;
                    }
                    cur = RT_NONE;
                    if (stop) sp = 0;
                    while (sp > 0) {
                        --sp;
                        const uint2 e = wide_top(wlds, wdeep, sp);
                        if (bw.closest > __uint_as_float(e.y)) { cur = (int)e.x; break; }
                    }
                }
            }
            continue;
        }
        if (NF == 2) {
            if (cur != RT_NONE) {
                int budget = RT_DESCENT_CAP4;
                while ((unsigned)cur < (unsigned)RT_NONE && budget-- > 0) {
                    Visit4 v;
                    visit4(S.nodes4 + cur, lo, linv, __any(parL), tMin, bw.closest, v);
                    RT_WORK_LANES(4);
                    if (!__any(sp > RT_WIDE_LDS_LEVELS - 3)) {
                        // the usual case, without branches: every entry is written to the lane's next free slot and
                        // the slot is kept only if the child is to be visited
                        lds_uint2* slot = wlds + sp * 64;
                        slot->x = (uint32_t)v.R3; slot->y = __float_as_uint(v.E3);
                        if (v.E3 < INFINITY) { ++sp; slot += 64; }
                        slot->x = (uint32_t)v.R2; slot->y = __float_as_uint(v.E2);
                        if (v.E2 < INFINITY) { ++sp; slot += 64; }
                        slot->x = (uint32_t)v.R1; slot->y = __float_as_uint(v.E1);
                        if (v.E1 < INFINITY) ++sp;
                    } else {
                        if (v.E3 < INFINITY) wide_push(wlds, wdeep, sp, v.R3, v.E3);
                        if (v.E2 < INFINITY) wide_push(wlds, wdeep, sp, v.R2, v.E2);
                        if (v.E1 < INFINITY) wide_push(wlds, wdeep, sp, v.R1, v.E1);
                    }
                    cur = v.R0;
                    if (!(v.E0 < INFINITY)) {
                        cur = RT_NONE;
                        while (sp > 0) {
                            --sp;
                            const uint2 e = wide_top(wlds, wdeep, sp);
                            if (bw.closest > __uint_as_float(e.y)) { cur = (int)e.x; break; }
                        }
                    }
                }
                if (cur < 0) {      // at a leaf (a lane whose descent was cut short goes on in the next round)
                    bool stop = false;
                    // the next record is fetched while the current one is tested (the array ends in spare bytes)
                    const float4* tp = reinterpret_cast<const float4*>(S.tris + ~cur);
                    float4 t0 = tp[0], t1 = tp[1], t2 = tp[2];
                    for (;;) {
                        // (only lanes whose leaf goes on fetch ahead: the address units are the busiest part of the memory path)
                        const int remaining = __float_as_int(t0.w);
                        float4 n0 = t0, n1 = t1, n2 = t2;
                        if (RT_PREFETCH_ALL || remaining > 1) { tp += 3; n0 = tp[0]; n1 = tp[1]; n2 = tp[2]; }
                        RT_WORK_LANES(6);
                        if (triangle_hit(t0, t1, t2, lo, ld, ob->fwd, org, dirN, tMin, obj, bw) && ANYHIT) { stop = true; break; }
                        if (remaining <= 1) break;
                        t0 = n0; t1 = n1; t2 = n2;
                    }
                    cur = RT_NONE;
                    if (stop) sp = 0;
                    while (sp > 0) {
                        --sp;
                        const uint2 e = wide_top(wlds, wdeep, sp);
                        if (bw.closest > __uint_as_float(e.y)) { cur = (int)e.x; break; }
                    }
                }
            }
            continue;
        }
        if (cur != RT_NONE) {
            int budget = RT_DESCENT_CAP;
            while ((unsigned)cur < (unsigned)RT_NONE && budget-- > 0) {
                NodeVisit nv;
                visit_node(S, cur, lo, linv, __any(parL), tMin, nv);
                int2 refs = nv.refs;
                const bool goF = refs.x > RT_REF_LIVE_MIN && nv.okF && (bw.closest > nv.EF);
                bool canS = refs.y > RT_REF_LIVE_MIN && nv.okS;
                if (NF && nearFirst) {
                    canS = canS && (bw.closest > nv.ES);
                    if (goF && canS && nv.ES < nv.EF) {
                        const int r = refs.x; refs.x = refs.y; refs.y = r;
                        const float e = nv.EF; nv.EF = nv.ES; nv.ES = e;
                    }
                }
                if (goF) {
                    if (canS) wide_push(wlds, wdeep, sp, refs.y, nv.ES);
                    cur = refs.x;
                } else if (canS && (bw.closest > nv.ES)) {
                    cur = refs.y;
                } else {
                    cur = RT_NONE;
                    while (sp > 0) {
                        --sp;
                        const uint2 e = wide_top(wlds, wdeep, sp);
                        if (bw.closest > __uint_as_float(e.y)) { cur = (int)e.x; break; }
                    }
                }
            }
            if (cur < 0) {      // at a leaf (a lane whose descent was cut short goes on in the next round)
                int k = ~cur;
                bool stop = false;
                for (;;) {
                    const float4* tp = reinterpret_cast<const float4*>(S.tris + k);
                    const float4 t0 = tp[0], t1 = tp[1], t2 = tp[2];
                    const int remaining = __float_as_int(t0.w);
                    if (triangle_hit(t0, t1, t2, lo, ld, ob->fwd, org, dirN, tMin, obj, bw) && ANYHIT) { stop = true; break; }
                    if (remaining <= 1) break;
                    ++k;
                }
                cur = RT_NONE;
                if (stop) sp = 0;
                while (sp > 0) {
                    --sp;
                    const uint2 e = wide_top(wlds, wdeep, sp);
                    if (bw.closest > __uint_as_float(e.y)) { cur = (int)e.x; break; }
                }
            }
        }
    }
}

// The BVH parents of a wide scene through the hierarchy over their world boxes (DevScene::tlasRoot): ONE walk per
// ray that starts in the hierarchy's records (world-space ray), turns into an object's own tree at an instance
// reference (the ray taken into the object's space there) and comes back to the hierarchy when the traversal
// stack hands it one of its entries again - the same loop, the same stack, every record's children nearest
// first, so that the objects themselves are entered nearest first and the ones behind the closest hit not at all.
// A lane walks ray A, then ray B, while other lanes are wherever their own rays have taken them. (Hits do not
// depend on the visiting order in these scenes: DevScene::orderFree.)
template <bool ANYHIT>
__device__ __forceinline__ void walk_tlas_pair(const DevScene& S, V3 oA, V3 dA, V3 winvA, bool liveA, Best& bA,
                                               V3 oB, V3 dB, V3 winvB, bool liveB, Best& bB, bool anyParW, float tMin, uint2* stack)
{
    if (!__any(liveA || liveB)) return;
    bool active = liveA || liveB;
    bool second = !liveA;
    V3 org = second ? oB : oA, dirN = second ? dB : dA, winv = second ? winvB : winvA;
    Best bw = second ? bB : bA;
    V3 lo = org, ld = dirN, linv = winv;
    bool parL = false;
    int cur = active ? S.tlasRoot : RT_NONE, sp = 0, obj = 0;
    const DevObject* ob = S.objects;
    lds_uint2* const wlds = (lds_uint2*)stack;
    WideDeep wdeep;
    const int tlasBase = S.tlasBase;
    for (;;) {
        if (active && cur == RT_NONE) {
            // this ray is done: the lane's other ray, or nothing
            if (!second) {
                bA = bw;
                second = true;
                if (liveB) {
                    org = oB; dirN = dB; winv = winvB;
                    bw = bB;
                    cur = S.tlasRoot; sp = 0;
                } else active = false;
            } else {
                bB = bw;
                active = false;
            }
        }
        if (!__any(active)) break;
        if (cur == RT_NONE) continue;
        int budget = RT_DESCENT_CAP4;
        while ((unsigned)cur < (unsigned)RT_REF_INSTANCE && budget-- > 0) {
            const bool top = cur >= tlasBase;
            const V3 ro = top ? org : lo, ri = top ? winv : linv;
            Visit4 v;
            visit4(S.nodes4 + cur, ro, ri, anyParW || __any(parL), tMin, bw.closest, v);
            RT_WORK_LANES(4);
            if (!__any(sp > RT_WIDE_LDS_LEVELS - 3)) {
                lds_uint2* slot = wlds + sp * 64;
                slot->x = (uint32_t)v.R3; slot->y = __float_as_uint(v.E3);
                if (v.E3 < INFINITY) { ++sp; slot += 64; }
                slot->x = (uint32_t)v.R2; slot->y = __float_as_uint(v.E2);
                if (v.E2 < INFINITY) { ++sp; slot += 64; }
                slot->x = (uint32_t)v.R1; slot->y = __float_as_uint(v.E1);
                if (v.E1 < INFINITY) ++sp;
            } else {
                if (v.E3 < INFINITY) wide_push(wlds, wdeep, sp, v.R3, v.E3);
                if (v.E2 < INFINITY) wide_push(wlds, wdeep, sp, v.R2, v.E2);
                if (v.E1 < INFINITY) wide_push(wlds, wdeep, sp, v.R1, v.E1);
            }
            cur = v.R0;
            if (!(v.E0 < INFINITY)) {
                cur = RT_NONE;
                while (sp > 0) {
                    --sp;
                    const uint2 e = wide_top(wlds, wdeep, sp);
                    if (bw.closest > __uint_as_float(e.y)) { cur = (int)e.x; break; }
                }
            }
        }
        if (cur >= RT_REF_INSTANCE && cur != RT_NONE) {
            // into an object: its tree is walked by the ray in the object's space
            obj = cur - RT_REF_INSTANCE;
            ob = S.objects + obj;
            lo = xform(ob->inv, org, 1.0f);
            ld = normalize3(xform(ob->inv, dirN, 0.0f));
            linv = ld;
            parL = lean_inverse(ld, linv);
            cur = ob->root;
        } else if (cur < 0) {      // at a leaf (a lane whose descent was cut short goes on in the next round)
            bool stop = false;
            const float4* tp = reinterpret_cast<const float4*>(S.tris + ~cur);
            float4 t0 = tp[0], t1 = tp[1], t2 = tp[2];
            for (;;) {
                const int remaining = __float_as_int(t0.w);
                float4 n0 = t0, n1 = t1, n2 = t2;
                if (RT_PREFETCH_ALL || remaining > 1) { tp += 3; n0 = tp[0]; n1 = tp[1]; n2 = tp[2]; }
                RT_WORK_LANES(6);
                if (triangle_hit(t0, t1, t2, lo, ld, ob->fwd, org, dirN, tMin, obj, bw) && ANYHIT) { stop = true; break; }
                if (remaining <= 1) break;
                t0 = n0; t1 = n1; t2 = n2;
            }
            cur = RT_NONE;
            if (stop) sp = 0;
            while (sp > 0) {
                --sp;
                const uint2 e = wide_top(wlds, wdeep, sp);
                if (bw.closest > __uint_as_float(e.y)) { cur = (int)e.x; break; }
            }
        }
    }
}

// The leaf parents (objects that are one leaf: the walls of a room, a floor) of a scene whose hits do not depend on the
// visiting order, every lane for itself. The wave-uniform loop of trace_pair enters a wall when ANY lane's ray meets its
// box, and a ray in a closed room meets the boxes of several walls before the one it hits: five of six walls entered per
// call with a fifth of the lanes busy in each. Here a first pass (still wave-uniform: boxes through scalar loads) gives
// every ray the wall whose box it enters FIRST, the runner-up and a mask of the rest; then every lane tests its own
// ray's nearest wall - object record and triangles through per-lane loads - and goes on to the next one only if that box
// begins in front of the closest hit so far, which for a ray inside a room it almost never does: one test per ray with
// every lane busy. Walls [base, base + count), count <= 32. A then B, as in the walks.
#ifndef RT_LANE_WALLS
#define RT_LANE_WALLS 1
#endif
// One ray's walls: candidates (mask of the walls whose boxes it enters in front of its closest hit), the nearest one
// and the runner-up with the distance at which the runner-up's box begins. k: the wall to test next, -1 none, -2 "measure
// what is left again".
struct WallRay { unsigned m; int k, k2; float e2; };

template <bool ANYHIT>
__device__ __forceinline__ void wall_step(const DevScene& S, int base, WallRay& w, V3 org, V3 dirN, float tMin, Best& best)
{
    const DevObject* ob = S.objects + base + w.k;      // per lane
    RT_WORK_LANES(8);
    const V3 lo = xform(ob->inv, org, 1.0f), ld = normalize3(xform(ob->inv, dirN, 0.0f));
    const float4* tp = reinterpret_cast<const float4*>(S.tris + ~ob->root);
    bool hit = false;
    for (;;) {
        const float4 t0 = tp[0], t1 = tp[1], t2 = tp[2];
        if (triangle_hit(t0, t1, t2, lo, ld, ob->fwd, org, dirN, tMin, base + w.k, best)) { hit = true; if (ANYHIT) break; }
        if (__float_as_int(t0.w) <= 1) break;
        tp += 3;
    }
    w.m &= ~(1u << w.k);
    // the next wall of this ray: the runner-up if its box begins in front of the closest hit (nothing else that is left
    // begins before it); beyond the two nearest the boxes are measured again (rare)
    int next = -1;
    if (!(ANYHIT && hit) && w.m != 0) {
        if (w.k2 >= 0 && (w.m & (1u << w.k2))) { if (best.closest > w.e2) next = w.k2; }
        else next = -2;
    }
    if (next == -1) w.m = 0;
    w.k = next;
}

template <bool ANYHIT>
__device__ __forceinline__ void walls_pair(const DevScene& S, int base, int count, V3 oA, V3 dA, V3 winvA, bool liveA, Best& bA,
                                           V3 oB, V3 dB, V3 winvB, bool liveB, Best& bB, bool anyParA, bool anyParB, float tMin)
{
    float e1A = INFINITY, e1B = INFINITY;
    WallRay wA, wB;
    wA.m = 0; wA.k = -1; wA.k2 = -1; wA.e2 = INFINITY;
    wB = wA;
    for (int k = 0; k < count; ++k) {
        const DevObject* ob = S.objects + base + k;      // wave-uniform address
        if (!(ob->root > RT_REF_LIVE_MIN)) continue;
        const V3 bmn = ld3(ob->bmin), bmx = ld3(ob->bmax);
        RT_WORK(0, (liveA ? 1 : 0) + (liveB ? 1 : 0));
        float E;
        if (liveA && (anyParA ? lean_box<true>(oA, winvA, bmn, bmx, tMin, E) : lean_box<false>(oA, winvA, bmn, bmx, tMin, E)) && (bA.closest > E)) {
            wA.m |= 1u << k;
            if (E < e1A) { wA.e2 = e1A; wA.k2 = wA.k; e1A = E; wA.k = k; }
            else if (E < wA.e2) { wA.e2 = E; wA.k2 = k; }
        }
        if (liveB && (anyParB ? lean_box<true>(oB, winvB, bmn, bmx, tMin, E) : lean_box<false>(oB, winvB, bmn, bmx, tMin, E)) && (bB.closest > E)) {
            wB.m |= 1u << k;
            if (E < e1B) { wB.e2 = e1B; wB.k2 = wB.k; e1B = E; wB.k = k; }
            else if (E < wB.e2) { wB.e2 = E; wB.k2 = k; }
        }
    }
    while (__any((wA.m | wB.m) != 0)) {
        if (wA.m != 0 && wA.k >= 0) wall_step<ANYHIT>(S, base, wA, oA, dA, tMin, bA);
        if (wB.m != 0 && wB.k >= 0) wall_step<ANYHIT>(S, base, wB, oB, dB, tMin, bB);
        if (__any((wA.m != 0 && wA.k == -2) || (wB.m != 0 && wB.k == -2))) {
            // measure the remaining boxes of the rays that need it again and take the nearest one in front of the closest hit
            const bool needA = wA.m != 0 && wA.k == -2, needB = wB.m != 0 && wB.k == -2;
            float eA = INFINITY, eB = INFINITY;
            int kA = -1, kB = -1;
            for (int j = 0; j < count; ++j) {
                const DevObject* ob = S.objects + base + j;      // wave-uniform address
                const V3 bmn = ld3(ob->bmin), bmx = ld3(ob->bmax);
                float E;
                if (needA && (wA.m & (1u << j)) && (anyParA ? lean_box<true>(oA, winvA, bmn, bmx, tMin, E) : lean_box<false>(oA, winvA, bmn, bmx, tMin, E)) &&
                    (bA.closest > E) && E < eA) { eA = E; kA = j; }
                if (needB && (wB.m & (1u << j)) && (anyParB ? lean_box<true>(oB, winvB, bmn, bmx, tMin, E) : lean_box<false>(oB, winvB, bmn, bmx, tMin, E)) &&
                    (bB.closest > E) && E < eB) { eB = E; kB = j; }
            }
            if (needA) { wA.k = kA; wA.k2 = -1; if (kA < 0) wA.m = 0; }
            if (needB) { wB.k = kB; wB.k2 = -1; if (kB < 0) wB.m = 0; }
        }
    }
}

// GetHit for two rays per lane (see above). onA / onB: the lane has such a ray. ANYHIT: shadow rays,
// leaf parents first, a ray stops at its first accepted hit.
// NF: 0 = the reference-order scene (near-first handling compiled out); 1 = the scene holds rebuilt two-child trees
// (DevScene::nearFirstBase set); 2 = wide scene, four-child records, parents found box by box; 3 = wide scene with the
// hierarchy over the parents (walk_tlas_pair); 4 = the reference-order scene with its trees folded into four-child records
// whose slots stand in the reference's visiting order (same hits, bit for bit).
template <bool ANYHIT, int NF>
__device__ __forceinline__ void trace_pair(const DevScene& S, int parentCount, V3 oA, V3 dA, bool onA, Best& bA, V3 oB, V3 dB, bool onB,
                                           Best& bB, float tMin, float tMax, uint2* stack)
{
    RT_SEC_BEGIN(secTrace);
    RT_WORK(2, (onA ? 1 : 0) + (onB ? 1 : 0));
    best_reset(bA, tMax);
    best_reset(bB, tMax);
    V3 winvA, winvB;
    const bool parWA = lean_inverse(dA, winvA) && onA, parWB = lean_inverse(dB, winvB) && onB;
    const bool permute = ANYHIT && parentCount == S.objectCount;
    const bool tightScene = NF != 0 && NF != 4;   // the near-first scene: boxes are only culls
    int pos = 0;
    if ((NF == 2 || NF == 3) && RT_LANE_WALLS) {
        // a wide scene stores its leaf parents first (in both orders): every lane goes through its own rays' walls, nearest first
        RT_SEC_BEGIN(secWall);
        for (; pos < S.anyLeafCount; pos += 32) {
            const bool liveA = onA && !(ANYHIT && bA.object >= 0), liveB = onB && !(ANYHIT && bB.object >= 0);
            walls_pair<ANYHIT>(S, pos, min(32, S.anyLeafCount - pos), oA, dA, winvA, liveA, bA, oB, dB, winvB, liveB, bB, __any(parWA), __any(parWB), tMin);
        }
        pos = S.anyLeafCount;
        RT_SEC_END(ANYHIT ? 7 : 2, secWall);
    }
    for (; pos < parentCount; ++pos) {
        const int i = permute ? S.anyOrder[pos] : pos;
        const DevObject* ob = S.objects + i;   // wave-uniform address
        const bool liveA = onA && !(ANYHIT && bA.object >= 0), liveB = onB && !(ANYHIT && bB.object >= 0);
        if (ANYHIT && !__any(liveA || liveB)) break;
        if (NF == 3 && ob->root >= 0) {
            // a wide scene keeps its BVH parents behind the leaf parents, in both orders: all of them in one walk
            RT_SEC_BEGIN(secRun);
            walk_tlas_pair<ANYHIT>(S, oA, dA, winvA, liveA, bA, oB, dB, winvB, liveB, bB, __any(parWA || parWB), tMin, stack);
            RT_SEC_END(ANYHIT ? 6 : 3, secRun);
            break;
        }
        if (NF != 3 && ob->root >= 0) {
            // consecutive BVH parents (up to 64): RT_RUN_MIN or more are walked lane by lane
            int count = permute ? S.runLenAny[pos] : S.runLen[pos];
            count = count < parentCount - pos ? count : parentCount - pos;
            if (count >= (NF == 2 || NF == 4 ? 1 : RT_RUN_MIN)) {      // (four-child records are only walked by the run walker)
                RT_SEC_BEGIN(secRun);
                walk_run_pair<ANYHIT, NF>(S, permute, pos, count, oA, dA, winvA, liveA, bA, oB, dB, winvB, liveB, bB, __any(parWA || parWB), tMin, stack);
                RT_SEC_END(ANYHIT ? 6 : 3, secRun);
                pos += count - 1;
                continue;
            }
        }
        if ((NF == 2 || NF == 3) && RT_LANE_WALLS) continue;      // not reached: leaf parents were handled above, BVH parents in runs or under the hierarchy
        RT_WORK(0, (liveA ? 1 : 0) + (liveB ? 1 : 0));
        const V3 bmn = ld3(ob->bmin), bmx = ld3(ob->bmax);
        float E;
        bool okA = liveA && (__any(parWA) ? lean_box<true>(oA, winvA, bmn, bmx, tMin, E) : lean_box<false>(oA, winvA, bmn, bmx, tMin, E)) &&
                   (bA.closest > E);
        if (okA && !tightScene) okA = !(distance_to_aabb(oA, bmn, bmx) > bA.closest);
        bool okB = liveB && (__any(parWB) ? lean_box<true>(oB, winvB, bmn, bmx, tMin, E) : lean_box<false>(oB, winvB, bmn, bmx, tMin, E)) &&
                   (bB.closest > E);
        if (okB && !tightScene) okB = !(distance_to_aabb(oB, bmn, bmx) > bB.closest);
        const int root = ob->root;   // wave-uniform
        if (!(root > RT_REF_LIVE_MIN) || !__any(okA || okB)) continue;
        if (root < 0) {
            // one leaf (a wall): records through the scalar cache, both rays test them
            RT_SEC_BEGIN(secWall);
            const V3 loA = xform(ob->inv, oA, 1.0f), ldA = normalize3(xform(ob->inv, dA, 0.0f));
            const V3 loB = xform(ob->inv, oB, 1.0f), ldB = normalize3(xform(ob->inv, dB, 0.0f));
            const float4* tp = reinterpret_cast<const float4*>(S.tris + ~root);
            for (;;) {
                const float4 t0 = tp[0], t1 = tp[1], t2 = tp[2];
                if (okA && triangle_hit(t0, t1, t2, loA, ldA, ob->fwd, oA, dA, tMin, i, bA) && ANYHIT) okA = false;
                if (okB && triangle_hit(t0, t1, t2, loB, ldB, ob->fwd, oB, dB, tMin, i, bB) && ANYHIT) okB = false;
                if (__float_as_int(t0.w) <= 1) break;
                tp += 3;
            }
            RT_SEC_END(ANYHIT ? 7 : 2, secWall);
            continue;
        }
        if (!(okA || okB)) continue;

        // the ray being walked: A first, then B
        bool second = !okA;
        V3 org = second ? oB : oA, dirN = second ? dB : dA;
        Best bw = second ? bB : bA;
        V3 lo = xform(ob->inv, org, 1.0f);
        V3 ld = normalize3(xform(ob->inv, dirN, 0.0f));
        V3 linv = ld;
        bool parL = lean_inverse(ld, linv);
        int cur = root, sp = 0;
        const bool nearFirst = NF == 1 && root >= S.nearFirstBase;   // wave-uniform: a rebuilt tree, nearer child first
        if (NF >= 2) continue;      // not reached: every BVH parent of a wide scene is in a run or under the hierarchy
        lds_uint2* const wlds = (lds_uint2*)stack;
        WideDeep wdeep;
        for (;;) {
            int budget = RT_DESCENT_CAP;
            while ((unsigned)cur < (unsigned)RT_NONE && budget-- > 0) {
                NodeVisit nv;
                visit_node(S, cur, lo, linv, __any(parL), tMin, nv);
                int2 refs = nv.refs;
                const bool goF = refs.x > RT_REF_LIVE_MIN && nv.okF && (bw.closest > nv.EF);
                bool canS = refs.y > RT_REF_LIVE_MIN && nv.okS;
                if (nearFirst) {
                    canS = canS && (bw.closest > nv.ES);
                    if (goF && canS && nv.ES < nv.EF) {
                        const int r = refs.x; refs.x = refs.y; refs.y = r;
                        const float e = nv.EF; nv.EF = nv.ES; nv.ES = e;
                    }
                }
                if (goF) {
                    if (canS) wide_push(wlds, wdeep, sp, refs.y, nv.ES);
                    cur = refs.x;
                } else if (canS && (bw.closest > nv.ES)) {
                    cur = refs.y;
                } else {
                    cur = RT_NONE;
                    while (sp > 0) {
                        --sp;
                        uint2 e = wide_top(wlds, wdeep, sp);
                        if (bw.closest > __uint_as_float(e.y)) { cur = (int)e.x; break; }
                    }
                }
            }
            if (cur < 0) {      // at a leaf (a lane whose descent was cut short goes on in the next round)
                int k = ~cur;
                bool stop = false;
                for (;;) {
                    const float4* tp = reinterpret_cast<const float4*>(S.tris + k);
                    float4 t0 = tp[0], t1 = tp[1], t2 = tp[2];
                    const int remaining = __float_as_int(t0.w);
                    if (triangle_hit(t0, t1, t2, lo, ld, ob->fwd, org, dirN, tMin, i, bw) && ANYHIT) { stop = true; break; }
                    if (remaining <= 1) break;
                    ++k;
                }
                cur = RT_NONE;
                if (stop) sp = 0;
                while (sp > 0) {
                    --sp;
                    uint2 e = wide_top(wlds, wdeep, sp);
                    if (bw.closest > __uint_as_float(e.y)) { cur = (int)e.x; break; }
                }
            }
            if (cur == RT_NONE) {
                // this ray has nothing left in this object
                if (second) { bB = bw; break; }
                bA = bw;
                if (!okB) break;
                second = true;
                org = oB; dirN = dB; bw = bB;
                lo = xform(ob->inv, org, 1.0f);
                ld = normalize3(xform(ob->inv, dirN, 0.0f));
                linv = ld;
                parL = lean_inverse(ld, linv);
                cur = root; sp = 0;
            }
        }
    }
    RT_SEC_END(1, secTrace);
}

// The part of RayColour's loop body between two GetHit calls (raytrace.comp:701-766) for one path:
// takes the regular hit, adds sky / emission, ends the path or prepares the shadow ray (sun term kept
// aside until the shadow ray's answer is in) and the next regular ray.
// RES: result, throughput and the scatter vectors of the path live in LDS at `res` (see above); the fields of `p`
// are then only scratch.
template <bool RES>
__device__ __forceinline__ void pair_shade(const DevScene& S, const rt_push_constants& pc, PathState& p, V3 dirN, const Best& best, uint32_t seed, float* res)
{
    p.shadow = false;
    if (!p.alive) return;
    HitRec rec;
    finish_hit(S, p.o, dirN, best, rec);
    if (RES && (RT_RES_SET & 2)) p.throughput = res_ld(res, RES_THROUGHPUT);
    if (!rec.hit) {
        const V3 sky = p.throughput * environment_colour<true>(pc, p.d);
        if (RES && (RT_RES_SET & 1)) res_st(res, RES_RESULT, res_ld(res, RES_RESULT) + sky);
        else p.result = p.result + sky;
        p.alive = false;
        return;
    }
    if (!rec.frontFace) { p.alive = false; return; }
    DevMaterial m = load_material(S, rec.matIndex);
    V3 emitted = ld3(m.colour) * m.emission;
    if (RES && (RT_RES_SET & 1)) res_st(res, RES_RESULT, res_ld(res, RES_RESULT) + p.throughput * emitted);
    else p.result = p.result + p.throughput * emitted;
    if (m.emission > 0.0f) { p.alive = false; return; }
    V3 negSun = -ld3(pc.sunDirection);
    float nDotL = gmax(dot3(rec.normal, negSun), 0.0f);
    if (nDotL > 0.0f) {
        p.shadow = true;
        p.so = rec.point + rec.normal * 0.01f;
        V3 sun = ld3(pc.sunColour) * pc.sunIntensity;
        const float INV_PI = 0.31830988618f;
        V3 direct = sun * (nDotL * INV_PI);
        if (m.refractionIndex == 0.0f) p.sun = (p.throughput * ld3(m.colour)) * direct;
        else p.sun = (p.throughput * reflectance(nDotL, m.refractionIndex)) * direct;
    }
    V3 attenuation;
    Ray scattered;
    if (RES && (RT_RES_SET & 4)) { p.draws.lambert = res_ld(res, RES_LAMBERT); p.draws.metal = res_ld(res, RES_METAL); }
    material_scatter<true>(m, p.d, rec, seed, attenuation, scattered, &p.draws);
    p.throughput = p.throughput * attenuation;
    if (RES && (RT_RES_SET & 2)) res_st(res, RES_THROUGHPUT, p.throughput);
    p.o = scattered.o;
    p.d = normalize3(scattered.d);
}

// SeedFromCoords of entry m of a pixel's sample sequence (frame m / rpp of the launch, sample m % rpp)
__device__ __forceinline__ uint32_t pair_seed(const rt_push_constants& pc, const DevLaunch& L, int x, int y, int m, int rpp)
{
    const int fr = m / rpp, i = m - fr * rpp;
    return seed_from_coords((uint32_t)x, (uint32_t)y, (uint32_t)(pc.frame + fr * L.frameStride), (uint32_t)i);
}

__device__ __forceinline__ float linear_to_gamma(float c) { return c > 0.0f ? sqrtf(c) : 0.0f; }

__device__ __forceinline__ int wave_sum(int v)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

} // namespace

// main(), raytrace.comp:857-875. One 64-lane block renders one 8x8 pixel tile.
// statsOnly != 0: write no image, only report the work counters: 1 = added to *counters, 2 = every
// wave stores its own record at counters[blockIdx.x] (the caller sums them).
#ifndef RT_MEGA_WAVES
#define RT_MEGA_WAVES 4
#endif
template <bool COUNT>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(RT_MEGA_WAVES))) void RT_SYM(rt_megakernel)(DevScene S, rt_push_constants pc, DevLaunch L,
                                                            float4* __restrict__ accum, float4* __restrict__ out,
                                                            DevCounters* counters, int statsOnly)
{
    extern __shared__ uint2 stackMem[];
    const int lane = threadIdx.x;
    uint2* stack = stackMem + lane;

    const int wavesX = L.tileSize >> 3;
    const int wavesPerTile = wavesX * wavesX;
    const int ownedTile = blockIdx.x / wavesPerTile;
    const int sub = blockIdx.x - ownedTile * wavesPerTile;
    const int tile = L.tileRank + ownedTile * L.tileWorld;
    int tx, ty;
    rt_tile_of_position(tile, L.tilesX, &tx, &ty);
    const int sy = sub / wavesX, sx = sub - sy * wavesX;
    const int x = tx * L.tileSize + sx * 8 + (lane & 7);
    const int y = ty * L.tileSize + sy * 8 + (lane >> 3);
    const bool inside = x < L.width && y < L.height;

    Counters cnt;
    cnt.rays = 0; cnt.nodeTests = 0; cnt.triTests = 0;
    int samples = 0;
#ifdef RT_SECTION_TIMERS
    if (lane < 8) rt_sec_lds()[lane] = 0;
    __syncthreads();
#endif
    RT_SEC_BEGIN(secKernel);

    if (inside) {
        const size_t idx = (size_t)y * (size_t)L.width + (size_t)x;
        const int mode = pc.renderMode;
        if (mode == 0 || mode == 3 || mode == 4) {
            // Pathtrace, raytrace.comp:771-817
            // One dispatch of the shader renders one frame. L.frames > 1 (path tracing only) runs that
            // many consecutive dispatches of this pixel back to back - frames pc.frame,
            // pc.frame + L.frameStride, ... - with the accumulation pixel held in registers in
            // between: the same additions in the same order as separate dispatches, the draw image
            // written once, for the last frame.
            const int frames = (mode == 0 && !statsOnly) ? L.frames : 1;
            const int lastFrame = pc.frame + (frames - 1) * L.frameStride;
            float raysPerPixelRatio = 1.0f / (float)pc.raysPerPixel;
            float accumulationMultiplier = pc.accumulateFrames == 1 ? 1.0f / (float)(lastFrame + 1) : 1.0f;
            V3 nc = mk(0.0f, 0.0f, 0.0f);
            float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            for (int fr = 0; fr < frames; ++fr) {
                const uint32_t frame = (uint32_t)(pc.frame + fr * L.frameStride);
                nc = mk(0.0f, 0.0f, 0.0f);
                for (int i = 0; i < pc.raysPerPixel; ++i) {
                    uint32_t seed = seed_from_coords((uint32_t)x, (uint32_t)y, frame, (uint32_t)i);
                    Ray ray = get_ray(pc, x, y, seed);
                    nc = nc + ray_colour<COUNT>(S, pc, ray, seed, cnt, stack);
                    samples++;
                }
                if (mode == 0 && !statsOnly) {
                    // the pixel goes through memory between frames (it stays in the cache): keeping it in
                    // registers across the whole path costs more than the round trip
                    acc = accum[idx];
                    nc = nc * raysPerPixelRatio;
                    acc.x += nc.x; acc.y += nc.y; acc.z += nc.z;
                    accum[idx] = acc;
                }
            }
            if (!statsOnly) {
                if (mode == 0) {
                    float4 avg = pc.accumulateFrames == 1 ? acc : make_float4(nc.x, nc.y, nc.z, 1.0f);
                    avg.x *= accumulationMultiplier; avg.y *= accumulationMultiplier; avg.z *= accumulationMultiplier;
                    out[idx] = make_float4(linear_to_gamma(avg.x), linear_to_gamma(avg.y), linear_to_gamma(avg.z), avg.w);
                } else {
                    int threshold = mode == 3 ? pc.triangleTestThreshold : pc.bvhNodeTestThreshold;
                    int target = mode == 3 ? cnt.triTests : cnt.nodeTests;
                    int grey = (int)((float)target / (float)threshold * 255.0f);
                    grey = grey < 0 ? 0 : (grey > 255 ? 255 : grey);
                    float g = (float)grey / 255.0f;
                    float4 c = make_float4(g, g, g, 1.0f);
                    if (cnt.triTests > threshold) c = make_float4(1.0f, 0.0f, 0.0f, 1.0f);
                    out[idx] = c;
                }
            }
        } else if (mode == 1) {
            // RenderBoundingBoxes, raytrace.comp:819-835
            Ray ray = get_ray(pc, x, y, 123u);
            SlabRay wr = make_slab_ray(ray.o, ray.d);
            float red = 0.0f;
            for (int i = 0; i < pc.parentBVHCount; ++i) {
                const DevObject* ob = S.objects + i;
                float E;
                if (slab_box(wr, ld3(ob->bmin), ld3(ob->bmax), 0.001f, E) && (1e20f > E)) { red = 1.0f; break; }
            }
            if (!statsOnly) out[idx] = make_float4(red, 0.0f, 0.0f, 1.0f);
        } else if (mode == 2) {
            // RenderDepth, raytrace.comp:837-855
            Ray ray = get_ray(pc, x, y, 123u);
            V3 dirN = normalize3(ray.d);
            Best best;
            trace<COUNT, false>(S, pc.parentBVHCount, ray.o, dirN, 0.001f, 1e20f, best, cnt, stack);
            HitRec rec;
            finish_hit(S, ray.o, dirN, best, rec);
            float d = 0.0f;
            if (rec.hit) {
                d = length3(rec.point - ld3(pc.cameraPosition));
                d = gclamp(d / pc.depthDebugScale, 0.0f, 1.0f);
            }
            if (!statsOnly) out[idx] = make_float4(d, d, d, 1.0f);
        }
    }

#ifdef RT_SECTION_TIMERS
    __syncthreads();
    if (lane == 0) {
        unsigned* a = rt_sec_lds();
        a[0] = (unsigned)__builtin_amdgcn_s_memtime() - secKernel;
        for (int k = 0; k < 8; ++k) atomicAdd(&g_sectionCycles[k], (unsigned long long)a[k]);
    }
#endif
    if (COUNT && counters != nullptr) {
        int r = wave_sum(cnt.rays), n = wave_sum(cnt.nodeTests), t = wave_sum(cnt.triTests), s = wave_sum(samples);
        if (lane == 0 && statsOnly == 2) {
            // every wave stores its own record (no record is shared, nothing has to start at zero)
            DevCounters c;
            c.rays = (unsigned long long)r; c.nodeTests = (unsigned long long)n; c.triTests = (unsigned long long)t;
            c.samples = (unsigned long long)s;
            counters[blockIdx.x] = c;
        } else if (lane == 0) {
            atomicAdd(&counters->rays, (unsigned long long)r);
            atomicAdd(&counters->nodeTests, (unsigned long long)n);
            atomicAdd(&counters->triTests, (unsigned long long)t);
            atomicAdd(&counters->samples, (unsigned long long)s);
        }
    }
}

// Pathtrace (raytrace.comp:771-817) with two samples per lane: the samples of a pixel - frames
// pc.frame, pc.frame + L.frameStride, ... (L.frames of them), pc.raysPerPixel samples each - are taken
// two at a time, traced together by trace_pair() and added to the pixel in sequence order.
#ifndef RT_PAIR_WAVES
#define RT_PAIR_WAVES 4
#endif
template <int NF>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(RT_PAIR_WAVES))) void RT_SYM(rt_pairkernel)(DevScene S, rt_push_constants pc, DevLaunch L,
                                                            float4* __restrict__ accum, float4* __restrict__ out, float4* __restrict__ partial)
{
    extern __shared__ uint2 stackMem[];
    const int lane = threadIdx.x;
    uint2* stack = stackMem + lane;
#ifdef RT_SECTION_TIMERS
    if (lane < 8) rt_sec_lds()[lane] = 0;
#endif
    RT_SEC_BEGIN(secKernel);
    RT_WORK_BEGIN();
    constexpr bool RES = true;
    float* const res = reinterpret_cast<float*>(stackMem + S.stackDepth * 64) + lane;      // (wide kernels: LDS-resident path state)

    const int wavesX = L.tileSize >> 3;
    const int wavesPerTile = wavesX * wavesX;
    // a launch in pieces: this block renders piece `piece` of the launch's frames for its 8x8 pixels
    const int piece = partial ? (int)blockIdx.x % L.pieces : 0;
    const int wave = partial ? (int)blockIdx.x / L.pieces : (int)blockIdx.x;
    const int ownedTile = wave / wavesPerTile;
    const int sub = wave - ownedTile * wavesPerTile;
    const int tile = L.tileRank + ownedTile * L.tileWorld;
    int tx, ty;
    rt_tile_of_position(tile, L.tilesX, &tx, &ty);
    const int sy = sub / wavesX, sx = sub - sy * wavesX;
    const int x = tx * L.tileSize + sx * 8 + (lane & 7);
    const int y = ty * L.tileSize + sy * 8 + (lane >> 3);
    if (!(x < L.width && y < L.height)) return;
    const size_t idx = (size_t)y * (size_t)L.width + (size_t)x;

    const int rpp = pc.raysPerPixel;
    const int framesHere = partial ? min(L.pieceFrames, L.frames - piece * L.pieceFrames) : L.frames;
    const int total = framesHere * rpp;               // samples of this pixel in this launch (in this piece of it)
    const int mBase = piece * L.pieceFrames * rpp;    // entry of the launch's sample sequence this block starts at
    const size_t pixels = (size_t)L.width * (size_t)L.height;
    const int bounces = pc.maxBounces + 1;
    const float raysPerPixelRatio = 1.0f / (float)rpp;
    const V3 sunDirN = normalize3(-ld3(pc.sunDirection));
    V3 nc = mk(0.0f, 0.0f, 0.0f);
    // the frame's running sum waits in LDS while the lane's samples are traced, in the (otherwise unused) result slot of path 0
    constexpr bool NC_LDS = RES && !(RT_RES_SET & 1);
    if (NC_LDS) res_st(res, RES_RESULT, nc);

    for (int n = 0; n < total; n += 2) {
        PathState P[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int m = n + q;
            PathState& p = P[q];
            p.alive = m < total;
            p.shadow = false;
            p.throughput = mk(1.0f, 1.0f, 1.0f);
            p.result = mk(0.0f, 0.0f, 0.0f);
            p.sun = mk(0.0f, 0.0f, 0.0f);
            p.so = mk(0.0f, 0.0f, 0.0f);
            const uint32_t seed = pair_seed(pc, L, x, y, mBase + m, rpp);
            Ray ray = get_ray(pc, x, y, seed);
            p.o = ray.o; p.d = normalize3(ray.d);
            p.draws = scatter_draws(seed);
            if (RES) {
                float* const r = res + q * RES_PER_PATH * 64;
                if (RT_RES_SET & 1) res_st(r, RES_RESULT, p.result);
                if (RT_RES_SET & 2) res_st(r, RES_THROUGHPUT, p.throughput);
                if (RT_RES_SET & 4) { res_st(r, RES_LAMBERT, p.draws.lambert); res_st(r, RES_METAL, p.draws.metal); }
            }
        }
        for (int bounce = 0; bounce < bounces; ++bounce) {
            if (!(P[0].alive || P[1].alive)) break;
            const V3 dirA = P[0].d, dirB = P[1].d;
            Best bA, bB;
            trace_pair<false, NF>(S, pc.parentBVHCount, P[0].o, dirA, P[0].alive, bA, P[1].o, dirB, P[1].alive, bB, 0.001f, 1e20f, stack);
            // (the seed of a sample is recomputed where it is needed rather than carried through the walks)
            RT_SEC_BEGIN(secShade);
            pair_shade<RES>(S, pc, P[0], dirA, bA, pair_seed(pc, L, x, y, mBase + n, rpp), res);
            pair_shade<RES>(S, pc, P[1], dirB, bB, pair_seed(pc, L, x, y, mBase + n + 1, rpp), res + RES_PER_PATH * 64);
            RT_SEC_END(4, secShade);
            if (P[0].shadow || P[1].shadow) {
                trace_pair<true, NF>(S, pc.parentBVHCount, P[0].so, sunDirN, P[0].shadow, bA, P[1].so, sunDirN, P[1].shadow, bB, 0.001f, 1e20f, stack);
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    if (!(P[q].shadow && (q ? bB : bA).object < 0)) continue;
                    if (RES && (RT_RES_SET & 1)) res_st(res + q * RES_PER_PATH * 64, RES_RESULT, res_ld(res + q * RES_PER_PATH * 64, RES_RESULT) + P[q].sun);
                    else P[q].result = P[q].result + P[q].sun;
                }
            }
        }
        if (NC_LDS) nc = res_ld(res, RES_RESULT);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int m = n + q;
            if (m >= total) break;
            nc = nc + (RES && (RT_RES_SET & 1) ? res_ld(res + q * RES_PER_PATH * 64, RES_RESULT) : P[q].result);
            const int fr = m / rpp, i = m - fr * rpp;
            if (i == rpp - 1) {
                // end of a frame: the pixel goes through memory as in the one-sample kernel
                nc = nc * raysPerPixelRatio;
                if (partial) {
                    partial[(size_t)(piece * L.pieceFrames + fr) * pixels + idx] = make_float4(nc.x, nc.y, nc.z, 0.0f);
                } else {
                    float4 acc = accum[idx];
                    acc.x += nc.x; acc.y += nc.y; acc.z += nc.z;
                    accum[idx] = acc;
                }
                if (m + 1 < total) nc = mk(0.0f, 0.0f, 0.0f);
            }
        }
        if (NC_LDS) res_st(res, RES_RESULT, nc);
    }
    if (partial) { RT_WORK_END(); return; }      // rt_sum_pieces adds the pieces up and writes the draw image
    if (NC_LDS) nc = res_ld(res, RES_RESULT);
    const int lastFrame = pc.frame + (L.frames - 1) * L.frameStride;
    const float accumulationMultiplier = pc.accumulateFrames == 1 ? 1.0f / (float)(lastFrame + 1) : 1.0f;
    // (the pixel as this lane stored it after its last sample: read back rather than kept in registers across the samples)
    float4 avg = pc.accumulateFrames == 1 ? accum[idx] : make_float4(nc.x, nc.y, nc.z, 1.0f);
    avg.x *= accumulationMultiplier; avg.y *= accumulationMultiplier; avg.z *= accumulationMultiplier;
    out[idx] = make_float4(linear_to_gamma(avg.x), linear_to_gamma(avg.y), linear_to_gamma(avg.z), avg.w);
    RT_WORK_END();
#ifdef RT_SECTION_TIMERS
    if (lane == __ffsll((long long)__ballot(true)) - 1) {
        unsigned* a = rt_sec_lds();
        a[0] = (unsigned)__builtin_amdgcn_s_memtime() - secKernel;
        for (int k = 0; k < 8; ++k) atomicAdd(&g_sectionCycles[k], (unsigned long long)a[k]);
    }
#endif
}

// rt_pairkernel for launches that hold a single sample per pixel (one frame of 1 ray per pixel, the
// reference's interactive dispatch): the lane's two paths are the same sample of two PIXELS, (x, y) and
// (x, y + 1) - the wave covers 8 x 16 pixels - instead of two samples of one pixel. Same arithmetic per
// sample. (A kernel of its own rather than a flag of rt_pairkernel: sharing the body cost that kernel 0.7 %.)
template <int NF>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(RT_PAIR_WAVES))) void RT_SYM(rt_pixelpairkernel)(DevScene S, rt_push_constants pc, DevLaunch L,
                                                            float4* __restrict__ accum, float4* __restrict__ out)
{
    constexpr bool PIX = true;
    extern __shared__ uint2 stackMem[];
    const int lane = threadIdx.x;
    uint2* stack = stackMem + lane;
#ifdef RT_SECTION_TIMERS
    if (lane < 8) rt_sec_lds()[lane] = 0;
#endif
    RT_SEC_BEGIN(secKernel);
    RT_WORK_BEGIN();
    constexpr bool RES = true;
    float* const res = reinterpret_cast<float*>(stackMem + S.stackDepth * 64) + lane;      // (wide kernels: LDS-resident path state)

    const int wavesX = L.tileSize >> 3;
    const int wavesPerTile = PIX ? wavesX * (wavesX >> 1) : wavesX * wavesX;
    const int ownedTile = blockIdx.x / wavesPerTile;
    const int sub = blockIdx.x - ownedTile * wavesPerTile;
    const int tile = L.tileRank + ownedTile * L.tileWorld;
    int tx, ty;
    rt_tile_of_position(tile, L.tilesX, &tx, &ty);
    const int sy = sub / wavesX, sx = sub - sy * wavesX;
    const int x = tx * L.tileSize + sx * 8 + (lane & 7);
    const int y = PIX ? ty * L.tileSize + sy * 16 + 2 * (lane >> 3) : ty * L.tileSize + sy * 8 + (lane >> 3);
    if (!(x < L.width && y < L.height)) return;
    // q-th path of the lane: pixel (x, yq[q]); without PIX both are the lane's one pixel
    const int yq[2] = { y, PIX ? y + 1 : y };
    const bool have[2] = { true, PIX ? (y + 1 < L.height) : true };
    const size_t idxq[2] = { (size_t)y * (size_t)L.width + (size_t)x, (size_t)(have[1] ? yq[1] : y) * (size_t)L.width + (size_t)x };
    const size_t idx = idxq[0];

    const int rpp = pc.raysPerPixel;
    const int total = L.frames * rpp;                 // samples of a pixel in this launch
    const int bounces = pc.maxBounces + 1;
    const float raysPerPixelRatio = 1.0f / (float)rpp;
    const V3 sunDirN = normalize3(-ld3(pc.sunDirection));
    V3 nc = mk(0.0f, 0.0f, 0.0f), ncB = mk(0.0f, 0.0f, 0.0f);
    float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f), accB = acc;

    for (int n = 0; n < total; n += PIX ? 1 : 2) {
        PathState P[2];
        int mq[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int m = PIX ? n : n + q;
            mq[q] = m;
            PathState& p = P[q];
            p.alive = PIX ? have[q] : m < total;
            p.shadow = false;
            p.throughput = mk(1.0f, 1.0f, 1.0f);
            p.result = mk(0.0f, 0.0f, 0.0f);
            p.sun = mk(0.0f, 0.0f, 0.0f);
            p.so = mk(0.0f, 0.0f, 0.0f);
            const uint32_t seed = pair_seed(pc, L, x, yq[q], m, rpp);
            Ray ray = get_ray(pc, x, yq[q], seed);
            p.o = ray.o; p.d = normalize3(ray.d);
            p.draws = scatter_draws(seed);
            if (RES) {
                float* const r = res + q * RES_PER_PATH * 64;
                if (RT_RES_SET & 1) res_st(r, RES_RESULT, p.result);
                if (RT_RES_SET & 2) res_st(r, RES_THROUGHPUT, p.throughput);
                if (RT_RES_SET & 4) { res_st(r, RES_LAMBERT, p.draws.lambert); res_st(r, RES_METAL, p.draws.metal); }
            }
        }
        for (int bounce = 0; bounce < bounces; ++bounce) {
            if (!(P[0].alive || P[1].alive)) break;
            const V3 dirA = P[0].d, dirB = P[1].d;
            Best bA, bB;
            trace_pair<false, NF>(S, pc.parentBVHCount, P[0].o, dirA, P[0].alive, bA, P[1].o, dirB, P[1].alive, bB, 0.001f, 1e20f, stack);
            // (the seed of a sample is recomputed where it is needed rather than carried through the walks)
            RT_SEC_BEGIN(secShade);
            pair_shade<RES>(S, pc, P[0], dirA, bA, pair_seed(pc, L, x, yq[0], mq[0], rpp), res);
            pair_shade<RES>(S, pc, P[1], dirB, bB, pair_seed(pc, L, x, yq[1], mq[1], rpp), res + RES_PER_PATH * 64);
            RT_SEC_END(4, secShade);
            if (P[0].shadow || P[1].shadow) {
                trace_pair<true, NF>(S, pc.parentBVHCount, P[0].so, sunDirN, P[0].shadow, bA, P[1].so, sunDirN, P[1].shadow, bB, 0.001f, 1e20f, stack);
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    if (!(P[q].shadow && (q ? bB : bA).object < 0)) continue;
                    if (RES && (RT_RES_SET & 1)) res_st(res + q * RES_PER_PATH * 64, RES_RESULT, res_ld(res + q * RES_PER_PATH * 64, RES_RESULT) + P[q].sun);
                    else P[q].result = P[q].result + P[q].sun;
                }
            }
        }
        if (PIX) {
            // one sample of each of the two pixels
            const int fr = n / rpp, i = n - fr * rpp;
            nc = nc + (RES && (RT_RES_SET & 1) ? res_ld(res, RES_RESULT) : P[0].result);
            ncB = ncB + (RES && (RT_RES_SET & 1) ? res_ld(res + RES_PER_PATH * 64, RES_RESULT) : P[1].result);
            if (i == rpp - 1) {
                acc = accum[idxq[0]];
                nc = nc * raysPerPixelRatio;
                acc.x += nc.x; acc.y += nc.y; acc.z += nc.z;
                accum[idxq[0]] = acc;
                if (have[1]) {
                    accB = accum[idxq[1]];
                    ncB = ncB * raysPerPixelRatio;
                    accB.x += ncB.x; accB.y += ncB.y; accB.z += ncB.z;
                    accum[idxq[1]] = accB;
                }
                if (n + 1 < total) { nc = mk(0.0f, 0.0f, 0.0f); ncB = nc; }
            }
        } else {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int m = n + q;
                if (m >= total) break;
                nc = nc + (RES && (RT_RES_SET & 1) ? res_ld(res + q * RES_PER_PATH * 64, RES_RESULT) : P[q].result);
                const int fr = m / rpp, i = m - fr * rpp;
                if (i == rpp - 1) {
                    // end of a frame: the pixel goes through memory as in the one-sample kernel
                    acc = accum[idx];
                    nc = nc * raysPerPixelRatio;
                    acc.x += nc.x; acc.y += nc.y; acc.z += nc.z;
                    accum[idx] = acc;
                    if (m + 1 < total) nc = mk(0.0f, 0.0f, 0.0f);
                }
            }
        }
    }
    const int lastFrame = pc.frame + (L.frames - 1) * L.frameStride;
    const float accumulationMultiplier = pc.accumulateFrames == 1 ? 1.0f / (float)(lastFrame + 1) : 1.0f;
    float4 avg = pc.accumulateFrames == 1 ? acc : make_float4(nc.x, nc.y, nc.z, 1.0f);
    avg.x *= accumulationMultiplier; avg.y *= accumulationMultiplier; avg.z *= accumulationMultiplier;
    out[idx] = make_float4(linear_to_gamma(avg.x), linear_to_gamma(avg.y), linear_to_gamma(avg.z), avg.w);
    if (PIX && have[1]) {
        float4 avgB = pc.accumulateFrames == 1 ? accB : make_float4(ncB.x, ncB.y, ncB.z, 1.0f);
        avgB.x *= accumulationMultiplier; avgB.y *= accumulationMultiplier; avgB.z *= accumulationMultiplier;
        out[idxq[1]] = make_float4(linear_to_gamma(avgB.x), linear_to_gamma(avgB.y), linear_to_gamma(avgB.z), avgB.w);
    }
    RT_WORK_END();
#ifdef RT_SECTION_TIMERS
    if (lane == __ffsll((long long)__ballot(true)) - 1) {
        unsigned* a = rt_sec_lds();
        a[0] = (unsigned)__builtin_amdgcn_s_memtime() - secKernel;
        for (int k = 0; k < 8; ++k) atomicAdd(&g_sectionCycles[k], (unsigned long long)a[k]);
    }
#endif
}

// ---------------------------------------------------------------------------------
// Wave-scheduled persistent megakernel.
//
// The lock-step kernel above keeps all 64 lanes inside the same GetHit call, so a
// wave pays for its slowest lane in every loop (PMC: 16 % of VALU lanes active).
// Here every lane carries its own small state machine over the same arithmetic and
// is in one of two states:
//   K_OS    outside a BVH walk: GetHit's object loop (parent boxes, leaf parents
//           such as the six walls) and everything between two GetHit calls (hit
//           record, emission, sun term, scatter, sample / pixel book-keeping,
//           fetching a new pixel)
//   K_TRAV  inside the walk of one BVH parent: ray into object space, then one
//           two-child record or one triangle per step
// The wave repeatedly votes (ballot + popcount) which of the two to run next. A BVH
// round steps only the walking lanes and ends when fewer than a quorum are left;
// the stragglers keep their place and continue in the next BVH round together with
// the lanes that arrived meanwhile, so one long walk does not hold 63 lanes up.
// Lanes of a finished path take the next pixel from a global counter, so a wave
// stays full until the image is done. A pixel is owned by one lane for all its
// samples, which keeps the reference's in-order accumulation (raytrace.comp:779-792).
//
// The loop-carried state is kept free of booleans (they would live in scalar
// register pairs and spill): a ray's slab constants are just origin and
// reciprocal direction. The "invD < 0" swap of raytrace.comp:230-236 is taken
// by min/max of the two plane distances, which is the same thing for a box with
// min <= max (the upload marks inverted boxes, which no ray can hit, as dead),
// and an axis with abs(dir) < 1e-8 gets a NaN reciprocal, which makes both
// plane distances NaN and leaves the interval untouched in fmaxf/fminf exactly
// as the shader's parallel branch does; its containment test runs in a separate
// wave-uniform slow path.
// Per-lane traversal state of the scheduled kernel.
struct LaneTrace {
    V3 rayO, dirN;      // world ray (dirN normalised)
    V3 lo, ld, linv;    // ray in the space of object `obj`, reciprocal direction (NaN = parallel axis)
    Best best;          // closest hit so far
    int obj;            // index in the parent array
    int cur;            // inside a walk: >= 0 two-child record, < 0 leaf triangle ~cur
    int sp;
};

// One triangle: IntersectTriangle (raytrace.comp:278-328) plus the acceptance test of
// raytrace.comp:559-572. Returns true when the hit was accepted.
__device__ __forceinline__ bool lane_triangle_loaded(LaneTrace& t, const float* fwd, float4 t0, float4 t1, float4 t2, float tMin, int& remaining);

__device__ __forceinline__ bool lane_triangle(const DevScene& S, LaneTrace& t, const float* fwd, int triIndex, float tMin, int& remaining)
{
    const float4* tp = reinterpret_cast<const float4*>(S.tris + triIndex);
    return lane_triangle_loaded(t, fwd, tp[0], tp[1], tp[2], tMin, remaining);
}

__device__ __forceinline__ bool lane_triangle_loaded(LaneTrace& t, const float* fwd, float4 t0, float4 t1, float4 t2, float tMin, int& remaining)
{
    remaining = __float_as_int(t0.w);
    return triangle_hit(t0, t1, t2, t.lo, t.ld, fwd, t.rayO, t.dirN, tMin, t.obj, t.best);
}

// Next stack entry whose box still reaches inside the closest hit, or RT_NONE.
__device__ __forceinline__ int lane_pop(LaneTrace& t, const uint2* stack)
{
    while (t.sp > 0) {
        --t.sp;
        uint2 e = stack[t.sp * 64];
        if (t.best.closest > __uint_as_float(e.y)) return (int)e.x;
    }
    return RT_NONE;
}

enum : int { K_OS = 0, K_TRAV = 1, K_IDLE = 2 };
#define RT_ENTER 0x7ffffffe   /* LaneTrace::cur: about to test the parent box of object `obj` */

template <bool COUNT, bool DIAG>
__global__ __launch_bounds__(64) void RT_SYM(rt_schedkernel)(DevScene S, rt_push_constants pc, DevLaunch L,
                                                            float4* __restrict__ accum, float4* __restrict__ out,
                                                            DevCounters* counters, int statsOnly,
                                                            unsigned long long* __restrict__ queue,
                                                            unsigned long long* __restrict__ diag)
{
    extern __shared__ uint2 stackMem[];
    const int lane = threadIdx.x;
    uint2* stack = stackMem + lane;

    const int bounces = pc.maxBounces + 1;
    const int parentCount = pc.parentBVHCount;
    const int rpp = pc.raysPerPixel;
    const int mode = pc.renderMode;
    const float tMin = 0.001f;
    const int wavesX = L.tileSize >> 3;
    const unsigned pixelsPerTile = (unsigned)L.tileSize * (unsigned)L.tileSize;
    const unsigned long long totalPixels = (unsigned long long)L.ownedTiles * pixelsPerTile;
    // shadow rays may visit leaf parents first (see trace()); not in the counting build, whose
    // counters are those of the reference order
    const bool permute = !COUNT && parentCount == S.objectCount;

    // ---- per-lane state (no booleans) ----
    int state = K_OS, flags = 0;
    int px = 0, py = 0, sample = 0, bounce = 0;
    uint32_t seed0 = 0;
    V3 nc = mk(0.0f, 0.0f, 0.0f);            // newColour.rgb of the pixel
    V3 result = mk(0.0f, 0.0f, 0.0f), throughput = mk(1.0f, 1.0f, 1.0f);
    V3 rayD = mk(0.0f, 0.0f, 1.0f);
    V3 winv = mk(1.0f, 1.0f, 1.0f);
    V3 pendD = mk(0.0f, 0.0f, 0.0f);          // sun term waiting for its shadow ray
    V3 pendO = mk(0.0f, 0.0f, 0.0f), pendDir = rayD;   // scattered ray waiting behind the shadow ray
    LaneTrace t;
    t.rayO = mk(0.0f, 0.0f, 0.0f); t.dirN = rayD; t.lo = t.rayO; t.ld = rayD; t.linv = winv;
    t.best.closest = 1e20f; t.best.tLocal = t.best.u = t.best.v = 0.0f; t.best.object = -1; t.best.tri = 0;
    t.obj = parentCount;                      // "trace complete": the first shade stage fetches a pixel
    t.cur = RT_ENTER; t.sp = 0;
    int cRays = 0, cNode = 0, cTri = 0;
    long long totRays = 0, totNode = 0, totTri = 0, totSamples = 0;
    unsigned long long dExec[2] = { 0, 0 }, dLanes[2] = { 0, 0 };   // wave-uniform

    for (;;) {
        // ---- vote: BVH round or object/shade iteration ----
        const int nT = __popcll(__ballot(state == K_TRAV));
        const int nS = __popcll(__ballot(state == K_OS));
        if ((nT | nS) == 0) break;

        if (nT > 0 && (nS == 0 || nT - L.biasTrav >= nS - L.biasShade)) {
            // BVH round: lanes walk the object whose box they entered; the round ends when fewer than
            // the quorum are still at it - the stragglers keep their place and carry on in the next
            // BVH round together with the lanes that arrive meanwhile
            const int quorum = max(min(L.quorumMin, nT), nT / L.quorumDiv);
            for (;;) {
                const int active = __popcll(__ballot(state == K_TRAV));
                if (DIAG) { dExec[K_TRAV] += 1; dLanes[K_TRAV] += active; }
                if (active < quorum || active == 0) break;
                if (state == K_TRAV) {
                    int next = t.cur;
                    if (t.cur == RT_ENTER) {
                        // ray into the object's space, raytrace.comp:480-486
                        const float4* op = reinterpret_cast<const float4*>(S.objects + t.obj);
                        float4 b0 = op[0];
                        float4 i0 = op[2], i1 = op[3], i2 = op[4];
                        t.lo = mk((i0.x * t.rayO.x + i0.y * t.rayO.y) + (i0.z * t.rayO.z + i0.w * 1.0f),
                                  (i1.x * t.rayO.x + i1.y * t.rayO.y) + (i1.z * t.rayO.z + i1.w * 1.0f),
                                  (i2.x * t.rayO.x + i2.y * t.rayO.y) + (i2.z * t.rayO.z + i2.w * 1.0f));
                        t.ld = normalize3(mk((i0.x * t.dirN.x + i0.y * t.dirN.y) + (i0.z * t.dirN.z + i0.w * 0.0f),
                                             (i1.x * t.dirN.x + i1.y * t.dirN.y) + (i1.z * t.dirN.z + i1.w * 0.0f),
                                             (i2.x * t.dirN.x + i2.y * t.dirN.y) + (i2.z * t.dirN.z + i2.w * 0.0f)));
                        flags = lean_inverse(t.ld, t.linv) ? (flags | F_PAR_LOCAL) : (flags & ~F_PAR_LOCAL);
                        t.sp = 0;
                        next = __float_as_int(b0.w);
                    } else if (t.cur >= 0) {
                        NodeVisit nv;
                        visit_node(S, t.cur, t.lo, t.linv, __any(flags & F_PAR_LOCAL), tMin, nv);
                        const int2 refs = nv.refs;
                        const float EF = nv.EF, ES = nv.ES;
                        const bool okF = nv.okF, okS = nv.okS;
                        if (COUNT) cNode += (refs.x != RT_REF_ABSENT ? 1 : 0) + (refs.y != RT_REF_ABSENT ? 1 : 0);
                        const bool goF = refs.x > RT_REF_LIVE_MIN && okF && (t.best.closest > EF);
                        const bool canS = refs.y > RT_REF_LIVE_MIN && okS;
                        if (goF) {
                            if (canS) { stack[t.sp * 64] = make_uint2((uint32_t)refs.y, __float_as_uint(ES)); ++t.sp; }
                            next = refs.x;
                        } else if (canS && (t.best.closest > ES)) {
                            next = refs.y;
                        } else {
                            next = lane_pop(t, stack);
                        }
                    } else {
                        int remaining;
                        if (COUNT) cTri++;
                        const bool hit = lane_triangle(S, t, S.objects[t.obj].fwd, ~t.cur, tMin, remaining);
                        if (!COUNT && hit && (flags & F_SHADOW)) { t.obj = parentCount; state = K_OS; }   // any hit settles a shadow ray
                        else if (remaining <= 1) next = lane_pop(t, stack);
                        else next = t.cur - 1;   // ~(k+1)
                    }
                    if (next == RT_NONE) {   // this object is done: back to the object loop
                        t.obj = (permute && (flags & F_SHADOW)) ? S.nextAny[t.obj] : t.obj + 1;
                        state = K_OS;
                    }
                    t.cur = next;
                }
            }
        } else {
            // object/shade iteration for the lanes outside the BVH walks: (S) shade every ray whose
            // GetHit is complete and start the next ray, then (W) sweep the parent array with a
            // wave-uniform index; lanes leave for the BVH rounds when they enter a BVH parent's box
            if (DIAG) { dExec[K_OS] += 1; dLanes[K_OS] += nS; }
            // ---- S: shade rays whose GetHit is complete, start the next ray ----
            if (state == K_OS && t.obj >= parentCount) {
                int pathEnd = 1;
                if (flags & F_HAVE_PATH) {
                    pathEnd = 0;
                    int startTrace = 0;
                    if (flags & F_SHADOW) {
                        // back from ShadowTransmission, raytrace.comp:733-753
                        if (t.best.object < 0) result = result + pendD;
                        flags &= ~F_SHADOW;
                        t.rayO = pendO; rayD = pendDir;
                        ++bounce;
                        if (bounce < bounces) startTrace = 1; else pathEnd = 1;
                    } else {
                        HitRec rec;
                        finish_hit(S, t.rayO, t.dirN, t.best, rec);
                        if (!rec.hit) {
                            result = result + throughput * environment_colour(pc, rayD);
                            pathEnd = 1;
                        } else if (!rec.frontFace) {
                            pathEnd = 1;
                        } else {
                            DevMaterial m = load_material(S, rec.matIndex);
                            V3 emitted = ld3(m.colour) * m.emission;
                            result = result + throughput * emitted;
                            if (m.emission > 0.0f) {
                                pathEnd = 1;
                            } else {
                                V3 negSun = -ld3(pc.sunDirection);
                                float nDotL = gmax(dot3(rec.normal, negSun), 0.0f);
                                V3 attenuation;
                                Ray scattered;
                                material_scatter(m, rayD, rec, seed0, attenuation, scattered);
                                if (nDotL > 0.0f) {
                                    V3 sun = ld3(pc.sunColour) * pc.sunIntensity;
                                    const float INV_PI = 0.31830988618f;
                                    V3 direct = sun * (nDotL * INV_PI);
                                    if (m.refractionIndex == 0.0f) pendD = (throughput * ld3(m.colour)) * direct;
                                    else pendD = (throughput * reflectance(nDotL, m.refractionIndex)) * direct;
                                    pendO = scattered.o; pendDir = scattered.d;
                                    t.rayO = rec.point + rec.normal * 0.01f;
                                    rayD = negSun;
                                    flags |= F_SHADOW;
                                    startTrace = 1;
                                } else {
                                    t.rayO = scattered.o; rayD = scattered.d;
                                    ++bounce;
                                    if (bounce < bounces) startTrace = 1; else pathEnd = 1;
                                }
                                throughput = throughput * attenuation;
                            }
                        }
                    }
                    (void)startTrace;
                }
                if (pathEnd) {
                    if (flags & F_HAVE_PATH) {
                        // end of one RayColour call, raytrace.comp:784
                        nc = nc + result;
                        ++sample;
                        if (sample >= rpp) {
                            // rest of Pathtrace, raytrace.comp:787-816
                            const size_t idx = (size_t)py * (size_t)L.width + (size_t)px;
                            if (!statsOnly) {
                                if (mode == 0) {
                                    float raysPerPixelRatio = 1.0f / (float)rpp;
                                    float accumulationMultiplier = pc.accumulateFrames == 1 ? 1.0f / (float)(pc.frame + 1) : 1.0f;
                                    float4 acc = accum[idx];
                                    V3 ncs = nc * raysPerPixelRatio;
                                    acc.x += ncs.x; acc.y += ncs.y; acc.z += ncs.z;
                                    accum[idx] = acc;
                                    float4 avg = pc.accumulateFrames == 1 ? acc : make_float4(ncs.x, ncs.y, ncs.z, 1.0f);
                                    avg.x *= accumulationMultiplier; avg.y *= accumulationMultiplier; avg.z *= accumulationMultiplier;
                                    out[idx] = make_float4(linear_to_gamma(avg.x), linear_to_gamma(avg.y), linear_to_gamma(avg.z), avg.w);
                                } else {
                                    int threshold = mode == 3 ? pc.triangleTestThreshold : pc.bvhNodeTestThreshold;
                                    int target = mode == 3 ? cTri : cNode;
                                    int grey = (int)((float)target / (float)threshold * 255.0f);
                                    grey = grey < 0 ? 0 : (grey > 255 ? 255 : grey);
                                    float g = (float)grey / 255.0f;
                                    float4 c = make_float4(g, g, g, 1.0f);
                                    if (cTri > threshold) c = make_float4(1.0f, 0.0f, 0.0f, 1.0f);
                                    out[idx] = c;
                                }
                            }
                            if (COUNT) { totRays += cRays; totNode += cNode; totTri += cTri; totSamples += sample; }
                            flags &= ~F_HAVE_PATH;
                        }
                    }
                    // lanes without a pixel take the next ones from the global counter
                    while (!(flags & F_HAVE_PATH) && state != K_IDLE) {
                        const unsigned long long need = __ballot(1);
                        const int n = __popcll(need);
                        const int leader = __ffsll((long long)need) - 1;
                        unsigned long long base = 0;
                        if (lane == leader) base = atomicAdd(queue, (unsigned long long)n);
                        base = (unsigned long long)__shfl((long long)base, leader, 64);
                        const unsigned long long idx = base + (unsigned long long)__popcll(need & ((1ull << lane) - 1ull));
                        if (idx >= totalPixels) {
                            state = K_IDLE;
                        } else {
                            const unsigned ownedTile = (unsigned)(idx / pixelsPerTile);
                            const unsigned local = (unsigned)(idx - (unsigned long long)ownedTile * pixelsPerTile);
                            const int tile = L.tileRank + (int)ownedTile * L.tileWorld;
                            int tx, ty;
    rt_tile_of_position(tile, L.tilesX, &tx, &ty);
                            const int sub = (int)(local >> 6), l = (int)(local & 63u);
                            const int sy = sub / wavesX, sx = sub - sy * wavesX;
                            px = tx * L.tileSize + sx * 8 + (l & 7);
                            py = ty * L.tileSize + sy * 8 + (l >> 3);
                            if (px < L.width && py < L.height) {
                                flags |= F_HAVE_PATH;
                                sample = 0;
                                nc = mk(0.0f, 0.0f, 0.0f);
                                cRays = cNode = cTri = 0;
                            }
                        }
                    }
                    if (flags & F_HAVE_PATH) {
                        // next sample of the pixel, raytrace.comp:781-783
                        seed0 = seed_from_coords((uint32_t)px, (uint32_t)py, (uint32_t)pc.frame, (uint32_t)sample);
                        Ray r0 = get_ray(pc, px, py, seed0);
                        t.rayO = r0.o; rayD = r0.d;
                        result = mk(0.0f, 0.0f, 0.0f);
                        throughput = mk(1.0f, 1.0f, 1.0f);
                        bounce = 0;
                        flags &= ~F_SHADOW;
                    }
                }
                if (state == K_OS) {
                    // start of GetHit for the ray just made, raytrace.comp:450-456
                    t.dirN = normalize3(rayD);
                    flags = lean_inverse(t.dirN, winv) ? (flags | F_PAR_WORLD) : (flags & ~F_PAR_WORLD);
                    t.best.closest = 1e20f;
                    t.best.object = -1;
                    t.obj = (permute && (flags & F_SHADOW)) ? S.anyFirst : 0;
                    if (COUNT) cRays++;
                }
            }
            // ---- W: parent array in order; a second pass over the BVH parents serves the shadow rays,
            // which take the leaf parents first ----
            const int sweep = parentCount + (permute ? parentCount - S.anyLeafCount : 0);
            for (int pos = 0; pos < sweep; ++pos) {
                const int j = pos < parentCount ? pos : S.anyOrder[pos - parentCount + S.anyLeafCount];
                const bool mine = state == K_OS && t.obj == j;
                if (!__any(mine)) continue;
                const DevObject* ob = S.objects + j;
                const int root = ob->root;
                const int nextShadow = permute ? S.nextAny[j] : j + 1;
                if (mine) {
                    const int nextObj = (flags & F_SHADOW) ? nextShadow : j + 1;
                    if (COUNT) cNode++;
                    V3 bmn = ld3(ob->bmin), bmx = ld3(ob->bmax);
                    float E;
                    bool ok = ((flags & F_PAR_WORLD) ? lean_box<true>(t.rayO, winv, bmn, bmx, tMin, E)
                                                     : lean_box<false>(t.rayO, winv, bmn, bmx, tMin, E)) && (t.best.closest > E);
                    if (ok) ok = !(distance_to_aabb(t.rayO, bmn, bmx) > t.best.closest);
                    ok = ok && root > RT_REF_LIVE_MIN;
                    if (!ok) {
                        t.obj = nextObj;
                    } else if (root >= 0) {
                        state = K_TRAV; t.cur = RT_ENTER;   // BVH parent: walked in the BVH rounds
                    } else {
                        // leaf parent, raytrace.comp:488-520
                        t.lo = xform(ob->inv, t.rayO, 1.0f);
                        t.ld = normalize3(xform(ob->inv, t.dirN, 0.0f));
                        int k = ~root, remaining;
                        for (;;) {
                            if (COUNT) cTri++;
                            const bool hit = lane_triangle(S, t, ob->fwd, k, tMin, remaining);
                            if (!COUNT && hit && (flags & F_SHADOW)) { remaining = -1; break; }
                            if (remaining <= 1) break;
                            ++k;
                        }
                        t.obj = remaining < 0 ? parentCount : nextObj;   // a shadow ray that hit is finished
                    }
                }
            }
        }
    }

    if (DIAG && diag != nullptr && lane == 0) {
        for (int k = 0; k < 2; ++k) {
            atomicAdd(&diag[2 * k], dExec[k]);
            atomicAdd(&diag[2 * k + 1], dLanes[k]);
        }
        atomicAdd(&diag[8], 1ull);
    }
    if (COUNT && counters != nullptr) {
        for (int off = 32; off > 0; off >>= 1) {
            totRays += __shfl_down(totRays, off, 64);
            totNode += __shfl_down(totNode, off, 64);
            totTri += __shfl_down(totTri, off, 64);
            totSamples += __shfl_down(totSamples, off, 64);
        }
        if (lane == 0) {
            atomicAdd(&counters->rays, (unsigned long long)totRays);
            atomicAdd(&counters->nodeTests, (unsigned long long)totNode);
            atomicAdd(&counters->triTests, (unsigned long long)totTri);
            atomicAdd(&counters->samples, (unsigned long long)totSamples);
        }
    }
}

// ---------------------------------------------------------------------------------
// Wavefront pipeline: the path loop and the BVH walks run in different kernels.
//
// In the lock-step kernel two thirds of the issue slots go to BVH walks whose lengths differ
// wildly between the 64 rays of a wave. Here a ray that enters the box of a BVH parent leaves the
// path kernel as a job (DevWavefront), and so does every ray that has just been started:
//   rt_wf_path   round 0: one lane per pixel - camera ray and the first part of GetHit's object
//                loop (parent boxes, leaf parents such as the six walls) up to the first BVH
//                parent the ray enters;
//                round r > 0: takes the jobs of round r-1, walked, and runs, for 64 of them in
//                step, the rest of the object loop, shading, the shadow ray and the start of the
//                next ray;
//                last round: BVH parents are walked in place, so nothing is left over.
//   rt_wf_walk   walks the BVH parent a job has entered and the BVH parents that follow it in the
//                object loop. Every lane runs its own small state machine (one parent box / one
//                two-child record / one triangle per step) and takes the next job from the list
//                when it is done, so the wave stays full whatever the lengths of the walks. The
//                answer is written back into the job's own record, so the next round of the path
//                kernel reads jobs and answers in list order, with consecutive addresses.
// Kernel boundaries on one stream order the hand-over of jobs and results; within a kernel only
// atomic counters are shared. Per ray the arithmetic and its order are those of the lock-step
// kernel, so the images are the same bit for bit.

// Walk of one BVH parent for one ray (raytrace.comp:523-606) with the lock-step kernel's loop.
// Returns true when `early` is set and a hit was accepted.
template <bool COUNT>
__device__ __forceinline__ bool walk_in_place(const DevScene& S, const DevObject* ob, int objIndex, bool early, V3 org, V3 dirN,
                                              float tMin, Best& best, Counters& cnt, uint2* stack)
{
    V3 lo = xform(ob->inv, org, 1.0f);
    V3 ld = normalize3(xform(ob->inv, dirN, 0.0f));
    V3 linv;
    const bool parL = lean_inverse(ld, linv);
    int cur = ob->root;
    int sp = 0;
    for (;;) {
        if (cur >= 0) {
            NodeVisit nv;
            visit_node(S, cur, lo, linv, __any(parL), tMin, nv);
            const int2 refs = nv.refs;
            const float EF = nv.EF, ES = nv.ES;
            const bool okF = nv.okF, okS = nv.okS;
            if (COUNT) cnt.nodeTests += (refs.x != RT_REF_ABSENT ? 1 : 0) + (refs.y != RT_REF_ABSENT ? 1 : 0);
            const bool goF = refs.x > RT_REF_LIVE_MIN && okF && (best.closest > EF);
            const bool canS = refs.y > RT_REF_LIVE_MIN && okS;
            if (goF) {
                if (canS) { stack[sp * 64] = make_uint2((uint32_t)refs.y, __float_as_uint(ES)); ++sp; }
                cur = refs.x;
                continue;
            }
            if (canS && (best.closest > ES)) { cur = refs.y; continue; }
        } else {
            int k = ~cur;
            for (;;) {
                const float4* tp = reinterpret_cast<const float4*>(S.tris + k);
                float4 t0 = tp[0], t1 = tp[1], t2 = tp[2];
                const int remaining = __float_as_int(t0.w);
                if (COUNT) cnt.triTests++;
                if (triangle_hit(t0, t1, t2, lo, ld, ob->fwd, org, dirN, tMin, objIndex, best)) {
                    if (early) return true;
                }
                if (remaining <= 1) break;
                ++k;
            }
        }
        cur = RT_NONE;
        while (sp > 0) {
            --sp;
            uint2 e = stack[sp * 64];
            if (best.closest > __uint_as_float(e.y)) { cur = (int)e.x; break; }
        }
        if (cur == RT_NONE) return false;
    }
}

// GetHit's object loop (raytrace.comp:458-520) up to the first BVH parent a lane enters: that lane
// stops there (returned object index, -1 otherwise) with `best` as it is at that point. fresh =
// false continues a GetHit call at object `next` with the `best` passed in (a ray back from a
// walk). `anyhit` is wave-uniform. Leaf parents are tested in place.
// INLINE: nobody stops; BVH parents are walked in place.
template <bool COUNT, bool INLINE>
__device__ __forceinline__ int trace_until_walk(const DevScene& S, int parentCount, bool anyhit, bool fresh, int next, V3 org,
                                                V3 dirN, float tMin, Best& best, Counters& cnt, uint2* stack)
{
    const bool early = anyhit && !COUNT;
    const bool permute = early && parentCount == S.objectCount;
    if (fresh) {
        best.closest = 1e20f;
        best.object = -1;
        best.tri = 0; best.u = 0.0f; best.v = 0.0f; best.tLocal = 0.0f;
        if (COUNT) cnt.rays++;
        next = permute ? S.anyFirst : 0;
    }
    V3 winv;
    const bool parW = lean_inverse(dirN, winv);
    int suspended = -1;
    for (int pos = 0; pos < parentCount; ++pos) {
        const int i = permute ? S.anyOrder[pos] : pos;
        const DevObject* ob = S.objects + i;   // wave-uniform address
        const int root = ob->root;
        const int after = permute ? S.nextAny[i] : i + 1;
        if (next != i) continue;
        next = after;
        if (COUNT) cnt.nodeTests++;
        V3 bmn = ld3(ob->bmin), bmx = ld3(ob->bmax);
        float E;
        bool ok = (__any(parW) ? lean_box<true>(org, winv, bmn, bmx, tMin, E) : lean_box<false>(org, winv, bmn, bmx, tMin, E)) &&
                  (best.closest > E);
        if (ok) ok = !(distance_to_aabb(org, bmn, bmx) > best.closest);
        if (!ok || !(root > RT_REF_LIVE_MIN)) continue;
        if (root >= 0) {
            if (INLINE) { if (walk_in_place<COUNT>(S, ob, i, early, org, dirN, tMin, best, cnt, stack)) next = parentCount; }
            else { suspended = i; next = parentCount; }
            continue;
        }
        V3 lo = xform(ob->inv, org, 1.0f);
        V3 ld = normalize3(xform(ob->inv, dirN, 0.0f));
        int k = ~root;
        for (;;) {
            const float4* tp = reinterpret_cast<const float4*>(S.tris + k);
            float4 t0 = tp[0], t1 = tp[1], t2 = tp[2];
            const int remaining = __float_as_int(t0.w);
            if (COUNT) cnt.triTests++;
            if (triangle_hit(t0, t1, t2, lo, ld, ob->fwd, org, dirN, tMin, i, best)) {
                if (early) { next = parentCount; break; }
            }
            if (remaining <= 1) break;
            ++k;
        }
    }
    return suspended;
}

// List space is handed out to a wave in blocks of RT_WF_BLOCK entries (one atomic per block; one
// per entry or per wave instruction would make the counter the bottleneck). A wave fills its
// block in order; what it leaves unused is marked as a hole, which the readers skip.
#define RT_WF_BLOCK 256
struct WaveBlock { unsigned base, left; };   // wave-uniform

// Index for each lane of `mask` (other lanes get an unspecified value). Whole wave.
// `holes` is the array whose record marks an entry unused, `hole` that record.
__device__ __forceinline__ unsigned wf_append(WaveBlock& blk, uint32_t* counter, float4* holes, float4 hole, unsigned long long mask)
{
    const int lane = threadIdx.x & 63;
    const unsigned n = (unsigned)__popcll(mask);
    if (n > blk.left) {
        for (unsigned i = (unsigned)lane; i < blk.left; i += 64u) holes[blk.base + i] = hole;
        unsigned base = 0;
        if (lane == 0) base = atomicAdd(counter, (unsigned)RT_WF_BLOCK);
        blk.base = (unsigned)__shfl((int)base, 0, 64);
        blk.left = RT_WF_BLOCK;
    }
    const unsigned idx = blk.base + (unsigned)__popcll(mask & ((1ull << lane) - 1ull));
    blk.base += n;
    blk.left -= n;
    return idx;
}
__device__ __forceinline__ void wf_close(WaveBlock& blk, float4* holes, float4 hole)
{
    const int lane = threadIdx.x & 63;
    for (unsigned i = (unsigned)lane; i < blk.left; i += 64u) holes[blk.base + i] = hole;
    blk.left = 0;
}

// Job records stream through the chip once per round; they are written and read with the
// non-temporal hint so that they do not push the scene out of the L2.
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 wf_load(const float4* p)
{
    const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void wf_store(float4* p, float4 v)
{
    const f32x4 x = { v.x, v.y, v.z, v.w };
    __builtin_nontemporal_store(x, reinterpret_cast<f32x4*>(p));
}

__device__ __forceinline__ float4* wf_job(const DevWavefront& W, int parity, int part)
{
    return reinterpret_cast<float4*>(W.jobs[parity]) + (size_t)part * (size_t)W.listCap;
}
__device__ __forceinline__ float4 wf_job_hole() { return make_float4(0.0f, 0.0f, 0.0f, __int_as_float(-1)); }

struct PathVars {
    V3 rayO, rayD;
    V3 result, throughput;
    int bounce, sample;
    V3 nc;
    V3 pendD, pendO, pendDir;
};

// The lanes with `emit` leave the path kernel: each becomes a job of this round. obj is the BVH
// parent whose box the ray entered, or parentCount when its GetHit call is already complete.
// Must be called by the whole wave.
__device__ __forceinline__ void wf_emit(const DevWavefront& W, int round, WaveBlock& blk, bool emit, int slot, bool shadow, int obj,
                                        V3 org, V3 dir, const Best& best, const PathVars& v, const Counters& cnt, bool wide)
{
    const unsigned long long m = __ballot(emit);
    if (m == 0) return;
    const int par = round & 1;
    const unsigned idx = wf_append(blk, &W.counts[round * RT_WF_WORDS_PER_ROUND + 0], wf_job(W, par, 0), wf_job_hole(), m);
    if (emit) {
        const uint32_t meta = (uint32_t)obj | (shadow ? 0x80000000u : 0u);
        wf_store(wf_job(W, par, 0) + idx, make_float4(org.x, org.y, org.z, __int_as_float(slot)));
        wf_store(wf_job(W, par, 1) + idx, make_float4(dir.x, dir.y, dir.z, __uint_as_float(meta)));
        wf_store(wf_job(W, par, 2) + idx, make_float4(best.closest, best.tLocal, best.u, best.v));
        wf_store(wf_job(W, par, 3) + idx, make_float4(__int_as_float(best.object + 1), __int_as_float(best.tri),
                                             __int_as_float(cnt.nodeTests), __int_as_float(cnt.triTests)));
        wf_store(wf_job(W, par, 4) + idx, make_float4(v.throughput.x, v.throughput.y, v.throughput.z, __int_as_float(v.sample)));
        wf_store(wf_job(W, par, 5) + idx, make_float4(v.result.x, v.result.y, v.result.z, __int_as_float(v.bounce)));
        if (shadow) {
            wf_store(wf_job(W, par, 6) + idx, make_float4(v.pendD.x, v.pendD.y, v.pendD.z, 0.0f));
            wf_store(wf_job(W, par, 7) + idx, make_float4(v.pendO.x, v.pendO.y, v.pendO.z, 0.0f));
            wf_store(wf_job(W, par, 8) + idx, make_float4(v.pendDir.x, v.pendDir.y, v.pendDir.z, 0.0f));
        }
        if (wide) wf_store(wf_job(W, par, 9) + idx, make_float4(v.nc.x, v.nc.y, v.nc.z, __int_as_float(cnt.rays)));
    }
}

// Persistent grid. Round 0 takes 8x8 pixel tiles, round > 0 chunks of 64 results of the previous
// round's walks, from a counter. One pass over a chunk is: the rest of the GetHit call the ray
// came back from, shading, the shadow ray, the first part of the next ray's GetHit call - then
// every lane that still has a path leaves as a job of this round, so a pass is never continued
// for a few lanes only. INLINE (the last round): BVH parents are walked in place and the paths
// run to their end (needs the traversal stack in LDS).
template <bool COUNT, bool INLINE>
__global__ __launch_bounds__(64) void RT_SYM(rt_wf_path)(DevScene S, rt_push_constants pc, DevLaunch L, DevWavefront W,
                                                        float4* __restrict__ accum, float4* __restrict__ out,
                                                        DevCounters* counters, int statsOnly, int round)
{
    extern __shared__ uint2 stackMem[];
    const int lane = threadIdx.x;
    uint2* stack = stackMem + lane;
    const int bounces = pc.maxBounces + 1;
    const int parentCount = pc.parentBVHCount;
    const int rpp = pc.raysPerPixel;
    const int mode = pc.renderMode;
    const float tMin = 0.001f;
    const bool wide = COUNT || rpp > 1;
    const int prev = (round + 1) & 1;   // parity of the previous round's job buffer
    const bool leadingBvh = parentCount > 0 && S.objects[0].root >= 0;
    long long totRays = 0, totNode = 0, totTri = 0, totSamples = 0;

    WaveBlock blk;
    blk.base = 0; blk.left = 0;
    unsigned grabNext = 0, grabLeft = 0;
    unsigned jobsIn = 0, chunks = 0;
    uint32_t* workCounter;
    if (round > 0) {
        jobsIn = W.counts[(round - 1) * RT_WF_WORDS_PER_ROUND + 0];
        chunks = (jobsIn + 63u) >> 6;
        workCounter = &W.counts[(round - 1) * RT_WF_WORDS_PER_ROUND + 4];
    } else {
        const int wavesX = L.tileSize >> 3;
        chunks = (unsigned)(L.ownedTiles * wavesX * wavesX);
        workCounter = &W.counts[(RT_WF_MAX_ROUNDS + 1) * RT_WF_WORDS_PER_ROUND];
    }

    for (;;) {
        // ---- lanes of this pass: a pixel tile (round 0) or 64 jobs of the previous round, walked ----
        // (chunks are taken four at a time: atomics on one address are served one by one)
        if (grabLeft == 0) {
            unsigned g = 0;
            if (lane == 0) g = atomicAdd(workCounter, 4u);
            grabNext = (unsigned)__shfl((int)g, 0, 64);
            grabLeft = 4;
        }
        const unsigned chunk = grabNext;
        ++grabNext; --grabLeft;
        if (chunk >= chunks) break;

        int kind = 0;           // per lane: 0 new path, 1 back from a regular ray's walk, 2 back from a shadow ray's walk
        bool alive = false;
        int slot = 0, px = 0, py = 0;
        int resumeObj = 0;                       // object at which a resumed GetHit call continues
        V3 resumeO = mk(0.0f, 0.0f, 0.0f);       // origin of a resumed shadow ray
        uint32_t seed = 0;
        PathVars v;
        Counters cnt;
        Best best;
        cnt.rays = cnt.nodeTests = cnt.triTests = 0;
        v.nc = mk(0.0f, 0.0f, 0.0f);
        v.result = mk(0.0f, 0.0f, 0.0f); v.throughput = mk(1.0f, 1.0f, 1.0f);
        v.pendD = v.pendO = v.pendDir = mk(0.0f, 0.0f, 0.0f);
        v.rayO = mk(0.0f, 0.0f, 0.0f); v.rayD = mk(0.0f, 0.0f, 1.0f);
        v.bounce = 0; v.sample = 0;
        best.closest = 1e20f; best.tLocal = 0.0f; best.u = 0.0f; best.v = 0.0f; best.object = -1; best.tri = 0;
        if (round == 0) {
            const int wavesX = L.tileSize >> 3;
            const int wavesPerTile = wavesX * wavesX;
            const int ownedTile = (int)chunk / wavesPerTile;
            const int sub = (int)chunk - ownedTile * wavesPerTile;
            const int tile = L.tileRank + ownedTile * L.tileWorld;
            int tx, ty;
    rt_tile_of_position(tile, L.tilesX, &tx, &ty);
            const int sy = sub / wavesX, sx = sub - sy * wavesX;
            px = tx * L.tileSize + sx * 8 + (lane & 7);
            py = ty * L.tileSize + sy * 8 + (lane >> 3);
            alive = px < L.width && py < L.height && rpp >= 1;
            slot = py * L.width + px;
            if (alive) {
                seed = seed_from_coords((uint32_t)px, (uint32_t)py, (uint32_t)pc.frame, 0u);
                Ray r0 = get_ray(pc, px, py, seed);
                v.rayO = r0.o; v.rayD = r0.d;
            }
        } else {
            const unsigned job = chunk * 64u + (unsigned)lane;
            float4 a = make_float4(0.0f, 0.0f, 0.0f, __int_as_float(-1));
            if (job < jobsIn) a = wf_load(wf_job(W, prev, 0) + job);
            slot = __float_as_int(a.w);
            alive = slot >= 0;   // -1: unused entry of a writer's block
            if (alive) {
                float4 b = wf_load(wf_job(W, prev, 1) + job), r0 = wf_load(wf_job(W, prev, 2) + job), r1 = wf_load(wf_job(W, prev, 3) + job);
                float4 e = wf_load(wf_job(W, prev, 4) + job), f = wf_load(wf_job(W, prev, 5) + job);
                kind = (__float_as_uint(b.w) & 0x80000000u) ? 2 : 1;
                py = slot / L.width; px = slot - py * L.width;
                best.closest = r0.x; best.tLocal = r0.y; best.u = r0.z; best.v = r0.w;
                best.object = (__float_as_int(r1.x) & 0xffff) - 1; best.tri = __float_as_int(r1.y);
                resumeObj = (int)((unsigned)__float_as_int(r1.x) >> 16);
                if (COUNT) { cnt.nodeTests = __float_as_int(r1.z); cnt.triTests = __float_as_int(r1.w); }
                v.throughput = mk(e.x, e.y, e.z); v.sample = __float_as_int(e.w);
                v.result = mk(f.x, f.y, f.z); v.bounce = __float_as_int(f.w);
                seed = seed_from_coords((uint32_t)px, (uint32_t)py, (uint32_t)pc.frame, (uint32_t)v.sample);
                if (kind == 2) {
                    float4 g = wf_load(wf_job(W, prev, 6) + job), h = wf_load(wf_job(W, prev, 7) + job), k = wf_load(wf_job(W, prev, 8) + job);
                    v.pendD = mk(g.x, g.y, g.z); v.pendO = mk(h.x, h.y, h.z); v.pendDir = mk(k.x, k.y, k.z);
                    resumeO = mk(a.x, a.y, a.z);
                } else {
                    v.rayO = mk(a.x, a.y, a.z); v.rayD = mk(b.x, b.y, b.z);
                }
                if (wide) {
                    float4 j = wf_load(wf_job(W, prev, 9) + job);
                    v.nc = mk(j.x, j.y, j.z); cnt.rays = __float_as_int(j.w);
                }
            }
        }

        // ---- the path loop: RayColour (raytrace.comp:692-769) inside Pathtrace's sample loop ----
        bool firstPass = true;
        while (__any(alive)) {
            bool pathEnd = false, needShadow = false;
            V3 so = v.rayO;
            const V3 negSun = -ld3(pc.sunDirection);
            // in the first pass of a resume round every lane continues a call it came back from:
            // regular rays here, shadow rays further down
            const bool fresh = !(firstPass && round > 0);
            const bool backShd = !fresh && kind == 2;
            const bool inReg = alive && !backShd;
            const V3 dirN = normalize3(v.rayD);
            if (!INLINE && fresh && leadingBvh) {
                // rays that have just started, in a scene whose first parent is a BVH parent: the
                // walk kernel starts their GetHit call, box test of that parent included
                best.closest = 1e20f; best.object = -1; best.tri = 0; best.u = 0.0f; best.v = 0.0f; best.tLocal = 0.0f;
                if (COUNT && inReg) cnt.rays++;
                wf_emit(W, round, blk, inReg, slot, false, 0x40000000, v.rayO, v.rayD, best, v, cnt, wide);
                if (inReg) alive = false;
            } else {
                int obj = -1;
                if (inReg)
                    obj = trace_until_walk<COUNT, INLINE>(S, parentCount, false, fresh, resumeObj, v.rayO, dirN, tMin, best, cnt, stack);
                if (!INLINE) {
                    // a ray that has just started leaves in any case; one that came back from a
                    // walk leaves only if it enters another BVH parent
                    const bool emit = inReg && (fresh || obj >= 0);
                    wf_emit(W, round, blk, emit, slot, false, obj >= 0 ? obj : parentCount, v.rayO, v.rayD, best, v, cnt, wide);
                    if (emit) alive = false;
                }
            }
            if (backShd) needShadow = alive;   // back from the walk of a shadow ray
            {
                if (inReg && alive) {
                    HitRec rec;
                    finish_hit(S, v.rayO, dirN, best, rec);
                    if (!rec.hit) {
                        v.result = v.result + v.throughput * environment_colour(pc, v.rayD);
                        pathEnd = true;
                    } else if (!rec.frontFace) {
                        pathEnd = true;
                    } else {
                        DevMaterial m = load_material(S, rec.matIndex);
                        V3 emitted = ld3(m.colour) * m.emission;
                        v.result = v.result + v.throughput * emitted;
                        if (m.emission > 0.0f) {
                            pathEnd = true;
                        } else {
                            float nDotL = gmax(dot3(rec.normal, negSun), 0.0f);
                            V3 attenuation;
                            Ray scattered;
                            material_scatter(m, v.rayD, rec, seed, attenuation, scattered);
                            if (nDotL > 0.0f) {
                                V3 sun = ld3(pc.sunColour) * pc.sunIntensity;
                                const float INV_PI = 0.31830988618f;
                                V3 direct = sun * (nDotL * INV_PI);
                                if (m.refractionIndex == 0.0f) v.pendD = (v.throughput * ld3(m.colour)) * direct;
                                else v.pendD = (v.throughput * reflectance(nDotL, m.refractionIndex)) * direct;
                                so = rec.point + rec.normal * 0.01f;
                                needShadow = true;
                            }
                            v.pendO = scattered.o; v.pendDir = scattered.d;
                            v.throughput = v.throughput * attenuation;
                        }
                    }
                }
            }
            // ShadowTransmission, raytrace.comp:612-621,733-753
            if (__any(needShadow)) {
                if (backShd) so = resumeO;
                int obj = -1;
                if (needShadow)
                    obj = trace_until_walk<COUNT, INLINE>(S, parentCount, true, !backShd, resumeObj, so, normalize3(negSun), tMin, best, cnt, stack);
                if (!INLINE) {
                    const bool emit = needShadow && obj >= 0;
                    wf_emit(W, round, blk, emit, slot, true, obj, so, negSun, best, v, cnt, wide);
                    if (emit) { alive = false; needShadow = false; }
                }
                if (needShadow && best.object < 0) v.result = v.result + v.pendD;
            }
            if (alive && !pathEnd) {
                // the scattered ray is the next ray of the path
                v.rayO = v.pendO; v.rayD = v.pendDir;
                ++v.bounce;
                if (v.bounce >= bounces) pathEnd = true;
            }
            if (alive && pathEnd) {
                // end of one RayColour call, raytrace.comp:784
                v.nc = v.nc + v.result;
                ++v.sample;
                if (v.sample < rpp) {
                    seed = seed_from_coords((uint32_t)px, (uint32_t)py, (uint32_t)pc.frame, (uint32_t)v.sample);
                    Ray r0 = get_ray(pc, px, py, seed);
                    v.rayO = r0.o; v.rayD = r0.d;
                    v.result = mk(0.0f, 0.0f, 0.0f);
                    v.throughput = mk(1.0f, 1.0f, 1.0f);
                    v.bounce = 0;
                } else {
                    // rest of Pathtrace, raytrace.comp:787-816
                    const size_t idx = (size_t)py * (size_t)L.width + (size_t)px;
                    if (!statsOnly) {
                        if (mode == 0) {
                            float raysPerPixelRatio = 1.0f / (float)rpp;
                            float accumulationMultiplier = pc.accumulateFrames == 1 ? 1.0f / (float)(pc.frame + 1) : 1.0f;
                            float4 acc = accum[idx];
                            V3 ncs = v.nc * raysPerPixelRatio;
                            acc.x += ncs.x; acc.y += ncs.y; acc.z += ncs.z;
                            accum[idx] = acc;
                            float4 avg = pc.accumulateFrames == 1 ? acc : make_float4(ncs.x, ncs.y, ncs.z, 1.0f);
                            avg.x *= accumulationMultiplier; avg.y *= accumulationMultiplier; avg.z *= accumulationMultiplier;
                            out[idx] = make_float4(linear_to_gamma(avg.x), linear_to_gamma(avg.y), linear_to_gamma(avg.z), avg.w);
                        } else {
                            int threshold = mode == 3 ? pc.triangleTestThreshold : pc.bvhNodeTestThreshold;
                            int target = mode == 3 ? cnt.triTests : cnt.nodeTests;
                            int grey = (int)((float)target / (float)threshold * 255.0f);
                            grey = grey < 0 ? 0 : (grey > 255 ? 255 : grey);
                            float g = (float)grey / 255.0f;
                            float4 c = make_float4(g, g, g, 1.0f);
                            if (cnt.triTests > threshold) c = make_float4(1.0f, 0.0f, 0.0f, 1.0f);
                            out[idx] = c;
                        }
                    }
                    if (COUNT) { totRays += cnt.rays; totNode += cnt.nodeTests; totTri += cnt.triTests; totSamples += v.sample; }
                    alive = false;
                }
            }
            firstPass = false;
        }
    }
    if (!INLINE) wf_close(blk, wf_job(W, round & 1, 0), wf_job_hole());

    if (COUNT && counters != nullptr) {
        for (int off = 32; off > 0; off >>= 1) {
            totRays += __shfl_down(totRays, off, 64);
            totNode += __shfl_down(totNode, off, 64);
            totTri += __shfl_down(totTri, off, 64);
            totSamples += __shfl_down(totSamples, off, 64);
        }
        if (lane == 0 && (totRays | totNode | totTri | totSamples) != 0) {
            atomicAdd(&counters->rays, (unsigned long long)totRays);
            atomicAdd(&counters->nodeTests, (unsigned long long)totNode);
            atomicAdd(&counters->triTests, (unsigned long long)totTri);
            atomicAdd(&counters->samples, (unsigned long long)totSamples);
        }
    }
}

// The walk kernel keeps only the first RT_WF_LDS_LEVELS levels of a lane's traversal stack in LDS
// (deeper entries are rare) and the rest in a per-wave region of HBM, so that twice as many waves
// fit on a compute unit and hide each other's record fetches.
#define RT_WF_LDS_LEVELS 12
__device__ __forceinline__ void wf_push(LaneTrace& t, uint2* lds, uint2* deep, uint2 e)
{
    if (t.sp < RT_WF_LDS_LEVELS) lds[t.sp * 64] = e; else deep[(t.sp - RT_WF_LDS_LEVELS) * 64] = e;
    ++t.sp;
}
// Next stack entry whose box still reaches inside the closest hit, or RT_NONE.
__device__ __forceinline__ int wf_pop(LaneTrace& t, const uint2* lds, const uint2* deep)
{
    while (t.sp > 0) {
        --t.sp;
        const uint2 e = t.sp < RT_WF_LDS_LEVELS ? lds[t.sp * 64] : deep[(t.sp - RT_WF_LDS_LEVELS) * 64];
        if (t.best.closest > __uint_as_float(e.y)) return (int)e.x;
    }
    return RT_NONE;
}

#define RT_WF_BOX 0x7ffffffd   /* LaneTrace::cur: at the parent box of object `obj` */

// Persistent grid. Lane states: 0 free, 1 walking, 2 finished (result still in registers), 3 retired.
template <bool COUNT>
__global__ __launch_bounds__(64) void RT_SYM(rt_wf_walk)(DevScene S, rt_push_constants pc, DevWavefront W, int round)
{
    extern __shared__ uint2 stackMem[];
    const int lane = threadIdx.x;
    uint2* stack = stackMem + lane;
    uint2* deep = reinterpret_cast<uint2*>(W.deepStack) + (size_t)blockIdx.x * (RT_MAX_STACK - RT_WF_LDS_LEVELS) * 64 + lane;
    const int parentCount = pc.parentBVHCount;
    const float tMin = 0.001f;
    const bool permute = !COUNT && parentCount == S.objectCount;
    const int par = round & 1;
    uint32_t* ctl = W.counts + round * RT_WF_WORDS_PER_ROUND;
    const unsigned jobs = ctl[0];
    const int refillMin = W.refillMin;

    int st = 0, job = 0, flags = 0;
    int cNode = 0, cTri = 0;
    V3 winv = mk(1.0f, 1.0f, 1.0f);
    LaneTrace t;
    t.rayO = mk(0.0f, 0.0f, 0.0f); t.dirN = mk(0.0f, 0.0f, 1.0f); t.lo = t.rayO; t.ld = t.dirN; t.linv = winv;
    t.best.closest = 1e20f; t.best.tLocal = t.best.u = t.best.v = 0.0f; t.best.object = -1; t.best.tri = 0;
    t.obj = 0; t.cur = RT_WF_BOX; t.sp = 0;
    bool exhausted = false;   // wave-uniform: the list has no more entries
    unsigned inNext = 0, inLeft = 0;   // this wave's range of the job list

    for (;;) {
        const int walking = __popcll(__ballot(st == 1));
        if (walking == 0 || (!exhausted && walking <= 64 - refillMin)) {
            // hand finished rays on: the answer goes back into parts 2 and 3 of the job's own record
            if (st == 2) {
                wf_store(wf_job(W, par, 2) + job, make_float4(t.best.closest, t.best.tLocal, t.best.u, t.best.v));
                wf_store(wf_job(W, par, 3) + job, make_float4(__int_as_float((t.best.object + 1) | (t.obj << 16)), __int_as_float(t.best.tri),
                                                     __int_as_float(cNode), __int_as_float(cTri)));
                st = 0;
            }
            // take new jobs from the list, of which this wave reserves 256 entries at a time
            if (!exhausted) {
                const unsigned long long need = __ballot(st == 0);
                const unsigned n = (unsigned)__popcll(need);
                const unsigned rank = (unsigned)__popcll(need & ((1ull << lane) - 1ull));
                unsigned fresh = 0;
                if (n > inLeft) {
                    if (lane == 0) fresh = atomicAdd(&ctl[1], 256u);
                    fresh = (unsigned)__shfl((int)fresh, 0, 64);
                }
                const unsigned idx = rank < inLeft ? inNext + rank : fresh + (rank - inLeft);
                if (n > inLeft) { inNext = fresh + (n - inLeft); inLeft = 256u - (n - inLeft); }
                else { inNext += n; inLeft -= n; }
                if (st == 0 && idx >= jobs) st = 3;
                exhausted = __any(st == 3);
                if (st == 0) {
                    float4 a = wf_load(wf_job(W, par, 0) + idx), b = wf_load(wf_job(W, par, 1) + idx);
                    if (__float_as_int(a.w) >= 0) {   // else an unused entry of a writer's block
                        float4 c = wf_load(wf_job(W, par, 2) + idx), d = wf_load(wf_job(W, par, 3) + idx);
                        const uint32_t meta = __float_as_uint(b.w);
                        job = (int)idx;
                        t.rayO = mk(a.x, a.y, a.z);
                        t.dirN = normalize3(mk(b.x, b.y, b.z));
                        flags = (meta & 0x80000000u) ? F_SHADOW : 0;
                        if (lean_inverse(t.dirN, winv)) flags |= F_PAR_WORLD;
                        t.best.closest = c.x; t.best.tLocal = c.y; t.best.u = c.z; t.best.v = c.w;
                        t.best.object = __float_as_int(d.x) - 1; t.best.tri = __float_as_int(d.y);
                        cNode = __float_as_int(d.z); cTri = __float_as_int(d.w);
                        t.obj = (int)(meta & 0x3fffffffu);
                        st = 2;   // a GetHit call that was already complete passes straight through
                        if (meta & 0x40000000u) {
                            // a ray that has just started: begin with the box of parent 0
                            t.obj = 0; t.cur = RT_WF_BOX; st = 1;
                        } else if (t.obj < parentCount) {
                            // the box of this object was entered by the path kernel: straight to its root
                            const float4* op = reinterpret_cast<const float4*>(S.objects + t.obj);
                            float4 b0 = op[0], i0 = op[2], i1 = op[3], i2 = op[4];
                            t.lo = mk((i0.x * t.rayO.x + i0.y * t.rayO.y) + (i0.z * t.rayO.z + i0.w * 1.0f),
                                      (i1.x * t.rayO.x + i1.y * t.rayO.y) + (i1.z * t.rayO.z + i1.w * 1.0f),
                                      (i2.x * t.rayO.x + i2.y * t.rayO.y) + (i2.z * t.rayO.z + i2.w * 1.0f));
                            t.ld = normalize3(mk((i0.x * t.dirN.x + i0.y * t.dirN.y) + (i0.z * t.dirN.z + i0.w * 0.0f),
                                                 (i1.x * t.dirN.x + i1.y * t.dirN.y) + (i1.z * t.dirN.z + i1.w * 0.0f),
                                                 (i2.x * t.dirN.x + i2.y * t.dirN.y) + (i2.z * t.dirN.z + i2.w * 0.0f)));
                            flags = lean_inverse(t.ld, t.linv) ? (flags | F_PAR_LOCAL) : (flags & ~F_PAR_LOCAL);
                            t.sp = 0;
                            t.cur = __float_as_int(b0.w);
                            st = 1;
                        }
                    }
                }
            }
            if (!__any(st == 1)) {
                if (exhausted && !__any(st == 2)) break;
                continue;   // finished lanes are handed on, free lanes try the list again
            }
        }

        if (st == 1) {
            int next = t.cur;
            // Whatever the lane is at - a parent (DevObject), a two-child record (DevNode) or a
            // triangle (DevTri) - the first 64 bytes of that record are fetched here, before the
            // three kinds of step part ways, so an iteration waits for memory once, not three times.
            const bool atBox = t.cur == RT_WF_BOX;
            const bool atEnd = atBox && t.obj >= parentCount;
            const float4* rec = atBox ? reinterpret_cast<const float4*>(S.objects + (atEnd ? 0 : t.obj))
                                      : (t.cur >= 0 ? reinterpret_cast<const float4*>(S.nodes + t.cur)
                                                    : reinterpret_cast<const float4*>(S.tris + ~t.cur));
            const float4 r0 = rec[0], r1 = rec[1], r2 = rec[2], r3 = rec[3];
            if (atBox) {
                // next parent, raytrace.comp:458-486: a BVH parent is walked here, at anything else
                // the ray goes back to the path kernel, which continues the object loop there
                const int root = atEnd ? -1 : __float_as_int(r0.w);
                if (root < 0) {
                    st = 2;
                } else {
                    if (COUNT) cNode++;
                    V3 bmn = mk(r0.x, r0.y, r0.z), bmx = mk(r1.x, r1.y, r1.z);
                    float E;
                    bool ok = (__any(flags & F_PAR_WORLD) ? lean_box<true>(t.rayO, winv, bmn, bmx, tMin, E)
                                                          : lean_box<false>(t.rayO, winv, bmn, bmx, tMin, E)) && (t.best.closest > E);
                    if (ok) ok = !(distance_to_aabb(t.rayO, bmn, bmx) > t.best.closest);
                    if (ok) {
                        const float4 i0 = r2, i1 = r3, i2 = rec[4];
                        t.lo = mk((i0.x * t.rayO.x + i0.y * t.rayO.y) + (i0.z * t.rayO.z + i0.w * 1.0f),
                                  (i1.x * t.rayO.x + i1.y * t.rayO.y) + (i1.z * t.rayO.z + i1.w * 1.0f),
                                  (i2.x * t.rayO.x + i2.y * t.rayO.y) + (i2.z * t.rayO.z + i2.w * 1.0f));
                        t.ld = normalize3(mk((i0.x * t.dirN.x + i0.y * t.dirN.y) + (i0.z * t.dirN.z + i0.w * 0.0f),
                                             (i1.x * t.dirN.x + i1.y * t.dirN.y) + (i1.z * t.dirN.z + i1.w * 0.0f),
                                             (i2.x * t.dirN.x + i2.y * t.dirN.y) + (i2.z * t.dirN.z + i2.w * 0.0f)));
                        flags = lean_inverse(t.ld, t.linv) ? (flags | F_PAR_LOCAL) : (flags & ~F_PAR_LOCAL);
                        t.sp = 0;
                        next = root;
                    } else {
                        t.obj = (permute && (flags & F_SHADOW)) ? S.nextAny[t.obj] : t.obj + 1;
                    }
                }
            } else if (t.cur >= 0) {
                NodeVisit nv;
                const int2 recRefs = make_int2(__float_as_int(r3.x), __float_as_int(r3.y));
                if (__any(flags & F_PAR_LOCAL)) node_boxes_loaded<true>(r0, r1, r2, recRefs, t.lo, t.linv, tMin, nv);
                else node_boxes_loaded<false>(r0, r1, r2, recRefs, t.lo, t.linv, tMin, nv);
                const int2 refs = nv.refs;
                const float EF = nv.EF, ES = nv.ES;
                const bool okF = nv.okF, okS = nv.okS;
                if (COUNT) cNode += (refs.x != RT_REF_ABSENT ? 1 : 0) + (refs.y != RT_REF_ABSENT ? 1 : 0);
                const bool goF = refs.x > RT_REF_LIVE_MIN && okF && (t.best.closest > EF);
                const bool canS = refs.y > RT_REF_LIVE_MIN && okS;
                if (goF) {
                    if (canS) wf_push(t, stack, deep, make_uint2((uint32_t)refs.y, __float_as_uint(ES)));
                    next = refs.x;
                } else if (canS && (t.best.closest > ES)) {
                    next = refs.y;
                } else {
                    next = wf_pop(t, stack, deep);
                }
            } else {
                int remaining;
                if (COUNT) cTri++;
                const bool hit = lane_triangle_loaded(t, S.objects[t.obj].fwd, r0, r1, r2, tMin, remaining);
                if (!COUNT && hit && (flags & F_SHADOW)) { st = 2; t.obj = parentCount; }   // any hit settles a shadow ray
                else if (remaining <= 1) next = wf_pop(t, stack, deep);
                else next = t.cur - 1;   // ~(k+1)
            }
            if (next == RT_NONE) {   // this object is done: on to the next parent
                t.obj = (permute && (flags & F_SHADOW)) ? S.nextAny[t.obj] : t.obj + 1;
                next = RT_WF_BOX;
            }
            t.cur = next;
        }
    }
}

#include "rt_team.inl"

// draw = gamma(accum.rgb * 1/total_frames), draw.a = accum.a (raytrace.comp:794-801)
// Second half of a launch in pieces (DevLaunch::pieces): the frames' contributions are added to the accumulation pixel in
// frame order, one after the other as separate dispatches would add them, and the draw pixel is written for the last
// frame (Pathtrace, raytrace.comp:787-803). Same block -> pixel map as the kernels that wrote the pieces.
__global__ __launch_bounds__(64) void RT_SYM(rt_sum_pieces)(rt_push_constants pc, DevLaunch L, float4* __restrict__ accum, float4* __restrict__ out,
                                                            const float4* __restrict__ partial)
{
    const int lane = threadIdx.x;
    const int wavesX = L.tileSize >> 3;
    const int wavesPerTile = wavesX * wavesX;
    const int ownedTile = blockIdx.x / wavesPerTile;
    const int sub = blockIdx.x - ownedTile * wavesPerTile;
    const int tile = L.tileRank + ownedTile * L.tileWorld;
    int tx, ty;
    rt_tile_of_position(tile, L.tilesX, &tx, &ty);
    const int sy = sub / wavesX, sx = sub - sy * wavesX;
    const int x = tx * L.tileSize + sx * 8 + (lane & 7);
    const int y = ty * L.tileSize + sy * 8 + (lane >> 3);
    if (!(x < L.width && y < L.height)) return;
    const size_t idx = (size_t)y * (size_t)L.width + (size_t)x;
    const size_t pixels = (size_t)L.width * (size_t)L.height;
    float4 acc = accum[idx];
    float4 nc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    for (int f = 0; f < L.frames; ++f) {
        nc = partial[(size_t)f * pixels + idx];
        acc.x += nc.x; acc.y += nc.y; acc.z += nc.z;
    }
    accum[idx] = acc;
    const int lastFrame = pc.frame + (L.frames - 1) * L.frameStride;
    const float accumulationMultiplier = pc.accumulateFrames == 1 ? 1.0f / (float)(lastFrame + 1) : 1.0f;
    float4 avg = pc.accumulateFrames == 1 ? acc : make_float4(nc.x, nc.y, nc.z, 1.0f);
    avg.x *= accumulationMultiplier; avg.y *= accumulationMultiplier; avg.z *= accumulationMultiplier;
    out[idx] = make_float4(linear_to_gamma(avg.x), linear_to_gamma(avg.y), linear_to_gamma(avg.z), avg.w);
}

__global__ __launch_bounds__(256) void RT_SYM(rt_resolve_kernel)(const float4* __restrict__ accum, float4* __restrict__ out,
                                                                 size_t pixels, float multiplier)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < pixels; i += stride) {
        float4 a = accum[i];
        a.x *= multiplier; a.y *= multiplier; a.z *= multiplier;
        out[i] = make_float4(linear_to_gamma(a.x), linear_to_gamma(a.y), linear_to_gamma(a.z), a.w);
    }
}

// rt_math.h over an array (which: 0 exp2, 1 exp, 2 log2, 3 acos, 4 div10000): lets a test compare the device's
// results with the host's bit for bit.
__global__ __launch_bounds__(256) void RT_SYM(rt_math_kernel)(int which, const float* __restrict__ x, float* __restrict__ y, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride)
        y[i] = which == 0 ? rt_exp2f(x[i]) : which == 1 ? rt_expf(x[i]) : which == 2 ? rt_log2f(x[i]) : which == 3 ? rt_acosf(x[i]) : rt_div10000f(x[i]);
}

// ---- launchers (called from rt_host.cpp) ---------------------------------------

#if defined(RT_SECTION_TIMERS) && defined(RT_FLAVOUR_STRICT)
// diagnostic build: read (and clear) the per-section cycle sums of the lock-step kernel
extern "C" int rt_debug_sections(unsigned long long out[8])
{
    unsigned long long zero[8] = { 0 };
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sectionCycles), sizeof(zero)) != hipSuccess) return -1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_sectionCycles), zero, sizeof(zero)) != hipSuccess) return -1;
    return 0;
}
#endif

#if defined(RT_WORK_COUNTERS) && defined(RT_FLAVOUR_STRICT)
// reads and clears the work counters of the strict flavour's two-samples-per-lane kernels (the device must be idle)
extern "C" int rt_debug_work_counters(unsigned long long out[RT_WORK_SLOTS])
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_workCounters), RT_WORK_SLOTS * sizeof(unsigned long long)) != hipSuccess) return -1;
    unsigned long long zero[RT_WORK_SLOTS] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_workCounters), zero, sizeof(zero)) != hipSuccess) return -1;
    return 0;
}
#endif

extern "C" hipError_t RT_SYM(rt_launch_megakernel)(const DevScene* S, const rt_push_constants* pc, const DevLaunch* L,
                                                   void* accum, void* out, DevCounters* counters, int countBuild,
                                                   int statsOnly, hipStream_t stream)
{
    (void)hipGetLastError();      // an error left behind by an earlier, unrelated HIP call is not this launch's
    const int wavesX = L->tileSize / 8;
    const long long blocks = (long long)L->ownedTiles * wavesX * wavesX;
    if (blocks <= 0) return hipSuccess;
    if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;
    const size_t lds = (size_t)S->stackDepth * 64 * sizeof(uint2);
    dim3 grid((unsigned)blocks), block(64);
    if (countBuild)
        hipLaunchKernelGGL(RT_SYM(rt_megakernel)<true>, grid, block, lds, stream, *S, *pc, *L, (float4*)accum, (float4*)out,
                           counters, statsOnly);
    else
        hipLaunchKernelGGL(RT_SYM(rt_megakernel)<false>, grid, block, lds, stream, *S, *pc, *L, (float4*)accum, (float4*)out,
                           counters, statsOnly);
    return hipGetLastError();
}

extern "C" hipError_t RT_SYM(rt_launch_pairkernel)(const DevScene* S, const rt_push_constants* pc, const DevLaunch* L,
                                                   void* accum, void* out, int pixelPairs, void* partial, hipStream_t stream)
{
    (void)hipGetLastError();      // an error left behind by an earlier, unrelated HIP call is not this launch's
    const int wavesX = L->tileSize / 8;
    if (pixelPairs && (L->tileSize & 15)) return hipErrorInvalidValue;      // 8 x 16 pixel footprints need whole 16-row bands
    const long long waves = (long long)L->ownedTiles * wavesX * (pixelPairs ? wavesX / 2 : wavesX);
    if (partial && (pixelPairs || L->pieces < 1 || L->pieceFrames < 1 || (long long)L->pieces * L->pieceFrames < L->frames)) return hipErrorInvalidValue;
    const long long blocks = partial ? waves * L->pieces : waves;
    if (blocks <= 0) return hipSuccess;
    if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;
    // wide scenes: the walk keeps RT_WIDE_LDS_LEVELS entries of a lane's stack in LDS (deeper ones in private memory), the LDS-resident path state behind them
    DevScene scene = *S;
    if (scene.stackDepth > RT_WIDE_LDS_LEVELS) scene.stackDepth = RT_WIDE_LDS_LEVELS;
    S = &scene;
    const size_t lds = (size_t)S->stackDepth * 64 * sizeof(uint2) + (size_t)RT_RES_DWORDS * 64 * sizeof(float);
    const dim3 grid((unsigned)blocks), block(64);
    // <true> knows rebuilt trees (near-first traversal), <false> has that handling compiled out
    // <2> walks four-child records (the wide twin of an order-free near-first scene)
    // <3> walks them from the hierarchy over the parents' boxes (wide scenes with many BVH parents)
    const bool nf = S->nearFirstBase != 0x7fffffff, wide = S->nodes4 != nullptr && nf, tlas = wide && S->tlasRoot != RT_REF_EMPTY;
    // <4> walks the reference's trees folded into four-child records, in the reference's order
    const bool refWide = S->nodes4 != nullptr && !nf;
    if (refWide && pixelPairs) hipLaunchKernelGGL(RT_SYM(rt_pixelpairkernel)<4>, grid, block, lds, stream, *S, *pc, *L, (float4*)accum, (float4*)out);
    else if (refWide) hipLaunchKernelGGL(RT_SYM(rt_pairkernel)<4>, grid, block, lds, stream, *S, *pc, *L, (float4*)accum, (float4*)out, (float4*)partial);
    else
    if (tlas && pixelPairs) hipLaunchKernelGGL(RT_SYM(rt_pixelpairkernel)<3>, grid, block, lds, stream, *S, *pc, *L, (float4*)accum, (float4*)out);
    else if (tlas) hipLaunchKernelGGL(RT_SYM(rt_pairkernel)<3>, grid, block, lds, stream, *S, *pc, *L, (float4*)accum, (float4*)out, (float4*)partial);
    else if (wide && pixelPairs) hipLaunchKernelGGL(RT_SYM(rt_pixelpairkernel)<2>, grid, block, lds, stream, *S, *pc, *L, (float4*)accum, (float4*)out);
    else if (wide) hipLaunchKernelGGL(RT_SYM(rt_pairkernel)<2>, grid, block, lds, stream, *S, *pc, *L, (float4*)accum, (float4*)out, (float4*)partial);
    else if (nf && pixelPairs) hipLaunchKernelGGL(RT_SYM(rt_pixelpairkernel)<1>, grid, block, lds, stream, *S, *pc, *L, (float4*)accum, (float4*)out);
    else if (nf) hipLaunchKernelGGL(RT_SYM(rt_pairkernel)<1>, grid, block, lds, stream, *S, *pc, *L, (float4*)accum, (float4*)out, (float4*)partial);
    else if (pixelPairs) hipLaunchKernelGGL(RT_SYM(rt_pixelpairkernel)<0>, grid, block, lds, stream, *S, *pc, *L, (float4*)accum, (float4*)out);
    else hipLaunchKernelGGL(RT_SYM(rt_pairkernel)<0>, grid, block, lds, stream, *S, *pc, *L, (float4*)accum, (float4*)out, (float4*)partial);
    if (partial) {
        if (hipGetLastError() != hipSuccess) return hipErrorLaunchFailure;
        hipLaunchKernelGGL(RT_SYM(rt_sum_pieces), dim3((unsigned)waves), block, 0, stream, *pc, *L, (float4*)accum, (float4*)out, (const float4*)partial);
    }
    return hipGetLastError();
}

extern "C" hipError_t RT_SYM(rt_launch_resolve)(const void* accum, void* out, size_t pixels, float multiplier, hipStream_t stream)
{
    (void)hipGetLastError();      // an error left behind by an earlier, unrelated HIP call is not this launch's
    if (pixels == 0) return hipSuccess;
    size_t blocks = (pixels + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(RT_SYM(rt_resolve_kernel), dim3((unsigned)blocks), dim3(256), 0, stream, (const float4*)accum,
                       (float4*)out, pixels, multiplier);
    return hipGetLastError();
}

extern "C" hipError_t RT_SYM(rt_launch_schedkernel)(const DevScene* S, const rt_push_constants* pc, const DevLaunch* L,
                                                    void* accum, void* out, DevCounters* counters, int countBuild,
                                                    int statsOnly, unsigned long long* queue, int blocks, hipStream_t stream,
                                                    unsigned long long* diag)
{
    (void)hipGetLastError();      // an error left behind by an earlier, unrelated HIP call is not this launch's
    if (L->ownedTiles <= 0 || blocks <= 0) return hipSuccess;
    const size_t lds = (size_t)S->stackDepth * 64 * sizeof(uint2);
    hipError_t e = hipMemsetAsync(queue, 0, sizeof(unsigned long long), stream);
    if (e != hipSuccess) return e;
    dim3 grid((unsigned)blocks), block(64);
    if (diag)
        hipLaunchKernelGGL((RT_SYM(rt_schedkernel)<false, true>), grid, block, lds, stream, *S, *pc, *L, (float4*)accum,
                           (float4*)out, counters, statsOnly, queue, diag);
    else if (countBuild)
        hipLaunchKernelGGL((RT_SYM(rt_schedkernel)<true, false>), grid, block, lds, stream, *S, *pc, *L, (float4*)accum,
                           (float4*)out, counters, statsOnly, queue, diag);
    else
        hipLaunchKernelGGL((RT_SYM(rt_schedkernel)<false, false>), grid, block, lds, stream, *S, *pc, *L, (float4*)accum,
                           (float4*)out, counters, statsOnly, queue, diag);
    return hipGetLastError();
}

extern "C" hipError_t RT_SYM(rt_launch_wf_path)(const DevScene* S, const rt_push_constants* pc, const DevLaunch* L,
                                                const DevWavefront* W, void* accum, void* out, DevCounters* counters,
                                                int countBuild, int statsOnly, int round, int inlineWalks, int blocks,
                                                hipStream_t stream)
{
    (void)hipGetLastError();      // an error left behind by an earlier, unrelated HIP call is not this launch's
    if (L->ownedTiles <= 0 || blocks <= 0) return hipSuccess;
    const size_t lds = inlineWalks ? (size_t)S->stackDepth * 64 * sizeof(uint2) : 0;
    dim3 grid((unsigned)blocks), block(64);
    if (countBuild) {
        if (inlineWalks)
            hipLaunchKernelGGL((RT_SYM(rt_wf_path)<true, true>), grid, block, lds, stream, *S, *pc, *L, *W, (float4*)accum,
                               (float4*)out, counters, statsOnly, round);
        else
            hipLaunchKernelGGL((RT_SYM(rt_wf_path)<true, false>), grid, block, lds, stream, *S, *pc, *L, *W, (float4*)accum,
                               (float4*)out, counters, statsOnly, round);
    } else {
        if (inlineWalks)
            hipLaunchKernelGGL((RT_SYM(rt_wf_path)<false, true>), grid, block, lds, stream, *S, *pc, *L, *W, (float4*)accum,
                               (float4*)out, counters, statsOnly, round);
        else
            hipLaunchKernelGGL((RT_SYM(rt_wf_path)<false, false>), grid, block, lds, stream, *S, *pc, *L, *W, (float4*)accum,
                               (float4*)out, counters, statsOnly, round);
    }
    return hipGetLastError();
}

extern "C" hipError_t RT_SYM(rt_launch_wf_walk)(const DevScene* S, const rt_push_constants* pc, const DevWavefront* W,
                                                int countBuild, int round, int blocks, hipStream_t stream)
{
    (void)hipGetLastError();      // an error left behind by an earlier, unrelated HIP call is not this launch's
    if (blocks <= 0) return hipSuccess;
    const size_t lds = (size_t)(S->stackDepth < RT_WF_LDS_LEVELS ? S->stackDepth : RT_WF_LDS_LEVELS) * 64 * sizeof(uint2);
    dim3 grid((unsigned)blocks), block(64);
    if (countBuild)
        hipLaunchKernelGGL(RT_SYM(rt_wf_walk)<true>, grid, block, lds, stream, *S, *pc, *W, round);
    else
        hipLaunchKernelGGL(RT_SYM(rt_wf_walk)<false>, grid, block, lds, stream, *S, *pc, *W, round);
    return hipGetLastError();
}

extern "C" hipError_t RT_SYM(rt_launch_math)(int which, const float* x, float* y, size_t n, hipStream_t stream)
{
    (void)hipGetLastError();      // an error left behind by an earlier, unrelated HIP call is not this launch's
    if (n == 0) return hipSuccess;
    size_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(RT_SYM(rt_math_kernel), dim3((unsigned)blocks), dim3(256), 0, stream, which, x, y, n);
    return hipGetLastError();
}
