/*
 * rt_oracle.c - CPU restatement of the reference path tracer. TEST INFRASTRUCTURE
 * ONLY: nothing in the product (raytracer_amd/, include/) links, loads or calls
 * this file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg use it, and only as the checker / reported baseline.
 *
 * PARITY UNPINNED: the reference ships no tests, no golden images and no
 * known-answer vectors for this path, and neither its Vulkan compute shader nor
 * its MSVC/glm host code can be built or run in this environment. This file is
 * therefore a line-by-line restatement of the reference's sources in strict
 * IEEE-754 binary32 (compile with -ffp-contract=off), pinned only by (a) exact
 * integer known-answer values of the RNG derived from Random.glsl by hand,
 * (b) the camera numbers implied by the shipped defaults, (c) the reference's own
 * OBJ parser (tiny_obj_loader.h, which does compile here; see oracle/Makefile
 * target _ref) for the vertex stream, and (d) structure counts of the BVH.
 *
 * What is restated, with the reference location each part follows:
 *   Random.glsl:4-88                    -> rng_*, sample_square, random_* below
 *   raytrace.comp:205-276               -> intersect_aabb
 *   raytrace.comp:278-328               -> intersect_triangle
 *   raytrace.comp:345-355               -> distance_to_aabb
 *   raytrace.comp:357-430               -> near_zero, reflectance, dielectric_scatter,
 *                                          material_scatter
 *   raytrace.comp:448-610               -> get_hit
 *   raytrace.comp:623-690               -> environment_colour, get_ray
 *   raytrace.comp:692-769               -> ray_colour
 *   raytrace.comp:771-875               -> pathtrace_pixel, render_boxes, render_depth,
 *                                          shade_pixel (main)
 *   HardwareRenderer.cpp:793-824        -> orc_camera_push_constants
 *   HardwareRenderer.cpp:502-551        -> convert_scene_object
 *   HardwareRenderer.cpp:574-649        -> orc_default_scene
 *   HardwareRenderer.cpp:1157-1314      -> orc_load_model
 *   ModelLoader.cpp:10-120,262-330      -> build_bvh, split_bvh_node, expand_node_aabb
 *   tiny_obj_loader.h:891-1021,1161-1212,1488-1584,2558-2810 -> obj_* below
 * glm is not present in the reference checkout (empty submodule); glm::translate,
 * rotate, scale, inverse, normalize and mat4*vec4 are restated from glm's
 * published definitions (glm 0.9.9 / 1.0: ext/matrix_transform.inl,
 * detail/func_matrix.inl compute_inverse<4,4>, detail/type_mat4x4.inl).
 *
 * GLSL built-ins whose precision the GLSL spec leaves open are fixed here as:
 *   normalize(v) = v * (1/sqrt(dot(v,v)));  length(v) = sqrt(dot(v,v))
 *   dot = left-to-right sum;  cross = (a.y*b.z-b.y*a.z, a.z*b.x-b.z*a.x, a.x*b.y-b.x*a.y)
 *   mix(a,b,t) = a*(1-t)+b*t;  reflect(I,N) = I - (2*dot(N,I))*N
 *   refract: k = 1-eta*eta*(1-d*d), k<0 ? 0 : eta*I-(eta*d+sqrt(k))*N
 *   pow(x,y) = exp2(y*log2(x))  (NaN for x<0, as GPU drivers and Mesa lower it)
 *   exp, exp2, log2, acos = the fixed-arithmetic versions of include/rt_math.h (<= 4 ulp), not the
 *   host libm's, so that the kernels - which compile the same header - agree in the last bit
 *   max(a,b) = a<b?b:a;  min(a,b) = b<a?b:a;  mat4*vec4 = (c0*x+c1*y)+(c2*z+c3*w)
 *   inverse(mat4) = glm's cofactor form
 */
#define _GNU_SOURCE
#include <ctype.h>
#include <float.h>
#include <limits.h>
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/rt_math.h"   /* exp / exp2 / log2 / acos in fixed binary32 arithmetic, shared with the kernels */

/* ------------------------------------------------------------------------- */
/* data contract (own copies of the layouts; RaytracerTypes.h:22-149)         */
/* ------------------------------------------------------------------------- */

typedef struct { float x, y, z; } v3;
typedef struct { float c[4][4]; } m4; /* c[column][row], glm::mat4 memory order */

typedef struct {
    float pixel00[3]; int32_t raysPerPixel;
    float deltaU[3];  int32_t maxBounces;
    float deltaV[3];  float defocusAngle;
    float camPos[3];  int32_t parentBVHCount;
    float defocusU[3]; int32_t accumulateFrames;
    float defocusV[3]; int32_t frame;
    float sunDir[3];  float sunStrength;
    float sunColour[3]; int32_t renderMode;
    int32_t triangleTestThreshold; int32_t bvhNodeTestThreshold;
    float depthDebugScale; int32_t pad;
} orc_pc;

typedef struct { float mn[3]; float p0; float mx[3]; float p1; } orc_aabb;
typedef struct { orc_aabb aabb; int32_t left, right, triStart, triCount; } orc_node;
typedef struct { orc_node node; int32_t objectIndex; int32_t pad[3]; } orc_parent;
typedef struct { int32_t triStart, triCount, materialIndex; float pad; float inv[16]; } orc_object;
typedef struct { float colour[3]; float smoothness; float fuzziness; float emission;
                 float refractionIndex; float pad1; float absorbtion[3]; float pad3; } orc_material;

typedef struct {
    const orc_parent* parents; const orc_object* objects; const orc_material* materials;
    const float *v0, *v1, *v2, *n0, *n1, *n2;          /* vec4 arrays */
    const float *aabbMins, *aabbMaxs;                  /* vec4 arrays */
    const int32_t *left, *right, *triStart, *triCount; /* per node */
} orc_buffers;

_Static_assert(sizeof(orc_pc) == 144, "pc");
_Static_assert(sizeof(orc_parent) == 64, "parent");
_Static_assert(sizeof(orc_object) == 80, "object");
_Static_assert(sizeof(orc_material) == 48, "material");

/* ------------------------------------------------------------------------- */
/* small vector helpers                                                       */
/* ------------------------------------------------------------------------- */

static inline v3 V(float x, float y, float z) { v3 r = { x, y, z }; return r; }
static inline v3 vadd(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 vsub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 vmul(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 vscale(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
static inline v3 vneg(v3 a) { return V(-a.x, -a.y, -a.z); }
static inline float vdot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline v3 vcross(v3 a, v3 b)
{
    return V(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);
}
static inline float vlength(v3 a) { return sqrtf(vdot(a, a)); }
static inline v3 vnormalize(v3 a) { float inv = 1.0f / sqrtf(vdot(a, a)); return vscale(a, inv); }
static inline float fmax_glsl(float a, float b) { return a < b ? b : a; }
static inline float fmin_glsl(float a, float b) { return b < a ? b : a; }
static inline float clamp_glsl(float x, float lo, float hi) { return fmin_glsl(fmax_glsl(x, lo), hi); }
static inline float mixf(float a, float b, float t) { return a * (1.0f - t) + b * t; }
static inline v3 vmix(v3 a, v3 b, float t) { return V(mixf(a.x, b.x, t), mixf(a.y, b.y, t), mixf(a.z, b.z, t)); }
static inline float pow_glsl(float x, float y) { return rt_exp2f(y * rt_log2f(x)); }
static inline v3 ld3(const float* p) { return V(p[0], p[1], p[2]); }
static inline float comp(v3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

static inline v3 reflect_glsl(v3 I, v3 N) { return vsub(I, vscale(N, 2.0f * vdot(N, I))); }
static inline v3 refract_glsl(v3 I, v3 N, float eta)
{
    float d = vdot(N, I);
    float k = 1.0f - eta * eta * (1.0f - d * d);
    if (k < 0.0f) return V(0.0f, 0.0f, 0.0f);
    return vsub(vscale(I, eta), vscale(N, eta * d + sqrtf(k)));
}

/* mat4 * vec4, glm's pairing: (c0*x + c1*y) + (c2*z + c3*w) */
static inline void m4_mul_v4(const m4* m, float x, float y, float z, float w, float out[4])
{
    for (int r = 0; r < 4; ++r) {
        float a = m->c[0][r] * x + m->c[1][r] * y;
        float b = m->c[2][r] * z + m->c[3][r] * w;
        out[r] = a + b;
    }
}

/* glm::inverse(mat4) / GLSL inverse(mat4): cofactor expansion */
static m4 m4_inverse(const m4* mm)
{
    const float (*m)[4] = mm->c;
    float Coef00 = m[2][2] * m[3][3] - m[3][2] * m[2][3];
    float Coef02 = m[1][2] * m[3][3] - m[3][2] * m[1][3];
    float Coef03 = m[1][2] * m[2][3] - m[2][2] * m[1][3];
    float Coef04 = m[2][1] * m[3][3] - m[3][1] * m[2][3];
    float Coef06 = m[1][1] * m[3][3] - m[3][1] * m[1][3];
    float Coef07 = m[1][1] * m[2][3] - m[2][1] * m[1][3];
    float Coef08 = m[2][1] * m[3][2] - m[3][1] * m[2][2];
    float Coef10 = m[1][1] * m[3][2] - m[3][1] * m[1][2];
    float Coef11 = m[1][1] * m[2][2] - m[2][1] * m[1][2];
    float Coef12 = m[2][0] * m[3][3] - m[3][0] * m[2][3];
    float Coef14 = m[1][0] * m[3][3] - m[3][0] * m[1][3];
    float Coef15 = m[1][0] * m[2][3] - m[2][0] * m[1][3];
    float Coef16 = m[2][0] * m[3][2] - m[3][0] * m[2][2];
    float Coef18 = m[1][0] * m[3][2] - m[3][0] * m[1][2];
    float Coef19 = m[1][0] * m[2][2] - m[2][0] * m[1][2];
    float Coef20 = m[2][0] * m[3][1] - m[3][0] * m[2][1];
    float Coef22 = m[1][0] * m[3][1] - m[3][0] * m[1][1];
    float Coef23 = m[1][0] * m[2][1] - m[2][0] * m[1][1];

    float Fac0[4] = { Coef00, Coef00, Coef02, Coef03 };
    float Fac1[4] = { Coef04, Coef04, Coef06, Coef07 };
    float Fac2[4] = { Coef08, Coef08, Coef10, Coef11 };
    float Fac3[4] = { Coef12, Coef12, Coef14, Coef15 };
    float Fac4[4] = { Coef16, Coef16, Coef18, Coef19 };
    float Fac5[4] = { Coef20, Coef20, Coef22, Coef23 };
    float Vec0[4] = { m[1][0], m[0][0], m[0][0], m[0][0] };
    float Vec1[4] = { m[1][1], m[0][1], m[0][1], m[0][1] };
    float Vec2[4] = { m[1][2], m[0][2], m[0][2], m[0][2] };
    float Vec3[4] = { m[1][3], m[0][3], m[0][3], m[0][3] };
    static const float SignA[4] = { +1.0f, -1.0f, +1.0f, -1.0f };
    static const float SignB[4] = { -1.0f, +1.0f, -1.0f, +1.0f };

    m4 inv;
    for (int i = 0; i < 4; ++i) {
        float Inv0 = Vec1[i] * Fac0[i] - Vec2[i] * Fac1[i] + Vec3[i] * Fac2[i];
        float Inv1 = Vec0[i] * Fac0[i] - Vec2[i] * Fac3[i] + Vec3[i] * Fac4[i];
        float Inv2 = Vec0[i] * Fac1[i] - Vec1[i] * Fac3[i] + Vec3[i] * Fac5[i];
        float Inv3 = Vec0[i] * Fac2[i] - Vec1[i] * Fac4[i] + Vec2[i] * Fac5[i];
        inv.c[0][i] = Inv0 * SignA[i];
        inv.c[1][i] = Inv1 * SignB[i];
        inv.c[2][i] = Inv2 * SignA[i];
        inv.c[3][i] = Inv3 * SignB[i];
    }
    float d0 = m[0][0] * inv.c[0][0];
    float d1 = m[0][1] * inv.c[1][0];
    float d2 = m[0][2] * inv.c[2][0];
    float d3 = m[0][3] * inv.c[3][0];
    float Dot1 = (d0 + d1) + (d2 + d3);
    float OneOverDeterminant = 1.0f / Dot1;
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 4; ++r) inv.c[c][r] = inv.c[c][r] * OneOverDeterminant;
    return inv;
}

/* ------------------------------------------------------------------------- */
/* Random.glsl                                                                */
/* ------------------------------------------------------------------------- */

/* Random.glsl:4-12 */
static inline float rng_random_seed(uint32_t* seed)
{
    uint32_t s = *seed;
    s = (s ^ 61u) ^ (s >> 16);
    s *= 9u;
    s = s ^ (s >> 4);
    s *= 0x27d4eb2du;
    s = s ^ (s >> 15);
    *seed = s;
    return (float)(s % 10000u) / 10000.0f;
}

/* Random.glsl:28-32 */
static inline float rng_float_in_range(float mn, float mx, uint32_t* seed)
{
    *seed += 1u;
    return mixf(mn, mx, rng_random_seed(seed));
}

/* Random.glsl:34-41 */
uint32_t orc_seed_from_coords(uint32_t x, uint32_t y, uint32_t frame, uint32_t i)
{
    uint32_t seed = x * 374761393u + y * 668265263u + frame * 1442695040u + i * 88963407u;
    seed = (seed ^ (seed >> 16)) * 0x85ebca6bu;
    seed = (seed ^ (seed >> 13)) * 0xc2b2ae35u;
    seed = seed ^ (seed >> 16);
    return seed;
}

/* One RandomFloatInRange(0,1) draw, exported for the known-answer tests:
 * returns the new state, writes state % 10000. */
uint32_t orc_rng_draw(uint32_t state, uint32_t* mod10000)
{
    uint32_t s = state;
    (void)rng_float_in_range(0.0f, 1.0f, &s);
    if (mod10000) *mod10000 = s % 10000u;
    return s;
}

/* Random.glsl:43-49 */
static inline v3 sample_square(uint32_t* seed)
{
    *seed += 1u;
    float x = rng_float_in_range(0.0f, 1.0f, seed);
    float y = rng_float_in_range(0.0f, 1.0f, seed);
    return V(x - 0.5f, y - 0.5f, 0.0f);
}

/* The reference's two rejection loops have no exit but acceptance, and its generator is a fixed
 * permutation of the 32-bit states: a state on a short cycle whose draws are all rejected loops for
 * ever (the shader would hang the GPU). That does happen at production sample counts - at 3840x2160,
 * frame 1010, pixel (3563, 2062), sample 0, the Lambert branch never leaves RandomUnitVector (found by
 * enumerating frames 0..1023 of that size). The reference defines no value for such a sample, so this
 * restatement and the kernels give up after REJECTION_CAP rounds and return the zero vector: no lens
 * offset, no fuzz, and a Lambert direction equal to the normal (MaterialScatter's NearZero branch). */
#define REJECTION_CAP 65536

/* Random.glsl:51-60 */
static inline v3 random_in_unit_disk(uint32_t* seed)
{
    *seed += 1u;
    for (int round = 0; round < REJECTION_CAP; ++round) {
        float px = rng_float_in_range(-1.0f, 1.0f, seed);
        float py = rng_float_in_range(-1.0f, 1.0f, seed);
        v3 p = V(px, py, 0.0f);
        if (sqrtf(vlength(p)) < 1.0f) return p;
    }
    return V(0.0f, 0.0f, 0.0f);
}

/* Random.glsl:62-66 */
static inline v3 random_vector(uint32_t* seed)
{
    *seed += 1u;
    float x = rng_float_in_range(-1.0f, 1.0f, seed);
    float y = rng_float_in_range(-1.0f, 1.0f, seed);
    float z = rng_float_in_range(-1.0f, 1.0f, seed);
    return V(x, y, z);
}

/* Random.glsl:68-78 ; the literal 1e-160 is below the f32 range and reads as 0 */
static inline v3 random_unit_vector(uint32_t* seed)
{
    *seed += 1u;
    for (int round = 0; round < REJECTION_CAP; ++round) {
        v3 p = random_vector(seed);
        float lensq = vlength(p) * vlength(p);
        if (0.0f < lensq && lensq <= 1.0f) {
            float l = sqrtf(lensq);
            return V(p.x / l, p.y / l, p.z / l);
        }
    }
    return V(0.0f, 0.0f, 0.0f);
}

/* exported for tests: RandomUnitVector from a generator state */
void orc_random_unit_vector(uint32_t seed, float out3[3])
{
    v3 u = random_unit_vector(&seed);
    out3[0] = u.x; out3[1] = u.y; out3[2] = u.z;
}

/* Random.glsl:80-88 */
static inline v3 random_on_hemisphere(v3 normal, uint32_t* seed)
{
    *seed += 1u;
    v3 u = random_unit_vector(seed);
    if (vdot(u, normal) > 0.0f) return u;
    return vneg(u);
}

/* exported for tests: SampleSquare offsets for a seed */
void orc_sample_square(uint32_t seed, float out2[2])
{
    v3 o = sample_square(&seed);
    out2[0] = o.x; out2[1] = o.y;
}

/* rt_math.h functions over an array (which: 0 exp2, 1 exp, 2 log2, 3 acos, 4 div10000), for the accuracy tests */
void orc_math_apply(int which, const float* x, float* y, size_t n)
{
    for (size_t i = 0; i < n; ++i)
        y[i] = which == 0 ? rt_exp2f(x[i]) : which == 1 ? rt_expf(x[i]) : which == 2 ? rt_log2f(x[i]) : which == 3 ? rt_acosf(x[i])
                                                                                                    : rt_div10000f(x[i]);
}

/* ------------------------------------------------------------------------- */
/* raytrace.comp                                                              */
/* ------------------------------------------------------------------------- */

typedef struct { v3 origin, direction; } ray_t;
typedef struct { v3 point, normal; int matIndex; float t; int frontFace; int hitObject; } hit_t;
typedef struct { long long rays, nodeTests, triTests; } counters_t;

/* raytrace.comp:205-276 (both overloads share this body) */
static int intersect_aabb(const ray_t* ray, v3 boxMin, v3 boxMax, float* tmin, float* tmax)
{
    for (int a = 0; a < 3; ++a) {
        float originA = comp(ray->origin, a);
        float dirA = comp(ray->direction, a);
        float minA = comp(boxMin, a);
        float maxA = comp(boxMax, a);
        if (fabsf(dirA) < 1e-8f) {
            if (originA < minA || originA > maxA) return 0;
        } else {
            float invD = 1.0f / dirA;
            float t0 = (minA - originA) * invD;
            float t1 = (maxA - originA) * invD;
            if (invD < 0.0f) { float tmp = t0; t0 = t1; t1 = tmp; }
            *tmin = fmax_glsl(*tmin, t0);
            *tmax = fmin_glsl(*tmax, t1);
            if (*tmax <= *tmin) return 0;
        }
    }
    return 1;
}

/* raytrace.comp:278-328 */
static int intersect_triangle(const orc_buffers* b, const ray_t* ray, int tri, float tmin, float tmax, hit_t* rec)
{
    v3 triV0 = ld3(b->v0 + 4 * (size_t)tri);
    v3 triV1 = ld3(b->v1 + 4 * (size_t)tri);
    v3 triV2 = ld3(b->v2 + 4 * (size_t)tri);
    v3 n0 = ld3(b->n0 + 4 * (size_t)tri);
    v3 n1 = ld3(b->n1 + 4 * (size_t)tri);
    v3 n2 = ld3(b->n2 + 4 * (size_t)tri);

    v3 edge1 = vsub(triV1, triV0);
    v3 edge2 = vsub(triV2, triV0);
    v3 h = vcross(ray->direction, edge2);
    float a = vdot(edge1, h);
    const float eps = 1e-5f;
    if (fabsf(a) < eps) return 0;

    float invDet = 1.0f / a;
    v3 s = vsub(ray->origin, triV0);
    float u = invDet * vdot(s, h);
    v3 sCross = vcross(s, edge1);
    float v = invDet * vdot(ray->direction, sCross);
    if (u < -eps || u > 1.0f + eps) return 0;
    if (v < -eps || u + v > 1.0f + eps) return 0;

    float hitT = invDet * vdot(edge2, sCross);
    if (!(hitT >= tmin && hitT <= tmax)) return 0;

    if (hitT > eps) {
        rec->t = hitT;
        rec->point = vadd(ray->origin, vscale(ray->direction, hitT));
        float w = 1.0f - u - v;
        v3 interp = vnormalize(vadd(vadd(vscale(n0, w), vscale(n1, u)), vscale(n2, v)));
        rec->frontFace = vdot(ray->direction, interp) < 0.0f;
        rec->normal = rec->frontFace ? interp : vneg(interp);
        rec->hitObject = 1;
        return 1;
    }
    return 0;
}

/* raytrace.comp:345-355 */
static float distance_to_aabb(v3 p, v3 boxMin, v3 boxMax)
{
    float sq = 0.0f;
    for (int i = 0; i < 3; ++i) {
        float v = comp(p, i), lo = comp(boxMin, i), hi = comp(boxMax, i);
        if (v < lo) sq += (lo - v) * (lo - v);
        if (v > hi) sq += (v - hi) * (v - hi);
    }
    return sqrtf(sq);
}

static void accept_hit(const ray_t* r, const m4* objectTransform, const m4* invT, const orc_object* obj,
                       const hit_t* temp, float tMin, float* closestSoFar, hit_t* rec)
{
    float wp[4];
    m4_mul_v4(objectTransform, temp->point.x, temp->point.y, temp->point.z, 1.0f, wp);
    v3 worldPoint = V(wp[0], wp[1], wp[2]);
    /* normalMatrix = mat3(transpose(inverseTransform)); mat3*vec3 sums left to right */
    v3 n = temp->normal;
    v3 wn = V(invT->c[0][0] * n.x + invT->c[0][1] * n.y + invT->c[0][2] * n.z,
              invT->c[1][0] * n.x + invT->c[1][1] * n.y + invT->c[1][2] * n.z,
              invT->c[2][0] * n.x + invT->c[2][1] * n.y + invT->c[2][2] * n.z);
    v3 worldNormal = vnormalize(wn);
    float worldT = vdot(vsub(worldPoint, r->origin), r->direction);
    if (worldT > tMin && worldT < *closestSoFar) {
        *closestSoFar = worldT;
        rec->t = worldT;
        rec->point = worldPoint;
        rec->normal = worldNormal;
        rec->matIndex = obj->materialIndex;
        rec->frontFace = vdot(r->direction, worldNormal) < 0.0f;
        rec->hitObject = 1;
    }
}

/* raytrace.comp:448-610 */
static int get_hit(const orc_buffers* b, const orc_pc* pc, const ray_t* r_in, float tMin, float tMax,
                   hit_t* rec, counters_t* cnt)
{
    ray_t r;
    r.origin = r_in->origin;
    r.direction = vnormalize(r_in->direction);
    cnt->rays++;

    float closestSoFar = tMax;
    rec->t = tMax;
    rec->hitObject = 0;

    for (int i = 0; i < pc->parentBVHCount; ++i) {
        const orc_parent* P = &b->parents[i];
        float imin = tMin, imax = closestSoFar;
        cnt->nodeTests++;
        v3 pmin = ld3(P->node.aabb.mn), pmax = ld3(P->node.aabb.mx);
        if (!intersect_aabb(&r, pmin, pmax, &imin, &imax)) continue;

        float distanceToParentNode = distance_to_aabb(r.origin, pmin, pmax);
        if (distanceToParentNode > closestSoFar) continue;

        const orc_object* obj = &b->objects[P->objectIndex];
        m4 invT; memcpy(&invT, obj->inv, sizeof(invT));

        float lo[4], ld[4];
        m4_mul_v4(&invT, r.origin.x, r.origin.y, r.origin.z, 1.0f, lo);
        m4_mul_v4(&invT, r.direction.x, r.direction.y, r.direction.z, 0.0f, ld);
        ray_t localRay;
        localRay.origin = V(lo[0], lo[1], lo[2]);
        localRay.direction = vnormalize(V(ld[0], ld[1], ld[2]));

        m4 objectTransform = m4_inverse(&invT);

        if (P->node.left == -1 && P->node.right == -1) {
            for (int t = 0; t < P->node.triCount; ++t) {
                cnt->triTests++;
                int tri = P->node.triStart + t;
                hit_t temp;
                if (intersect_triangle(b, &localRay, tri, tMin, closestSoFar, &temp))
                    accept_hit(&r, &objectTransform, &invT, obj, &temp, tMin, &closestSoFar, rec);
            }
            continue;
        }

        int stack[32];
        int sp = 0;
        if (P->node.left != -1) stack[sp++] = P->node.left;
        if (P->node.right != -1) stack[sp++] = P->node.right;

        while (sp > 0) {
            int nodeIdx = stack[--sp];
            float nmin = tMin, nmax = closestSoFar;
            cnt->nodeTests++;
            if (!intersect_aabb(&localRay, ld3(b->aabbMins + 4 * (size_t)nodeIdx),
                                ld3(b->aabbMaxs + 4 * (size_t)nodeIdx), &nmin, &nmax))
                continue;
            int leftChild = b->left[nodeIdx];
            int rightChild = b->right[nodeIdx];
            if (leftChild == -1 && rightChild == -1) {
                int start = b->triStart[nodeIdx];
                int count = b->triCount[nodeIdx];
                for (int t = 0; t < count; ++t) {
                    cnt->triTests++;
                    hit_t temp;
                    if (intersect_triangle(b, &localRay, start + t, tMin, closestSoFar, &temp))
                        accept_hit(&r, &objectTransform, &invT, obj, &temp, tMin, &closestSoFar, rec);
                }
            } else {
                if (sp > 30) { fprintf(stderr, "rt_oracle: stack[32] overflow (BVH too deep)\n"); abort(); }
                if (rightChild != -1) stack[sp++] = rightChild;
                if (leftChild != -1) stack[sp++] = leftChild;
            }
        }
    }
    return rec->hitObject;
}

/* raytrace.comp:363-369 */
static float reflectance(float cosine, float refIdx)
{
    float r0 = (1.0f - refIdx) / (1.0f + refIdx);
    r0 = r0 * r0;
    return r0 + (1.0f - r0) * pow_glsl(1.0f - cosine, 5.0f);
}

/* raytrace.comp:371-388 ; seed arrives by value */
static void dielectric_scatter(const orc_material* m, const ray_t* rayIn, const hit_t* rec, float refIdx,
                               v3* attenuation, ray_t* scattered, uint32_t seed)
{
    v3 e = V(rt_expf(-m->absorbtion[0] * rec->t), rt_expf(-m->absorbtion[1] * rec->t), rt_expf(-m->absorbtion[2] * rec->t));
    *attenuation = vmul(ld3(m->colour), e);
    float etai = rec->frontFace ? (1.0f / refIdx) : refIdx;
    v3 unit = vnormalize(rayIn->direction);
    float cosTheta = fmin_glsl(vdot(vneg(unit), rec->normal), 1.0f);
    float sinTheta = sqrtf(1.0f - cosTheta * cosTheta);
    int cannotRefract = etai * sinTheta > 1.0f;
    v3 dir;
    if (cannotRefract || reflectance(cosTheta, refIdx) > rng_float_in_range(0.0f, 1.0f, &seed))
        dir = reflect_glsl(unit, rec->normal);
    else
        dir = refract_glsl(unit, rec->normal, etai);
    scattered->origin = rec->point;
    scattered->direction = vnormalize(dir);
}

/* raytrace.comp:391-423 ; seed arrives by value */
static void material_scatter(const orc_buffers* b, const ray_t* rayIn, const hit_t* rec, v3* attenuation,
                             ray_t* scattered, uint32_t seed)
{
    float scatterOrReflect = rng_float_in_range(0.0f, 1.0f, &seed);
    const orc_material* m = &b->materials[rec->matIndex];
    if (m->refractionIndex > 0.0f) {
        dielectric_scatter(m, rayIn, rec, m->refractionIndex, attenuation, scattered, seed);
        return;
    }
    const float EPSILON = 1e-4f;
    if (scatterOrReflect < m->smoothness) {
        v3 reflected = reflect_glsl(vnormalize(rayIn->direction), rec->normal);
        reflected = vadd(reflected, vscale(random_on_hemisphere(rec->normal, &seed), m->fuzziness));
        scattered->origin = vadd(rec->point, vscale(reflected, EPSILON));
        scattered->direction = vnormalize(reflected);
        *attenuation = ld3(m->colour);
    } else {
        v3 sd = vadd(rec->normal, random_unit_vector(&seed));
        const float s = 1e-8f;
        if (fabsf(sd.x) < s && fabsf(sd.y) < s && fabsf(sd.z) < s) sd = rec->normal;
        scattered->origin = vadd(rec->point, vscale(sd, EPSILON));
        scattered->direction = vnormalize(sd);
        *attenuation = ld3(m->colour);
    }
}

/* raytrace.comp:623-668 */
static v3 environment_colour(const orc_pc* pc, const ray_t* ray)
{
    v3 unitDir = vnormalize(ray->direction);
    v3 sunDir = vnormalize(ld3(pc->sunDir));
    float sunDotUp = vdot(vneg(sunDir), V(0.0f, 1.0f, 0.0f));
    v3 dayZenith = V(0.5f, 0.7f, 1.0f), dayHorizon = V(1.0f, 1.0f, 1.0f), sunrise = V(1.0f, 0.4f, 0.2f);
    v3 nightZenith = V(0.0f, 0.0f, 0.0f), nightHorizon = V(0.0f, 0.0f, 0.0f);
    float viewT = clamp_glsl(0.5f * (unitDir.y + 1.0f), 0.0f, 1.0f);
    v3 baseSky;
    if (sunDotUp > 0.1f) {
        baseSky = vmix(dayHorizon, dayZenith, viewT);
    } else if (sunDotUp > -0.1f) {
        float blend = (sunDotUp + 0.1f) / 0.2f;
        v3 daySky = vmix(dayHorizon, dayZenith, viewT);
        v3 sunSky = vmix(sunrise, vscale(sunrise, 0.5f), viewT);
        baseSky = vmix(sunSky, daySky, blend);
    } else {
        baseSky = vmix(nightHorizon, nightZenith, viewT);
    }
    float cosTheta = vdot(unitDir, vneg(sunDir));
    cosTheta = clamp_glsl(cosTheta, 0.0f, 1.0f);
    const float halfDegree = 0.5f * 0.017453292519943295f; /* radians(0.5) */
    float sunDisk = rt_expf(-pow_glsl(rt_acosf(cosTheta) / halfDegree, 2.0f));
    v3 sunGlow = vscale(vscale(ld3(pc->sunColour), pc->sunStrength), sunDisk);
    return vadd(baseSky, sunGlow);
}

/* raytrace.comp:670-690 ; seed arrives by value */
static ray_t get_ray(const orc_pc* pc, int px, int py, uint32_t seed)
{
    v3 offset = sample_square(&seed);
    v3 ps = vadd(vadd(ld3(pc->pixel00), vscale(ld3(pc->deltaU), (float)px + offset.x)),
                 vscale(ld3(pc->deltaV), (float)py + offset.y));
    v3 origin;
    if (pc->defocusAngle <= 0.0f) {
        origin = ld3(pc->camPos);
    } else {
        uint32_t s2 = seed; /* DefocusDiskSample takes the seed by value */
        v3 p = random_in_unit_disk(&s2);
        origin = vadd(vadd(ld3(pc->camPos), vscale(ld3(pc->defocusU), p.x)), vscale(ld3(pc->defocusV), p.y));
    }
    ray_t ray;
    ray.origin = origin;
    ray.direction = vnormalize(vsub(ps, origin));
    return ray;
}

/* raytrace.comp:692-769 ; `seed` is inout in the shader but never advanced */
static v3 ray_colour(const orc_buffers* b, const orc_pc* pc, ray_t ray, uint32_t seed, counters_t* cnt)
{
    v3 result = V(0.0f, 0.0f, 0.0f);
    v3 throughput = V(1.0f, 1.0f, 1.0f);
    hit_t rec;
    memset(&rec, 0, sizeof(rec));
    for (int bounce = 0; bounce < pc->maxBounces + 1; ++bounce) {
        get_hit(b, pc, &ray, 0.001f, 1e20f, &rec, cnt);
        if (!rec.hitObject) {
            v3 bg = environment_colour(pc, &ray);
            result = vadd(result, vmul(throughput, bg));
            break;
        }
        if (!rec.frontFace) break;

        const orc_material* m = &b->materials[rec.matIndex];
        v3 emitted = vscale(ld3(m->colour), m->emission);
        result = vadd(result, vmul(throughput, emitted));
        if (m->emission > 0.0f) break;

        v3 negSun = vneg(ld3(pc->sunDir));
        float nDotL = fmax_glsl(vdot(rec.normal, negSun), 0.0f);
        if (nDotL > 0.0f) {
            ray_t shadow;
            shadow.origin = vadd(rec.point, vscale(rec.normal, 0.01f));
            shadow.direction = negSun;
            hit_t srec;
            memset(&srec, 0, sizeof(srec));
            int blocked = get_hit(b, pc, &shadow, 0.001f, 1e20f, &srec, cnt);
            if (!blocked) {
                v3 sun = vscale(ld3(pc->sunColour), pc->sunStrength);
                const float INV_PI = 0.31830988618f;
                v3 direct = vscale(sun, nDotL * INV_PI);
                if (m->refractionIndex == 0.0f) {
                    result = vadd(result, vmul(vmul(throughput, ld3(m->colour)), direct));
                } else {
                    float fresnel = reflectance(nDotL, m->refractionIndex);
                    result = vadd(result, vmul(vscale(throughput, fresnel), direct));
                }
            }
        }
        ray_t scattered;
        v3 attenuation;
        material_scatter(b, &ray, &rec, &attenuation, &scattered, seed);
        throughput = vmul(throughput, attenuation);
        ray = scattered;
    }
    return result;
}

static inline float linear_to_gamma(float c) { return c > 0.0f ? sqrtf(c) : 0.0f; }

/* raytrace.comp:857-875 with Pathtrace/RenderBoundingBoxes/RenderDepth inlined.
 * accum/out are W*H*4 floats, row 0 = top. tri/node/rays (optional) receive the
 * per-pixel counters. */
static void shade_pixel(const orc_buffers* b, const orc_pc* pc, int W, int H, int x, int y,
                        float* accum, float* out, int32_t* triOut, int32_t* nodeOut, int32_t* rayOut)
{
    (void)H;
    size_t idx = ((size_t)y * (size_t)W + (size_t)x);
    float raysPerPixelRatio = 1.0f / (float)pc->raysPerPixel;
    float accumulationMultiplier = pc->accumulateFrames == 1 ? 1.0f / (float)(pc->frame + 1) : 1.0f;
    counters_t cnt = { 0, 0, 0 };
    int mode = pc->renderMode;

    if (mode == 0 || mode == 3 || mode == 4) {
        float nr = 0.0f, ng = 0.0f, nb = 0.0f, na = 1.0f;
        float ar = accum[4 * idx + 0], ag = accum[4 * idx + 1], ab = accum[4 * idx + 2], aa = accum[4 * idx + 3];
        for (int i = 0; i < pc->raysPerPixel; ++i) {
            uint32_t seed = orc_seed_from_coords((uint32_t)x, (uint32_t)y, (uint32_t)pc->frame, (uint32_t)i);
            ray_t ray = get_ray(pc, x, y, seed);
            v3 c = ray_colour(b, pc, ray, seed, &cnt);
            nr += c.x; ng += c.y; nb += c.z;
        }
        if (mode == 0) {
            nr *= raysPerPixelRatio; ng *= raysPerPixelRatio; nb *= raysPerPixelRatio;
            ar += nr; ag += ng; ab += nb;
            accum[4 * idx + 0] = ar; accum[4 * idx + 1] = ag; accum[4 * idx + 2] = ab; accum[4 * idx + 3] = aa;
            float vr, vg, vb, va;
            if (pc->accumulateFrames == 1) { vr = ar; vg = ag; vb = ab; va = aa; }
            else { vr = nr; vg = ng; vb = nb; va = na; }
            vr *= accumulationMultiplier; vg *= accumulationMultiplier; vb *= accumulationMultiplier;
            out[4 * idx + 0] = linear_to_gamma(vr);
            out[4 * idx + 1] = linear_to_gamma(vg);
            out[4 * idx + 2] = linear_to_gamma(vb);
            out[4 * idx + 3] = va;
        } else {
            int threshold = mode == 3 ? pc->triangleTestThreshold : pc->bvhNodeTestThreshold;
            int target = mode == 3 ? (int)cnt.triTests : (int)cnt.nodeTests;
            int grey = (int)((float)target / (float)threshold * 255.0f);
            grey = grey < 0 ? 0 : (grey > 255 ? 255 : grey);
            float g = (float)grey / 255.0f;
            float r_ = g, g_ = g, b_ = g;
            if ((int)cnt.triTests > threshold) { r_ = 1.0f; g_ = 0.0f; b_ = 0.0f; }
            out[4 * idx + 0] = r_; out[4 * idx + 1] = g_; out[4 * idx + 2] = b_; out[4 * idx + 3] = 1.0f;
        }
    } else if (mode == 1) {
        ray_t ray = get_ray(pc, x, y, 123u);
        float r_ = 0.0f;
        for (int i = 0; i < pc->parentBVHCount; ++i) {
            float tmin = 0.001f, tmax = 1e20f;
            if (intersect_aabb(&ray, ld3(b->parents[i].node.aabb.mn), ld3(b->parents[i].node.aabb.mx), &tmin, &tmax)) {
                r_ = 1.0f;
                break;
            }
        }
        out[4 * idx + 0] = r_; out[4 * idx + 1] = 0.0f; out[4 * idx + 2] = 0.0f; out[4 * idx + 3] = 1.0f;
    } else if (mode == 2) {
        ray_t ray = get_ray(pc, x, y, 123u);
        hit_t rec;
        memset(&rec, 0, sizeof(rec));
        float d = 0.0f;
        if (get_hit(b, pc, &ray, 0.001f, 1e20f, &rec, &cnt)) {
            d = vlength(vsub(rec.point, ld3(pc->camPos)));
            d = clamp_glsl(d / pc->depthDebugScale, 0.0f, 1.0f);
        }
        out[4 * idx + 0] = d; out[4 * idx + 1] = d; out[4 * idx + 2] = d; out[4 * idx + 3] = 1.0f;
    }
    if (triOut) triOut[idx] = (int32_t)cnt.triTests;
    if (nodeOut) nodeOut[idx] = (int32_t)cnt.nodeTests;
    if (rayOut) rayOut[idx] = (int32_t)cnt.rays;
}

typedef struct {
    const orc_buffers* b; const orc_pc* pc; int W, H, x0, y0, x1, y1;
    float *accum, *out; int32_t *tri, *node, *rays;
    volatile int* nextRow;
} job_t;

static void* worker(void* arg)
{
    job_t* j = (job_t*)arg;
    for (;;) {
        int y = __sync_fetch_and_add(j->nextRow, 1);
        if (y >= j->y1) break;
        for (int x = j->x0; x < j->x1; ++x)
            shade_pixel(j->b, j->pc, j->W, j->H, x, y, j->accum, j->out, j->tri, j->node, j->rays);
    }
    return NULL;
}

/* One dispatch over the window [x0,x1) x [y0,y1) of a W x H image. The 18
 * buffers arrive as an array of 18 pointers in binding order (UV slots unused). */
void orc_render(const void* const* bindings, const void* pc144, int W, int H,
                int x0, int y0, int x1, int y1, float* accum, float* out,
                int32_t* triTests, int32_t* nodeTests, int32_t* rays, int nthreads)
{
    orc_buffers b;
    b.parents = (const orc_parent*)bindings[0];
    b.objects = (const orc_object*)bindings[1];
    b.materials = (const orc_material*)bindings[2];
    b.v0 = (const float*)bindings[3]; b.v1 = (const float*)bindings[4]; b.v2 = (const float*)bindings[5];
    b.n0 = (const float*)bindings[6]; b.n1 = (const float*)bindings[7]; b.n2 = (const float*)bindings[8];
    b.aabbMins = (const float*)bindings[12]; b.aabbMaxs = (const float*)bindings[13];
    b.left = (const int32_t*)bindings[14]; b.right = (const int32_t*)bindings[15];
    b.triStart = (const int32_t*)bindings[16]; b.triCount = (const int32_t*)bindings[17];
    if (x0 < 0) x0 = 0;
    if (y0 < 0) y0 = 0;
    if (x1 > W) x1 = W;
    if (y1 > H) y1 = H;
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    volatile int next = y0;
    job_t j = { &b, (const orc_pc*)pc144, W, H, x0, y0, x1, y1, accum, out, triTests, nodeTests, rays, &next };
    pthread_t th[256];
    for (int i = 1; i < nthreads; ++i) pthread_create(&th[i], NULL, worker, &j);
    worker(&j);
    for (int i = 1; i < nthreads; ++i) pthread_join(th[i], NULL);
}

/* ------------------------------------------------------------------------- */
/* host side: camera (HardwareRenderer.cpp:793-824)                           */
/* ------------------------------------------------------------------------- */

void orc_camera_push_constants(const float camPos[3], const float look[3], float fov, float focusDistance,
                               float defocusAngle, int W, int H, void* pc144)
{
    orc_pc* pc = (orc_pc*)pc144;
    float theta = fov * 0.01745329251994329576923690768489f; /* glm::radians(float) */
    double h = tan((double)theta / 2.0);
    float viewportHeight = (float)(2 * h * (double)focusDistance);
    float viewportWidth = (float)((double)viewportHeight * ((double)W / (double)H));

    v3 up = V(0.0f, 1.0f, 0.0f);
    v3 w = vnormalize(vneg(ld3(look)));
    v3 u = vnormalize(vcross(up, w));
    v3 v = vcross(w, u);

    v3 viewportU = vscale(u, viewportWidth);
    v3 viewportV = vscale(vneg(v), viewportHeight);
    float fw = (float)W, fh = (float)H;
    v3 pixelDeltaU = V(viewportU.x / fw, viewportU.y / fw, viewportU.z / fw);
    v3 pixelDeltaV = V(viewportV.x / fh, viewportV.y / fh, viewportV.z / fh);

    v3 pos = ld3(camPos);
    v3 halfU = V(viewportU.x / 2.0f, viewportU.y / 2.0f, viewportU.z / 2.0f);
    v3 halfV = V(viewportV.x / 2.0f, viewportV.y / 2.0f, viewportV.z / 2.0f);
    v3 upperLeft = vsub(vsub(vsub(pos, vscale(w, focusDistance)), halfU), halfV);
    v3 pixel00 = vadd(upperLeft, vscale(vadd(pixelDeltaU, pixelDeltaV), 0.5f));

    double defocusRadius = (double)focusDistance * tan((double)defocusAngle / 2.0 * 0.01745329251994329576923690768489);
    float dr = (float)defocusRadius;

    pc->pixel00[0] = pixel00.x; pc->pixel00[1] = pixel00.y; pc->pixel00[2] = pixel00.z;
    pc->deltaU[0] = pixelDeltaU.x; pc->deltaU[1] = pixelDeltaU.y; pc->deltaU[2] = pixelDeltaU.z;
    pc->deltaV[0] = pixelDeltaV.x; pc->deltaV[1] = pixelDeltaV.y; pc->deltaV[2] = pixelDeltaV.z;
    pc->camPos[0] = pos.x; pc->camPos[1] = pos.y; pc->camPos[2] = pos.z;
    pc->defocusAngle = defocusAngle;
    pc->defocusU[0] = dr * u.x; pc->defocusU[1] = dr * u.y; pc->defocusU[2] = dr * u.z;
    pc->defocusV[0] = dr * v.x; pc->defocusV[1] = dr * v.y; pc->defocusV[2] = dr * v.z;
}

/* ------------------------------------------------------------------------- */
/* host side: OBJ reader (tiny_obj_loader.h semantics, triangles and quads)    */
/* ------------------------------------------------------------------------- */

typedef struct { float* d; size_t n, cap; } fvec;
typedef struct { int32_t* d; size_t n, cap; } ivec;
static void fpush(fvec* v, float x)
{
    if (v->n == v->cap) { v->cap = v->cap ? v->cap * 2 : 256; v->d = (float*)realloc(v->d, v->cap * sizeof(float)); }
    v->d[v->n++] = x;
}
static void ipush(ivec* v, int32_t x)
{
    if (v->n == v->cap) { v->cap = v->cap ? v->cap * 2 : 256; v->d = (int32_t*)realloc(v->d, v->cap * sizeof(int32_t)); }
    v->d[v->n++] = x;
}

/* tiny_obj_loader.h:891-1021 (tryParseDouble).
 * The OBJ number parser and the face triangulation below follow tinyobjloader v2.0.0 statement by
 * statement (the byte-exact vertex stream depends on it), which is
 *   Copyright (c) 2012-Present, Syoyo Fujita and many contributors, released under the MIT License:
 *   "Permission is hereby granted, free of charge, to any person obtaining a copy of this software and
 *   associated documentation files (the "Software"), to deal in the Software without restriction,
 *   including without limitation the rights to use, copy, modify, merge, publish, distribute,
 *   sublicense, and/or sell copies of the Software, and to permit persons to whom the Software is
 *   furnished to do so, subject to the following conditions: The above copyright notice and this
 *   permission notice shall be included in all copies or substantial portions of the Software.
 *   THE SOFTWARE IS PROVIDED "AS IS", WITHOUT WARRANTY OF ANY KIND, EXPRESS OR IMPLIED, INCLUDING BUT
 *   NOT LIMITED TO THE WARRANTIES OF MERCHANTABILITY, FITNESS FOR A PARTICULAR PURPOSE AND
 *   NONINFRINGEMENT. IN NO EVENT SHALL THE AUTHORS OR COPYRIGHT HOLDERS BE LIABLE FOR ANY CLAIM, DAMAGES
 *   OR OTHER LIABILITY, WHETHER IN AN ACTION OF CONTRACT, TORT OR OTHERWISE, ARISING FROM, OUT OF OR IN
 *   CONNECTION WITH THE SOFTWARE OR THE USE OR OTHER DEALINGS IN THE SOFTWARE." */
static int obj_parse_double(const char* s, const char* s_end, double* result)
{
    if (s >= s_end) return 0;
    double mantissa = 0.0;
    int exponent = 0;
    char sign = '+', exp_sign = '+';
    const char* curr = s;
    int read = 0;
    int end_not_reached = 0;
    int leading_dot = 0;

    if (*curr == '+' || *curr == '-') {
        sign = *curr; curr++;
        if (curr != s_end && *curr == '.') leading_dot = 1;
    } else if (isdigit((unsigned char)*curr)) {
    } else if (*curr == '.') {
        leading_dot = 1;
    } else {
        return 0;
    }
    end_not_reached = (curr != s_end);
    if (!leading_dot) {
        while (end_not_reached && isdigit((unsigned char)*curr)) {
            mantissa *= 10;
            mantissa += (int)(*curr - 0x30);
            curr++; read++;
            end_not_reached = (curr != s_end);
        }
        if (read == 0) return 0;
    }
    if (!end_not_reached) goto assemble;
    if (*curr == '.') {
        static const double pow_lut[] = { 1.0, 0.1, 0.01, 0.001, 0.0001, 0.00001, 0.000001, 0.0000001 };
        curr++; read = 1;
        end_not_reached = (curr != s_end);
        while (end_not_reached && isdigit((unsigned char)*curr)) {
            mantissa += (int)(*curr - 0x30) * (read < 8 ? pow_lut[read] : pow(10.0, -read));
            read++; curr++;
            end_not_reached = (curr != s_end);
        }
    } else if (*curr == 'e' || *curr == 'E') {
    } else {
        goto assemble;
    }
    if (!end_not_reached) goto assemble;
    if (*curr == 'e' || *curr == 'E') {
        curr++;
        end_not_reached = (curr != s_end);
        if (end_not_reached && (*curr == '+' || *curr == '-')) { exp_sign = *curr; curr++; }
        else if (isdigit((unsigned char)*curr)) { }
        else return 0;
        read = 0;
        end_not_reached = (curr != s_end);
        while (end_not_reached && isdigit((unsigned char)*curr)) {
            if (exponent > (2147483647 / 10)) return 0;
            exponent *= 10;
            exponent += (int)(*curr - 0x30);
            curr++; read++;
            end_not_reached = (curr != s_end);
        }
        exponent *= (exp_sign == '+' ? 1 : -1);
        if (read == 0) return 0;
    }
assemble:
    *result = (sign == '+' ? 1 : -1) * (exponent ? ldexp(mantissa * pow(5.0, exponent), exponent) : mantissa);
    return 1;
}

/* tiny_obj_loader.h:1023-1031 (parseReal with default 0) */
static float obj_parse_real(const char** token)
{
    *token += strspn(*token, " \t");
    const char* end = *token + strcspn(*token, " \t\r");
    double val = 0.0;
    obj_parse_double(*token, end, &val);
    *token = end;
    return (float)val;
}

/* tiny_obj_loader.h:816-846 */
static int obj_fix_index(int idx, int n, int* ret, int allow_zero)
{
    if (idx > 0) { *ret = idx - 1; return 1; }
    if (idx == 0) { *ret = idx - 1; return allow_zero; }
    *ret = n + idx;
    return *ret >= 0;
}

/* tiny_obj_loader.h:1161-1212 */
static int obj_parse_triple(const char** token, int vsize, int vnsize, int vtsize, int* vi, int* vti, int* vni)
{
    *vi = -1; *vti = -1; *vni = -1;
    if (!obj_fix_index(atoi(*token), vsize, vi, 0)) return 0;
    *token += strcspn(*token, "/ \t\r");
    if ((*token)[0] != '/') return 1;
    (*token)++;
    if ((*token)[0] == '/') {
        (*token)++;
        if (!obj_fix_index(atoi(*token), vnsize, vni, 1)) return 0;
        *token += strcspn(*token, "/ \t\r");
        return 1;
    }
    if (!obj_fix_index(atoi(*token), vtsize, vti, 1)) return 0;
    *token += strcspn(*token, "/ \t\r");
    if ((*token)[0] != '/') return 1;
    (*token)++;
    if (!obj_fix_index(atoi(*token), vnsize, vni, 1)) return 0;
    *token += strcspn(*token, "/ \t\r");
    return 1;
}

/* Reads an OBJ into the flat per-corner stream LoadObjFile produces
 * (ModelLoader.cpp:292-330): pos xyz + normal xyz per corner, 3 corners per
 * triangle, in file order; quads are split by tinyobj's shorter-diagonal rule
 * (tiny_obj_loader.h:1488-1584). Faces with more than 4 corners are rejected
 * (tinyobj's ear clipping is not restated). Returns corner count or -1. */
static long obj_load(const char* path, fvec* outPos, fvec* outNrm)
{
    FILE* f = fopen(path, "rb");
    if (!f) return -1;
    fvec v = { 0 }, vn = { 0 };
    size_t vtCount = 0;
    ivec fv = { 0 }, fn = { 0 }; /* corner indices after triangulation */
    char* line = NULL;
    size_t cap = 0;
    ssize_t len;
    long status = 0;
    while ((len = getline(&line, &cap, f)) != -1) {
        while (len > 0 && (line[len - 1] == '\n' || line[len - 1] == '\r')) line[--len] = 0;
        const char* token = line;
        token += strspn(token, " \t");
        if (token[0] == 0 || token[0] == '#') continue;
        if (token[0] == 'v' && (token[1] == ' ' || token[1] == '\t')) {
            token += 2;
            float x = obj_parse_real(&token), y = obj_parse_real(&token), z = obj_parse_real(&token);
            fpush(&v, x); fpush(&v, y); fpush(&v, z);
            continue;
        }
        if (token[0] == 'v' && token[1] == 'n' && (token[2] == ' ' || token[2] == '\t')) {
            token += 3;
            float x = obj_parse_real(&token), y = obj_parse_real(&token), z = obj_parse_real(&token);
            fpush(&vn, x); fpush(&vn, y); fpush(&vn, z);
            continue;
        }
        if (token[0] == 'v' && token[1] == 't' && (token[2] == ' ' || token[2] == '\t')) { vtCount++; continue; }
        if (token[0] == 'f' && (token[1] == ' ' || token[1] == '\t')) {
            token += 2;
            token += strspn(token, " \t");
            int cv[8], cn[8], nc = 0;
            while (!(token[0] == '\r' || token[0] == '\n' || token[0] == 0)) {
                int vi, vti, vni;
                if (!obj_parse_triple(&token, (int)(v.n / 3), (int)(vn.n / 3), (int)vtCount, &vi, &vti, &vni)) { status = -1; goto done; }
                if (nc < 8) { cv[nc] = vi; cn[nc] = vni; }
                nc++;
                token += strspn(token, " \t\r");
            }
            if (nc < 3) continue;
            if (nc == 3) {
                for (int k = 0; k < 3; ++k) { ipush(&fv, cv[k]); ipush(&fn, cn[k]); }
            } else if (nc == 4) {
                int bad = 0;
                for (int k = 0; k < 4; ++k) if ((size_t)(3 * (size_t)cv[k] + 2) >= v.n) bad = 1;
                if (bad) continue;
                const float* p0 = v.d + 3 * (size_t)cv[0]; const float* p1 = v.d + 3 * (size_t)cv[1];
                const float* p2 = v.d + 3 * (size_t)cv[2]; const float* p3 = v.d + 3 * (size_t)cv[3];
                float e02x = p2[0] - p0[0], e02y = p2[1] - p0[1], e02z = p2[2] - p0[2];
                float e13x = p3[0] - p1[0], e13y = p3[1] - p1[1], e13z = p3[2] - p1[2];
                float sqr02 = e02x * e02x + e02y * e02y + e02z * e02z;
                float sqr13 = e13x * e13x + e13y * e13y + e13z * e13z;
                static const int A[6] = { 0, 1, 2, 0, 2, 3 }, B[6] = { 0, 1, 3, 1, 2, 3 };
                const int* order = sqr02 < sqr13 ? A : B;
                for (int k = 0; k < 6; ++k) { ipush(&fv, cv[order[k]]); ipush(&fn, cn[order[k]]); }
            } else {
                status = -2; goto done;
            }
            continue;
        }
        /* g / o / s / usemtl / mtllib / vt... do not change the emitted order */
    }
    for (size_t i = 0; i < fv.n; ++i) {
        int vi = fv.d[i], ni = fn.d[i];
        if (vi < 0 || (size_t)(3 * (size_t)vi + 2) >= v.n || (ni >= 0 && (size_t)(3 * (size_t)ni + 2) >= vn.n)) { status = -3; goto done; }
        fpush(outPos, v.d[3 * (size_t)vi + 0]); fpush(outPos, v.d[3 * (size_t)vi + 1]); fpush(outPos, v.d[3 * (size_t)vi + 2]);
        if (vn.n != 0 && ni >= 0) {
            fpush(outNrm, vn.d[3 * (size_t)ni + 0]); fpush(outNrm, vn.d[3 * (size_t)ni + 1]); fpush(outNrm, vn.d[3 * (size_t)ni + 2]);
        } else {
            fpush(outNrm, 0.0f); fpush(outNrm, 0.0f); fpush(outNrm, 0.0f);
        }
    }
    status = (long)fv.n;
done:
    free(line); free(v.d); free(vn.d); free(fv.d); free(fn.d);
    fclose(f);
    return status;
}

/* exported for the parser cross-check against the reference's own tinyobj */
long orc_obj_corners(const char* path, float** pos, float** nrm)
{
    fvec p = { 0 }, n = { 0 };
    long c = obj_load(path, &p, &n);
    if (c < 0) { free(p.d); free(n.d); return c; }
    *pos = p.d; *nrm = n.d;
    return c;
}
void orc_free(void* p) { free(p); }

/* ------------------------------------------------------------------------- */
/* host side: LoadModel + BVH (HardwareRenderer.cpp:1157-1314, ModelLoader.cpp)*/
/* ------------------------------------------------------------------------- */

typedef struct { v3 v0, v1, v2, n0, n1, n2, centroid; } tri_t;
typedef struct { v3 mn, mx; int left, right, triStart, triCount; } bnode_t;
typedef struct { bnode_t* d; size_t n, cap; } nvec;

static size_t npush(nvec* v, bnode_t x)
{
    if (v->n == v->cap) { v->cap = v->cap ? v->cap * 2 : 256; v->d = (bnode_t*)realloc(v->d, v->cap * sizeof(bnode_t)); }
    v->d[v->n] = x;
    return v->n++;
}

/* RaytracerTypes.h:61-70 */
static int longest_axis(v3 mn, v3 mx)
{
    v3 e = vsub(mx, mn);
    if (e.x > e.y && e.x > e.z) return 0;
    else if (e.y > e.z) return 1;
    return 2;
}

/* ModelLoader.cpp:262-290 */
static void expand_node_aabb(const tri_t* tris, bnode_t* n)
{
    n->mn = V(FLT_MAX, FLT_MAX, FLT_MAX);
    n->mx = V(-FLT_MAX, -FLT_MAX, -FLT_MAX);
    for (int i = 0; i < n->triCount; ++i) {
        const tri_t* t = &tris[n->triStart + i];
        const v3* vs[3] = { &t->v0, &t->v1, &t->v2 };
        for (int k = 0; k < 3; ++k) {
            if (vs[k]->x < n->mn.x) n->mn.x = vs[k]->x;
            if (vs[k]->y < n->mn.y) n->mn.y = vs[k]->y;
            if (vs[k]->z < n->mn.z) n->mn.z = vs[k]->z;
            if (vs[k]->x > n->mx.x) n->mx.x = vs[k]->x;
            if (vs[k]->y > n->mx.y) n->mx.y = vs[k]->y;
            if (vs[k]->z > n->mx.z) n->mx.z = vs[k]->z;
        }
    }
}

/* the swap-to-back partition shared by ModelLoader.cpp:19-32 and :78-91 */
static int partition_tris(tri_t* tris, int first, int count, int axis, float splitPos)
{
    int i = first, j = first + count - 1;
    while (i <= j) {
        if (comp(tris[i].centroid, axis) < splitPos) {
            i++;
        } else {
            tri_t tmp = tris[i]; tris[i] = tris[j]; tris[j] = tmp;
            j--;
        }
    }
    return i;
}

/* ModelLoader.cpp:64-120 */
static void split_bvh_node(tri_t* tris, nvec* nodes, size_t idx)
{
    nodes->d[idx].left = -1;
    nodes->d[idx].right = -1;
    bnode_t node = nodes->d[idx];
    if (node.triCount <= 2) return;
    int axis = longest_axis(node.mn, node.mx);
    float splitPos = (comp(node.mn, axis) + comp(node.mx, axis)) * 0.5f;
    int i = partition_tris(tris, node.triStart, node.triCount, axis, splitPos);
    int leftCount = i - node.triStart;
    if (leftCount == 0 || leftCount == node.triCount) return;

    size_t li = nodes->n, ri = nodes->n + 1;
    nodes->d[idx].left = (int)li;
    nodes->d[idx].right = (int)ri;
    bnode_t l, r;
    l.triStart = node.triStart; l.triCount = leftCount; l.left = l.right = -1;
    r.triStart = i; r.triCount = node.triCount - leftCount; r.left = r.right = -1;
    expand_node_aabb(tris, &l);
    expand_node_aabb(tris, &r);
    npush(nodes, l);
    npush(nodes, r);
    split_bvh_node(tris, nodes, li);
    split_bvh_node(tris, nodes, ri);
}

typedef struct {
    char* path;
    int triStart, triCount;
    orc_parent parent; /* model-space parent with global child / triangle indices */
    int nodeStart, nodeCount;
} model_t;

typedef struct { char* model; float pos[3], rot[3], scale[3]; int materialIndex; } sobject_t;

typedef struct orc_scene {
    fvec v0, v1, v2, n0, n1, n2, uv; /* vec4 arrays (uv: zeros) */
    fvec aabbMins, aabbMaxs;
    ivec left, right, triStart, triCount;
    model_t* models; size_t nModels;
    sobject_t* objects; size_t nObjects;
    orc_material* materials; size_t nMaterials;
    orc_parent* parents; orc_object* gpuObjects; size_t nParents;
} orc_scene;

orc_scene* orc_scene_new(void) { return (orc_scene*)calloc(1, sizeof(orc_scene)); }

void orc_scene_free(orc_scene* s)
{
    if (!s) return;
    free(s->v0.d); free(s->v1.d); free(s->v2.d); free(s->n0.d); free(s->n1.d); free(s->n2.d); free(s->uv.d);
    free(s->aabbMins.d); free(s->aabbMaxs.d); free(s->left.d); free(s->right.d); free(s->triStart.d); free(s->triCount.d);
    for (size_t i = 0; i < s->nModels; ++i) free(s->models[i].path);
    for (size_t i = 0; i < s->nObjects; ++i) free(s->objects[i].model);
    free(s->models); free(s->objects); free(s->materials); free(s->parents); free(s->gpuObjects);
    free(s);
}

static model_t* find_model(orc_scene* s, const char* path)
{
    for (size_t i = 0; i < s->nModels; ++i) if (strcmp(s->models[i].path, path) == 0) return &s->models[i];
    return NULL;
}

static void push4(fvec* v, v3 a) { fpush(v, a.x); fpush(v, a.y); fpush(v, a.z); fpush(v, 0.0f); }

/* HardwareRenderer.cpp:1157-1314 */
int orc_load_model(orc_scene* s, const char* path)
{
    if (find_model(s, path)) return 0;
    fvec pos = { 0 }, nrm = { 0 };
    long corners = obj_load(path, &pos, &nrm);
    if (corners < 0) { free(pos.d); free(nrm.d); return (int)corners; }

    int triCount = (int)(corners / 3);
    tri_t* tris = (tri_t*)malloc(sizeof(tri_t) * (size_t)(triCount > 0 ? triCount : 1));
    float minX = (float)INT_MAX, minY = (float)INT_MAX, minZ = (float)INT_MAX;
    float maxX = (float)INT_MIN, maxY = (float)INT_MIN, maxZ = (float)INT_MIN;
    for (int t = 0; t < triCount; ++t) {
        tri_t T;
        const float* p = pos.d + 9 * (size_t)t;
        const float* n = nrm.d + 9 * (size_t)t;
        T.v0 = ld3(p); T.v1 = ld3(p + 3); T.v2 = ld3(p + 6);
        T.centroid = vscale(vadd(vadd(T.v0, T.v1), T.v2), 0.33f);
        v3 a = ld3(n), b = ld3(n + 3), c = ld3(n + 6);
        int nzA = !(a.x == 0.0f && a.y == 0.0f && a.z == 0.0f);
        int nzB = !(b.x == 0.0f && b.y == 0.0f && b.z == 0.0f);
        int nzC = !(c.x == 0.0f && c.y == 0.0f && c.z == 0.0f);
        if (nzA && nzB && nzC) {
            T.n0 = vnormalize(a); T.n1 = vnormalize(b); T.n2 = vnormalize(c);
        } else {
            v3 e1 = vsub(T.v1, T.v0), e2 = vsub(T.v2, T.v0);
            v3 fnrm = vnormalize(vcross(e1, e2));
            T.n0 = fnrm; T.n1 = fnrm; T.n2 = fnrm;
        }
        const v3* vs[3] = { &T.v0, &T.v1, &T.v2 };
        for (int k = 0; k < 3; ++k) {
            if (vs[k]->x < minX) minX = (float)(int)floorf(vs[k]->x);
            if (vs[k]->y < minY) minY = (float)(int)floorf(vs[k]->y);
            if (vs[k]->z < minZ) minZ = (float)(int)floorf(vs[k]->z);
            if (vs[k]->x > maxX) maxX = (float)(int)ceilf(vs[k]->x);
            if (vs[k]->y > maxY) maxY = (float)(int)ceilf(vs[k]->y);
            if (vs[k]->z > maxZ) maxZ = (float)(int)ceilf(vs[k]->z);
        }
        tris[t] = T;
    }
    free(pos.d); free(nrm.d);
    if (maxX - minX == 0.0f) { maxX += 1.0f; minX -= 1.0f; }
    if (maxY - minY == 0.0f) { maxY += 1.0f; minY -= 1.0f; }
    if (maxZ - minZ == 0.0f) { maxZ += 1.0f; minZ -= 1.0f; }

    orc_parent parent;
    memset(&parent, 0, sizeof(parent));
    parent.node.aabb.mn[0] = minX; parent.node.aabb.mn[1] = minY; parent.node.aabb.mn[2] = minZ;
    parent.node.aabb.mx[0] = maxX; parent.node.aabb.mx[1] = maxY; parent.node.aabb.mx[2] = maxZ;
    parent.objectIndex = -1;
    parent.node.left = -1; parent.node.right = -1;
    parent.node.triStart = 0; parent.node.triCount = triCount;

    int bvhGlobalOffset = (int)(s->aabbMins.n / 4);
    int triGlobalOffset = (int)(s->v0.n / 4);
    nvec nodes = { 0 };

    /* ModelLoader.cpp:10-62 (BuildBVH) */
    if (triCount > 2) {
        v3 pmn = ld3(parent.node.aabb.mn), pmx = ld3(parent.node.aabb.mx);
        int axis = longest_axis(pmn, pmx);
        float splitPos = (comp(pmn, axis) + comp(pmx, axis)) * 0.5f;
        int leftCount = partition_tris(tris, 0, triCount, axis, splitPos);
        if (!(leftCount == 0 || leftCount == triCount)) {
            bnode_t l, r;
            l.triStart = 0; l.triCount = leftCount; l.left = l.right = -1;
            r.triStart = leftCount; r.triCount = triCount - leftCount; r.left = r.right = -1;
            expand_node_aabb(tris, &l);
            expand_node_aabb(tris, &r);
            npush(&nodes, l);
            npush(&nodes, r);
            split_bvh_node(tris, &nodes, 0);
            split_bvh_node(tris, &nodes, 1);
            parent.node.left = bvhGlobalOffset;
            parent.node.right = bvhGlobalOffset + 1;
        }
    }

    for (size_t i = 0; i < nodes.n; ++i) {
        bnode_t* n = &nodes.d[i];
        int l = n->left != -1 ? n->left + bvhGlobalOffset : -1;
        int r = n->right != -1 ? n->right + bvhGlobalOffset : -1;
        push4(&s->aabbMins, n->mn);
        push4(&s->aabbMaxs, n->mx);
        ipush(&s->left, l); ipush(&s->right, r);
        ipush(&s->triStart, n->triStart + triGlobalOffset);
        ipush(&s->triCount, n->triCount);
    }
    parent.node.triStart += triGlobalOffset;
    for (int t = 0; t < triCount; ++t) {
        push4(&s->v0, tris[t].v0); push4(&s->v1, tris[t].v1); push4(&s->v2, tris[t].v2);
        push4(&s->n0, tris[t].n0); push4(&s->n1, tris[t].n1); push4(&s->n2, tris[t].n2);
        push4(&s->uv, V(0.0f, 0.0f, 0.0f));
    }
    s->models = (model_t*)realloc(s->models, sizeof(model_t) * (s->nModels + 1));
    model_t* m = &s->models[s->nModels++];
    m->path = strdup(path);
    m->triStart = triGlobalOffset; m->triCount = triCount;
    m->parent = parent;
    m->nodeStart = bvhGlobalOffset; m->nodeCount = (int)nodes.n;
    free(nodes.d); free(tris);
    return 0;
}

int orc_add_material(orc_scene* s, const void* material48)
{
    s->materials = (orc_material*)realloc(s->materials, sizeof(orc_material) * (s->nMaterials + 1));
    memcpy(&s->materials[s->nMaterials], material48, sizeof(orc_material));
    return (int)s->nMaterials++;
}

/* HardwareRenderer.cpp:490-500 with the material already resolved to an index */
int orc_add_object(orc_scene* s, const char* model, const float pos[3], const float rot[3], const float scale[3], int materialIndex)
{
    s->objects = (sobject_t*)realloc(s->objects, sizeof(sobject_t) * (s->nObjects + 1));
    sobject_t* o = &s->objects[s->nObjects];
    o->model = strdup(model);
    memcpy(o->pos, pos, 12); memcpy(o->rot, rot, 12); memcpy(o->scale, scale, 12);
    o->materialIndex = materialIndex;
    return (int)s->nObjects++;
}

static m4 m4_identity(void)
{
    m4 m; memset(&m, 0, sizeof(m));
    m.c[0][0] = m.c[1][1] = m.c[2][2] = m.c[3][3] = 1.0f;
    return m;
}

/* glm::translate: Result[3] = m[0]*v[0] + m[1]*v[1] + m[2]*v[2] + m[3] */
static m4 glm_translate(m4 m, const float v[3])
{
    m4 r = m;
    for (int k = 0; k < 4; ++k) r.c[3][k] = m.c[0][k] * v[0] + m.c[1][k] * v[1] + m.c[2][k] * v[2] + m.c[3][k];
    return r;
}

/* glm::rotate(m, angle, axis) */
static m4 glm_rotate(m4 m, float angle, v3 ax)
{
    float c = cosf(angle), s = sinf(angle);
    v3 axis = vnormalize(ax);
    v3 temp = vscale(axis, 1.0f - c);
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
    m4 r;
    for (int col = 0; col < 3; ++col)
        for (int k = 0; k < 4; ++k)
            r.c[col][k] = m.c[0][k] * R[col][0] + m.c[1][k] * R[col][1] + m.c[2][k] * R[col][2];
    for (int k = 0; k < 4; ++k) r.c[3][k] = m.c[3][k];
    return r;
}

static m4 glm_scale(m4 m, const float v[3])
{
    m4 r;
    for (int k = 0; k < 4; ++k) {
        r.c[0][k] = m.c[0][k] * v[0];
        r.c[1][k] = m.c[1][k] * v[1];
        r.c[2][k] = m.c[2][k] * v[2];
        r.c[3][k] = m.c[3][k];
    }
    return r;
}

static inline float glm_radians(float deg) { return deg * 0.01745329251994329576923690768489f; }

/* HardwareRenderer.cpp:502-551 */
static int convert_scene_object(orc_scene* s, const sobject_t* o, size_t index)
{
    model_t* mdl = find_model(s, o->model);
    if (!mdl) return -1;
    m4 M = m4_identity();
    M = glm_translate(M, o->pos);
    M = glm_rotate(M, glm_radians(o->rot[0]), V(1.0f, 0.0f, 0.0f));
    M = glm_rotate(M, glm_radians(o->rot[1]), V(0.0f, 1.0f, 0.0f));
    M = glm_rotate(M, glm_radians(o->rot[2]), V(0.0f, 0.0f, 1.0f));
    M = glm_scale(M, o->scale);

    const float* mn = mdl->parent.node.aabb.mn;
    const float* mx = mdl->parent.node.aabb.mx;
    float corners[8][3] = {
        { mn[0], mn[1], mn[2] }, { mx[0], mn[1], mn[2] }, { mn[0], mx[1], mn[2] }, { mx[0], mx[1], mn[2] },
        { mn[0], mn[1], mx[2] }, { mx[0], mn[1], mx[2] }, { mn[0], mx[1], mx[2] }, { mx[0], mx[1], mx[2] } };
    float newMin[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, newMax[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
    for (int i = 0; i < 8; ++i) {
        float t[4];
        m4_mul_v4(&M, corners[i][0], corners[i][1], corners[i][2], 1.0f, t);
        for (int k = 0; k < 3; ++k) {
            newMin[k] = t[k] < newMin[k] ? t[k] : newMin[k]; /* glm::min(a,b) = b<a?b:a */
            newMax[k] = newMax[k] < t[k] ? t[k] : newMax[k]; /* glm::max(a,b) = a<b?b:a */
        }
    }
    orc_object g;
    memset(&g, 0, sizeof(g));
    m4 inv = m4_inverse(&M);
    memcpy(g.inv, &inv, sizeof(inv));
    g.materialIndex = o->materialIndex;
    g.triStart = mdl->triStart;
    g.triCount = mdl->triCount;

    orc_parent p = mdl->parent;
    for (int k = 0; k < 3; ++k) { p.node.aabb.mn[k] = newMin[k]; p.node.aabb.mx[k] = newMax[k]; }
    p.objectIndex = (int)index;
    s->gpuObjects[index] = g;
    s->parents[index] = p;
    return 0;
}

/* HardwareRenderer.cpp:653-657 */
int orc_buffer_scene_data(orc_scene* s)
{
    free(s->parents); free(s->gpuObjects);
    s->parents = (orc_parent*)calloc(s->nObjects ? s->nObjects : 1, sizeof(orc_parent));
    s->gpuObjects = (orc_object*)calloc(s->nObjects ? s->nObjects : 1, sizeof(orc_object));
    s->nParents = s->nObjects;
    for (size_t i = 0; i < s->nObjects; ++i)
        if (convert_scene_object(s, &s->objects[i], i) != 0) return -1;
    return 0;
}

/* binding-order accessor: data pointer + element count */
const void* orc_scene_binding(orc_scene* s, int binding, size_t* count)
{
    switch (binding) {
    case 0: *count = s->nParents; return s->parents;
    case 1: *count = s->nParents; return s->gpuObjects;
    case 2: *count = s->nMaterials; return s->materials;
    case 3: *count = s->v0.n / 4; return s->v0.d;
    case 4: *count = s->v1.n / 4; return s->v1.d;
    case 5: *count = s->v2.n / 4; return s->v2.d;
    case 6: *count = s->n0.n / 4; return s->n0.d;
    case 7: *count = s->n1.n / 4; return s->n1.d;
    case 8: *count = s->n2.n / 4; return s->n2.d;
    case 9: case 10: case 11: *count = s->uv.n / 4; return s->uv.d;
    case 12: *count = s->aabbMins.n / 4; return s->aabbMins.d;
    case 13: *count = s->aabbMaxs.n / 4; return s->aabbMaxs.d;
    case 14: *count = s->left.n; return s->left.d;
    case 15: *count = s->right.n; return s->right.d;
    case 16: *count = s->triStart.n; return s->triStart.d;
    case 17: *count = s->triCount.n; return s->triCount.d;
    default: *count = 0; return NULL;
    }
}

static int node_depth(const orc_scene* s, int node, int* maxLeaf)
{
    int l = s->left.d[node], r = s->right.d[node];
    if (l == -1 && r == -1) {
        if (s->triCount.d[node] > *maxLeaf) *maxLeaf = s->triCount.d[node];
        return 1;
    }
    int dl = l != -1 ? node_depth(s, l, maxLeaf) : 0;
    int dr = r != -1 ? node_depth(s, r, maxLeaf) : 0;
    return 1 + (dl > dr ? dl : dr);
}

/* triangles / child nodes / depth below the parent / largest leaf of one model */
int orc_model_info(orc_scene* s, const char* path, int* tris, int* nodes, int* depth, int* maxLeaf)
{
    model_t* m = find_model(s, path);
    if (!m) return -1;
    *tris = m->triCount; *nodes = m->nodeCount; *maxLeaf = 0; *depth = 0;
    if (m->parent.node.left == -1 && m->parent.node.right == -1) { *maxLeaf = m->triCount; return 0; }
    int dl = node_depth(s, m->parent.node.left, maxLeaf);
    int dr = node_depth(s, m->parent.node.right, maxLeaf);
    *depth = dl > dr ? dl : dr;
    return 0;
}

/* HardwareRenderer.cpp:574-649 */
int orc_default_scene(orc_scene* s, const char* modelsDir, const char* dragonFile)
{
    orc_material base;
    memset(&base, 0, sizeof(base));
    base.colour[0] = base.colour[1] = base.colour[2] = 1.0f; /* GPUMaterial defaults */
    orc_material m[9];
    for (int i = 0; i < 9; ++i) m[i] = base;
    m[0].colour[0] = m[0].colour[1] = m[0].colour[2] = 0.5f;                         /* ground */
    m[1].colour[0] = m[1].colour[1] = m[1].colour[2] = 0.8f;                         /* diffuse */
    m[2].refractionIndex = 1.5f;                                                     /* glass */
    m[3].smoothness = 1.0f;                                                          /* mirror */
    m[4].emission = 15.0f;                                                           /* emissive */
    m[5].colour[0] = 1.0f; m[5].colour[1] = 0.71f; m[5].colour[2] = 0.29f; m[5].smoothness = 0.8f; m[5].fuzziness = 0.4f;
    m[6].colour[0] = 1.0f; m[6].colour[1] = 0.71f; m[6].colour[2] = 0.29f; m[6].smoothness = 1.0f;
    m[7].colour[0] = 0.8f; m[7].colour[1] = 0.1f; m[7].colour[2] = 0.1f;
    m[8].colour[0] = 0.1f; m[8].colour[1] = 0.8f; m[8].colour[2] = 0.1f;
    for (int i = 0; i < 9; ++i) orc_add_material(s, &m[i]);

    char quad[1024], sphere[1024], dragon[1024];
    snprintf(quad, sizeof(quad), "%s/flat_quad.obj", modelsDir);
    snprintf(sphere, sizeof(sphere), "%s/icosphere.obj", modelsDir);
    snprintf(dragon, sizeof(dragon), "%s/%s", modelsDir, dragonFile);
    if (orc_load_model(s, quad) || orc_load_model(s, sphere) || orc_load_model(s, dragon)) return -1;

    struct { const char* model; float p[3], r[3], sc[3]; int mat; } objs[8] = {
        { sphere, { 0, 10, 0 }, { 0, 0, 0 }, { 2, 2, 2 }, 4 },
        { dragon, { 0, 2, 0 }, { 0, 195, 0 }, { 1, 1, 1 }, 5 },
        { quad, { 0, 0, 0 }, { 0, 0, 0 }, { 5, 5, 5 }, 0 },
        { quad, { 0, 10, 0 }, { 0, 0, 180 }, { 5, 5, 5 }, 0 },
        { quad, { 0, 5, -5 }, { 90, 0, 0 }, { 5, 5, 5 }, 0 },
        { quad, { -5, 5, 0 }, { 0, 0, 90 }, { 5, 5, 5 }, 7 },
        { quad, { 5, 5, 0 }, { 0, 0, -90 }, { 5, 5, 5 }, 8 },
        { quad, { 0, 5, 5 }, { -90, 0, 0 }, { 5, 5, 5 }, 0 } };
    for (int i = 0; i < 8; ++i) orc_add_object(s, objs[i].model, objs[i].p, objs[i].r, objs[i].sc, objs[i].mat);
    return orc_buffer_scene_data(s);
}
