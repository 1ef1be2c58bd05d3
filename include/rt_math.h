/*
 * rt_math.h - exp, exp2, log2 and acos in plain binary32 arithmetic.
 *
 * The reference shader takes these from GLSL (raytrace.comp:366 pow, :404 exp, :661 acos/pow/exp),
 * whose precision the specification leaves to the implementation. A host libm and a device libm
 * differ in the last bit now and then, and one such bit in a sky colour or a Fresnel term is all it
 * takes for two otherwise identical renderers to disagree. These versions use only IEEE-754
 * add / subtract / multiply / divide / sqrt / floor on floats and integer bit operations, in a
 * fixed order, so every compiler that does not contract multiply-add pairs (build with
 * -ffp-contract=off) and every machine gives the same bits. Accuracy: exp2 and exp within 2 ulp,
 * log2 within 3 ulp (relative to the result, away from x = 1), acos within 4 ulp on [-1, 1]
 * (tests/test_rt_math.py). Coefficients: scripts/gen_math_coeffs.py. rt_div10000f is the one
 * function here that asks for a fused multiply-add by name (fmaf); contraction stays off.
 *
 * Usable from C, C++ and HIP device code.
 */
#ifndef RT_MATH_H
#define RT_MATH_H

#include <math.h>
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__) || defined(__CUDACC__)
#define RT_MATH_FN static __host__ __device__ __forceinline__
#else
#define RT_MATH_FN static inline
#endif

RT_MATH_FN uint32_t rt_float_bits(float f) { uint32_t u; memcpy(&u, &f, sizeof u); return u; }
RT_MATH_FN float rt_bits_float(uint32_t u) { float f; memcpy(&f, &u, sizeof f); return f; }

/* x * 2^n for any integer n in [-300, 300], in two exactly representable factors so that
 * results in the subnormal range are rounded once. */
RT_MATH_FN float rt_scale2(float x, int n)
{
    const int a = n / 2, b = n - a;   /* |a|, |b| <= 150 would overflow one exponent field: clamp below */
    const int ea = a < -126 ? -126 : (a > 127 ? 127 : a);
    const int eb = b < -126 ? -126 : (b > 127 ? 127 : b);
    return (x * rt_bits_float((uint32_t)(ea + 127) << 23)) * rt_bits_float((uint32_t)(eb + 127) << 23);
}

RT_MATH_FN float rt_exp2f(float x)
{
    if (x != x) return x;
    if (x >= 128.0f) return INFINITY;
    if (x < -150.0f) return 0.0f;
    const float n = floorf(x + 0.5f);
    const float f = x - n;   /* exact, in [-0.5, 0.5] */
    float p = 1.546973217e-04f;
    p = p * f + 1.340043265e-03f;
    p = p * f + 9.618025273e-03f;
    p = p * f + 5.550327152e-02f;
    p = p * f + 2.402265072e-01f;
    p = p * f + 6.931471825e-01f;
    p = p * f + 1.0f;
    return rt_scale2(p, (int)n);
}

RT_MATH_FN float rt_expf(float x)
{
    if (x != x) return x;
    if (x > 88.8f) return INFINITY;
    if (x < -104.0f) return 0.0f;
    const float k = floorf(x * 1.44269504089f + 0.5f);
    /* ln 2 in two pieces; the first has nine significant bits, so k times it is exact */
    const float r = (x - k * 0.693359375f) - k * -2.12194440e-4f;
    float p = 1.394976978e-03f;
    p = p * r + 8.375961334e-03f;
    p = p * r + 4.166619852e-02f;
    p = p * r + 1.666640490e-01f;
    p = p * r + 0.5f;
    p = p * r + 1.0f;
    p = p * r + 1.0f;
    return rt_scale2(p, (int)k);
}

RT_MATH_FN float rt_log2f(float x)
{
    if (x != x) return x;
    if (x < 0.0f) return rt_bits_float(0x7fc00000u);
    if (x == 0.0f) return -INFINITY;
    if (x == INFINITY) return x;
    int e = 0;
    uint32_t u = rt_float_bits(x);
    if (u < 0x00800000u) { u = rt_float_bits(x * 8388608.0f); e = -23; }   /* subnormal */
    e += (int)(u >> 23) - 127;
    float m = rt_bits_float((u & 0x007fffffu) | 0x3f800000u);   /* [1, 2) */
    if (m > 1.41421356f) { m = m * 0.5f; e += 1; }              /* [0.7071, 1.4142] */
    const float s = (m - 1.0f) / (m + 1.0f);
    const float z = s * s;
    float p = 1.181265637e-01f;
    p = p * z + 1.426729113e-01f;
    p = p * z + 2.000019550e-01f;
    p = p * z + 3.333333135e-01f;
    p = p * z + 1.0f;
    const float ln = (s + s) * p;            /* log(m) */
    return (float)e + ln * 1.44269504089f;
}

RT_MATH_FN float rt_asin_tail(float z)
{
    float p = 3.380628675e-02f;
    p = p * z + 1.707748882e-02f;
    p = p * z + 3.111624904e-02f;
    p = p * z + 4.459802061e-02f;
    p = p * z + 7.500098646e-02f;
    p = p * z + 1.666666567e-01f;
    return p;   /* (asin(x)/x - 1) / x^2 with z = x^2 in [0, 0.25] */
}

RT_MATH_FN float rt_acosf(float x)
{
    if (x != x) return x;
    const float ax = x < 0.0f ? -x : x;
    if (ax > 1.0f) return rt_bits_float(0x7fc00000u);
    if (ax <= 0.5f) {
        const float z = x * x;
        return 1.57079637f - (x + x * (z * rt_asin_tail(z)));
    }
    /* acos(|x|) = 2 asin(sqrt((1 - |x|) / 2)) */
    const float z = (1.0f - ax) * 0.5f;
    const float s = sqrtf(z);
    const float t = s + s * (z * rt_asin_tail(z));
    return x > 0.0f ? t + t : 3.14159274f - (t + t);
}

/* k / 10000 for k = 0 ... 9999, correctly rounded like the division the shader writes
 * (`float(seed % 10000) / 10000.0`, Random.glsl:11), without the general division sequence: the product
 * with the rounded reciprocal plus one fused correction step. Exact for all 10 000 dividends
 * (tests/test_rt_math.py goes through every one of them); not meant for other arguments. */
RT_MATH_FN float rt_div10000f(float k)
{
    const float c = 1.0f / 10000.0f;
    const float q0 = k * c;
    return fmaf(fmaf(-q0, 10000.0f, k), c, q0);
}

#endif /* RT_MATH_H */
