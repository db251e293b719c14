// psk_libm.h -- the three libm functions of the hot path, restated so that the GPU returns the
// same floats as the reference's CPU build does.
//
// The reference calls, through libstdc++'s std::arg and std::polar,
//     atan2f   reference cpp/psk_soft.cpp:474 (raw phase) and :547 (8-PSK slicing)
//     cosf, sinf  reference cpp/psk_soft.cpp:499 (de-rotation phasor)
// of glibc 2.35 (third-party, not in /root/reference; pinned version = the oracle image's).
// Restated here from the published algorithms:
//   * atan2f / atanf: the fdlibm single-precision routines glibc 2.35 ships
//     (sysdeps/ieee754/flt-32/e_atan2f.c, s_atanf.c): argument reduction to one of five
//     ranges, an 11-term odd polynomial evaluated in float as two interleaved Horner chains,
//     hi/lo table recombination.  All float operations, unfused.
//   * sinf / cosf: the ARM "optimized routines" implementation glibc 2.35 ships
//     (sysdeps/ieee754/flt-32/s_sinf.c, s_cosf.c, sincosf.h): reduction and polynomial in
//     double; x86-64 hosts with FMA run the ifunc variant built with contraction, which the
//     explicit fma() calls below reproduce.
// tests/test_libm_pin.py compiles this header for the host and checks all three functions
// bit-for-bit against the oracle image's libm (10^8 arguments, including every special
// range), so device results equal the oracle's wherever the arguments are equal.
//
// The header is host/device neutral: PSK_HD expands to __host__ __device__ under hipcc.
#ifndef PSK_LIBM_H
#define PSK_LIBM_H

#include <stdint.h>

#if defined(__HIPCC__)
#define PSK_HD __host__ __device__ __forceinline__
#define PSK_HDM __host__ __device__ __forceinline__  // member functions
#else
#define PSK_HD static inline
#define PSK_HDM inline
#endif

namespace psk {

// A double constant for use inside the block loop of a kernel.  Left to itself the compiler hoists
// such constants out of the loop into scalar register pairs, runs out of scalar registers, and
// parks them in lanes of a vector register -- every use then costs vector-ALU instructions to
// fetch them back.  PSK_KD materialises the constant with two scalar moves where it is used;
// `dep` is any value that changes per loop iteration (it only pins the moves inside the loop).
#if defined(__HIP_DEVICE_COMPILE__)
template <uint64_t BITS>
__device__ __forceinline__ double lm_kd(int dep)
{
    int lo, hi;
    asm("s_mov_b32 %0, %1" : "=s"(lo) : "i"((uint32_t)BITS), "s"(dep));
    asm("s_mov_b32 %0, %1" : "=s"(hi) : "i"((uint32_t)(BITS >> 32)), "s"(dep));
    return __builtin_bit_cast(double, ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
}
#define PSK_KD(val, dep) psk::lm_kd<__builtin_bit_cast(uint64_t, (double)(val))>(dep)
#else
#define PSK_KD(val, dep) ((double)(val))
#endif

PSK_HD uint32_t lm_asuint(float f) { return __builtin_bit_cast(uint32_t, f); }
// the value, opaque to the optimiser on the device: keeps integer arithmetic on bit patterns from being folded back into the
// compares and selects it was written to replace (no instruction is emitted)
PSK_HD uint32_t lm_opaque(uint32_t v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    asm("" : "+v"(v));
#endif
    return v;
}
PSK_HD float lm_asfloat(uint32_t u) { return __builtin_bit_cast(float, u); }

// ---------------------------------------------------------------------------------
// atanf / atan2f (fdlibm float)
// ---------------------------------------------------------------------------------
PSK_HD float lm_atanf(float x)
{
    const float hi0 = 4.6364760399e-01f, hi1 = 7.8539812565e-01f, hi2 = 9.8279368877e-01f, hi3 = 1.5707962513e+00f;
    const float lo0 = 5.0121582440e-09f, lo1 = 3.7748947079e-08f, lo2 = 3.4473217170e-08f, lo3 = 7.5497894159e-08f;
    const float aT0 = 3.3333334327e-01f, aT1 = -2.0000000298e-01f, aT2 = 1.4285714924e-01f, aT3 = -1.1111110449e-01f,
                aT4 = 9.0908870101e-02f, aT5 = -7.6918758452e-02f, aT6 = 6.6610731184e-02f, aT7 = -5.8335702866e-02f,
                aT8 = 4.9768779427e-02f, aT9 = -3.6531571299e-02f, aT10 = 1.6285819933e-02f;
    const int32_t hx = (int32_t)lm_asuint(x);
    const int32_t ix = hx & 0x7fffffff;
    if (ix >= 0x4c000000) {  // |x| >= 2^25
        if (ix > 0x7f800000)
            return x + x;  // NaN
        return hx > 0 ? hi3 + lo3 : -hi3 - lo3;
    }
    // argument reduction: one division whatever the range (the reference code has one per branch)
    const float ax = __builtin_fabsf(x);
    float num, den, hi, lo;
    int id;
    if (ix < 0x3ee00000) {  // |x| < 0.4375
        if (ix < 0x31000000)  // |x| < 2^-29
            return x;
        id = -1;
        num = x;
        den = 1.0f;
        hi = 0.0f;
        lo = 0.0f;
    } else if (ix < 0x3f300000) {  // 7/16 <= |x| < 11/16
        id = 0;
        num = 2.0f * ax - 1.0f;
        den = 2.0f + ax;
        hi = hi0;
        lo = lo0;
    } else if (ix < 0x3f980000) {  // 11/16 <= |x| < 19/16
        id = 1;
        num = ax - 1.0f;
        den = ax + 1.0f;
        hi = hi1;
        lo = lo1;
    } else if (ix < 0x401c0000) {  // |x| < 2.4375
        id = 2;
        num = ax - 1.5f;
        den = 1.0f + 1.5f * ax;
        hi = hi2;
        lo = lo2;
    } else {
        id = 3;
        num = -1.0f;
        den = ax;
        hi = hi3;
        lo = lo3;
    }
    const float t = (id < 0) ? num : num / den;
    const float z = t * t;
    const float w = z * z;
    const float s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
    const float s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
    if (id < 0)
        return t - t * (s1 + s2);
    const float r = hi - ((t * (s1 + s2) - lo) - t);
    return hx < 0 ? -r : r;
}

PSK_HD float lm_atan2f(float y, float x)
{
    const float tiny = 1.0e-30f, pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f,
                pi_lo = -8.7422776573e-08f;
    const int32_t hx = (int32_t)lm_asuint(x), hy = (int32_t)lm_asuint(y);
    const int32_t ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
    if (ix > 0x7f800000 || iy > 0x7f800000)
        return x + y;  // NaN
    if (hx == 0x3f800000)
        return lm_atanf(y);  // x == 1.0
    const int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);  // 2*sign(x) + sign(y)
    if (iy == 0) {
        switch (m) {
        case 0:
        case 1: return y;
        case 2: return pi + tiny;
        default: return -pi - tiny;
        }
    }
    if (ix == 0)
        return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    if (ix == 0x7f800000) {
        if (iy == 0x7f800000) {
            switch (m) {
            case 0: return pi_o_4 + tiny;
            case 1: return -pi_o_4 - tiny;
            case 2: return 3.0f * pi_o_4 + tiny;
            default: return -3.0f * pi_o_4 - tiny;
            }
        }
        switch (m) {
        case 0: return 0.0f;
        case 1: return -0.0f;
        case 2: return pi + tiny;
        default: return -pi - tiny;
        }
    }
    if (iy == 0x7f800000)
        return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    const int32_t k = (iy - ix) >> 23;
    float z;
    if (k > 60)
        z = pi_o_2 + 0.5f * pi_lo;
    else if (hx < 0 && k < -60)
        z = 0.0f;
    else
        z = lm_atanf(__builtin_fabsf(y / x));
    switch (m) {
    case 0: return z;
    case 1: return lm_asfloat(lm_asuint(z) ^ 0x80000000u);
    case 2: return pi - (z - pi_lo);
    default: return (z - pi_lo) - pi;
    }
}

// ---------------------------------------------------------------------------------
// sinf / cosf (glibc 2.35 flt-32, FMA variant), both from one range reduction
// ---------------------------------------------------------------------------------
PSK_HD uint32_t lm_abstop12(float x) { return (lm_asuint(x) >> 20) & 0x7ff; }

// sinf_poly of sincosf.h: n even -> sine polynomial, n odd -> cosine polynomial;
// neg selects the second table entry (cosine coefficients negated)
PSK_HD float lm_sincos_poly(double x, double x2, bool neg, int n)
{
    const double C0 = 0x1p0, C1 = -0x1.ffffffd0c621cp-2, C2 = 0x1.55553e1068f19p-5, C3 = -0x1.6c087e89a359dp-10,
                 C4 = 0x1.99343027bf8c3p-16;
    const double S1 = -0x1.555545995a603p-3, S2 = 0x1.1107605230bc4p-7, S3 = -0x1.994eb3774cf24p-13;
    if ((n & 1) == 0) {
        double x3 = x * x2;
        double s1 = __builtin_fma(x2, S3, S2);
        double x7 = x3 * x2;
        double s = __builtin_fma(x3, S1, x);
        return (float)__builtin_fma(x7, s1, s);
    }
    const double sg = neg ? -1.0 : 1.0;
    double x4 = x2 * x2;
    double c2 = __builtin_fma(x2, sg * C4, sg * C3);
    double c1 = __builtin_fma(x2, sg * C1, sg * C0);
    double x6 = x4 * x2;
    double c = __builtin_fma(x4, sg * C2, c1);
    return (float)__builtin_fma(x6, c2, c);
}

// *sp = sinf(y), *cp = cosf(y)
PSK_HD void lm_sincosf(float y, float *sp, float *cp)
{
    const double HPI_INV = 0x1.45F306DC9C883p+23;  // 2/pi * 2^24
    const double HPI = 0x1.921FB54442D18p0;        // pi/2
    const double x = (double)y;
    const uint32_t top = lm_abstop12(y);
    if (top < lm_abstop12(0x1.921FB6p-1f)) {  // |y| < pi/4
        if (top < lm_abstop12(0x1p-12f)) {
            *sp = y;
            *cp = 1.0f;
            return;
        }
        const double x2 = x * x;
        *sp = lm_sincos_poly(x, x2, false, 0);
        *cp = lm_sincos_poly(x, x2, false, 1);
        return;
    }
    int n, q;
    double xr;
    if (top < lm_abstop12(120.0f)) {  // reduce_fast
        double r = x * HPI_INV;
        n = ((int32_t)r + 0x800000) >> 24;
        xr = __builtin_fma(-(double)n, HPI, x);
        q = n;
    } else if (top < lm_abstop12(__builtin_inff())) {  // reduce_large: 192 bits of 4/pi
        static constexpr uint32_t inv_pio4[24] = {0xa2,       0xa2f9,     0xa2f983,   0xa2f9836e, 0xf9836e4e, 0x836e4e44,
                                       0x6e4e4415, 0x4e441529, 0x441529fc, 0x1529fc27, 0x29fc2757, 0xfc2757d1,
                                       0x2757d1f5, 0x57d1f534, 0xd1f534dd, 0xf534ddc0, 0x34ddc0db, 0xddc0db62,
                                       0xc0db6295, 0xdb629599, 0x6295993c, 0x95993c43, 0x993c4390, 0x3c439041};
        uint32_t xi = lm_asuint(y);
        const int sign = (int)(xi >> 31);
        const uint32_t *arr = &inv_pio4[(xi >> 26) & 15];
        const int shift = (int)((xi >> 23) & 7);
        xi = (xi & 0xffffff) | 0x800000;
        xi <<= shift;
        uint64_t res0 = (uint64_t)(uint32_t)(xi * arr[0]);
        uint64_t res1 = (uint64_t)xi * arr[4];
        uint64_t res2 = (uint64_t)xi * arr[8];
        res0 = (res2 >> 32) | (res0 << 32);
        res0 += res1;
        uint64_t nn = (res0 + (1ULL << 61)) >> 62;
        res0 -= nn << 62;
        xr = (double)(int64_t)res0 * 0x1.921FB54442D18p-62;
        n = (int)nn;
        q = n + sign;
    } else {  // inf / NaN -> NaN
        *sp = y - y;
        *cp = y - y;
        return;
    }
    const double s = ((q & 3) == 1 || (q & 3) == 2) ? -1.0 : 1.0;  // sign[q & 3] = {1,-1,-1,1}
    const bool neg = (q & 2) != 0;
    const double xs = xr * s, x2 = xr * xr;
    *sp = lm_sincos_poly(xs, x2, neg, n);
    *cp = lm_sincos_poly(xs, x2, neg, n ^ 1);
}

// ---------------------------------------------------------------------------------
// Straight-line forms for SIMD execution.  lm_atan2f_ordinary returns the same bits as lm_atan2f
// for every finite operand pair and sets *special for an infinite or NaN operand (the caller then
// takes the general routine above); lm_sincosf_ordinary covers every argument.
// tests/support/libm_pin.cpp checks both against glibc wherever *special is false.
// ---------------------------------------------------------------------------------
// The five argument ranges of s_atanf.c as one table: with t = (c1*a + c0) / (d1*a + d0) and the
// result hi - ((t*(s1+s2) - lo) - t), every range -- the first one (t = a, result t - t*(s1+s2))
// included -- runs the same instructions; products with 0, 1 and 2 are exact, so each entry
// rounds exactly where the reference expression does.  Row `which`, entry `id`:
//   which: 0 c1, 1 c0, 2 d1, 3 d0, 4 hi, 5 lo;   id: 0 a<7/16, 1 <11/16, 2 <19/16, 3 <39/16, 4 rest
struct LmAtanTabHost {
    PSK_HDM float get(int which, int id) const
    {
        static const float T[6][5] = {
            {1.0f, 2.0f, 1.0f, 1.0f, 0.0f},
            {0.0f, -1.0f, -1.0f, -1.5f, -1.0f},
            {0.0f, 1.0f, 1.0f, 1.5f, 1.0f},
            {1.0f, 2.0f, 1.0f, 1.0f, 0.0f},
            {0.0f, 4.6364760399e-01f, 7.8539812565e-01f, 9.8279368877e-01f, 1.5707962513e+00f},
            {0.0f, 5.0121582440e-09f, 3.7748947079e-08f, 3.4473217170e-08f, 7.5497894159e-08f},
        };
        return T[which][id];
    }
    PSK_HDM static bool any(bool v) { return v; }
};

// Tab::get(which, id) looks a table entry up (the device version keeps each row in the lanes of one
// register and reads it with ds_bpermute: no select chains); Tab::any(v) is "v holds in some lane".
template <class Tab>
PSK_HD float lm_atan2f_ordinary_t(float y, float x, bool *special, const Tab &tab)
{
    const float aT0 = 3.3333334327e-01f, aT1 = -2.0000000298e-01f, aT2 = 1.4285714924e-01f, aT3 = -1.1111110449e-01f,
                aT4 = 9.0908870101e-02f, aT5 = -7.6918758452e-02f, aT6 = 6.6610731184e-02f, aT7 = -5.8335702866e-02f,
                aT8 = 4.9768779427e-02f, aT9 = -3.6531571299e-02f, aT10 = 1.6285819933e-02f;
    const float pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f, pi_o_2 = 1.5707963705e+00f;
    const uint32_t ux = lm_asuint(x), uy = lm_asuint(y);
    const uint32_t ix = ux & 0x7fffffffu, iy = uy & 0x7fffffffu;
    // only an infinite or NaN operand needs the general routine
    *special = (ix > 0x7f7fffffu) || (iy > 0x7f7fffffu);
    const float a = __builtin_fabsf(y / x);  // atanf argument, >= 0 (x == 0: inf or NaN, patched below)
    const uint32_t ia = lm_asuint(a);
    // atanf(a), a >= 0.  The first range also serves a < 2^-29, where s_atanf.c returns its
    // argument: t - t*(s1+s2) rounds to t there.
    // (the range index by subtractions and sign bits -- ia and the bounds are below 2^31 -- instead of four compares and the
    // selects behind them: compares and selects cost gfx950 1.6 to 4 times a subtraction, tools/micro/valu_rate_probe.hip)
    const int id = -(((int)lm_opaque(0x3edfffffu - ia) >> 31) + ((int)lm_opaque(0x3f2fffffu - ia) >> 31) +
                     ((int)lm_opaque(0x3f97ffffu - ia) >> 31) + ((int)lm_opaque(0x401bffffu - ia) >> 31));
    const float c1a = tab.get(0, id) * a, d1a = tab.get(2, id) * a;
    const float num = c1a + tab.get(1, id);
    const float den = d1a + tab.get(3, id);
    const float hi = tab.get(4, id), lo = tab.get(5, id);
    const float t = num / den;
    const float z = t * t;
    const float w = z * z;
    const float s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
    const float s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
    const float p = t * (s1 + s2);
    float zat = hi - ((p - lo) - t);  // atanf(|y/x|)
    // quadrant, e_atan2f.c switch (m); x == 1.0 needs no case of its own: atanf(y) is odd in y
    // (x - y and y - x round to the same magnitude: the lower half plane is the upper one with the sign of y -- one select
    // and one sign transfer instead of two differences and three selects)
    const bool xneg = (ux >> 31) != 0, yneg = (uy >> 31) != 0;
    float q2 = pi - (zat - pi_lo);
    float r = lm_asfloat((lm_asuint(xneg ? q2 : zat) & 0x7fffffffu) | (uy & 0x80000000u));
    // rare operands, patched afterwards:
    //  * a >= 2^25 (also a = inf): s_atanf.c returns hi3 + lo3, and e_atan2f.c's shortcut for an
    //    exponent difference above 60, pi/2 + 0.5*pi_lo, is the same float.  (Its other shortcut,
    //    z = 0 for x < 0 and an exponent difference below -60, changes nothing after z - pi_lo.)
    //  * zeros (e_atan2f.c "when y = 0" / "when x = 0"; the +-tiny there is absorbed by rounding)
    const bool rare = (ix == 0) || (iy == 0) || (ia >= 0x4c000000u);
    if (tab.any(rare)) {
        zat = (ia >= 0x4c000000u) ? pi_o_2 : zat;
        q2 = pi - (zat - pi_lo);
        r = lm_asfloat((lm_asuint(xneg ? q2 : zat) & 0x7fffffffu) | (uy & 0x80000000u));
        const float ry0 = xneg ? (yneg ? -pi : pi) : y;
        const float rx0 = yneg ? -pi_o_2 : pi_o_2;
        r = (ix == 0) ? rx0 : r;
        r = (iy == 0) ? ry0 : r;
    }
    return r;
}
PSK_HD float lm_atan2f_ordinary(float y, float x, bool *special)
{
    return lm_atan2f_ordinary_t(y, x, special, LmAtanTabHost());
}

// atan2f for the operand pairs lm_atan2f_ordinary_t leaves out (*special): a NaN or an infinity
// somewhere.  The handful of constants of e_atan2f.c, nothing else.
PSK_HD float lm_atan2f_nonfinite(float y, float x)
{
    const float tiny = 1.0e-30f, pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f;
    const uint32_t ux = lm_asuint(x), uy = lm_asuint(y);
    const uint32_t ix = ux & 0x7fffffffu, iy = uy & 0x7fffffffu;
    const bool xneg = (ux >> 31) != 0, yneg = (uy >> 31) != 0;
    if (ix > 0x7f800000u || iy > 0x7f800000u)
        return x + y;  // NaN
    if (ix == 0x7f800000u) {
        if (iy == 0x7f800000u)
            return xneg ? (yneg ? -3.0f * pi_o_4 - tiny : 3.0f * pi_o_4 + tiny) : (yneg ? -pi_o_4 - tiny : pi_o_4 + tiny);
        return xneg ? (yneg ? -pi - tiny : pi + tiny) : (yneg ? -0.0f : 0.0f);
    }
    return yneg ? -pi_o_2 - tiny : pi_o_2 + tiny;  // y infinite, x finite
}

// sinf / cosf as one straight line: reduce_fast with n = 0 is the identity for |y| < pi/4, so the
// first two ranges of s_sinf.c share the code, tiny |y| is patched at the end; for |y| >= 120
// (ordinary business once a carrier offset has run the phase estimate up) the 192-bit reduction
// replaces (n, xr) in the lanes that need it, behind a test the whole wave shares.
// *special is never set any more (kept for the call sites' sake).
#if defined(__HIP_DEVICE_COMPILE__)
#define PSK_LM_ANY(x) __any(x)
#else
#define PSK_LM_ANY(x) (x)
#endif
PSK_HD void lm_sincosf_ordinary(float y, float *sp, float *cp, bool *special, int dep = 0)
{
    const double HPI_INV = PSK_KD(0x1.45F306DC9C883p+23, dep), HPI = PSK_KD(0x1.921FB54442D18p0, dep);
    const double C1 = PSK_KD(-0x1.ffffffd0c621cp-2, dep), C2 = PSK_KD(0x1.55553e1068f19p-5, dep),
                 C3 = PSK_KD(-0x1.6c087e89a359dp-10, dep), C4 = PSK_KD(0x1.99343027bf8c3p-16, dep);
    const double S1 = PSK_KD(-0x1.555545995a603p-3, dep), S2 = PSK_KD(0x1.1107605230bc4p-7, dep),
                 S3 = PSK_KD(-0x1.994eb3774cf24p-13, dep);
    const double C0 = 0x1p0;
    const uint32_t top = lm_abstop12(y);
    *special = false;
    const double x = (double)y;
    const double r = x * HPI_INV;
    int n = ((int32_t)r + 0x800000) >> 24;  // (garbage for huge |y|: replaced below)
    double xr = __builtin_fma(-(double)n, HPI, x);
    int q = n;  // selects sign and table entry; the polynomial's parity goes by n
    const bool big = top >= lm_abstop12(120.0f);
    if (PSK_LM_ANY(big)) {
        if (big) {  // reduce_large of sincosf.h: 192 bits of 4/pi; inf / NaN come out as NaN at the end
            static constexpr uint32_t inv_pio4[24] = {0xa2,       0xa2f9,     0xa2f983,   0xa2f9836e, 0xf9836e4e, 0x836e4e44,
                                                      0x6e4e4415, 0x4e441529, 0x441529fc, 0x1529fc27, 0x29fc2757, 0xfc2757d1,
                                                      0x2757d1f5, 0x57d1f534, 0xd1f534dd, 0xf534ddc0, 0x34ddc0db, 0xddc0db62,
                                                      0xc0db6295, 0xdb629599, 0x6295993c, 0x95993c43, 0x993c4390, 0x3c439041};
            uint32_t xi = lm_asuint(y);
            const int sign = (int)(xi >> 31);
            const uint32_t *arr = &inv_pio4[(xi >> 26) & 15];
            const int shift = (int)((xi >> 23) & 7);
            xi = (xi & 0xffffff) | 0x800000;
            xi <<= shift;
            uint64_t res0 = (uint64_t)(uint32_t)(xi * arr[0]);
            uint64_t res1 = (uint64_t)xi * arr[4];
            uint64_t res2 = (uint64_t)xi * arr[8];
            res0 = (res2 >> 32) | (res0 << 32);
            res0 += res1;
            uint64_t nn = (res0 + (1ULL << 61)) >> 62;
            res0 -= nn << 62;
            xr = (double)(int64_t)res0 * 0x1.921FB54442D18p-62;
            n = (int)nn;
            q = n + sign;
        }
    }
    // sign[q & 3] = {1,-1,-1,1} on the sine's argument, and the second table entry (cosine coefficients negated) for q & 2:
    // both polynomials are evaluated for the positive signs and the sign put on the result -- every operation in them is
    // odd (sine) resp. linear (cosine) in the sign and rounding to nearest is symmetric, so the bits are the same; it saves
    // a multiplication and six selects per symbol
    const double x2 = xr * xr;
    // sine polynomial
    const double x3 = xr * x2;
    const double s1 = __builtin_fma(x2, S3, S2);
    const double x7 = x3 * x2;
    const double s = __builtin_fma(x3, S1, xr);
    const float ps = lm_asfloat(lm_asuint((float)__builtin_fma(x7, s1, s)) ^ (((uint32_t)(q + 1) & 2u) << 30));
    // cosine polynomial
    const double x4 = x2 * x2;
    const double c2 = __builtin_fma(x2, C4, C3);
    const double c1 = __builtin_fma(x2, C1, C0);
    const double x6 = x4 * x2;
    const double c = __builtin_fma(x4, C2, c1);
    const float pc = lm_asfloat(lm_asuint((float)__builtin_fma(x6, c2, c)) ^ (((uint32_t)q & 2u) << 30));
    const bool odd = (n & 1) != 0;
    const bool tiny = top < lm_abstop12(0x1p-12f);
    const bool nonfinite = top >= lm_abstop12(__builtin_inff());
    float sv = tiny ? y : (odd ? pc : ps);
    float cv = tiny ? 1.0f : (odd ? ps : pc);
    if (PSK_LM_ANY(nonfinite)) {
        sv = nonfinite ? y - y : sv;
        cv = nonfinite ? y - y : cv;
    }
    *sp = sv;
    *cp = cv;
}

// ---------------------------------------------------------------------------------
// 8-PSK sector of a point without the arctangent.  The reference slices by
// round(atan2f(im, re) / pi * 4) (cpp/psk_soft.cpp:547-555): the decision boundaries are the rays at
// odd multiples of pi/8, i.e. |im| = tan(pi/8) |re| and |im| = tan(3 pi/8) |re|.  Away from them the
// sector follows from two compares and the signs; *near is set where the point lies within 4e-5
// (relative) of a boundary, is the origin, or is not finite -- there the caller takes the
// arctangent, whose last-bit behaviour decides.  (atan2f is good to an ulp and the float / double
// steps after it to 2^-24: seven hundred times finer than the margin.)
// tests/support/libm_pin.cpp checks the result against the reference expression wherever !*near.
// ---------------------------------------------------------------------------------
PSK_HD unsigned lm_slice8_fast(float re, float im, bool *near)
{
    const float T1 = 0.41421356f, T3 = 2.41421356f, eps = 4.0e-5f;
    const float u = __builtin_fabsf(re), t = __builtin_fabsf(im);
    const float a1 = T1 * u, a3 = T3 * u;
    const bool lo = t < a1, hi = t > a3;
    const bool neg_re = lm_asuint(re) >> 31, neg_im = lm_asuint(im) >> 31;
    // sectors: 0 around +re, 2 around +im, 4 around -re, 6 around -im, odd ones in between
    const unsigned diag = neg_re ? (neg_im ? 5u : 3u) : (neg_im ? 7u : 1u);
    const unsigned axis_re = neg_re ? 4u : 0u, axis_im = neg_im ? 6u : 2u;
    const unsigned s = lo ? axis_re : (hi ? axis_im : diag);
    const float d1 = __builtin_fabsf(t - a1), d3 = __builtin_fabsf(t - a3);
    const bool fin = (lm_asuint(u) < 0x7f800000u) && (lm_asuint(t) < 0x7f800000u);
    // (!(x > y) rather than x <= y: a NaN anywhere lands in *near)
    *near = !fin || !(d1 > eps * (t + a1)) || !(d3 > eps * (t + a3));
    return s;
}

// ---------------------------------------------------------------------------------
// a / b for a divisor whose correctly rounded reciprocal rb = 1.0 / b is known
// (Markstein: q = RN(a*rb), r = a - q*b exactly by fma, RN(q + r*rb) is the correctly
// rounded quotient; the excluded case, a divisor significand of all ones, cannot occur for
// the divisors used here: 2*pi, float-valued doubles and small integers).  Bit-identical to
// the reference's IEEE double divisions at cpp/psk_soft.cpp:157-158 and :477.
// ---------------------------------------------------------------------------------
PSK_HD double lm_div_known(double a, double b, double rb)
{
    double q = a * rb;
    double r = __builtin_fma(-q, b, a);
    return __builtin_fma(r, rb, q);
}

}  // namespace psk
#endif
