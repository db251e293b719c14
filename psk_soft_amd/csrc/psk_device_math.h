// psk_device_math.h -- gfx950 device arithmetic for the psk_soft hot path.
//
// Every function states which reference expression (and which libstdc++ / libgcc routine
// behind it) it reproduces, with the float / double rounding points of SURVEY.md
// section 8(a).  The translation unit is built with -ffp-contract=off, so a*b+c below
// is a rounded product followed by a rounded sum (quirk Q9), never an fma.
#ifndef PSK_DEVICE_MATH_H
#define PSK_DEVICE_MATH_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "psk_libm.h"

namespace psk {

#define PSK_DEV __device__ __forceinline__

constexpr double kTwoPi = 6.283185307179586476925286766559;  // cpp/psk_soft.h:65  2*M_PI
constexpr double kPi = 3.14159265358979323846;               // M_PI
constexpr double kPi4 = 0.78539816339744830962;              // M_PI_4
constexpr double kInvTwoPi = 1.0 / kTwoPi;                   // correctly rounded 1/(2*pi), for lm_div_known

struct cf32 {
    float re, im;
};

// std::norm(complex<float>) (cpp/psk_soft.cpp:448): x*x + y*y in float, unfused.
PSK_DEV float norm_f(float re, float im)
{
    float xx = re * re;
    float yy = im * im;
    return xx + yy;
}

PSK_DEV bool is_nan(float v) { return v != v; }
PSK_DEV bool is_inf(float v) { return __builtin_fabsf(v) == __builtin_inff(); }
PSK_DEV bool is_fin(float v) { return __builtin_fabsf(v) < __builtin_inff(); }

// complex<float> multiply = GCC complex multiply (cpp/psk_soft.cpp:474 via pow, :500):
// (ac-bd, ad+bc); when both parts are NaN, libgcc __mulsc3's C99 Annex G recovery.
PSK_DEV cf32 cmul_recover(float a, float b, float c, float d, float ac, float bd, float ad, float bc)
{
    cf32 r;
    r.re = ac - bd;
    r.im = ad + bc;
    bool recalc = false;
    if (is_inf(a) || is_inf(b)) {
        a = __builtin_copysignf(is_inf(a) ? 1.0f : 0.0f, a);
        b = __builtin_copysignf(is_inf(b) ? 1.0f : 0.0f, b);
        if (is_nan(c)) c = __builtin_copysignf(0.0f, c);
        if (is_nan(d)) d = __builtin_copysignf(0.0f, d);
        recalc = true;
    }
    if (is_inf(c) || is_inf(d)) {
        c = __builtin_copysignf(is_inf(c) ? 1.0f : 0.0f, c);
        d = __builtin_copysignf(is_inf(d) ? 1.0f : 0.0f, d);
        if (is_nan(a)) a = __builtin_copysignf(0.0f, a);
        if (is_nan(b)) b = __builtin_copysignf(0.0f, b);
        recalc = true;
    }
    if (!recalc && (is_inf(ac) || is_inf(bd) || is_inf(ad) || is_inf(bc))) {
        if (is_nan(a)) a = __builtin_copysignf(0.0f, a);
        if (is_nan(b)) b = __builtin_copysignf(0.0f, b);
        if (is_nan(c)) c = __builtin_copysignf(0.0f, c);
        if (is_nan(d)) d = __builtin_copysignf(0.0f, d);
        recalc = true;
    }
    if (recalc) {
        r.re = __builtin_inff() * (a * c - b * d);
        r.im = __builtin_inff() * (a * d + b * c);
    }
    return r;
}

template <bool RECOVER>
PSK_DEV cf32 cmul(cf32 x, cf32 y)
{
    float ac = x.re * y.re, bd = x.im * y.im, ad = x.re * y.im, bc = x.im * y.re;
    cf32 r;
    r.re = ac - bd;
    r.im = ad + bc;
    if (RECOVER) {
        const bool both_nan = is_nan(r.re) && is_nan(r.im);
        if (__any(both_nan)) {  // wave-uniform test first: the recovery code stays off the common path
            if (both_nan)
                r = cmul_recover(x.re, x.im, y.re, y.im, ac, bd, ad, bc);
        }
    }
    return r;
}

// pow(complex<float>, size_t) in gnu++98 = __complex_pow_unsigned (quirk Q6,
// cpp/psk_soft.cpp:474).  n is wave-uniform.
template <bool RECOVER>
PSK_DEV cf32 cpow_uint(cf32 x, unsigned n)
{
    cf32 y;
    if (n % 2) {
        y = x;
    } else {
        y.re = 1.0f;
        y.im = 0.0f;
    }
    while (n >>= 1) {
        x = cmul<RECOVER>(x, x);
        if (n % 2)
            y = cmul<RECOVER>(y, x);
    }
    return y;
}

// complex<float> divide = libgcc __divsc3 as a g++-linked binary resolves it in the oracle
// image (GCC 12 libgcc_s: quotient formed in double, one rounding per part, then the
// Annex G recovery) -- cpp/psk_soft.cpp:488, differential decoding only (quirk Q10).
template <bool RECOVER>
PSK_DEV cf32 cdiv(cf32 n, cf32 dn)
{
    float a = n.re, b = n.im, c = dn.re, d = dn.im;
    double aa = a, bb = b, cc = c, dd = d;
    double denom = (cc * cc) + (dd * dd);
    float x = (float)(((aa * cc) + (bb * dd)) / denom);
    float y = (float)(((bb * cc) - (aa * dd)) / denom);
    if (RECOVER && __any(is_nan(x) && is_nan(y)) && is_nan(x) && is_nan(y)) {
        if (c == 0.0f && d == 0.0f && (!is_nan(a) || !is_nan(b))) {
            x = __builtin_copysignf(__builtin_inff(), c) * a;
            y = __builtin_copysignf(__builtin_inff(), c) * b;
        } else if ((is_inf(a) || is_inf(b)) && is_fin(c) && is_fin(d)) {
            a = __builtin_copysignf(is_inf(a) ? 1.0f : 0.0f, a);
            b = __builtin_copysignf(is_inf(b) ? 1.0f : 0.0f, b);
            x = __builtin_inff() * (a * c + b * d);
            y = __builtin_inff() * (b * c - a * d);
        } else if ((is_inf(c) || is_inf(d)) && is_fin(a) && is_fin(b)) {
            c = __builtin_copysignf(is_inf(c) ? 1.0f : 0.0f, c);
            d = __builtin_copysignf(is_inf(d) ? 1.0f : 0.0f, d);
            x = 0.0f * (a * c + b * d);
            y = 0.0f * (b * c - a * d);
        }
    }
    cf32 r;
    r.re = x;
    r.im = y;
    return r;
}

// atan2f / sincosf for a whole wave: straight-line forms that cover every argument, the rare ones
// (NaN, infinities, huge angles) behind wave-uniform tests that are practically never taken.
// The range table of the straight-line atan2f (LmAtanTabHost) held in registers: row `which` lives
// in the first five lanes of one VGPR and an entry is fetched with ds_bpermute -- one LDS-crossbar
// instruction instead of a chain of selects, and fewer registers than the broadcast constants.
struct AtanTabDev {
    int packed;  // lane 5*which + id holds T[which][id]: all six rows in one register
    PSK_DEV float get(int which, int id) const
    {
        // (the row offset folds into the instruction's offset field)
        return __int_as_float(__builtin_amdgcn_ds_bpermute((id << 2) + 20 * which, packed));
    }
    PSK_DEV static bool any(bool v) { return __any(v); }
};
PSK_DEV AtanTabDev atan_tab_dev(int lane)
{
    AtanTabDev t;
    const LmAtanTabHost h;
    float v = 0.0f;
#pragma unroll
    for (int which = 0; which < 6; which++)
#pragma unroll
        for (int id = 0; id < 5; id++) v = (lane == 5 * which + id) ? h.get(which, id) : v;
    t.packed = __float_as_int(v);
    return t;
}
// (every lane of the wave must be active where the table form is used: ds_bpermute reads lanes 0-29)
struct AtanTabWave {
    PSK_DEV float get(int which, int id) const { return LmAtanTabHost().get(which, id); }
    PSK_DEV static bool any(bool v) { return __any(v); }
};

// atan2f for a whole wave: the table-driven straight line for every finite operand pair, a few
// constants for a NaN or an infinity (wave-uniform test first: practically never taken)
template <class Tab>
PSK_DEV float atan2f_wave(float y, float x, const Tab &tab)
{
    bool sp;
    float r = lm_atan2f_ordinary_t(y, x, &sp, tab);
    if (__any(sp)) {
        if (sp)
            r = lm_atan2f_nonfinite(y, x);
    }
    return r;
}
PSK_DEV float atan2f_wave(float y, float x) { return atan2f_wave(y, x, AtanTabWave()); }
// sincosf: the straight-line form covers every argument (the large-argument reduction sits behind
// a wave-uniform test: |theta| >= 120 is ordinary business, the phase estimate grows by the carrier
// offset times the symbols of the call before the end-of-call wrap brings it back).
PSK_DEV void sincosf_wave(float t, float *sn, float *cs, int dep)
{
    bool sp;
    lm_sincosf_ordinary(t, sn, cs, &sp, dep);
}

// (long) of a double as x86-64 cvttsd2si does it (cpp/psk_soft.cpp:477, 598)
PSK_DEV long long to_long_x86(double v, int dep = 0)
{
    const double lim = PSK_KD(9223372036854775808.0, dep);
    if (!(v < lim) || v < -lim)  // also NaN
        return (long long)0x8000000000000000ull;
    return (long long)v;
}

// numWraps = round((phaseEstimate-thisPhase)/M_2PI)  (cpp/psk_soft.cpp:477)
// (dep: see PSK_KD -- any per-iteration value when called inside a loop)
PSK_DEV long long unwrap_count(float phaseEstimate, double thisPhase, int dep = 0)
{
    // the quotient is bit-identical to the IEEE division (lm_div_known)
    return to_long_x86(
        __builtin_round(lm_div_known((double)phaseEstimate - thisPhase, PSK_KD(kTwoPi, dep), PSK_KD(kInvTwoPi, dep))), dep);
}

// LinearFit::calculateDenominator (cpp/psk_soft.cpp:176-185): C pow(double,double) on
// exactly representable arguments is exact, so p*p*p == pow(p,3) and xd*xd == pow(xd,2).
PSK_DEV void fit_denominator(float xdelta, unsigned pts, float &denominator, float &xAvg)
{
    if (pts <= 1)
        return;
    unsigned pts_m_1 = pts - 1;
    double p = (double)pts_m_1;
    double p2 = p * p;
    double p3 = p2 * p;
    double poly = p3 / 3.0 + p2 / 2.0 + p / 6.0 - p2 * (double)pts / 4.0;
    double xd = (double)xdelta;
    denominator = (float)((xd * xd) * poly);
    xAvg = xdelta * (float)pts_m_1 / 2;
}

// LinearFit::calculateFit (cpp/psk_soft.cpp:135-174) for pts > 1
PSK_DEV float fit_value(double ySum, double xySum, float xdelta, unsigned pts, float denominator, float xAvg,
                        float &m_out, float &b_out)
{
    unsigned pts_m_1 = pts - 1;
    float half_span = xdelta * (float)pts_m_1 / 2;
    float m = (float)((xySum - (double)half_span * ySum) / (double)denominator);
    float mx = m * xAvg;
    float b = (float)(ySum / (double)pts - (double)mx);
    float xVal = xdelta * (float)pts_m_1;
    float mxv = m * xVal;
    m_out = m;
    b_out = b;
    return mxv + b;
}

// A wave-uniform value, told to the compiler as such: it then lives in scalar registers instead
// of occupying a vector register in every lane (v_readfirstlane).
// (The builtin alone is folded away when the operand is already known to be uniform, which
// leaves the value in the vector register its floating-point producer wrote; or-ing in a zero
// the compiler cannot see through keeps it.)
PSK_DEV int uni(int v)
{
    int z;
    asm("v_mov_b32 %0, 0" : "=v"(z));
    return __builtin_amdgcn_readfirstlane(v | z);
}
PSK_DEV float uni(float v) { return __int_as_float(uni(__float_as_int(v))); }
PSK_DEV double uni(double v) { return __hiloint2double(uni(__double2hiint(v)), uni(__double2loint(v))); }

// calculateFit in the steady state (the window holds phaseAvg points): every operand that depends
// only on (xdelta, phaseAvg) is precomputed once per call, the two divisions become
// multiplications by correctly rounded reciprocals (lm_div_known).  All members are wave-uniform.
struct FitKnown {
    float xd;        // LinearFit::xdelta
    float sizef;     // (float)(yvals.size()) before the push, cpp/psk_soft.cpp:78
    float xavg;      // xAvg
    float xval;      // xdelta * (pts-1), the abscissa of the newest point
    double half_span_d, den_d, rden, pts_d, rpts;
};
PSK_DEV FitKnown fit_known(float xdelta, unsigned pts, float denominator, float xAvg)
{
    FitKnown k;
    const unsigned pts_m_1 = pts - 1;
    k.xd = uni(xdelta);
    k.sizef = uni((float)pts_m_1);
    k.xavg = uni(xAvg);
    k.xval = uni(xdelta * (float)pts_m_1);
    k.half_span_d = uni((double)(xdelta * (float)pts_m_1 / 2));
    k.den_d = uni((double)denominator);
    k.rden = uni(1.0 / (double)denominator);
    k.pts_d = uni((double)pts);
    k.rpts = uni(1.0 / (double)pts);
    return k;
}
PSK_DEV float fit_value_known(double ySum, double xySum, const FitKnown &k, float &m_out)
{
    float m = (float)lm_div_known(xySum - k.half_span_d * ySum, k.den_d, k.rden);
    m_out = m;
    float mx = m * k.xavg;
    float b = (float)(lm_div_known(ySum, k.pts_d, k.rpts) - (double)mx);
    float mxv = m * k.xval;
    return mxv + b;
}

// QPSK bit pair (cpp/psk_soft.cpp:523-526): `bool real = out.back().real()` is "!= 0" (quirk Q1,
// the default, bit-exact with the reference); sign_map = the mapping of the diagram at :516-521
PSK_DEV void qpsk_bits(float re, float im, bool sign_map, int &b0, int &b1)
{
    int r, m;
    if (sign_map) {  // (wave-uniform: a scalar branch, one of the two pairs is executed)
        r = re > 0.0f;
        m = im > 0.0f;
    } else {
        // "!= 0" (true for a NaN) on the bits: magnitude + 0x7fffffff carries into the sign bit unless the magnitude is zero
        // (integer additions and shifts instead of compares and selects: a third of their cost on gfx950)
        r = -((int)lm_opaque((__float_as_uint(re) & 0x7fffffffu) + 0x7fffffffu) >> 31);
        m = -((int)lm_opaque((__float_as_uint(im) & 0x7fffffffu) + 0x7fffffffu) >> 31);
    }
    b0 = r ^ m;
    b1 = m ^ 1;
}

// abs(phaseEstimate) > wrapValue with ::abs(int) (quirk Q5, cpp/psk_soft.cpp:596)
PSK_DEV bool wrap_test(float phaseEstimate, float wrapValue)
{
    int asInt;
    if (!(phaseEstimate < 2147483648.0f) || phaseEstimate < -2147483648.0f)
        asInt = (int)0x80000000;
    else
        asInt = (int)phaseEstimate;
    if (asInt != (int)0x80000000 && asInt < 0)
        asInt = -asInt;
    return (float)asInt > wrapValue;
}

// 8-PSK symbol index (cpp/psk_soft.cpp:547-555, quirk Q17) by the reference's own expression
template <class Tab>
PSK_DEV unsigned short slice_8psk_atan(float c_re, float c_im, const Tab &tab)
{
    // (NaN arrives here as a matter of course: the first symbol of every differentially decoded
    // stream divides by last = 0, cpp/psk_soft.cpp:486-491)
    float theta = atan2f_wave(c_im, c_re, tab);
    float softsym = (float)((double)theta / kPi * 4);
    if ((double)softsym < -.5)
        softsym = softsym + 8.0f;
    double r = __builtin_round((double)softsym);
    int asInt;
    if (!(r < 2147483648.0) || r < -2147483648.0)
        asInt = (int)0x80000000;
    else
        asInt = (int)r;
    return (unsigned short)(unsigned)asInt;
}
// ... and as the kernels compute it: the sector from two compares (lm_slice8_fast), the arctangent
// only when some lane's point lies next to a decision boundary.  The whole wave then runs it (its
// range table lives in lanes), the lanes concerned take its answer.
template <class Tab>
PSK_DEV unsigned short slice_8psk(float c_re, float c_im, const Tab &tab)
{
    bool nearb;
    unsigned s = lm_slice8_fast(c_re, c_im, &nearb);
    if (__any(nearb)) {
        const unsigned full = slice_8psk_atan(c_re, c_im, tab);
        s = nearb ? full : s;
    }
    return (unsigned short)s;
}

PSK_DEV unsigned short slice_8psk(float c_re, float c_im) { return slice_8psk(c_re, c_im, AtanTabWave()); }

}  // namespace psk
#endif
