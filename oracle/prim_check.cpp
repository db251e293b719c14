/*
 * prim_check.cpp -- pins the oracle's arithmetic primitives (TEST INFRASTRUCTURE ONLY).
 *
 * Built as gnu++98 -- the only language mode the reference compiles in (SURVEY.md
 * Q6) -- with the same two headers the reference's hot path sees ("complex",
 * <cmath>; reference cpp/psk_soft.cpp:29-31), so that overload resolution of the
 * unqualified pow/abs/round calls is the toolchain's, not ours.  Every oracle
 * primitive is compared BIT FOR BIT with the library routine the reference
 * calls at the cited line:
 *     std::norm(complex<float>)          cpp/psk_soft.cpp:448
 *     pow(complex<float>, size_t)        cpp/psk_soft.cpp:474
 *     complex<float> operator*           cpp/psk_soft.cpp:500   (libgcc __mulsc3)
 *     complex<float> operator/           cpp/psk_soft.cpp:488   (libgcc __divsc3)
 *     std::polar(float(1.0), float)      cpp/psk_soft.cpp:499
 *     abs(float) > float                 cpp/psk_soft.cpp:596
 *     pow(float,2)*(pow(size_t,3)/3.0..) cpp/psk_soft.cpp:183
 * This file contains no reference code: each probe is a one-line expression of
 * the same *shape* as the cited reference expression.
 */
#include "complex"
#include <cmath>

#include <stdio.h>
#include <string.h>
#include <stdint.h>

#include "psk_soft_oracle.h"

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rng_next()
{
    uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static float rng_float_any()
{ /* any bit pattern: covers inf / nan / denormals */
    uint32_t u = (uint32_t)rng_next();
    float f;
    memcpy(&f, &u, 4);
    return f;
}
static float rng_float_unit()
{
    return (float)((double)(rng_next() >> 11) / 9007199254740992.0 * 4.0 - 2.0);
}
static bool same_bits(float a, float b)
{
    uint32_t x, y;
    memcpy(&x, &a, 4);
    memcpy(&y, &b, 4);
    if (x == y)
        return true;
    /* any NaN matches any NaN: payload / sign of a NaN is not observable downstream */
    return (a != a) && (b != b);
}

static const float specials[] = {0.0f, -0.0f, 1.0f, -1.0f, 0.5f, -2.0f, 1e-30f, -1e-30f, 1e-45f,
                                 3e38f, -3e38f, 1e19f, 1.8446744e19f,
                                 __builtin_inff(), -__builtin_inff(), __builtin_nanf("")};
static const int n_specials = sizeof(specials) / sizeof(specials[0]);

static int fails = 0;
static void report(const char *what, long n, long bad)
{
    printf("%-28s %9ld cases  %ld mismatches\n", what, n, bad);
    if (bad)
        fails++;
}

/* shapes of the reference expressions, resolved by the toolchain */
static int shape_wrap_test(float phaseEstimate, float wrapValue) { return abs(phaseEstimate) > wrapValue; }
static size_t shape_pow_sizeof(std::complex<float> s, size_t numSyms) { return sizeof(pow(s, numSyms).real()); }
static double shape_arg_pow(std::complex<float> s, size_t numSyms) { return arg(pow(s, numSyms)); }
static float shape_denominator(float xdelta, size_t pts)
{
    size_t pts_m_1 = pts - 1;
    float denominator = pow(xdelta, 2) * (pow(pts_m_1, 3) / 3.0 + pow(pts_m_1, 2) / 2.0 + (pts_m_1) / 6.0 - pow(pts_m_1, 2) * pts / 4.0);
    return denominator;
}

int main()
{
    long n, bad;

    /* ---- language-mode facts ---- */
    printf("__cplusplus = %ldL\n", (long)__cplusplus);
    size_t pow_sz = shape_pow_sizeof(std::complex<float>(0.5f, 0.25f), 4);
    printf("sizeof(pow(complex<float>,size_t).real()) = %lu (4 => float repeated squaring)\n", (unsigned long)pow_sz);
    if (pow_sz != 4) fails++;
    printf("sizeof(abs(float)) = %lu, abs(-25.9f) = %d (4, 25 => ::abs(int))\n", (unsigned long)sizeof(abs(-25.9f)), (int)abs(-25.9f));
    if (sizeof(abs(-25.9f)) != 4 || abs(-25.9f) != 25) fails++;
    printf("sizeof(pow(float,2)) = %lu, sizeof(round(float)) = %lu (8, 8 => C double functions)\n",
           (unsigned long)sizeof(pow(1.5f, 2)), (unsigned long)sizeof(round(1.5f)));
    if (sizeof(pow(1.5f, 2)) != 8 || sizeof(round(1.5f)) != 8) fails++;
    double a1 = shape_arg_pow(std::complex<float>(0.7071234f, -0.7069871f), 4);
    printf("arg(pow((0.7071234,-0.7069871),4)) = %.17g (SURVEY A.1 gnu++98: -3.1412069797515869)\n", a1);
    if (a1 != -3.1412069797515869) fails++;
    {
        std::complex<float> q = std::complex<float>(1, 0) / std::complex<float>(0, 0);
        printf("(1,0)/(0,0) = (%g,%g) (SURVEY A.1: (inf,-nan))\n", q.real(), q.imag());
        if (!(q.real() == __builtin_inff()) || q.imag() == q.imag()) fails++;
    }

    /* ---- norm ---- */
    n = bad = 0;
    for (int i = 0; i < 2000000; i++) {
        float a = (i & 1) ? rng_float_any() : rng_float_unit(), b = (i & 2) ? rng_float_any() : rng_float_unit();
        float want = std::norm(std::complex<float>(a, b));
        if (!same_bits(want, psk_oracle_prim_norm(a, b))) bad++;
        n++;
    }
    report("norm", n, bad);

    /* ---- complex multiply ---- */
    n = bad = 0;
    for (int i = 0; i < 2000000 + n_specials * n_specials * n_specials * n_specials; i++) {
        float a, b, c, d;
        if (i < 2000000) {
            a = (i & 1) ? rng_float_any() : rng_float_unit();
            b = (i & 2) ? rng_float_any() : rng_float_unit();
            c = (i & 4) ? rng_float_any() : rng_float_unit();
            d = (i & 8) ? rng_float_any() : rng_float_unit();
        } else {
            int j = i - 2000000;
            a = specials[j % n_specials]; j /= n_specials;
            b = specials[j % n_specials]; j /= n_specials;
            c = specials[j % n_specials]; j /= n_specials;
            d = specials[j % n_specials];
        }
        std::complex<float> w = std::complex<float>(a, b) * std::complex<float>(c, d);
        float re, im;
        psk_oracle_prim_cmul(a, b, c, d, &re, &im);
        if (!same_bits(w.real(), re) || !same_bits(w.imag(), im)) bad++;
        n++;
    }
    report("complex multiply", n, bad);

    /* ---- complex divide ---- */
    n = bad = 0;
    for (int i = 0; i < 2000000 + n_specials * n_specials * n_specials * n_specials; i++) {
        float a, b, c, d;
        if (i < 2000000) {
            a = (i & 1) ? rng_float_any() : rng_float_unit();
            b = (i & 2) ? rng_float_any() : rng_float_unit();
            c = (i & 4) ? rng_float_any() : rng_float_unit();
            d = (i & 8) ? rng_float_any() : rng_float_unit();
        } else {
            int j = i - 2000000;
            a = specials[j % n_specials]; j /= n_specials;
            b = specials[j % n_specials]; j /= n_specials;
            c = specials[j % n_specials]; j /= n_specials;
            d = specials[j % n_specials];
        }
        std::complex<float> w = std::complex<float>(a, b) / std::complex<float>(c, d);
        float re, im;
        psk_oracle_prim_cdiv(a, b, c, d, &re, &im);
        if (!same_bits(w.real(), re) || !same_bits(w.imag(), im)) bad++;
        n++;
    }
    report("complex divide", n, bad);

    /* ---- pow(complex<float>, size_t) for every supported and a few odd M ---- */
    n = bad = 0;
    const size_t Ms[] = {0, 1, 2, 3, 4, 5, 7, 8, 16, 100};
    for (int i = 0; i < 1000000; i++) {
        float a = (i % 7 == 0) ? rng_float_any() : rng_float_unit(), b = (i % 11 == 0) ? rng_float_any() : rng_float_unit();
        if (i < n_specials * n_specials) { a = specials[i % n_specials]; b = specials[i / n_specials]; }
        for (unsigned m = 0; m < sizeof(Ms) / sizeof(Ms[0]); m++) {
            size_t numSyms = Ms[m];
            std::complex<float> w = pow(std::complex<float>(a, b), numSyms);
            float re, im;
            psk_oracle_prim_cpow(a, b, (unsigned)(int)numSyms, &re, &im);
            if (!same_bits(w.real(), re) || !same_bits(w.imag(), im)) bad++;
            n++;
        }
    }
    report("pow(complex<float>,size_t)", n, bad);

    /* ---- polar(1, theta) ---- */
    n = bad = 0;
    for (int i = 0; i < 2000000; i++) {
        float t = (i % 5 == 0) ? rng_float_any() : 8.0f * rng_float_unit();
        if (t != t || t < 0 || t >= 0 || true) {
            std::complex<float> w = std::polar(float(1.0), t);
            float re, im;
            psk_oracle_prim_polar1(t, &re, &im);
            if (!same_bits(w.real(), re) || !same_bits(w.imag(), im)) bad++;
            n++;
        }
    }
    report("polar(1,theta)", n, bad);

    /* ---- abs(phaseEstimate) > wrapValue ---- */
    n = bad = 0;
    for (int i = 0; i < 2000000; i++) {
        float p = (i % 3 == 0) ? rng_float_any() : 40.0f * rng_float_unit();
        float w = (i % 2) ? 12.566371f : ((i % 4) ? 25.132741f : 50.265482f);
        if (shape_wrap_test(p, w) != psk_oracle_prim_wrap_test(p, w)) bad++;
        n++;
    }
    report("abs(float)>wrapValue", n, bad);

    /* ---- LinearFit denominator ---- */
    n = bad = 0;
    const float xds[] = {0.01f, 1.0f, 1e-6f, 0.125f, 3.3333333e-5f};
    for (unsigned x = 0; x < sizeof(xds) / sizeof(xds[0]); x++)
        for (size_t pts = 2; pts <= 65535; pts++) {
            if (!same_bits(shape_denominator(xds[x], pts), psk_oracle_prim_denominator(xds[x], pts))) bad++;
            n++;
        }
    report("LinearFit denominator", n, bad);

    printf(fails ? "PRIM_CHECK FAIL\n" : "PRIM_CHECK OK\n");
    return fails ? 1 : 0;
}
