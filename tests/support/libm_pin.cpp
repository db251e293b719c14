// libm_pin.cpp -- compiles the PRODUCT header psk_soft_amd/csrc/psk_libm.h for the host and
// checks its atan2f / sinf / cosf restatements bit-for-bit against this image's glibc (the
// libm the oracle, like the reference, calls), and lm_div_known against IEEE division.
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "psk_libm.h"

static uint64_t st = 88172645463325252ull;
static uint64_t rnd()
{
    st ^= st << 13;
    st ^= st >> 7;
    st ^= st << 17;
    return st;
}
static double unit() { return (double)(rnd() >> 11) / 9007199254740992.0; }
static float anyf()
{
    uint32_t u = (uint32_t)rnd();
    float f;
    memcpy(&f, &u, 4);
    return f;
}
static bool same(float a, float b) { return psk::lm_asuint(a) == psk::lm_asuint(b) || (a != a && b != b); }

int main(int argc, char **argv)
{
    long n_each = argc > 1 ? atol(argv[1]) : 20000000;
    long bad_s = 0, bad_c = 0, bad_a2 = 0, bad_a = 0, bad_d = 0, bad_fs = 0, bad_fa = 0, n_sp_s = 0, n_sp_a = 0;
    for (long i = 0; i < n_each; i++) {
        float x;
        switch (i % 6) {
        case 0: x = anyf(); break;
        case 1: x = (float)(unit() * 240.0 - 120.0); break;
        case 2: x = (float)(unit() * 16.0 - 8.0); break;
        case 3: x = (float)(unit() * 2.0 - 1.0); break;
        case 4: x = (float)(unit() * 2e6 - 1e6); break;
        default: x = (float)((unit() * 2.0 - 1.0) * 1e-3); break;
        }
        float s, c;
        psk::lm_sincosf(x, &s, &c);
        if (!same(s, sinf(x))) bad_s++;
        if (!same(c, cosf(x))) bad_c++;
        bool sp;
        float fs, fc;
        psk::lm_sincosf_ordinary(x, &fs, &fc, &sp);
        if (sp) n_sp_s++;
        else if (!same(fs, s) || !same(fc, c)) bad_fs++;
    }
    for (long i = 0; i < n_each; i++) {
        float y, x;
        switch (i % 8) {
        case 0: y = anyf(); x = anyf(); break;
        case 1: y = (float)(unit() * 4 - 2); x = (float)(unit() * 4 - 2); break;
        case 2: y = (float)((unit() * 4 - 2) * 1e-3); x = (float)(unit() * 4 - 2); break;
        case 3: y = (float)(unit() * 4 - 2); x = (float)((unit() * 4 - 2) * 1e-4); break;
        case 4: case 5: case 6: {  // one operand from the table of special values, the other arbitrary
                   const float sp[] = {0.0f, -0.0f, 1.0f, -1.0f, INFINITY, -INFINITY, NAN, 1e-40f, 3e38f};
                   float o = (i % 8 == 4) ? anyf() : (float)((unit() * 4 - 2) * ldexp(1.0, (int)(rnd() % 100) - 50));
                   if (rnd() & 1) { y = sp[rnd() % 9]; x = o; } else { y = o; x = sp[rnd() % 9]; }
                   break; }
        default: { const float sp[] = {0.0f, -0.0f, 1.0f, -1.0f, INFINITY, -INFINITY, NAN, 1e-40f, 3e38f};
                   y = sp[rnd() % 9]; x = sp[rnd() % 9]; break; }
        }
        if (!same(psk::lm_atan2f(y, x), atan2f(y, x))) bad_a2++;
        if (!same(psk::lm_atanf(y), atanf(y))) bad_a++;
        bool sp;
        float fa = psk::lm_atan2f_ordinary(y, x, &sp);
        if (sp) {
            n_sp_a++;
            if (!same(psk::lm_atan2f_nonfinite(y, x), atan2f(y, x))) bad_fa++;
        } else if (!same(fa, atan2f(y, x))) bad_fa++;
    }
    for (long i = 0; i < n_each; i++) {
        double b;
        switch (i % 3) {
        case 0: { float f = (float)(unit() * 1e3 + 1e-6); b = (double)f; break; }
        case 1: b = (double)(1 + rnd() % 65535); break;
        default: b = 6.283185307179586476925286766559; break;
        }
        double a = (unit() - 0.5) * ldexp(1.0, (int)(rnd() % 80) - 40);
        if (psk::lm_div_known(a, b, 1.0 / b) != a / b) bad_d++;
    }
    // 8-PSK sector without the arctangent vs the reference expression (cpp/psk_soft.cpp:547-555)
    long bad_s8 = 0, n_near = 0;
    for (long i = 0; i < n_each; i++) {
        float re, im;
        switch (i % 4) {
        case 0: re = (float)(unit() * 4 - 2); im = (float)(unit() * 4 - 2); break;
        case 1: { double a = unit() * 6.283185307179586, r = ldexp(1.0, (int)(rnd() % 60) - 30);  // all angles, many scales
                  re = (float)(r * cos(a)); im = (float)(r * sin(a)); break; }
        case 2: { int k = (int)(rnd() % 16); double a = k * 0.39269908169872414 + (unit() - 0.5) * 1e-3;  // around the rays
                  re = (float)cos(a); im = (float)sin(a); break; }
        default: re = anyf(); im = anyf(); break;
        }
        bool nearb;
        unsigned fast = psk::lm_slice8_fast(re, im, &nearb);
        if (nearb) { n_near++; continue; }
        float theta = atan2f(im, re);
        float softsym = (float)((double)theta / M_PI * 4);
        if ((double)softsym < -.5) softsym = softsym + 8.0f;
        unsigned short sym = (unsigned short)round((double)softsym);
        if (sym != fast) bad_s8++;
    }
    printf("slice8: bad=%ld (near a boundary, left to atan2f: %ld of %ld)\n", bad_s8, n_near, n_each);
    if (bad_s8) return 1;
    printf("n=%ld sinf_bad=%ld cosf_bad=%ld atan2f_bad=%ld atanf_bad=%ld div_bad=%ld\n", n_each, bad_s, bad_c, bad_a2, bad_a, bad_d);
    printf("ordinary forms: sincos_bad=%ld (special %ld) atan2_bad=%ld (special %ld)\n", bad_fs, n_sp_s, bad_fa, n_sp_a);
    return (bad_s || bad_c || bad_a2 || bad_a || bad_d || bad_fs || bad_fa) ? 1 : 0;
}
