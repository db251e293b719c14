/*
 * psk_soft_oracle.c -- CPU oracle for the psk_soft hot path (TEST INFRASTRUCTURE ONLY).
 *
 * Sequential restatement, in plain C with every float/double rounding point
 * written out, of
 *     psk_soft_i::serviceFunction()   reference cpp/psk_soft.cpp:346-618
 *     psk_soft_i::resyncEnergy()      reference cpp/psk_soft.cpp:619-636
 *     property listeners              reference cpp/psk_soft.cpp:638-651
 *     class LinearFit                 reference cpp/psk_soft.cpp:35-185
 * and of the third-party arithmetic those lines call (libstdc++ 11 <complex>:
 * norm / pow(complex,int) / polar / arg / operator* / operator/;  libgcc
 * __mulsc3 / __divsc3;  glibc 2.35 libm atan2f / sinf / cosf / round / pow),
 * in the language mode the reference is built in (gnu++98: SURVEY.md Q5-Q7).
 *
 * Build: gcc -O2 -ffp-contract=off (no -march=native, no -ffast-math); see
 * oracle/Makefile.  See psk_soft_oracle.h for the pinning status.
 *
 * Not supported (undefined behaviour in the reference itself): samplesPerBaud
 * == 0 (index never wraps, cpp/psk_soft.cpp:441,454) and phaseAvg == 0
 * (front() of an empty deque, cpp/psk_soft.cpp:54-55,70).
 */
#include "psk_soft_oracle.h"

#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* cpp/psk_soft.h:65  static const double M_2PI = 2*M_PI */
#define ORC_M_2PI (2 * M_PI)
#define ORC_RESYNC_COUNT 1048576u /* cpp/psk_soft.cpp:51,582 */

/* ------------------------------------------------------------------ */
/* third-party arithmetic, restated                                    */
/* ------------------------------------------------------------------ */

/* std::norm(complex<float>) -- /usr/include/c++/11/complex:678-696:
 * x*x + y*y evaluated in float (two products, one sum, unfused). */
static inline float orc_norm(float re, float im)
{
    float xx = re * re;
    float yy = im * im;
    return xx + yy;
}

/* complex<float> operator*= -> GCC complex multiply -> (ac-bd, ad+bc) with the
 * C99 Annex G recovery of libgcc __mulsc3 when both parts come out NaN. */
static void orc_cmul(float a, float b, float c, float d, float *xr, float *xi)
{
    float ac = a * c, bd = b * d, ad = a * d, bc = b * c;
    float x = ac - bd;
    float y = ad + bc;
    if (isnan(x) && isnan(y)) {
        int recalc = 0;
        if (isinf(a) || isinf(b)) {
            a = copysignf(isinf(a) ? 1.0f : 0.0f, a);
            b = copysignf(isinf(b) ? 1.0f : 0.0f, b);
            if (isnan(c)) c = copysignf(0.0f, c);
            if (isnan(d)) d = copysignf(0.0f, d);
            recalc = 1;
        }
        if (isinf(c) || isinf(d)) {
            c = copysignf(isinf(c) ? 1.0f : 0.0f, c);
            d = copysignf(isinf(d) ? 1.0f : 0.0f, d);
            if (isnan(a)) a = copysignf(0.0f, a);
            if (isnan(b)) b = copysignf(0.0f, b);
            recalc = 1;
        }
        if (!recalc && (isinf(ac) || isinf(bd) || isinf(ad) || isinf(bc))) {
            if (isnan(a)) a = copysignf(0.0f, a);
            if (isnan(b)) b = copysignf(0.0f, b);
            if (isnan(c)) c = copysignf(0.0f, c);
            if (isnan(d)) d = copysignf(0.0f, d);
            recalc = 1;
        }
        if (recalc) {
            x = INFINITY * (a * c - b * d);
            y = INFINITY * (a * d + b * c);
        }
    }
    *xr = x;
    *xi = y;
}

/* complex<float> operator/= -> libgcc __divsc3.  A g++-linked program resolves
 * __divsc3 from libgcc_s.so.1, which in this image is GCC 12's: the quotient is
 * formed in DOUBLE by the plain formula (no Smith scaling), rounded to float
 * once per part, then the Annex G recovery runs.  (GCC <= 11's static libgcc.a
 * used Smith's algorithm in float; prim_check.cpp pins which one is live.)
 * (a+ib)/(c+id). */
static void orc_cdiv(float a, float b, float c, float d, float *xr, float *xi)
{
    float x, y;
    {
        double aa = a, bb = b, cc = c, dd = d;
        double denom = (cc * cc) + (dd * dd);
        x = (float)(((aa * cc) + (bb * dd)) / denom);
        y = (float)(((bb * cc) - (aa * dd)) / denom);
    }
    if (isnan(x) && isnan(y)) {
        if (c == 0.0f && d == 0.0f && (!isnan(a) || !isnan(b))) {
            x = copysignf(INFINITY, c) * a;
            y = copysignf(INFINITY, c) * b;
        } else if ((isinf(a) || isinf(b)) && isfinite(c) && isfinite(d)) {
            a = copysignf(isinf(a) ? 1.0f : 0.0f, a);
            b = copysignf(isinf(b) ? 1.0f : 0.0f, b);
            x = INFINITY * (a * c + b * d);
            y = INFINITY * (b * c - a * d);
        } else if ((isinf(c) || isinf(d)) && isfinite(a) && isfinite(b)) {
            c = copysignf(isinf(c) ? 1.0f : 0.0f, c);
            d = copysignf(isinf(d) ? 1.0f : 0.0f, d);
            x = 0.0f * (a * c + b * d);
            y = 0.0f * (b * c - a * d);
        }
    }
    *xr = x;
    *xi = y;
}

/* pow(complex<float>, size_t) under gnu++98 resolves to pow(complex<float>,int)
 * = __complex_pow_unsigned: /usr/include/c++/11/complex:997-1024 (Q6). */
static void orc_cpow(float xr, float xi, unsigned n, float *pr, float *pi)
{
    float yr, yi;
    if (n % 2) { yr = xr; yi = xi; } else { yr = 1.0f; yi = 0.0f; }
    while (n >>= 1) {
        orc_cmul(xr, xi, xr, xi, &xr, &xi);
        if (n % 2)
            orc_cmul(yr, yi, xr, xi, &yr, &yi);
    }
    *pr = yr;
    *pi = yi;
}

/* std::polar(float(1.0), theta): (rho*cosf(theta), rho*sinf(theta)),
 * /usr/include/c++/11/complex:699-705 */
static inline void orc_polar1(float theta, float *re, float *im)
{
    const float rho = 1.0f;
    *re = rho * cosf(theta);
    *im = rho * sinf(theta);
}

/* cpp/psk_soft.cpp:596  abs(phaseEstimate) > wrapValue.  Unqualified abs()
 * with only <complex>/<cmath> in scope is ::abs(int) (Q5): the float is
 * truncated to int first (x86 cvttss2si: NaN / out of range -> INT_MIN, and
 * abs(INT_MIN) stays INT_MIN), then compared as float. */
static int orc_wrap_test(float phaseEstimate, float wrapValue)
{
    int asInt;
    if (isnan(phaseEstimate) || phaseEstimate >= 2147483648.0f || phaseEstimate < -2147483648.0f)
        asInt = INT_MIN;
    else
        asInt = (int)phaseEstimate;
    if (asInt != INT_MIN && asInt < 0)
        asInt = -asInt;
    return (float)asInt > wrapValue;
}

/* (long) of a double the way x86-64 cvttsd2si does it */
static long orc_to_long(double v)
{
    if (isnan(v) || v >= 9223372036854775808.0 || v < -9223372036854775808.0)
        return LONG_MIN;
    return (long)v;
}

/* ------------------------------------------------------------------ */
/* small containers standing in for std::deque / std::vector           */
/* ------------------------------------------------------------------ */

typedef struct { float re, im; double e; } orc_sample_t; /* samples[i] + energy[i] */

typedef struct {
    orc_sample_t *buf;
    size_t head, len, cap;
} orc_sfifo_t;

static void sfifo_push(orc_sfifo_t *f, orc_sample_t v)
{
    if (f->head + f->len == f->cap) {
        if (f->head > 0 && f->head >= f->len) {
            memmove(f->buf, f->buf + f->head, f->len * sizeof *f->buf);
            f->head = 0;
        } else {
            size_t ncap = f->cap ? 2 * f->cap : 1024;
            f->buf = (orc_sample_t *)realloc(f->buf, ncap * sizeof *f->buf);
            f->cap = ncap;
        }
    }
    f->buf[f->head + f->len++] = v;
}
static inline orc_sample_t *sfifo_at(orc_sfifo_t *f, size_t i) { return &f->buf[f->head + i]; }
static inline void sfifo_pop_front(orc_sfifo_t *f, size_t n) { f->head += n; f->len -= n; if (!f->len) f->head = 0; }

typedef struct {
    float *buf;
    size_t head, len, cap;
} orc_ffifo_t;

static void ffifo_push(orc_ffifo_t *f, float v)
{
    if (f->head + f->len == f->cap) {
        if (f->head > 0 && f->head >= f->len) {
            memmove(f->buf, f->buf + f->head, f->len * sizeof *f->buf);
            f->head = 0;
        } else {
            size_t ncap = f->cap ? 2 * f->cap : 256;
            f->buf = (float *)realloc(f->buf, ncap * sizeof *f->buf);
            f->cap = ncap;
        }
    }
    f->buf[f->head + f->len++] = v;
}
static inline void ffifo_pop_front(orc_ffifo_t *f) { f->head++; f->len--; if (!f->len) f->head = 0; }

typedef struct { void *p; size_t len, cap; } orc_vec_t;
static void *vec_grow(orc_vec_t *v, size_t elem)
{
    if (v->len == v->cap) {
        v->cap = v->cap ? 2 * v->cap : 1024;
        v->p = realloc(v->p, v->cap * elem);
    }
    return (char *)v->p + elem * v->len++;
}
#define VEC_PUSH(vec, type, val) (*(type *)vec_grow(&(vec), sizeof(type)) = (val))

/* ------------------------------------------------------------------ */
/* LinearFit (cpp/psk_soft.h:33-53, cpp/psk_soft.cpp:35-185)           */
/* ------------------------------------------------------------------ */

typedef struct {
    orc_ffifo_t yvals;
    float m, b;
    double ySum, xySum;
    size_t n;
    float xdelta, denominator, xAvg;
    size_t count;
} orc_linfit_t;

/* cpp/psk_soft.cpp:176-185.  pow() here is C ::pow(double,double) (Q7). */
static void linfit_calc_denominator(orc_linfit_t *f)
{
    size_t pts = f->yvals.len;
    if (pts <= 1)
        return;
    size_t pts_m_1 = pts - 1;
    double p = (double)pts_m_1;
    double poly = pow(p, 3) / 3.0 + pow(p, 2) / 2.0 + p / 6.0 - pow(p, 2) * (double)pts / 4.0;
    f->denominator = (float)(pow((double)f->xdelta, 2) * poly);
    f->xAvg = f->xdelta * (float)pts_m_1 / 2;
}

/* cpp/psk_soft.cpp:135-174 */
static float linfit_calc_fit(orc_linfit_t *f)
{
    size_t pts = f->yvals.len;
    if (pts > 1) {
        size_t pts_m_1 = pts - 1;
        float half_span = f->xdelta * (float)pts_m_1 / 2;              /* float ops     */
        f->m = (float)((f->xySum - (double)half_span * f->ySum) / (double)f->denominator);
        float mx = f->m * f->xAvg;                                      /* float product */
        f->b = (float)(f->ySum / (double)pts - (double)mx);
        float xVal = f->xdelta * (float)pts_m_1;
        float mxv = f->m * xVal;
        return mxv + f->b;
    }
    f->m = 0;
    if (pts == 0)
        f->b = 0;
    else
        f->b = f->yvals.buf[f->yvals.head + f->yvals.len - 1];
    return f->b;
}

/* cpp/psk_soft.cpp:89-124 */
static float linfit_reset(orc_linfit_t *f, const size_t *numPts, const float *sampleRate,
                          int forceHistoryClear)
{
    if (sampleRate) {
        float newXdelta = (float)(1.0 / (double)*sampleRate);
        if (f->xdelta != newXdelta) {
            f->xdelta = newXdelta;
            forceHistoryClear = 1;
        }
    }
    if (forceHistoryClear) {
        f->yvals.len = 0;
        f->yvals.head = 0;
    }
    if (numPts && *numPts != f->n) {
        f->n = *numPts;
        while (f->yvals.len > f->n)
            ffifo_pop_front(&f->yvals);
    }
    unsigned int j = 0;
    f->ySum = 0;
    f->xySum = 0;
    for (size_t i = 0; i < f->yvals.len; i++, j++) {
        float y = f->yvals.buf[f->yvals.head + i];
        f->ySum += (double)y;
        float jx = (float)j * f->xdelta;
        float jxy = jx * y;
        f->xySum += (double)jxy;
    }
    linfit_calc_denominator(f);
    f->count = 0;
    return linfit_calc_fit(f);
}

/* cpp/psk_soft.cpp:48-87 */
static float linfit_next(orc_linfit_t *f, float yval)
{
    if (f->count == ORC_RESYNC_COUNT)
        linfit_reset(f, NULL, NULL, 0);
    int steadyState = (f->yvals.len == f->n);
    if (steadyState) {
        f->ySum -= (double)f->yvals.buf[f->yvals.head];
        ffifo_pop_front(&f->yvals);
        f->xySum -= (double)f->xdelta * f->ySum;
    }
    f->ySum += (double)yval;
    float t = yval * (float)f->yvals.len; /* size before the push: intentional in the reference */
    t = t * f->xdelta;
    f->xySum += (double)t;
    ffifo_push(&f->yvals, yval);
    if (!steadyState)
        linfit_calc_denominator(f);
    f->count++;
    return linfit_calc_fit(f);
}

/* cpp/psk_soft.cpp:126-133 */
static float linfit_subtract_const(orc_linfit_t *f, float yval)
{
    for (size_t i = 0; i < f->yvals.len; i++)
        f->yvals.buf[f->yvals.head + i] -= yval;
    return linfit_reset(f, NULL, NULL, 0);
}

/* ------------------------------------------------------------------ */
/* psk_soft_i                                                          */
/* ------------------------------------------------------------------ */

struct psk_oracle {
    /* properties: cpp/psk_soft_base.h:45-56, defaults cpp/psk_soft_base.cpp:96-148 */
    unsigned short samplesPerBaud;
    uint32_t numAvg;
    unsigned short constelationSize;
    unsigned short phaseAvg;
    int differentialDecoding;
    int resetState;
    /* members: cpp/psk_soft.h:66-86 */
    orc_sfifo_t samples;          /* samples + energy deques, always the same length */
    double *symbolEnergy;
    size_t symbolEnergySize;
    size_t index;
    float last_re, last_im;
    int resetSamplesPerBaud, resetNumSymbols, resetPhaseAvg;
    float phaseEstimate;
    float sampleRate;
    size_t count;
    orc_linfit_t phaseEstimator;
    /* per-call outputs */
    orc_vec_t out, bits, phase, sidx;
};

psk_oracle_t *psk_oracle_create(void)
{
    psk_oracle_t *o = (psk_oracle_t *)calloc(1, sizeof *o);
    o->samplesPerBaud = 10;
    o->numAvg = 100;
    o->constelationSize = 4;
    o->phaseAvg = 50;
    o->differentialDecoding = 0;
    o->resetState = 0;
    /* cpp/psk_soft.cpp:187-199 */
    o->symbolEnergySize = o->samplesPerBaud;
    o->symbolEnergy = (double *)calloc(o->symbolEnergySize ? o->symbolEnergySize : 1, sizeof(double));
    o->index = 0;
    o->last_re = 0;
    o->last_im = 0;
    o->resetSamplesPerBaud = 1;
    o->resetNumSymbols = 1;
    o->resetPhaseAvg = 1;
    o->phaseEstimate = 0.0f;
    o->sampleRate = 1.0f;
    o->count = 0;
    /* LinearFit(phaseAvg, sampleRate): cpp/psk_soft.cpp:35-46 */
    o->phaseEstimator.m = 0;
    o->phaseEstimator.b = 0;
    o->phaseEstimator.ySum = 0;
    o->phaseEstimator.xySum = 0;
    o->phaseEstimator.n = o->phaseAvg;
    o->phaseEstimator.xdelta = (float)(1.0 / (double)o->sampleRate);
    o->phaseEstimator.denominator = 1.0f;
    o->phaseEstimator.xAvg = 0.0f;
    o->phaseEstimator.count = 0;
    return o;
}

void psk_oracle_destroy(psk_oracle_t *o)
{
    if (!o)
        return;
    free(o->samples.buf);
    free(o->symbolEnergy);
    free(o->phaseEstimator.yvals.buf);
    free(o->out.p);
    free(o->bits.p);
    free(o->phase.p);
    free(o->sidx.p);
    free(o);
}

void psk_oracle_set_property(psk_oracle_t *o, int prop, uint32_t value, int fire)
{
    switch (prop) {
    case PSK_ORACLE_PROP_samplesPerBaud:
        o->samplesPerBaud = (unsigned short)value;
        if (fire) /* cpp/psk_soft.cpp:638-641 */
            o->resetSamplesPerBaud = (o->samplesPerBaud != o->symbolEnergySize);
        break;
    case PSK_ORACLE_PROP_numAvg:
        o->numAvg = value;
        break;
    case PSK_ORACLE_PROP_constelationSize:
        o->constelationSize = (unsigned short)value;
        if (fire) /* cpp/psk_soft.cpp:643-646 */
            o->resetNumSymbols = 1;
        break;
    case PSK_ORACLE_PROP_phaseAvg:
        o->phaseAvg = (unsigned short)value;
        if (fire) /* cpp/psk_soft.cpp:648-651 */
            o->resetPhaseAvg = 1;
        break;
    case PSK_ORACLE_PROP_differentialDecoding:
        o->differentialDecoding = value != 0;
        break;
    case PSK_ORACLE_PROP_resetState:
        o->resetState = value != 0;
        break;
    default:
        break;
    }
}

uint32_t psk_oracle_get_property(const psk_oracle_t *o, int prop)
{
    switch (prop) {
    case PSK_ORACLE_PROP_samplesPerBaud: return o->samplesPerBaud;
    case PSK_ORACLE_PROP_numAvg: return o->numAvg;
    case PSK_ORACLE_PROP_constelationSize: return o->constelationSize;
    case PSK_ORACLE_PROP_phaseAvg: return o->phaseAvg;
    case PSK_ORACLE_PROP_differentialDecoding: return (uint32_t)o->differentialDecoding;
    case PSK_ORACLE_PROP_resetState: return (uint32_t)o->resetState;
    default: return 0;
    }
}

/* cpp/psk_soft.cpp:619-636 */
static void resync_energy(psk_oracle_t *o, size_t samplesPerSymbol, size_t numDataPts)
{
    if (o->symbolEnergySize != samplesPerSymbol) {
        free(o->symbolEnergy);
        o->symbolEnergy = (double *)calloc(samplesPerSymbol ? samplesPerSymbol : 1, sizeof(double));
        o->symbolEnergySize = samplesPerSymbol;
    }
    for (size_t k = 0; k < samplesPerSymbol; k++)
        o->symbolEnergy[k] = 0.0;
    if (o->samples.len > numDataPts)
        o->samples.len = numDataPts; /* erase(begin()+numDataPts, end()): the NEWEST are dropped */
    o->index = 0;
    for (size_t i = 0; i < o->samples.len; i++) {
        o->symbolEnergy[o->index] += sfifo_at(&o->samples, i)->e;
        o->index++;
        if (o->index == samplesPerSymbol)
            o->index = 0;
    }
    o->count = 0;
}

/* the per-symbol body, cpp/psk_soft.cpp:457-585 */
static void emit_symbol(psk_oracle_t *o, float cur_re, float cur_im, size_t S, size_t numDataPts,
                        size_t numSyms, size_t bitsPerBaud, psk_oracle_result_t *res)
{
    float s_re, s_im;
    if (S > 1) {
        /* :462 std::max_element -> FIRST maximum, strict '<' */
        size_t best = 0;
        for (size_t k = 1; k < S; k++)
            if (o->symbolEnergy[best] < o->symbolEnergy[k])
                best = k;
        orc_sample_t *pick = sfifo_at(&o->samples, best); /* :465 */
        s_re = pick->re;
        s_im = pick->im;
        VEC_PUSH(o->sidx, short, (short)(unsigned short)best); /* :466 */
    } else {
        s_re = cur_re; /* :469 */
        s_im = cur_im;
    }

    /* :474  double thisPhase = arg(pow(sample,numSyms)) */
    float p_re, p_im;
    orc_cpow(s_re, s_im, (unsigned)(int)numSyms, &p_re, &p_im);
    double thisPhase = (double)atan2f(p_im, p_re);

    /* :477-478 */
    long numWraps = orc_to_long(round(((double)o->phaseEstimate - thisPhase) / ORC_M_2PI));
    thisPhase += (double)numWraps * ORC_M_2PI;

    /* :481-482 */
    o->phaseEstimate = linfit_next(&o->phaseEstimator, (float)thisPhase);
    VEC_PUSH(o->phase, float, o->phaseEstimate);

    float phaseCorrection = 0;
    if (o->differentialDecoding) { /* :486-491 */
        float d_re, d_im;
        orc_cdiv(s_re, s_im, o->last_re, o->last_im, &d_re, &d_im);
        o->last_re = s_re;
        o->last_im = s_im;
        s_re = d_re;
        s_im = d_im;
    } else {
        phaseCorrection = -o->phaseEstimate / (float)numSyms; /* :494 */
    }
    if (numSyms == 4) /* :497-498 */
        phaseCorrection = (float)((double)phaseCorrection + M_PI_4);
    float ph_re, ph_im;
    orc_polar1(phaseCorrection, &ph_re, &ph_im); /* :499 */
    float c_re, c_im;
    orc_cmul(s_re, s_im, ph_re, ph_im, &c_re, &c_im); /* :500 */
    VEC_PUSH(o->out, float, c_re);
    VEC_PUSH(o->out, float, c_im);

    if (bitsPerBaud == 1) { /* :503-513 */
        VEC_PUSH(o->bits, short, (short)(c_re < 0));
    } else if (bitsPerBaud == 2) { /* :514-527, Q1: float -> bool is '!= 0' */
        int real = (c_re != 0);
        int imag = (c_im != 0);
        VEC_PUSH(o->bits, short, (short)(real ^ imag));
        VEC_PUSH(o->bits, short, (short)(!imag));
    } else if (bitsPerBaud == 3) { /* :528-564 */
        float theta = atan2f(c_im, c_re);
        float softsym = (float)((double)theta / M_PI * 4);
        if ((double)softsym < -.5)
            softsym = softsym + 8.0f;
        double r = round((double)softsym);
        /* double -> unsigned short the way x86-64 gcc does it: cvttsd2si to a
         * 32-bit int, keep the low 16 bits (Q17: -1 -> 0xFFFF, NaN -> 0) */
        int asInt;
        if (isnan(r) || r >= 2147483648.0 || r < -2147483648.0)
            asInt = INT_MIN;
        else
            asInt = (int)r;
        unsigned short sym = (unsigned short)(unsigned)asInt;
        for (size_t j = 0; j != 3; j++) {
            VEC_PUSH(o->bits, short, (short)(sym & 1));
            sym = sym >> 1;
        }
    } else {
        res->n_warn++; /* :565-566 */
    }

    if (S > 1) { /* :568-584 */
        for (size_t k = 0; k < S; k++)
            o->symbolEnergy[k] -= sfifo_at(&o->samples, k)->e;
        sfifo_pop_front(&o->samples, S);
        o->count++;
        if (o->count == ORC_RESYNC_COUNT)
            resync_energy(o, S, numDataPts);
    }
}

int psk_oracle_service(psk_oracle_t *o, const psk_oracle_packet_t *pkt, psk_oracle_result_t *res)
{
    memset(res, 0, sizeof *res);
    o->out.len = o->bits.len = o->phase.len = o->sidx.len = 0;
    if (!pkt) { /* :350-352 */
        res->ret = PSK_ORACLE_NOOP;
        return res->ret;
    }
    res->ret = PSK_ORACLE_NORMAL;
    if (pkt->inputQueueFlushed) { /* :353-357 */
        res->n_warn++;
        o->resetState = 1;
    }
    if (pkt->mode != 1) { /* :359-363 */
        res->n_warn++;
        return res->ret;
    }
    if (o->resetState) { /* :365-372 */
        o->resetSamplesPerBaud = 1;
        o->resetNumSymbols = 1;
        o->resetPhaseAvg = 1;
        o->resetState = 0;
    }
    /* :376-390 */
    const size_t samplesPerSymbol = o->samplesPerBaud;
    const size_t numDataPts = samplesPerSymbol * (size_t)o->numAvg;
    const size_t numSyms = o->constelationSize;
    if (samplesPerSymbol == 0)
        return res->ret; /* undefined in the reference; refuse */
    if (numDataPts > o->samples.len)
        o->resetSamplesPerBaud = 1;
    size_t bitsPerBaud = 0;
    if (numSyms == 2)
        bitsPerBaud = 1;
    else if (numSyms == 4)
        bitsPerBaud = 2;
    else if (numSyms == 8)
        bitsPerBaud = 3;

    /* :393-405 */
    if (pkt->sriChanged || o->resetNumSymbols || o->resetSamplesPerBaud) {
        double xdelta = pkt->xdelta;
        if (xdelta != (double)o->sampleRate) { /* Q3: period compared with rate */
            o->sampleRate = (float)(1.0 / xdelta);
            linfit_reset(&o->phaseEstimator, NULL, &o->sampleRate, 0);
        }
        xdelta *= (double)samplesPerSymbol;
        res->sri_pushed = 1;
        res->sri_soft_xdelta = xdelta;
        xdelta /= (double)bitsPerBaud;
        res->sri_bits_xdelta = xdelta;
    }
    if (o->resetSamplesPerBaud) { /* :408-412 */
        resync_energy(o, samplesPerSymbol, numDataPts);
        o->resetSamplesPerBaud = 0;
    }
    if (o->resetNumSymbols) { /* :416-420 */
        linfit_reset(&o->phaseEstimator, NULL, NULL, 1);
        o->resetNumSymbols = 0;
    }
    if (o->resetPhaseAvg) { /* :421-426 */
        size_t numPts = o->phaseAvg;
        linfit_reset(&o->phaseEstimator, &numPts, NULL, 0);
        o->resetPhaseAvg = 0;
    }

    /* :428-591 */
    const size_t nComplex = pkt->n_floats / 2;
    const size_t lastSample = samplesPerSymbol - 1;
    for (size_t i = 0; i < nComplex; i++) {
        float re = pkt->data[2 * i], im = pkt->data[2 * i + 1];
        if (samplesPerSymbol > 1) { /* :445-452 */
            orc_sample_t s;
            s.re = re;
            s.im = im;
            s.e = (double)orc_norm(re, im);
            sfifo_push(&o->samples, s);
            o->symbolEnergy[o->index] += s.e;
        }
        if (o->index == lastSample) { /* :454 */
            if (o->samples.len == numDataPts) /* :457 */
                emit_symbol(o, re, im, samplesPerSymbol, numDataPts, numSyms, bitsPerBaud, res);
            o->index = 0; /* :587 */
        } else {
            o->index++; /* :590 */
        }
    }

    /* :592-603 */
    float wrapValue = (float)(ORC_M_2PI * (double)numSyms);
    if (orc_wrap_test(o->phaseEstimate, wrapValue)) {
        float q = o->phaseEstimate / wrapValue;
        long numWraps = orc_to_long(round((double)q));
        float c = (float)numWraps * wrapValue;
        o->phaseEstimate = linfit_subtract_const(&o->phaseEstimator, c);
    }

    res->soft = (const float *)o->out.p;
    res->n_soft_floats = o->out.len;
    res->bits = (const short *)o->bits.p;
    res->n_bits = o->bits.len;
    res->phase = (const float *)o->phase.p;
    res->n_phase = o->phase.len;
    res->index = (const short *)o->sidx.p;
    res->n_index = o->sidx.len;
    return res->ret;
}

size_t psk_oracle_ring_size(const psk_oracle_t *o) { return o->samples.len; }
size_t psk_oracle_index(const psk_oracle_t *o) { return o->index; }
float psk_oracle_phase_estimate(const psk_oracle_t *o) { return o->phaseEstimate; }
size_t psk_oracle_fit_history(const psk_oracle_t *o, float *dst, size_t cap)
{
    size_t n = o->phaseEstimator.yvals.len;
    for (size_t i = 0; i < n && i < cap; i++)
        dst[i] = o->phaseEstimator.yvals.buf[o->phaseEstimator.yvals.head + i];
    return n;
}

/* exported primitives (checked against the toolchain's own routines) */
void psk_oracle_prim_cmul(float a, float b, float c, float d, float *re, float *im) { orc_cmul(a, b, c, d, re, im); }
void psk_oracle_prim_cdiv(float a, float b, float c, float d, float *re, float *im) { orc_cdiv(a, b, c, d, re, im); }
void psk_oracle_prim_cpow(float a, float b, unsigned n, float *re, float *im) { orc_cpow(a, b, n, re, im); }
float psk_oracle_prim_norm(float a, float b) { return orc_norm(a, b); }
void psk_oracle_prim_polar1(float theta, float *re, float *im) { orc_polar1(theta, re, im); }
int psk_oracle_prim_wrap_test(float phaseEstimate, float wrapValue) { return orc_wrap_test(phaseEstimate, wrapValue); }
float psk_oracle_prim_denominator(float xdelta, size_t pts)
{
    orc_linfit_t f;
    memset(&f, 0, sizeof f);
    f.xdelta = xdelta;
    f.yvals.len = pts;
    f.denominator = 1.0f;
    linfit_calc_denominator(&f);
    return f.denominator;
}
