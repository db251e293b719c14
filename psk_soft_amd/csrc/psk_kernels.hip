// psk_kernels.hip -- CDNA4 (gfx950) kernels for the psk_soft hot path.
//
// One 64-lane wavefront owns one channel (= one psk_soft_i instance) for the whole call and
// walks its packet in blocks of 64 output symbols; the 4096-channel headline workload is
// 4096 single-wave workgroups = 16 waves per CU, no inter-workgroup communication.
//
//   psk_fast_kernel   the wave-scan kernel.  Lane l of block c owns output symbol 64c+l.
//       timing   (cpp/psk_soft.cpp:445-466, 568-584)  per-phase window energy
//                W_k(i) = W_k(i-1) + e_k(i+A-1) - e_k(i-1): one double-precision DPP prefix
//                scan per intra-symbol phase k, exact because the addends are float-valued
//                (quirk Q8) -- the exactness guard below refuses inputs where that fails;
//       phase    (cpp/psk_soft.cpp:474-482, 48-87, 135-174)  M-th power, atan2f, then the
//                feedback unwrap + sliding least-squares fit as two more double prefix scans
//                (ySum, xySum with the reference's float-rounded terms, quirk Q4), the unwrap
//                count speculated by consecutive differences and verified / corrected by a
//                fixed-point pass that fixes >= 1 more lane per pass;
//       output   (cpp/psk_soft.cpp:484-566)  de-rotation, hard bits, 4 output streams.
//   psk_seq_kernel    the reference-order kernel: lane 0 replays the reference's statement
//       order exactly (any property values, samplesPerBaud == 1, calls longer than 2^20
//       symbols, and every call the wave-scan kernel's exactness guard refused).
//
// Built with -ffp-contract=off (quirk Q9).  No MFMA: the path is a streaming complex-MAC
// with O(1) flop/byte, bound by HBM.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "psk_device_math.h"
#include "psk_plan.h"

namespace psk {

constexpr int kWave = 64;
constexpr int kYRing = 512;         // LDS ring of unwrapped phases per wave (floats)
constexpr int kYMask = kYRing - 1;
constexpr int kSeqMaxS = 1024;      // reference-order kernel: symbolEnergy[] lives in LDS
constexpr int kSeqChunk = 512;      // reference-order kernel: packet staging chunk (complex samples)
constexpr int kMaxUnwrapPasses = 80;

// ---------------------------------------------------------------------------------
// cross-lane primitives (wave64, DPP)
// ---------------------------------------------------------------------------------
template <int CTRL, int ROW_MASK>
PSK_DEV int dpp_zero(int v)
{
    // lanes without a source lane, and rows masked off, receive 0
    return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xF, false);
}
template <int CTRL, int ROW_MASK>
PSK_DEV double dpp_zero_f64(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = dpp_zero<CTRL, ROW_MASK>(lo);
    hi = dpp_zero<CTRL, ROW_MASK>(hi);
    return __hiloint2double(hi, lo);
}
// row_shr:N with bound_ctrl:0 -- lanes whose source falls outside their row of 16 read 0, so no
// "old" value has to be materialised (saves two v_mov per 64-bit step)
template <int CTRL>
PSK_DEV int dpp_shr0(int v)
{
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}
template <int CTRL>
PSK_DEV double dpp_shr0_f64(double v)
{
    int lo = dpp_shr0<CTRL>(__double2loint(v)), hi = dpp_shr0<CTRL>(__double2hiint(v));
    return __hiloint2double(hi, lo);
}
// inclusive prefix sum over the 64 lanes: row_shr 1,2,4,8 inside each row of 16, then
// row_bcast:15 into rows 1,3 and row_bcast:31 into rows 2,3
PSK_DEV double wave_scan_f64(double v)
{
    v += dpp_shr0_f64<0x111>(v);
    v += dpp_shr0_f64<0x112>(v);
    v += dpp_shr0_f64<0x114>(v);
    v += dpp_shr0_f64<0x118>(v);
    v += dpp_zero_f64<0x142, 0xA>(v);
    v += dpp_zero_f64<0x143, 0xC>(v);
    return v;
}
PSK_DEV int wave_scan_i32(int v)
{
    v += dpp_shr0<0x111>(v);
    v += dpp_shr0<0x112>(v);
    v += dpp_shr0<0x114>(v);
    v += dpp_shr0<0x118>(v);
    v += dpp_zero<0x142, 0xA>(v);
    v += dpp_zero<0x143, 0xC>(v);
    return v;
}
// value of lane-1 (wave_shr:1); lane 0 receives `carry`
PSK_DEV int wave_up1(int v, int carry) { return __builtin_amdgcn_update_dpp(carry, v, 0x138, 0xF, 0xF, false); }
PSK_DEV float wave_up1(float v, float carry)
{
    return __int_as_float(wave_up1(__float_as_int(v), __float_as_int(carry)));
}
PSK_DEV double wave_up1(double v, double carry)
{
    int lo = wave_up1(__double2loint(v), __double2loint(carry));
    int hi = wave_up1(__double2hiint(v), __double2hiint(carry));
    return __hiloint2double(hi, lo);
}
PSK_DEV float read_lane(float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); }
PSK_DEV double read_lane(double v, int lane)
{
    int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
PSK_DEV double wave_sum_f64(double v) { return read_lane(wave_scan_f64(v), 63); }
PSK_DEV unsigned wave_max_u32(unsigned v)
{
#pragma unroll
    for (int o = 32; o; o >>= 1) {
        unsigned t = (unsigned)__shfl_xor((int)v, o);
        v = t > v ? t : v;
    }
    return v;
}
PSK_DEV unsigned wave_min_u32(unsigned v)
{
#pragma unroll
    for (int o = 32; o; o >>= 1) {
        unsigned t = (unsigned)__shfl_xor((int)v, o);
        v = t < v ? t : v;
    }
    return v;
}
// orders LDS traffic of this wave: a later ds_read sees an earlier ds_write of another lane
PSK_DEV void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// ---------------------------------------------------------------------------------
// the virtual stream X = [ring of carried samples] ++ [packet]   (the `samples` deque)
// ---------------------------------------------------------------------------------
struct XView {
    const float2 *ring;
    const float2 *in;
    uint32_t L0;  // samples in the ring
};
PSK_DEV float2 x_at(const XView &X, uint64_t j) { return j < X.L0 ? X.ring[j] : X.in[j - X.L0]; }

template <int S>
PSK_DEV void load_symbol(const XView &X, uint64_t tau, bool valid, float2 (&x)[S])
{
#pragma unroll
    for (int k = 0; k < S; k++) x[k] = make_float2(0.0f, 0.0f);
    if (!valid)
        return;
    const uint64_t j0 = tau * (uint64_t)S;
    const float2 *p;
    if (j0 >= X.L0) {
        p = X.in + (j0 - X.L0);
    } else if (j0 + S <= X.L0) {
        p = X.ring + j0;
    } else {  // the one symbol that straddles ring and packet
#pragma unroll
        for (int k = 0; k < S; k++) x[k] = x_at(X, j0 + k);
        return;
    }
    if (S % 2 == 0) {
        // 16-byte loads; the address is only 8-byte aligned (gfx950 global loads allow that)
        typedef float f4u __attribute__((ext_vector_type(4), aligned(8)));
        const f4u *q = reinterpret_cast<const f4u *>(p);
#pragma unroll
        for (int k = 0; k < S / 2; k++) {
            f4u t = q[k];
            x[2 * k] = make_float2(t.x, t.y);
            x[2 * k + 1] = make_float2(t.z, t.w);
        }
    } else {
#pragma unroll
        for (int k = 0; k < S; k++) x[k] = p[k];
    }
}

// ---------------------------------------------------------------------------------
// LinearFit pieces shared by both kernels
// ---------------------------------------------------------------------------------
// LinearFit::reset() tail (cpp/psk_soft.cpp:110-122) on `len` values y(j), wave-parallel:
// ySum = sum y_j, xySum = sum fl32(fl32(j*xdelta)*y_j) accumulated in double.
template <class YAt>
PSK_DEV void fit_rebuild_sums(YAt y_at, uint32_t len, float xdelta, double &ySum, double &xySum)
{
    const int lane = threadIdx.x & 63;
    double ys = 0.0, xys = 0.0;
    for (uint32_t j = lane; j < len; j += kWave) {
        float y = y_at(j);
        ys += (double)y;
        float jx = (float)j * xdelta;
        float jxy = jx * y;
        xys += (double)jxy;
    }
    ySum = wave_sum_f64(ys);
    xySum = wave_sum_f64(xys);
}

// ---------------------------------------------------------------------------------
// wave-scan kernel
// ---------------------------------------------------------------------------------
}  // namespace psk

#include "psk_fast_loop.h"

namespace psk {

// SV = samplesPerBaud this instantiation handles, HV = ceil(numAvg / 128) (blocks of window
// history kept in registers); SV == 0 takes the channels of the batch that emit nothing this
// call (warm-up, stalled window) whatever their samplesPerBaud / numAvg.
template <int SV, int HV>
__global__ __launch_bounds__(64, (HV == 1 ? PSK_WAVES_H1 : 1)) void psk_fast_kernel(const ChanPlan *__restrict__ plans, uint32_t ch0,
                                                      ChanState *__restrict__ states, float2 *__restrict__ rings,
                                                      uint32_t ring_cap, float *__restrict__ yvs, uint32_t fit_cap)
{
    __shared__ float yring[kYRing];
    const int lane = threadIdx.x & 63;
    const ChanPlan &p = plans[blockIdx.x];
    if (p.mode != PLAN_FAST)
        return;
    if (SV == 0 ? (p.n_out != 0)
                : (p.n_out == 0 || p.S != (uint32_t)SV || (p.A + (uint32_t)kB - 1u) / (uint32_t)kB != (uint32_t)HV))
        return;
    const uint32_t ch = ch0 + blockIdx.x;
    ChanState *st = &states[ch];
    float2 *ring_base = rings + (size_t)ch * 2u * ring_cap;
    const float2 *ring_src = ring_base + (size_t)p.ring_src * ring_cap;
    float2 *ring_dst = ring_base + (size_t)(p.ring_src ^ 1u) * ring_cap;
    float *yv = yvs + (size_t)ch * fit_cap;

    XView X;
    X.ring = ring_src;
    X.in = reinterpret_cast<const float2 *>(p.in);
    X.L0 = p.ring_len0;

    // ---- prologue: LinearFit history into the LDS ring; LinearFit::reset() sums if it ran ----
    const uint32_t len0 = p.lf_len0, n = p.lf_n;
    for (uint32_t j = lane; j < len0; j += kWave) yring[j & kYMask] = yv[(p.lf_head + j) % fit_cap];
    wave_lds_fence();
    FastCarry cy;
    cy.ySum = st->lf_ySum;
    cy.xySum = st->lf_xySum;
    cy.est = st->phaseEstimate;
    cy.last_re = st->last_re;
    cy.last_im = st->last_im;
    cy.den = st->lf_den;
    cy.xavg = st->lf_xavg;
    cy.q = len0;
    cy.last_k = st->last_k < p.S ? st->last_k : 0u;
    cy.umax = 0u;
    cy.umin1 = 0xFFFFFFFFu;
    cy.refuse = false;
    cy.stat_blocks = 0;
    cy.stat_extra = 0;
    if (p.lf_flags & LF_RECOMPUTE) {
        fit_rebuild_sums([&](uint32_t j) { return yring[j & kYMask]; }, len0, p.lf_xdelta, cy.ySum, cy.xySum);
        fit_denominator(p.lf_xdelta, len0, cy.den, cy.xavg);
        if (len0 > 1) {
            (void)fit_value(cy.ySum, cy.xySum, p.lf_xdelta, len0, cy.den, cy.xavg, cy.m, cy.b);
        } else {
            cy.m = 0.0f;
            cy.b = len0 ? yring[(len0 - 1) & kYMask] : 0.0f;
        }
    }

    // ---- the symbol loop ----
    if constexpr (SV != 0)
        fast_main_loop<SV, HV>(p, X, yring, cy);

    // ---- exactness guard (quirk Q8): float-valued energies summed in double are exact, hence
    //      order-independent, only while 24 + exponent spread + log2(#terms) <= 53 ----
    {
        unsigned umax = wave_max_u32(cy.umax);
        unsigned umin1 = wave_min_u32(cy.umin1);
        if (umax >= 0x7F800000u)
            cy.refuse = true;  // inf / NaN energy
        if (umin1 != 0xFFFFFFFFu) {
            int emax = (int)(umax >> 23), emin = (int)((umin1 + 1u) >> 23);
            emax = emax < 1 ? 1 : emax;
            emin = emin < 1 ? 1 : emin;
            int terms_log2 = 32 - __builtin_clz((unsigned)(p.A + 2u * kB));
            if (24 + (emax - emin) + terms_log2 > 52)
                cy.refuse = true;
        }
        cy.refuse = __any(cy.refuse);
    }
    if (cy.refuse) {
        if (lane == 0)
            st->guard = 1u;  // nothing committed: psk_seq_kernel redoes this call from the old state
        return;
    }

    // ---- end-of-call wrap (cpp/psk_soft.cpp:592-603) ----
    const uint32_t grown = len0 + (uint32_t)p.n_out;  // n_out <= 2^20 on this path
    const uint32_t len1 = grown < n ? grown : n;
    const uint32_t first = cy.q - len1;  // ring position of yvals.front()
    float pe = cy.est;
    const float wrapValue = (float)(kTwoPi * (double)p.M);
    uint32_t count1 = 0;
    if (wrap_test(pe, wrapValue)) {
        float qv = pe / wrapValue;
        long long numWraps = to_long_x86(__builtin_round((double)qv));
        float cst = (float)numWraps * wrapValue;
        for (uint32_t j = lane; j < len1; j += kWave) {  // LinearFit::subtractConst :126-133
            float v = yring[(first + j) & kYMask];
            yring[(first + j) & kYMask] = v - cst;
        }
        wave_lds_fence();
        fit_rebuild_sums([&](uint32_t j) { return yring[(first + j) & kYMask]; }, len1, p.lf_xdelta, cy.ySum, cy.xySum);
        fit_denominator(p.lf_xdelta, len1, cy.den, cy.xavg);
        if (len1 > 1) {
            pe = fit_value(cy.ySum, cy.xySum, p.lf_xdelta, len1, cy.den, cy.xavg, cy.m, cy.b);
        } else {
            cy.m = 0.0f;
            cy.b = len1 ? yring[(first + len1 - 1) & kYMask] : 0.0f;
            pe = cy.b;
        }
        count1 = 1;  // informational only: the host mirrors LinearFit::count
    }
    (void)count1;

    // ---- commit the channel state ----
    {
        const uint32_t dropped = grown - len1;
        const uint32_t head1 = (uint32_t)(((uint64_t)p.lf_head + dropped) % fit_cap);
        for (uint32_t j = lane; j < len1; j += kWave) yv[(head1 + j) % fit_cap] = yring[(first + j) & kYMask];
        const uint64_t drop = p.n_out * (uint64_t)p.S;  // samples popped by the emissions (:579-580)
        for (uint32_t j = lane; j < p.ring_len1; j += kWave) ring_dst[j] = x_at(X, drop + j);
        if (lane == 0) {
            st->lf_ySum = cy.ySum;
            st->lf_xySum = cy.xySum;
            st->last_re = cy.last_re;
            st->last_im = cy.last_im;
            st->phaseEstimate = pe;
            st->lf_den = cy.den;
            st->lf_xavg = cy.xavg;
            st->guard = 0u;
            st->last_k = cy.last_k;
            st->stat_blocks = cy.stat_blocks;
            st->stat_extra = cy.stat_extra;
        }
    }
}

// ---------------------------------------------------------------------------------
// reference-order kernel
// ---------------------------------------------------------------------------------
struct SeqFit {  // class LinearFit (cpp/psk_soft.h:33-53) on the circular HBM buffer
    float *yv;
    uint32_t cap, head, len, n;
    float xdelta, den, xavg, m, b;
    double ySum, xySum;
    uint32_t count;

    __device__ float at(uint32_t j) const { return yv[(head + j) % cap]; }
    __device__ float calc_fit()
    {
        if (len > 1)
            return fit_value(ySum, xySum, xdelta, len, den, xavg, m, b);
        m = 0.0f;
        b = len ? at(len - 1) : 0.0f;
        return b;
    }
    __device__ float reset_sums()  // cpp/psk_soft.cpp:110-122
    {
        ySum = 0.0;
        xySum = 0.0;
        for (uint32_t j = 0; j < len; j++) {
            float y = at(j);
            ySum += (double)y;
            float jx = (float)j * xdelta;
            float jxy = jx * y;
            xySum += (double)jxy;
        }
        fit_denominator(xdelta, len, den, xavg);
        count = 0;
        return calc_fit();
    }
    __device__ float next(float yval)  // cpp/psk_soft.cpp:48-87
    {
        if (count == kResyncCount)
            reset_sums();
        bool steady = (len == n);
        if (steady) {
            ySum -= (double)at(0);
            head = (head + 1) % cap;
            len--;
            xySum -= (double)xdelta * ySum;
        }
        ySum += (double)yval;
        float t = yval * (float)len;
        t = t * xdelta;
        xySum += (double)t;
        yv[(head + len) % cap] = yval;
        len++;
        if (!steady)
            fit_denominator(xdelta, len, den, xavg);
        count++;
        return calc_fit();
    }
    __device__ float subtract_const(float c)  // cpp/psk_soft.cpp:126-133
    {
        for (uint32_t j = 0; j < len; j++) {
            uint32_t pos = (head + j) % cap;
            yv[pos] = yv[pos] - c;
        }
        return reset_sums();
    }
};

struct SeqEmit {
    const ChanPlan *p;
    SeqFit fit;
    float pe;
    cf32 last;
    uint64_t n_emit;
};

// the per-symbol body, cpp/psk_soft.cpp:471-566, in the reference's statement order
__device__ void seq_emit_symbol(SeqEmit &E, cf32 sample, int sampleIndex, bool have_index)
{
    const ChanPlan &p = *E.p;
    const uint64_t i = E.n_emit++;
    if (have_index && p.sidx)
        p.sidx[i] = (int16_t)(unsigned short)sampleIndex;
    cf32 pw = cpow_uint<true>(sample, p.M);
    double thisPhase = (double)lm_atan2f(pw.im, pw.re);
    long long numWraps = unwrap_count(E.pe, thisPhase);
    thisPhase += (double)numWraps * kTwoPi;
    E.pe = E.fit.next((float)thisPhase);
    if (p.phase)
        p.phase[i] = E.pe;
    float phaseCorrection = 0.0f;
    if (p.diff) {
        cf32 decoded = cdiv(sample, E.last);
        E.last = sample;
        sample = decoded;
    } else {
        phaseCorrection = -E.pe / (float)p.M;
    }
    if (p.M == 4)
        phaseCorrection = (float)((double)phaseCorrection + kPi4);
    float sn, cs;
    lm_sincosf(phaseCorrection, &sn, &cs);
    cf32 ph;
    ph.re = 1.0f * cs;
    ph.im = 1.0f * sn;
    cf32 corr = cmul<true>(sample, ph);
    if (p.soft) {
        p.soft[2 * i] = corr.re;
        p.soft[2 * i + 1] = corr.im;
    }
    if (!p.bits) {
    } else if (p.bpb == 1) {
        p.bits[i] = (int16_t)(corr.re < 0);
    } else if (p.bpb == 2) {
        int r = (corr.re != 0), im = (corr.im != 0);
        p.bits[2 * i] = (int16_t)(r ^ im);
        p.bits[2 * i + 1] = (int16_t)(!im);
    } else if (p.bpb == 3) {
        unsigned short sym = slice_8psk(corr.re, corr.im);
        for (int j = 0; j != 3; j++) {
            p.bits[3 * i + j] = (int16_t)(sym & 1);
            sym = sym >> 1;
        }
    }
}

__global__ __launch_bounds__(64) void psk_seq_kernel(const ChanPlan *__restrict__ plans, uint32_t ch0,
                                                     ChanState *__restrict__ states, float2 *__restrict__ rings,
                                                     uint32_t ring_cap, float *__restrict__ yvs, uint32_t fit_cap)
{
    __shared__ double symE[kSeqMaxS];
    __shared__ float2 chunk[kSeqChunk];
    __shared__ uint64_t sh_head;
    const int lane = threadIdx.x & 63;
    const ChanPlan &p = plans[blockIdx.x];
    const uint32_t ch = ch0 + blockIdx.x;
    ChanState *st = &states[ch];
    const bool redo = (p.mode == PLAN_FAST) && (st->guard != 0u);
    if (!(p.mode == PLAN_SEQ || p.mode == PLAN_SEQ_S1 || redo))
        return;
    float2 *ring_base = rings + (size_t)ch * 2u * ring_cap;
    const float2 *ring_src = ring_base + (size_t)p.ring_src * ring_cap;
    float2 *ring_dst = ring_base + (size_t)(p.ring_src ^ 1u) * ring_cap;
    XView X;
    X.ring = ring_src;
    X.in = reinterpret_cast<const float2 *>(p.in);
    X.L0 = p.ring_len0;
    const uint32_t S = p.S;
    const uint64_t N = p.n_in;
    const uint64_t D = (uint64_t)S * p.A;

    SeqEmit E;
    uint32_t idx = 0, count = p.count0;
    uint64_t head = 0, size = X.L0;
    if (lane == 0) {
        E.p = &p;
        E.fit.yv = yvs + (size_t)ch * fit_cap;
        E.fit.cap = fit_cap;
        E.fit.head = p.lf_head;
        E.fit.len = p.lf_len0;
        E.fit.n = p.lf_n;
        E.fit.xdelta = p.lf_xdelta;
        E.fit.den = st->lf_den;
        E.fit.xavg = st->lf_xavg;
        E.fit.m = st->lf_m;
        E.fit.b = st->lf_b;
        E.fit.ySum = st->lf_ySum;
        E.fit.xySum = st->lf_xySum;
        E.fit.count = p.lf_count0;
        E.pe = st->phaseEstimate;
        E.last.re = st->last_re;
        E.last.im = st->last_im;
        E.n_emit = 0;
        if (p.lf_flags & LF_RECOMPUTE)
            E.fit.reset_sums();
        if (p.mode != PLAN_SEQ_S1) {  // resyncEnergy ran in this call's prologue, :619-636
            for (uint32_t k = 0; k < S; k++) symE[k] = 0.0;
            for (uint64_t j = 0; j < X.L0; j++) {
                float2 v = X.ring[j];
                symE[idx] += (double)norm_f(v.x, v.y);
                idx++;
                if (idx == S)
                    idx = 0;
            }
            count = 0;
        }
    }
    for (uint64_t base = 0; base < N; base += kSeqChunk) {
        const uint32_t cnt = (N - base) < (uint64_t)kSeqChunk ? (uint32_t)(N - base) : (uint32_t)kSeqChunk;
        __syncthreads();
        for (uint32_t j = lane; j < cnt; j += kWave) chunk[j] = X.in[base + j];
        __syncthreads();
        if (lane == 0) {
            for (uint32_t jj = 0; jj < cnt; jj++) {
                float2 v = chunk[jj];
                cf32 cur;
                cur.re = v.x;
                cur.im = v.y;
                if (p.mode == PLAN_SEQ_S1) {  // samplesPerBaud == 1, :468-469
                    seq_emit_symbol(E, cur, 0, false);
                    continue;
                }
                symE[idx] += (double)norm_f(cur.re, cur.im);  // :447-451
                size++;
                if (idx == S - 1) {  // :454
                    if (size == D) {  // :457
                        uint32_t best = 0;
                        for (uint32_t k = 1; k < S; k++)
                            if (symE[best] < symE[k])
                                best = k;
                        float2 pk = x_at(X, head + best);
                        cf32 smp;
                        smp.re = pk.x;
                        smp.im = pk.y;
                        seq_emit_symbol(E, smp, (int)best, true);
                        for (uint32_t k = 0; k < S; k++) {  // :572-577
                            float2 o = x_at(X, head + k);
                            symE[k] -= (double)norm_f(o.x, o.y);
                        }
                        head += S;  // :579-580
                        size -= S;
                        count++;
                        if (count == kResyncCount) {  // :582-583
                            for (uint32_t k = 0; k < S; k++) symE[k] = 0.0;
                            uint32_t ix = 0;
                            for (uint64_t j = 0; j < size; j++) {
                                float2 o = x_at(X, head + j);
                                symE[ix] += (double)norm_f(o.x, o.y);
                                ix++;
                                if (ix == S)
                                    ix = 0;
                            }
                            count = 0;
                        }
                    }
                    idx = 0;  // :587
                } else {
                    idx++;  // :590
                }
            }
        }
    }
    if (lane == 0) {
        // end-of-call wrap, cpp/psk_soft.cpp:592-603
        const float wrapValue = (float)(kTwoPi * (double)p.M);
        if (wrap_test(E.pe, wrapValue)) {
            float qv = E.pe / wrapValue;
            long long numWraps = to_long_x86(__builtin_round((double)qv));
            E.pe = E.fit.subtract_const((float)numWraps * wrapValue);
        }
        st->lf_ySum = E.fit.ySum;
        st->lf_xySum = E.fit.xySum;
        st->last_re = E.last.re;
        st->last_im = E.last.im;
        st->phaseEstimate = E.pe;
        st->lf_den = E.fit.den;
        st->lf_xavg = E.fit.xavg;
        st->lf_m = E.fit.m;
        st->lf_b = E.fit.b;
        st->stat_blocks = 0;
        st->stat_extra = 0;
        st->guard = redo ? 2u : 0u;  // 2 = "the guard sent this call here" (statistics)
        sh_head = head;
    }
    __syncthreads();
    if (p.mode != PLAN_SEQ_S1) {
        const uint64_t h = sh_head;
        for (uint32_t j = lane; j < p.ring_len1; j += kWave) ring_dst[j] = x_at(X, h + j);
    }
}

}  // namespace psk

// ---------------------------------------------------------------------------------
// launchers (called from psk_capi.cpp through plain C++ declarations)
// ---------------------------------------------------------------------------------
namespace psk {
// S = 0 launches the append-only variant; otherwise S in {2,4,5,8,10,16} and H = ceil(numAvg/128) in {1,2,4}
// (H = 3 runs on the H = 4 instantiation's sibling below).
hipError_t launch_fast(int S, int H, const ChanPlan *plans, uint32_t ch0, uint32_t nch, ChanState *states,
                       float2 *rings, uint32_t ring_cap, float *yvs, uint32_t fit_cap, hipStream_t stream)
{
    if (!nch)
        return hipSuccess;
#define PSK_LAUNCH(SV, HV)                                                                                          \
    hipLaunchKernelGGL((psk_fast_kernel<SV, HV>), dim3(nch), dim3(kWave), 0, stream, plans, ch0, states, rings,      \
                       ring_cap, yvs, fit_cap)
#define PSK_LAUNCH_H(SV)                 \
    switch (H) {                         \
    case 1: PSK_LAUNCH(SV, 1); break;    \
    case 2: PSK_LAUNCH(SV, 2); break;    \
    case 3: PSK_LAUNCH(SV, 3); break;    \
    case 4: PSK_LAUNCH(SV, 4); break;    \
    default: return hipErrorInvalidValue; \
    }
#ifdef PSK_ONLY_S8H1
    if (S == 8 && H == 1) { PSK_LAUNCH(8, 1); return hipGetLastError(); }
    if (S == 0) { PSK_LAUNCH(0, 1); return hipGetLastError(); }
    return hipErrorInvalidValue;
#else
    switch (S) {
    case 0: PSK_LAUNCH(0, 1); break;
    case 2: PSK_LAUNCH_H(2); break;
    case 4: PSK_LAUNCH_H(4); break;
    case 5: PSK_LAUNCH_H(5); break;
    case 8: PSK_LAUNCH_H(8); break;
    case 10: PSK_LAUNCH_H(10); break;
    case 16: PSK_LAUNCH_H(16); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
#endif
#undef PSK_LAUNCH_H
#undef PSK_LAUNCH
}
hipError_t launch_seq(const ChanPlan *plans, uint32_t ch0, uint32_t nch, ChanState *states, float2 *rings,
                      uint32_t ring_cap, float *yvs, uint32_t fit_cap, hipStream_t stream)
{
    if (!nch)
        return hipSuccess;
    hipLaunchKernelGGL(psk_seq_kernel, dim3(nch), dim3(kWave), 0, stream, plans, ch0, states, rings, ring_cap, yvs,
                       fit_cap);
    return hipGetLastError();
}
}  // namespace psk
