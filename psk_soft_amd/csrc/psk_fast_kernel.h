// psk_fast_kernel.h -- the wave-scan kernel: per-call prologue (LinearFit history, reset sums),
// the symbol loop (psk_fast_loop.h), exactness guard, end-of-call wrap, state commit.
#ifndef PSK_FAST_KERNEL_H
#define PSK_FAST_KERNEL_H

#include "psk_fast_loop.h"

namespace psk {

// number of 128-symbol blocks of window history an instantiation keeps in registers for numAvg = A
PSK_HD int hist_blocks_for(uint32_t A) { return A <= 128u ? 1 : A <= 256u ? 2 : A <= 512u ? 4 : 8; }

// Per-call prologue of a channel (shared with the time-tiled fit kernel, psk_tile_kernel.h): LinearFit history into
// the LDS ring, the carried state into `cy`; LinearFit::reset() sums if it ran.
PSK_DEV void call_prologue(const ChanPlan &p, const ChanState *st, const float *yv, uint32_t fit_cap, float *yring, uint32_t ymask,
                           int lane, FastCarry &cy)
{
    const uint32_t len0 = p.lf_len0;
    for (uint32_t j = lane; j < len0; j += kWave) yring[j & ymask] = yv[(p.lf_head + j) % fit_cap];
    wave_lds_fence();
    cy.ySum = st->lf_ySum;
    cy.xySum = st->lf_xySum;
    cy.est = st->phaseEstimate;
    cy.last_re = st->last_re;
    cy.last_im = st->last_im;
    cy.den = st->lf_den;
    cy.xavg = st->lf_xavg;
    cy.slope = is_fin(st->lf_m) ? st->lf_m : 0.0f;  // (a hint only: the slope of the last fit, per symbol)
    cy.q = len0;
    cy.last_k = st->last_k < p.S ? st->last_k : 0u;
    cy.umax = 0u;
    cy.umin1 = 0xFFFFFFFFu;
    cy.wmax = 0.0f;
    cy.ambiguous = false;
    cy.refuse = false;
    cy.stat_blocks = 0;
    cy.stat_extra = 0;
    cy.stat_exact_blocks = 0;
    cy.stat_chain = 0;
    cy.chain_run = 0;
    cy.chain_streak = 0;
    if (p.lf_flags & LF_RECOMPUTE) {
        fit_rebuild_sums([&](uint32_t j) { return yring[j & ymask]; }, len0, p.lf_xdelta, cy.ySum, cy.xySum);
        fit_denominator(p.lf_xdelta, len0, cy.den, cy.xavg);
        if (len0 > 1) {
            (void)fit_value(cy.ySum, cy.xySum, p.lf_xdelta, len0, cy.den, cy.xavg, cy.m, cy.b);
        } else {
            cy.m = 0.0f;
            cy.b = len0 ? yring[(len0 - 1) & ymask] : 0.0f;
        }
    }
}

// End of a call that ran to completion (shared with the time-tiled fit kernel): end-of-call wrap, then the channel
// state is committed -- LinearFit history and sums, the surviving samples into the other ring buffer, `guard_done` into
// ChanState::guard.
PSK_DEV void call_epilogue(const ChanPlan &p, ChanState *st, float *yv, uint32_t fit_cap, float *yring, uint32_t ymask, const XView &X,
                           float2 *ring_dst, int lane, FastCarry &cy, uint32_t guard_done)
{
    const uint32_t len0 = p.lf_len0, n = p.lf_n;
    // ---- end-of-call wrap (cpp/psk_soft.cpp:592-603) ----
    const uint32_t grown = len0 + (uint32_t)p.n_out;  // n_out <= 2^20 on this path
    const uint32_t len1 = grown < n ? grown : n;
    const uint32_t first = cy.q - len1;  // ring position of yvals.front()
    float pe = cy.est;
    const float wrapValue = (float)(kTwoPi * (double)p.M);
    uint32_t count1 = 0;
    if (!(p.lf_flags & PLAN_NO_WRAP) && wrap_test(pe, wrapValue)) {
        float qv = pe / wrapValue;
        long long numWraps = to_long_x86(__builtin_round((double)qv));
        float cst = (float)numWraps * wrapValue;
        for (uint32_t j = lane; j < len1; j += kWave) {  // LinearFit::subtractConst :126-133
            float v = yring[(first + j) & ymask];
            yring[(first + j) & ymask] = v - cst;
        }
        wave_lds_fence();
        fit_rebuild_sums([&](uint32_t j) { return yring[(first + j) & ymask]; }, len1, p.lf_xdelta, cy.ySum, cy.xySum);
        fit_denominator(p.lf_xdelta, len1, cy.den, cy.xavg);
        if (len1 > 1) {
            pe = fit_value(cy.ySum, cy.xySum, p.lf_xdelta, len1, cy.den, cy.xavg, cy.m, cy.b);
        } else {
            cy.m = 0.0f;
            cy.b = len1 ? yring[(first + len1 - 1) & ymask] : 0.0f;
            pe = cy.b;
        }
        count1 = 1;  // informational only: the host mirrors LinearFit::count
    }
    (void)count1;

    // ---- commit the channel state ----
    {
        const uint32_t dropped = grown - len1;
        const uint32_t head1 = (uint32_t)(((uint64_t)p.lf_head + dropped) % fit_cap);
        for (uint32_t j = lane; j < len1; j += kWave) yv[(head1 + j) % fit_cap] = yring[(first + j) & ymask];
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
            st->lf_m = cy.slope;
            st->guard = guard_done;
            st->last_k = cy.last_k;
            st->stat_blocks = cy.stat_blocks;
            st->stat_extra = cy.stat_extra;
            st->stat_exact = cy.stat_exact_blocks;
            st->stat_chain = cy.stat_chain;
            st->stat_pfit = 0u;
        }
    }
}

// SV = samplesPerBaud this instantiation handles, HV = blocks of window history kept in
// registers (numAvg <= 128 HV), EXACT = timing by the exact double pass (else the float
// screening pass, see psk_fast_loop.h).  SV == 0 takes the channels of the batch that emit
// nothing this call (warm-up, stalled window) whatever their samplesPerBaud / numAvg.
// Register budget: the numAvg <= 128, samplesPerBaud <= 10 instantiations are held to 128 VGPRs
// (4 waves per SIMD = 16 single-wave workgroups per CU, so a 4096-channel batch is resident at
// once; samplesPerBaud = 9 and 10 pay for it with a handful of spilled VGPRs and win 20 % by the
// residency); the others keep what they need.
// Hand-over protocol through ChanState::guard: the screened kernel leaves 0 (done) or 1
// (refused); the exact kernel runs on 1 and leaves 3 (done) or 1; the reference-order kernel
// runs on 1 and leaves 2.  A call planned for the time-tiled kernels (PLAN_TILED) starts there: they
// leave 4 (done) or 1, and the screened kernel runs on 1 only.
#ifndef PSK_WAVES_PER_SIMD
#define PSK_WAVES_PER_SIMD 4
#endif
// two blocks of window history in registers (numAvg 129 .. 256): samplesPerBaud 8 and 10 come to 172 / 178 registers left to
// themselves, a hair over the 168 that let three waves share a SIMD instead of two -- and held to 168 they fit without a
// spill.  Measured SLOWER on a machine-filling batch (4096 channels, numAvg 200: 3.60 against 3.06 ms): 3072 waves and then
// 1024 is a full round and a third of one, two rounds of 2048 are two full ones.  Left at 1 (the compiler's own count).
// (no window history in registers, H == 0: see psk_fast_loop.h REREAD; samplesPerBaud <= 5 is bound by arithmetic and keeps four
// waves: samplesPerBaud 2 / 4 at two waves measured 12.0 / 6.8 ms against 6.9 / 4.4)
#ifndef PSK_WAVES_PER_SIMD_H0
#define PSK_WAVES_PER_SIMD_H0 2  /* (170 registers, nothing spilled, two full rounds of 2048 waves: numAvg 600 4.3 ms; 3: 5.9 -- a round and a third --; 4: 5.1, 37 registers spilled) */
#endif
#ifndef PSK_WAVES_PER_SIMD_H2
#define PSK_WAVES_PER_SIMD_H2 1
#endif
template <int SV, int HV, bool EXACT>
__global__ __launch_bounds__(64, ((HV == 0 && SV <= 5 && !EXACT) ? 4 : (HV == 0 && SV <= 10 && !EXACT) ? PSK_WAVES_PER_SIMD_H0 : (HV == 1 && SV <= 10 && !EXACT) ? PSK_WAVES_PER_SIMD : (HV == 2 && SV <= 10 && !EXACT) ? PSK_WAVES_PER_SIMD_H2 : (HV == 4 && SV <= 8 && !EXACT && PSK_DBUF) ? 2 : 1)) void psk_fast_kernel(const ChanPlan *__restrict__ plans, const uint32_t *__restrict__ list, uint32_t ch0,
                                                      ChanState *__restrict__ states, float2 *__restrict__ rings,
                                                      uint32_t ring_cap, float *__restrict__ yvs, uint32_t fit_cap,
                                                      uint32_t y_len, uint32_t r_len)
{
    // LDS: a ring of the last unwrapped phases (y_len floats, a power of two >= phaseAvg + 128 for every
    // channel of the launch, sized by the host: dynamic LDS) and, for numAvg <= 128, the energy ring of
    // SV rows -- 256 positions each, static, except samplesPerBaud = 9, 10 where the rows follow the
    // phase ring in the dynamic segment at r_len positions each (see psk_fast_loop.h).
    extern __shared__ __attribute__((aligned(16))) float lds_dyn[];
    constexpr bool kDyn = ering_dynamic(SV);
    float *const yring = lds_dyn;
    const uint32_t ymask = y_len - 1u;
    ERingT<kDyn> er;
    if constexpr (kDyn) {
        er.mem = lds_dyn + y_len;
        er.set_len((int)r_len);
    } else {
        __shared__ __attribute__((aligned(16))) float ering_s[(SV != 0 && HV == 1) ? SV * kERing : 4];
        er.mem = ering_s;
    }
    const int lane = threadIdx.x & 63;
#ifdef PSK_DIAG_STAMP  /* (diagnostic builds only: when each wave started and ended, 100 MHz ticks, into two statistics words) */
    const uint32_t diag_t0 = (uint32_t)wall_clock64();
#endif
#ifdef PSK_STAGGER  /* (experiment: waves of a launch start up to PSK_STAGGER * 0.43 us apart instead of all at once) */
    if (SV != 0 && !EXACT) {
        const uint32_t hsh = (blockIdx.x * 2654435761u) >> 24;
        for (uint32_t i = 0; i < (hsh * (uint32_t)(PSK_STAGGER)) >> 8; i++) __builtin_amdgcn_s_sleep(16);
    }
#endif
    if constexpr (EXACT) {
        if (plan_header(plans)[0] == 0u)
            return;  // nothing was handed over in this call so far (psk_plan.h)
    }
    // the launch covers the channels of the batch that this instantiation handles: list[workgroup] = index into the batch
    const uint32_t bi = list[blockIdx.x];
    const ChanPlan &p = plans[bi];
    if (p.mode != PLAN_FAST)
        return;
    if (SV == 0 ? (p.n_out != 0) : (p.n_out == 0 || p.S != (uint32_t)SV || (HV == 0 ? p.A <= (uint32_t)kB : hist_blocks_for(p.A) != HV)))
        return;
    if (EXACT && states[ch0 + bi].guard != 1u)
        return;  // the screened kernel finished this channel's call
    if (!EXACT && SV != 0 && (p.lf_flags & PLAN_TILED) && states[ch0 + bi].guard != 1u)
        return;  // ... or the time-tiled kernels did, launched in front of this one (psk_tile_kernel.h)
    const uint32_t ch = ch0 + bi;
    ChanState *st = &states[ch];
    float2 *ring_base = rings + (size_t)ch * 2u * ring_cap;
    const float2 *ring_src = ring_base + (size_t)p.ring_src * ring_cap;
    float2 *ring_dst = ring_base + (size_t)(p.ring_src ^ 1u) * ring_cap;
    float *yv = yvs + (size_t)ch * fit_cap;

    XView X;
    X.ring = reinterpret_cast<const f2g *>(ring_src);
    X.in = reinterpret_cast<const f2g *>(p.in);
    X.L0 = p.ring_len0;

    FastCarry cy;
    call_prologue(p, st, yv, fit_cap, yring, ymask, lane, cy);
#ifdef PSK_DIAG_STAMP2  /* (diagnostic builds only: where a short call's time goes -- after the prologue, after the loop) */
    const uint32_t diag_t1 = (uint32_t)wall_clock64();
#endif

    // ---- the symbol loop ----
    if constexpr (SV != 0) {
        // (a later piece of a call cut in time starts from the largest window sum the earlier pieces met: PLAN_CARRY_DRIFT)
        const float carried = (p.lf_flags & PLAN_CARRY_DRIFT) ? __uint_as_float(st->pad_state) : 0.0f;
        fast_main_loop<SV, HV, EXACT>(p, X, yring, ymask, er, cy, 0, -1, nullptr, nullptr, carried >= 0.0f ? carried : __builtin_inff());
        // (... and leaves its own for the piece behind it -- written here, not with the state: kept until the epilogue the value
        // cost samplesPerBaud 10 five per cent; a call that is refused further down is redone by a kernel that writes it again)
        if (lane == 0)
            st->pad_state = __float_as_uint(cy.wmax * 1.00001f);
        if constexpr (PSK_PACE_ON(false, EXACT))
            pace_post(pace_key(), 0u, lane);  // (on every way out of the loop: nothing left, whoever takes this wave slot next)
    }

#ifdef PSK_DIAG_STAMP2
    const uint32_t diag_t2 = (uint32_t)wall_clock64();
#endif
    // ---- exactness guard (quirk Q8): float-valued energies summed in double are exact, hence
    //      order-independent, only while 24 + exponent spread + log2(#terms) <= 53.  It matters only
    //      for calls in which an exact-timing pass met a best / runner-up pair closer than accumulated
    //      rounding could explain (cy.ambiguous, psk_fast_loop.h): everywhere else the argmax is the
    //      reference's whether or not the sums are exact ----
    {
        unsigned umax = wave_max_u32(cy.umax);
        unsigned umin1 = wave_min_u32(cy.umin1);
        if (umin1 != 0xFFFFFFFFu && __any(cy.ambiguous)) {
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
        if (lane == 0) {
            st->guard = 1u;  // nothing committed: psk_seq_kernel redoes this call from the old state
            if (!EXACT)  // (a call the exact tier refuses was counted when it was handed to it)
#ifdef PSK_HANDOVER_VIA_ARG  /* (A/B builds: the variant measured 10 % slower on one box) */
                atomicAdd(plan_header(plans), 1u);
#else
                atomicAdd(p.handed_over, 1u);
#endif
        }
        return;
    }

    call_epilogue(p, st, yv, fit_cap, yring, ymask, X, ring_dst, lane, cy, EXACT ? 3u : 0u);
#ifdef PSK_DIAG_STAMP
    if (lane == 0) {
#ifdef PSK_DIAG_HWID  /* (where the wave ran: HW_ID (register 4) and XCC_ID (register 20)) */
        st->stat_pfit = (uint32_t)__builtin_amdgcn_s_getreg((31 << 11) | 4);
        st->emax_hint = __uint_as_float((uint32_t)__builtin_amdgcn_s_getreg((31 << 11) | 20));
#else
        st->stat_pfit = diag_t0;
#endif
#ifdef PSK_DIAG_STAMP2
        st->emax_hint = __uint_as_float(diag_t1);
        st->stat_chain = diag_t2;
#endif
        st->pad_state = (uint32_t)wall_clock64();
    }
#endif
}

#define PSK_FAST_ARGS                                                                                          \
    const ChanPlan *plans, const uint32_t *list, uint32_t ch0, uint32_t nch, ChanState *states, float2 *rings, uint32_t ring_cap,    \
        float *yvs, uint32_t fit_cap, uint32_t y_len, uint32_t r_len, hipStream_t stream

// Dynamic LDS beyond the 64 KiB a kernel gets by default (a phase ring for phaseAvg in the thousands) has to be asked
// for, per kernel, device and size; LdsGrant is the kernel's own record of what it has asked for so far.
struct LdsGrant {
    size_t bytes[32] = {};  // per device
};
inline hipError_t lds_grant(const void *kernel, size_t bytes, LdsGrant &granted)
{
    if (bytes <= 65536)
        return hipSuccess;
    int dev = 0;
    if (const hipError_t e = hipGetDevice(&dev))
        return e;
    size_t &have = granted.bytes[dev & 31];
    if (bytes <= have)
        return hipSuccess;
    const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess)
        have = bytes;
    return e;
}

template <int SV, int HV, bool EXACT>
hipError_t launch_fast_inst(PSK_FAST_ARGS)
{
    if (!nch)
        return hipSuccess;
    const size_t lds_bytes = sizeof(float) * ((size_t)y_len + (ering_dynamic(SV) ? (size_t)SV * r_len : 0));
    static LdsGrant granted;
    if (const hipError_t e = lds_grant(reinterpret_cast<const void *>(&psk_fast_kernel<SV, HV, EXACT>), lds_bytes, granted))
        return e;
    hipLaunchKernelGGL((psk_fast_kernel<SV, HV, EXACT>), dim3(nch), dim3(kWave), lds_bytes, stream, plans, list, ch0, states,
                       rings, ring_cap, yvs, fit_cap, y_len, r_len);
    return hipGetLastError();
}

}  // namespace psk
#endif
