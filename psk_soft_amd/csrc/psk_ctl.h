// psk_ctl.h -- host-side control plane of one channel (= one psk_soft_i instance).
//
// Mirrors, value for value, the part of the reference's state that does not depend on
// sample values: the six properties, the three reset flags and their listeners
// (reference cpp/psk_soft.cpp:187-199, 638-651), samples.size(), symbolEnergy.size(),
// index, count, sampleRate, and LinearFit's n / xdelta / yvals.size() / count
// (cpp/psk_soft.h:43-52, 66-86).  plan_call() replays the control flow of one
// serviceFunction() call (cpp/psk_soft.cpp:353-426 prologue, the symbol clock of
// :454-457/:568-590 in closed form) and emits the ChanPlan the kernels execute.
// Pure C++, no HIP: unit-tested on the CPU against the oracle's counters.
#ifndef PSK_CTL_H
#define PSK_CTL_H

#include <stdint.h>

#include "psk_plan.h"
#include "psk_soft_hip.h"

namespace psk {

struct ChanCtl {
    // properties, defaults of cpp/psk_soft_base.cpp:96-148
    psk_soft_props_t props;
    // flags, cpp/psk_soft.cpp:191-193
    bool resetSamplesPerBaud = true;
    bool resetNumSymbols = true;
    bool resetPhaseAvg = true;
    // mirrored control state
    uint64_t ring_len = 0;        // samples.size() (may exceed the device ring in the stalled state)
    uint64_t symEnergySize = 10;  // symbolEnergy.size(), cpp/psk_soft.cpp:189
    uint64_t index = 0;           // cpp/psk_soft.cpp:190
    uint64_t count = 0;           // cpp/psk_soft.cpp:196
    float sampleRate = 1.0f;      // cpp/psk_soft.cpp:195
    uint64_t lf_n = 50;           // LinearFit(phaseAvg, sampleRate), cpp/psk_soft.cpp:197
    float lf_xdelta = 1.0f;       // 1.0/sampleRate, cpp/psk_soft.cpp:41
    uint64_t lf_len = 0;          // yvals.size()
    uint64_t lf_count = 0;
    // device bookkeeping
    uint32_t lf_head = 0;         // oldest entry of the circular yvals buffer in HBM
    uint32_t ring_src = 0;        // which ping-pong ring buffer is current
    bool lf_recompute_pending = false;  // a LinearFit::reset() was planned while no kernel ran

    ChanCtl()
    {
        props.samplesPerBaud = 10;
        props.constelationSize = 4;
        props.numAvg = 100;
        props.phaseAvg = 50;
        props.differentialDecoding = 0;
        props.resetState = 0;
    }

    // the three registered listeners, cpp/psk_soft.cpp:638-651
    void samplesPerBaudChanged() { resetSamplesPerBaud = (props.samplesPerBaud != symEnergySize); }
    void constelationSizeChanged() { resetNumSymbols = true; }
    void phaseAvgChanged() { resetPhaseAvg = true; }

    // configure(): store, then fire the listener of every property whose value changed
    void configure(const psk_soft_props_t &p)
    {
        const psk_soft_props_t old = props;
        props = p;
        if (old.samplesPerBaud != p.samplesPerBaud) samplesPerBaudChanged();
        if (old.constelationSize != p.constelationSize) constelationSizeChanged();
        if (old.phaseAvg != p.phaseAvg) phaseAvgChanged();
    }
};

struct Limits {
    uint32_t ring_cap;  // samples per ring buffer
    uint32_t fit_cap;   // floats in the circular yvals buffer
    uint32_t fast_fit_max;   // largest phaseAvg the wave-scan kernel holds in LDS
    bool force_seq;
};

// instantiations of the wave-scan kernel (window history of 1, 2, 4 or 8 blocks of 128 symbols in
// registers): samplesPerBaud 2 .. 16 with numAvg <= 1024, 17 .. 32 with numAvg <= 512.  (The deep
// histories of the wide symbols spill hundreds of registers and still run two orders of magnitude
// faster than the reference-order kernel.)
inline bool fast_kernel_has(uint32_t S, uint32_t A)
{
    if (S < 2 || S > 32)
        return false;
    return A <= (S <= 16 ? 1024u : 512u);
}
// window classes without an instantiation go through the time-tiled kernels behind a front stage that takes samplesPerBaud
// and numAvg at run time (psk_tile.hip: psk_tile_front_any_kernel; up to 16 timing phases per lane)
inline bool any_front_has(uint32_t S) { return S >= 2 && S <= 1024; }
// history blocks of the instantiation that takes a window of numAvg symbols
inline int fast_hist_blocks(uint32_t A) { return A <= 128u ? 1 : A <= 256u ? 2 : A <= 512u ? 4 : 8; }

// a / d and a % d for the symbol clock: 64-bit divisions cost the control plane more than everything else in plan_call (four
// of them per channel and call); samplesPerBaud is a power of two more often than not, and the operands fit 32 bits otherwise
inline uint64_t ctl_div(uint64_t a, uint64_t d)
{
    if ((d & (d - 1)) == 0)
        return a >> __builtin_ctzll(d);
    if (((a | d) >> 32) == 0)
        return (uint32_t)a / (uint32_t)d;
    return a / d;
}
inline uint64_t ctl_mod(uint64_t a, uint64_t d)
{
    if ((d & (d - 1)) == 0)
        return a & (d - 1);
    if (((a | d) >> 32) == 0)
        return (uint32_t)a % (uint32_t)d;
    return a % d;
}

// LinearFit::reset(numPts, sampleRate, forceHistoryClear) on the control state,
// cpp/psk_soft.cpp:89-124.  Returns true (the sums must be rebuilt).
inline void ctl_linfit_reset(ChanCtl &c, uint32_t fit_cap, const uint64_t *numPts, const float *sampleRate,
                             bool forceHistoryClear)
{
    if (sampleRate) {
        float newXdelta = (float)(1.0 / (double)*sampleRate);
        if (c.lf_xdelta != newXdelta) {
            c.lf_xdelta = newXdelta;
            forceHistoryClear = true;
        }
    }
    if (forceHistoryClear) {
        c.lf_len = 0;
        c.lf_head = 0;
    }
    if (numPts && *numPts != c.lf_n) {
        c.lf_n = *numPts;
        if (c.lf_len > c.lf_n) {  // pop_front until size <= n
            uint64_t dropped = c.lf_len - c.lf_n;
            c.lf_head = (uint32_t)((c.lf_head + dropped) % fit_cap);
            c.lf_len = c.lf_n;
        }
    }
    c.lf_count = 0;
    c.lf_recompute_pending = true;
}

// One serviceFunction() call, control flow only.  Fills `plan` (device work) and the
// result fields of `out`.  On a status other than PSK_SOFT_OK the caller discards `c`
// (it plans on a copy), so a refused call leaves the channel untouched.
inline uint64_t plan_lf_count0(const ChanCtl &c) { return c.lf_recompute_pending ? 0 : c.lf_count; }

// cont = true: the packet CONTINUES the serviceFunction() call of the packet before it -- the library cuts a call that emits
// more than 2^20 symbols (or that runs LinearFit::count past 2^20) at those boundaries and plans the pieces one after the other
// (psk_capi.cpp) -- so nothing of the call's prologue (cpp/psk_soft.cpp:353-426) runs again.
inline psk_soft_status plan_call(ChanCtl &c, const Limits &lim, const psk_soft_packet_t &pkt,
                                 psk_soft_output_t &out, ChanPlan &plan, bool cont = false)
{
    plan = ChanPlan();
    plan.mode = PLAN_SKIP;
    out.ret = PSK_SOFT_NORMAL;
    out.n_symbols = out.n_bits = out.n_sampleIndex = 0;
    out.sri_pushed = 0;
    out.sri_soft_xdelta = out.sri_bits_xdelta = 0.0;
    out.n_warn = 0;
    if (!pkt.present) {  // :350-352
        out.ret = PSK_SOFT_NOOP;
        return PSK_SOFT_OK;
    }
    if (pkt.inputQueueFlushed && !cont) {  // :353-357
        out.n_warn++;
        c.props.resetState = 1;
    }
    if (pkt.sri_mode != 1) {  // :359-363
        out.n_warn++;
        return PSK_SOFT_OK;
    }
    if (c.props.samplesPerBaud == 0 || c.props.phaseAvg == 0)
        return PSK_SOFT_ERR_UNSUPPORTED;
    if ((uint64_t)c.props.samplesPerBaud * c.props.numAvg > lim.ring_cap || c.props.phaseAvg > lim.fit_cap)
        return PSK_SOFT_ERR_LIMIT;
    if (c.props.resetState && !cont) {  // :365-372
        c.resetSamplesPerBaud = true;
        c.resetNumSymbols = true;
        c.resetPhaseAvg = true;
        c.props.resetState = 0;
    }
    // :376-390
    const uint64_t S = c.props.samplesPerBaud;
    const uint64_t D = S * (uint64_t)c.props.numAvg;
    const uint64_t M = c.props.constelationSize;
    if (D > c.ring_len && !cont)
        c.resetSamplesPerBaud = true;
    uint64_t bpb = 0;
    if (M == 2) bpb = 1;
    else if (M == 4) bpb = 2;
    else if (M == 8) bpb = 3;

    // :393-405
    if (!cont && (pkt.sriChanged || c.resetNumSymbols || c.resetSamplesPerBaud)) {
        double xdelta = pkt.sri_xdelta;
        if (xdelta != (double)c.sampleRate) {
            c.sampleRate = (float)(1.0 / xdelta);
            ctl_linfit_reset(c, lim.fit_cap, nullptr, &c.sampleRate, false);
        }
        xdelta *= (double)S;
        out.sri_pushed = 1;
        out.sri_soft_xdelta = xdelta;
        xdelta /= (double)bpb;
        out.sri_bits_xdelta = xdelta;
    }
    bool resynced = false;
    if (cont) {
        // (a continuation runs none of the three reset blocks: the flags were consumed by the call's first piece)
    } else if (c.resetSamplesPerBaud) {  // :408-412 -> resyncEnergy :619-636
        c.symEnergySize = S;
        if (c.ring_len > D)
            c.ring_len = D;
        c.index = ctl_mod(c.ring_len, S);
        c.count = 0;
        c.resetSamplesPerBaud = false;
        resynced = true;
    }
    if (c.resetNumSymbols && !cont) {  // :416-420
        ctl_linfit_reset(c, lim.fit_cap, nullptr, nullptr, true);
        c.resetNumSymbols = false;
    }
    if (c.resetPhaseAvg && !cont) {  // :421-426
        uint64_t numPts = c.props.phaseAvg;
        ctl_linfit_reset(c, lim.fit_cap, &numPts, nullptr, false);
        c.resetPhaseAvg = false;
    }

    const uint64_t N = pkt.n_floats / 2;  // :428
    const uint64_t dev_ring0 = c.ring_len < lim.ring_cap ? c.ring_len : lim.ring_cap;

    plan.in = pkt.data;
    plan.soft = out.soft;
    plan.bits = out.bits;
    plan.phase = out.phase;
    plan.sidx = out.sampleIndex;
    plan.n_in = N;
    plan.S = (uint32_t)S;
    plan.A = c.props.numAvg;
    plan.M = (uint32_t)M;
    plan.bpb = (uint32_t)bpb;
    plan.diff = c.props.differentialDecoding ? 1u : 0u;
    plan.ring_len0 = (uint32_t)dev_ring0;
    plan.ring_src = c.ring_src;
    plan.count0 = (uint32_t)c.count;

    // LinearFit::count standing at 1048576: the next next() opens with reset() (cpp/psk_soft.cpp:51-52) -- the sums rebuilt from
    // yvals, count = 0 -- which is what a plan flagged LF_RECOMPUTE has the kernels do before the call's first symbol
    if (c.lf_count == kResyncCount && !c.lf_recompute_pending) {
        c.lf_recompute_pending = true;
        c.lf_count = 0;
    }
    uint64_t n_out = 0;
    bool any_front = false;  // (regular window mode) the window class has no wave-scan instantiation
    if (S == 1) {
        // :445 nothing is pushed; :454 index==lastSample needs index==0; :457 size==numDataPts
        if (c.index == 0) {
            if (c.ring_len == D) {
                n_out = N;
                // (a symbol per sample, no timing recovery: the time-tiled kernels behind their run-time front stage, which has
                // nothing to pick there; the reference-order kernel for what the fit limits exclude)
                any_front = !lim.force_seq && n_out > 0 && c.lf_n <= lim.fast_fit_max && n_out <= kResyncCount &&
                            plan_lf_count0(c) + n_out <= kResyncCount;
                plan.mode = any_front ? PLAN_FAST : PLAN_SEQ_S1;
            }
        } else {
            c.index += N;  // lastSample==0 is never reached again
        }
        plan.ring_len1 = plan.ring_len0;
    } else if (c.ring_len >= D) {
        // window already full (or numAvg==0): size()==numDataPts can never hold again; the
        // deque only grows.  The device ring keeps the oldest ring_cap samples, which is all a
        // later resyncEnergy() can keep (it trims from the back, :622-626).
        c.index = ctl_mod(c.index + N, S);
        c.ring_len += N;
        uint64_t dev1 = c.ring_len < lim.ring_cap ? c.ring_len : lim.ring_cap;
        plan.ring_len1 = (uint32_t)dev1;
        if (dev1 > dev_ring0)
            plan.mode = PLAN_FAST;  // append-only: n_out == 0, any S
    } else {
        // regular window mode: resyncEnergy ran at the top of this call (D > size), so
        // index == size % S and the ring starts on a symbol boundary.
        (void)resynced;
        const uint64_t total = c.ring_len + N;
        const uint64_t complete = ctl_div(total, S);
        const uint64_t A = c.props.numAvg;  // A >= 1 here because D > ring_len >= 0
        n_out = complete >= A ? complete - (A - 1) : 0;
        // every emission pops S samples (:579-580)
        c.ring_len = total - n_out * S;
        c.index = ctl_mod(c.ring_len, S);
        c.count = (c.count + n_out) % kResyncCount;  // :581-583
        plan.ring_len1 = (uint32_t)c.ring_len;
        any_front = !fast_kernel_has((uint32_t)S, (uint32_t)A);
        bool fast_ok = !lim.force_seq && (!any_front || any_front_has((uint32_t)S)) && c.lf_n <= lim.fast_fit_max &&
                       n_out <= kResyncCount &&
                       ((plan_lf_count0(c)) + n_out <= kResyncCount);
        plan.mode = (n_out == 0 || fast_ok) ? PLAN_FAST : PLAN_SEQ;
        if (lim.force_seq && n_out > 0)
            plan.mode = PLAN_SEQ;
    }

    // LinearFit control for the kernel
    plan.lf_n = (uint32_t)c.lf_n;
    plan.lf_head = c.lf_head;
    plan.lf_len0 = (uint32_t)c.lf_len;
    plan.lf_count0 = (uint32_t)c.lf_count;
    plan.lf_xdelta = c.lf_xdelta;
    plan.lf_flags = 0;
    plan.n_out = n_out;
    if (plan.mode == PLAN_FAST && n_out && any_front)
        plan.lf_flags |= PLAN_ANYFRONT;
    if (plan.mode != PLAN_SKIP && c.lf_recompute_pending) {
        plan.lf_flags |= LF_RECOMPUTE;
        plan.lf_count0 = 0;
        c.lf_recompute_pending = false;
    }
    if (n_out) {
        // yvals after n_out calls of LinearFit::next(): circular, capacity fit_cap
        uint64_t grow = c.lf_len + n_out;
        uint64_t new_len = grow < c.lf_n ? grow : c.lf_n;
        uint64_t dropped = grow - new_len;
        c.lf_head = (uint32_t)ctl_mod(c.lf_head + dropped, lim.fit_cap);
        c.lf_len = new_len;
        // count==1048576 -> reset() -> count=0 at the top of next() (:51-52)
        uint64_t cnt = (plan.lf_flags & LF_RECOMPUTE) ? 0 : c.lf_count;
        cnt += n_out;
        while (cnt > kResyncCount) cnt -= kResyncCount;
        c.lf_count = cnt;
    }
    // these kernels write the surviving samples to the other ring buffer -- except at samplesPerBaud == 1, where the reference
    // leaves the deque alone (cpp/psk_soft.cpp:445: nothing is pushed, nothing popped): the carried samples stay where they
    // are, whatever the kernels of that call write into the other buffer
    if ((plan.mode == PLAN_FAST || plan.mode == PLAN_SEQ) && S != 1)
        c.ring_src ^= 1u;

    out.n_symbols = n_out;
    out.n_bits = n_out * bpb;
    out.n_sampleIndex = (S > 1) ? n_out : 0;
    if (n_out && bpb == 0)
        out.n_warn += (int32_t)(n_out > 0x7fffffff ? 0x7fffffff : n_out);  // :565-566, one per symbol
    if (n_out > out.cap_symbols && plan.mode != PLAN_SKIP && (out.soft || out.phase))
        return PSK_SOFT_ERR_CAPACITY;
    return PSK_SOFT_OK;
}

}  // namespace psk
#endif
