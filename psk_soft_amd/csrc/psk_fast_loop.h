// psk_fast_loop.h -- the symbol loop of the wave-scan kernels.
//
// One wave walks one channel in blocks of B = 128 output symbols; lane l owns the two
// consecutive symbols at block positions s = 2l and 2l+1 ("R = 2": every cross-lane scan,
// carry and fixed-point check is shared by two symbols, and the four output streams are
// written as 16/8/4/8-byte vectors).
//
// Input traffic is ONE pass: a symbol is loaded exactly once, when it becomes the newest
// symbol of a window (A-1 symbols before it is output).  What later steps need from it stays on chip:
//   * its S energies  -- subtracted from the window sums A symbols later
//                        (symbolEnergy[k] -= energy[k], reference cpp/psk_soft.cpp:572-577).
//                        numAvg <= 128 (H = 1): in an LDS ring of the last 256 positions;
//                        larger windows: in registers for H = ceil(A/128) blocks, moved across
//                        lanes with ds_bpermute;
//   * ONE of its samples, the one at the timing index the lane chose one block earlier.  When the
//     symbol is output and the true argmax equals that prediction -- timing is stationary, so
//     practically always -- the sample is already there; otherwise the lane re-reads it from
//     memory (exact either way).
//
// Timing argmax (reference cpp/psk_soft.cpp:445-466), two instantiations of the same loop:
//   EXACT = false  "screened": the window sums are scanned in FLOAT (one fused DPP add per step,
//                  no double arithmetic) together with a running bound E of their rounding error;
//                  the argmax is accepted only where the best sum beats the runner-up by more
//                  than 6E, i.e. where the exact argmax provably equals it.  A block in which
//                  some symbol fails that test is redone exactly on the spot from the LDS ring
//                  (numAvg <= 128); for larger windows the wave refuses the call (nothing
//                  committed) and the EXACT = true kernel, launched right behind it, redoes it.
//                  What this tier does not carry at all -- a sample that is inf or NaN, an M-th power
//                  that overflows -- it hands over the same way (every window class has both
//                  instantiations), stopping at the block where it notices.
//   EXACT = true   the sums are float-valued addends accumulated in double: exact, hence equal
//                  to the reference's whatever the summation order (quirk Q8), under the
//                  exponent-spread guard; near-ties resolve by std::max_element's first-maximum
//                  rule exactly as the reference's do.  The sums are updated symbol by symbol like
//                  the reference's, so an inf or NaN energy takes them through the same values (inf
//                  while the sample is in the window, NaN from the moment it leaves to the end of the
//                  call); this tier also carries libgcc's complex-multiply recovery (cmul<true>) and
//                  64-bit unwrap counts: calls with non-finite samples, or with the NaN / astronomically
//                  large phase estimate such a sample leaves behind in the channel, finish here at the
//                  wave-scan kernels' speed (the reference has no special case for them either).
#ifndef PSK_FAST_LOOP_H
#define PSK_FAST_LOOP_H

#include "psk_wave.h"

namespace psk {

constexpr int kR = 2;           // symbols per lane per block
constexpr int kB = kWave * kR;  // symbols per block
constexpr int kMaxUnwrapPasses = 160;
#ifndef PSK_TIES_IN_PLACE
#define PSK_TIES_IN_PLACE 1
#endif
#ifndef PSK_WIDE_FIRST
#define PSK_WIDE_FIRST 1
#endif
#ifndef PSK_DBUF
#define PSK_DBUF 1
#endif
constexpr int kScreenRefresh = 64;  // blocks between refreshes of the float window sums

struct FastCarry {
    double ySum, xySum;    // LinearFit sums after the last processed symbol
    float est;             // phaseEstimate
    float last_re, last_im;
    float den, xavg;       // LinearFit::denominator / xAvg
    float m, b;            // scratch for the prologue / epilogue fits
    float slope;           // slope of the last fit, per symbol: predicts the estimates of the next block
    uint32_t q;            // number of values ever written to the LDS y ring (history included)
    uint32_t last_k;       // timing index of the last emitted symbol (prediction seed)
    unsigned umax, umin1;  // exactness guard: max energy bits, min (energy bits - 1)
    float wmax;            // largest window sum seen so far in this call (exact-timing passes)
    bool ambiguous;        // (per lane) an exact-timing pass found best and runner-up closer than the
                           // reference's own accumulated rounding could be: only then does the
                           // exactness guard decide whether the call may stay here
    bool refuse;
    uint32_t stat_blocks, stat_extra, stat_exact_blocks;
    uint32_t stat_chain;   // blocks whose LinearFit sums went through the reference-order chain (fit_sums_chain)
    uint32_t chain_run;    // > 0: the chain ran on several blocks in a row -- the next blocks go straight to it (see fast_main_loop)
    uint32_t chain_streak; // low half: chained blocks in a row; high half: blocks left of the raised priority
    float gap_rel;         // (time-tiled front kernel only) smallest gap / relative bound among the exact re-decisions: a tile
                           // does not know the largest sum of the whole call, the fit kernel compares once it does
    float emax;            // (time-tiled front kernel only) largest sample energy loaded
    float cap;             // (time-tiled front kernel only) largest window sum anywhere in the call up to which the screening
                           // thresholds this tile used still exceed what the reference's running sums may have drifted by
};

// exactness guard bookkeeping: max of the energy bit patterns and min of (bits - 1); a zero
// energy gives bits 0 / 0xFFFFFFFF and so constrains neither
// An energy that is inf or NaN (energies carry no sign) is not a question of exactness: the screened tier leaves
// the call to the exact tier (EXACT_TIER), whose window sums are updated symbol by symbol like the reference's and
// so take the same values (inf while the sample is in the window, NaN from the moment it leaves, to the end of the
// call) and decide the same way (argtop_next).
template <bool EXACT_TIER>
PSK_DEV void guard_track(FastCarry &cy, float e)
{
    const unsigned eb = __float_as_uint(e);
    if (eb >= 0x7F800000u) {
        if (!EXACT_TIER)
            cy.refuse = true;
        return;
    }
    cy.umax = eb > cy.umax ? eb : cy.umax;
    const unsigned em = eb - 1u;
    cy.umin1 = em < cy.umin1 ? em : cy.umin1;
}

// what a block keeps of the symbols it loaded (positions s = 2*lane + r)
template <int S>
struct BlockKeep {
    float e[kR][S];  // energies
    float2 pk[kR];   // the sample at the predicted timing index
    int kp[kR];      // the predicted timing index
};

PSK_DEV float bperm_addr(int addr, float v)
{
    return __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(v)));
}
PSK_DEV int bperm_addr(int addr, int v) { return __builtin_amdgcn_ds_bpermute(addr, v); }

// Cross-lane fetch of "the value that sat D symbol positions earlier in the stream of loaded
// symbols" (D = numAvg for the energies leaving the window, numAvg-1 for the picked-from symbol).
// D = u*kB + v with 1 <= v <= kB (D = 0: u = v = 0): positions s >= v of this block find it in
// block c-u ("newer"), the others in block c-u-1 ("older").  The per-lane parameters depend only
// on numAvg, so they are computed once per call; the fetch itself is branch-free.
struct RotParam {
    int u;               // how many blocks back "newer" is (wave-uniform)
    bool odd;            // v odd: the two slots of a lane swap roles
    int src_addr[kR];    // byte address (lane*4) this lane pulls slot r from
    bool offer_old[kR];  // what this lane offers to the puller of its slot r: older or newer block
};
PSK_DEV RotParam rot_param(int lane, unsigned D)
{
    RotParam p;
    const int u = D ? (int)((D - 1) / kB) : 0;
    const int v = (int)D - u * kB;
    p.u = u;
    p.odd = (v & 1) != 0;
#pragma unroll
    for (int r = 0; r < kR; r++) {
        p.src_addr[r] = ((((2 * lane + r - v) & (kB - 1)) >> 1)) << 2;
        // this lane's slot r is pulled by some lane's result r ^ odd; that puller needs the older
        // block iff its own position < v, i.e. iff this slot's position >= kB - v
        p.offer_old[r] = (2 * lane + r) >= kB - v;
    }
    return p;
}
template <class T>
PSK_DEV T rot_pull(const RotParam &p, int r, const T (&newer)[kR], const T (&older)[kR])
{
    // compile-time array indices only (a runtime index would push the arrays to scratch)
    const T off0 = p.offer_old[0] ? older[0] : newer[0];
    const T off1 = p.offer_old[1] ? older[1] : newer[1];
    const T offered = (r == 0) ? (p.odd ? off1 : off0) : (p.odd ? off0 : off1);
    return bperm_addr(p.src_addr[r], offered);
}

// x[k] for a per-lane k without a runtime-indexed array: a binary tree of selects
template <int S, int LO, int SPAN>
PSK_DEV float sel_tree(const float2 (&x)[S], int k, bool want_y)
{
    if constexpr (SPAN == 1) {
        constexpr int idx = LO < S ? LO : S - 1;
        return want_y ? x[idx].y : x[idx].x;
    } else {
        constexpr int half = SPAN / 2;
        if constexpr (LO + half >= S) {
            return sel_tree<S, LO, half>(x, k, want_y);
        } else {
            float a = sel_tree<S, LO, half>(x, k, want_y);
            float b = sel_tree<S, LO + half, half>(x, k, want_y);
            return (k & half) ? b : a;
        }
    }
}
template <int S>
PSK_DEV float2 select_sample(const float2 (&x)[S], int k)
{
    constexpr int P = S <= 2 ? 2 : S <= 4 ? 4 : S <= 8 ? 8 : S <= 16 ? 16 : 32;
    return make_float2(sel_tree<S, 0, P>(x, k, false), sel_tree<S, 0, P>(x, k, true));
}

// x[k] for a WAVE-UNIFORM k: a scalar jump table and one move per component instead of the select tree.  (The timing index a
// lane predicts is the one its symbol had a block ago: on a signal whose timing stands still every lane predicts the same.
// The empty asm statements keep the compiler from folding the cases back into selects.)
template <int S>
PSK_DEV float2 select_sample_uniform(const float2 (&x)[S], int k)
{
    float rx = x[0].x, ry = x[0].y;
    switch (k) {
#define PSK_SEL_CASE(i)                              \
    case i:                                          \
        if constexpr (i < S) {                       \
            rx = x[i < S ? i : 0].x;                 \
            ry = x[i < S ? i : 0].y;                 \
            asm volatile("" : "+v"(rx), "+v"(ry));   \
        }                                            \
        break;
        PSK_SEL_CASE(1) PSK_SEL_CASE(2) PSK_SEL_CASE(3) PSK_SEL_CASE(4) PSK_SEL_CASE(5) PSK_SEL_CASE(6) PSK_SEL_CASE(7)
        PSK_SEL_CASE(8) PSK_SEL_CASE(9) PSK_SEL_CASE(10) PSK_SEL_CASE(11) PSK_SEL_CASE(12) PSK_SEL_CASE(13) PSK_SEL_CASE(14)
        PSK_SEL_CASE(15) PSK_SEL_CASE(16) PSK_SEL_CASE(17) PSK_SEL_CASE(18) PSK_SEL_CASE(19) PSK_SEL_CASE(20) PSK_SEL_CASE(21)
        PSK_SEL_CASE(22) PSK_SEL_CASE(23) PSK_SEL_CASE(24) PSK_SEL_CASE(25) PSK_SEL_CASE(26) PSK_SEL_CASE(27) PSK_SEL_CASE(28)
        PSK_SEL_CASE(29) PSK_SEL_CASE(30) PSK_SEL_CASE(31)
#undef PSK_SEL_CASE
    default: break;
    }
    return make_float2(rx, ry);
}

// loads the two symbols at positions 2*lane, 2*lane+1 of "new-symbol block" cblk:
// tau = kB*cblk + s + A - 1; symbols outside [tau_lo, tau_hi] are zero-filled.
//
// No lane takes a branch of its own here.  The block is either "steady" -- every symbol wanted and in the packet, a
// wave-uniform test in scalar arithmetic -- and loaded with 16-byte loads from one base address, or it is one of the
// few blocks of a call that are not (the first: the carried samples end and the packet begins somewhere inside; the
// last: partial; the window rebuilds of the prologue) and every lane loads sample by sample, picking buffer and
// address with selects and zeroing what is not wanted with selects.  An earlier version let each lane choose among
// three paths (packet / carried samples / the symbol that straddles both) and return early where nothing was wanted:
// correct as written, but in the instantiations that spill a thousand registers (samplesPerBaud 30 with four blocks
// of history: 1431 spills) the build of it read one of a symbol's samples as zero in partial blocks -- wrong timing
// picks in the last block of a call (found by the randomised comparison, rounds 129 and 148 of seed 20261004; the
// wrong tail block of round 1's <11,4,true> has the same signature).  With uniform control flow the compiler has no
// lane masks to carry spilled registers across.
template <int S>
PSK_DEV void load_symbol_any(const XView &X, uint64_t tau, bool wanted, float2 (&x)[S])
{
    const uint64_t j0 = tau * (uint64_t)S;
#pragma unroll
    for (int k = 0; k < S; k++) {
        const uint64_t j = j0 + (uint64_t)k;
        const f2g *p = j < X.L0 ? X.ring + j : X.in + (j - X.L0);  // (a select of two addresses)
        const f2g v = *mem_ptr<packet_global(S)>(p);
        x[k] = make_float2(wanted ? v.x : 0.0f, wanted ? v.y : 0.0f);
    }
}
template <int S>
PSK_DEV void load_block(const XView &X, long long cblk, uint32_t A, long long tau_lo, long long tau_hi, int lane,
                        float2 (&x)[kR][S])
{
    static_assert(kR == 2, "a lane's two symbols are one run of 2*S samples");
    const long long tau_first = cblk * kB + (long long)A - 1;  // symbol of lane 0, r = 0
    if (tau_first >= tau_lo && tau_first + (kB - 1) <= tau_hi && tau_first >= 0 &&
        (uint64_t)tau_first * (uint64_t)S >= (uint64_t)X.L0) {
        // steady: one wave-uniform base address plus a per-lane offset that does not change from block to block; a lane's
        // two symbols are contiguous and 16*S bytes long: S 16-byte loads (at 8-byte alignment, which gfx950 global
        // loads allow), whatever the parity of S
        const f2g *base = X.in + ((uint64_t)tau_first * (uint64_t)S - (uint64_t)X.L0);
        const typename F4Ptr<packet_global(S)>::type q =
            (typename F4Ptr<packet_global(S)>::type)base + (uint32_t)lane * (uint32_t)S;
#pragma unroll
        for (int k = 0; k < S; k++) {
#if PSK_NT_LOAD
            const f4g t = __builtin_nontemporal_load(q + k);
#else
            const f4g t = q[k];
#endif
            x[(2 * k) / S][(2 * k) % S] = make_float2(t.x, t.y);
            x[(2 * k + 1) / S][(2 * k + 1) % S] = make_float2(t.z, t.w);
        }
        return;
    }
#pragma unroll
    for (int r = 0; r < kR; r++) {
        const long long tau = cblk * kB + 2 * lane + r + (long long)A - 1;
        const bool ok = tau >= tau_lo && tau <= tau_hi;
        load_symbol_any<S>(X, (uint64_t)(ok ? tau : 0), ok, x[r]);  // (symbol 0 exists in every call that emits)
    }
}

// numAvg <= 128 (H == 1): the energies of the last R symbol positions live in an LDS ring,
// ering[k][position mod R], instead of registers.  A block writes its 128 positions (one 8-byte
// store per phase: a lane's two positions are adjacent and never straddle the wrap) and reads back
// the energies that sat numAvg positions earlier -- part of them written a moment ago by other
// lanes, the rest by the previous block.  No cross-lane permutes, no selects, and 2*S fewer live
// registers.
//   R = 256: every offset folds into the instructions (8 KiB at S = 8: 16 waves on a CU).
//   samplesPerBaud = 9, 10: R is chosen by the host per launch (dynamic LDS), even and at least
//   numAvg + 128 for every channel of the launch: the shorter ring is what lets this instantiation
//   keep 16 waves on a CU (8-PSK, S = 10: 2.92 -> 2.43 ms).  Its wrap is a compare and a select
//   instead of a mask, which costs where residency is not the limit (S = 8: +1.5 %, S = 12: +7 %,
//   S = 16: -1 %), hence only there.
constexpr int kERing = 2 * kB;
template <bool DYN>
struct ERingT;
template <>
struct ERingT<false> {
    float *mem;
    PSK_DEV void set_len(int) {}
    PSK_DEV int length() const { return kERing; }
    PSK_DEV int wrap(int i) const { return i & (kERing - 1); }
    PSK_DEV float *row(int k) const { return mem + k * kERing; }
};
template <>
struct ERingT<true> {
    float *mem;
    int len;  // R
    PSK_DEV void set_len(int r) { len = r; }
    PSK_DEV int length() const { return len; }
    PSK_DEV int wrap(int i) const { return i < 0 ? i + len : (i >= len ? i - len : i); }  // i in (-R, 2R)
    PSK_DEV float *row(int k) const { return mem + k * len; }
};
template <int S>
PSK_DEV void ering_put(const ERingT<ering_dynamic(S)> &er, int base, int lane, const float (&e)[kR][S])
{
    const int i = er.wrap(base + 2 * lane);
#pragma unroll
    for (int k = 0; k < S; k++) *reinterpret_cast<float2 *>(er.row(k) + i) = make_float2(e[0][k], e[1][k]);
}
template <int S>
PSK_DEV void ering_get(const ERingT<ering_dynamic(S)> &er, int base, int lane, uint32_t D, float (&e)[kR][S])
{
    const int i0 = er.wrap(base + 2 * lane - (int)D);
    if ((D & 1u) == 0) {  // the pair stays 8-byte aligned (and never straddles the wrap)
#pragma unroll
        for (int k = 0; k < S; k++) {
            const float2 t = *reinterpret_cast<const float2 *>(er.row(k) + i0);
            e[0][k] = t.x;
            e[1][k] = t.y;
        }
    } else {
        const int i1 = er.wrap(i0 + 1);
#pragma unroll
        for (int k = 0; k < S; k++) {
            e[0][k] = er.row(k)[i0];
            e[1][k] = er.row(k)[i1];
        }
    }
}

// value of `field` in the block `back` blocks back in time (0 = cur, j >= 1 = hist[j-1]); back is
// wave-uniform and at most H
template <int S, int H, class F>
PSK_DEV auto block_back(int back, const BlockKeep<S> &cur, const BlockKeep<S> (&hist)[H], F field)
{
    auto v = field(back == 0 ? cur : hist[0]);
#pragma unroll
    for (int j = 2; j <= H; j++)
        if (back == j)
            v = field(hist[j - 1]);
    return v;
}

// Between two exact passes of fit_block: settle the knock-on effects of the corrections a pass has
// just found, cheaply, so that the next exact pass (normally) only has to confirm.
// A count that changes by d_j at position j moves y_j by 2 pi d_j, and with it every later
// estimate that still has j in its window of n values: the endpoint of a least-squares line
// through n equidistant points weighs the point of age t with (4n-2)/(n(n+1)) - 6t/(n(n+1)), so
//     est'_s - est_s = 2 pi (alpha D0_s - beta D1_s),  D0_s = sum_{j in (s-n, s]} d_j,  D1_s = sum (s-j) d_j.
// Both windowed sums come from ONE integer prefix scan of d_j (65536 j + 1) and two ds_bpermute
// (low half: sum d_j, high half: sum j d_j).  Every position then re-derives its count from the
// shifted estimate of its predecessor, as a DIFFERENCE to what the unshifted estimate gives in
// the same float formula (so that where nothing moved, the exact count w2 stands), and the step
// repeats until nothing changes.  This is a guess: everything is verified by the exact pass that
// follows, an overflow of the packed sums or a float near-tie only costs another pass.
template <bool WIDE>
struct WrapInt {
    typedef int type;
};
template <>
struct WrapInt<true> {
    typedef long long type;
};
constexpr int kRefineMax = 12;
// A channel whose sums hover around zero (a stationary carrier at zero phase) needs the lane-after-lane chain on
// every block: after PSK_CHAIN_STREAK such blocks in a row the next PSK_CHAIN_RUN go straight to it (no candidates),
// and a wave that chained within its last 16 blocks keeps a raised priority through the arithmetic half of its
// blocks (measured on the headline: streak 1 / run 7 2.66 ms, streak 6 / run 15 2.64, no run at all 3.02).
#ifndef PSK_CHAIN_RUN
#define PSK_CHAIN_RUN 15
#endif
#ifndef PSK_CHAIN_STREAK
#define PSK_CHAIN_STREAK 6
#endif
#ifndef PSK_CHAIN_PRIO
#define PSK_CHAIN_PRIO 2
#endif
#ifndef PSK_CHAIN_UNROLL
#define PSK_CHAIN_UNROLL 4
#endif
PSK_DEV void refine_unwrap(int lane, uint32_t n, float est_prev0, const float (&est)[kR], const float (&raw)[kR],
                           const bool (&valid)[kR], const int (&w_base)[kR], int (&w2)[kR])
{
    const float inv2pi = 0.15915494f;
    const float nn = (float)n * (float)(n + 1u);
    const float alpha = 6.2831853f * (float)(4u * n - 2u) / nn, beta = 6.2831853f * 6.0f / nn;
    const int s0 = 2 * lane, s1 = s0 + 1;
    const float raw0 = raw[0], raw1 = raw[1];
    const float qb0 = __builtin_rintf((est_prev0 - raw0) * inv2pi), qb1 = __builtin_rintf((est[0] - raw1) * inv2pi);
    int d0 = valid[0] ? w2[0] - w_base[0] : 0, d1 = valid[1] ? w2[1] - w_base[1] : 0;
    // where position s - n lives (n < 128; otherwise the window reaches back past the block start)
    const bool windowed = n < (uint32_t)kB;
    const int la = lane - (int)((n + 1u) >> 1);  // n even: both positions from lane - n/2
    const int lb = (n & 1u) ? la + 1 : la;       // n odd: s0 - n = slot 1 of la, s1 - n = slot 0 of la + 1
#pragma unroll 1
    for (int it = 0; it < kRefineMax; it++) {
        const int v0 = d0 * (65536 * s0 + 1), v1 = d1 * (65536 * s1 + 1);
        const int incl = wave_scan_i32(v0 + v1);
        const int x0 = wave_up1(incl, 0) + v0, x1 = x0 + v1;
        int z0 = 0, z1 = 0;
        if (windowed) {
            const int a = __builtin_amdgcn_ds_bpermute(la << 2, (n & 1u) ? x1 : x0);
            const int b = __builtin_amdgcn_ds_bpermute(lb << 2, (n & 1u) ? x0 : x1);
            z0 = la >= 0 ? a : 0;
            z1 = lb >= 0 ? b : 0;
        }
        const int y0 = x0 - z0, y1 = x1 - z1;
        const int D0a = (y0 << 16) >> 16, D0b = (y1 << 16) >> 16;  // sum of d_j over the window
        const int J0 = (y0 - D0a) >> 16, J1 = (y1 - D0b) >> 16;    // sum of j d_j over the window
        const float de0 = alpha * (float)D0a - beta * (float)(s0 * D0a - J0);
        const float de1 = alpha * (float)D0b - beta * (float)(s1 * D0b - J1);
        const float dp0 = wave_up1(de1, 0.0f);  // shift of the estimate fed back at s0
        const float q0 = __builtin_rintf((est_prev0 + dp0 - raw0) * inv2pi);
        const float q1 = __builtin_rintf((est[0] + de0 - raw1) * inv2pi);
        const int nd0 = valid[0] ? w2[0] + (int)(q0 - qb0) - w_base[0] : 0;
        const int nd1 = valid[1] ? w2[1] + (int)(q1 - qb1) - w_base[1] : 0;
        const bool changed = nd0 != d0 || nd1 != d1;
        d0 = nd0;
        d1 = nd1;
        if (!__any(changed))
            break;
    }
    w2[0] = w_base[0] + d0;
    w2[1] = w_base[1] + d1;
}

// ---------------------------------------------------------------------------------------------
// LinearFit::next's running sums in the REFERENCE'S ORDER of additions (cpp/psk_soft.cpp:70-79):
//     ySum -= front;  xySum -= xdelta*ySum;  ySum += y;  xySum += y*size*xdelta      per symbol.
// xdelta*ySum is a rounded double product, so xySum depends on the order of its additions at the
// 2^-53 level, and through the float roundings of calculateFit that reaches the outputs (a one-ulp
// flip of phaseEstimate now and then: 6e-5 rad once the estimate has grown past 512 rad).  The sums of a
// block are therefore produced so that they ARE the sequential ones, in three steps:
//   1. candidates, wave-parallel: ySum by a prefix scan (exact whenever the reference's own additions are:
//      float addends of similar size); xySum by xysum_grid below;
//   2. a certificate (fit_sums_verify): every position re-runs the reference's statements
//      on its predecessor's sums and compares bits.  If all positions agree the candidates are the sequential
//      sums, by induction from the carried sums -- whatever produced them; nothing else is trusted;
//   3. otherwise fit_sums_chain runs the recurrence itself, lane after lane.
// ---------------------------------------------------------------------------------------------
PSK_DEV double pow2_biased(int eb) { return __hiloint2double((int)((unsigned)eb << 20), 0); }
PSK_DEV bool same_bits(double a, double b) { return __double_as_longlong(a) == __double_as_longlong(b); }
// wave votes straight from the lane mask of a condition (HIP's __ballot / __any / __all take an int: the compiler
// then turns the mask into 0 / 1 per lane and back)
PSK_DEV unsigned long long vote_mask(bool b) { return __builtin_amdgcn_ballot_w64(b); }
PSK_DEV bool vote_any(bool b) { return __builtin_amdgcn_ballot_w64(b) != 0ull; }
PSK_DEV bool vote_all(bool b) { return __builtin_amdgcn_ballot_w64(!b) == 0ull; }
PSK_DEV bool odd_f64(double n) { return __builtin_amdgcn_fract(n * 0.5) != 0.0; }  // n integer-valued

// xySum candidates for the 128 positions of a block, in the reference's order of roundings.
//     r_j = fl(s_{j-1} - c_j),  s_j = fl(r_j + t_j);   c_j = fl(xdelta*ySum) (53 bits), t_j a float.
// Let [2^p, 2^(p+1)) be the binade of the intermediates r_j and q = 2^(p-52) its ulp; everything is scaled
// to units of q (exact).
//   mode A, |s| < 2^(p+1): s_{j-1} is a multiple of q, so step j subtracts RN(c_j) whatever the state --
//     except on an exact tie (c_j has some five bits below q: one step in 32), which goes to the even
//     neighbour and so depends on the parity of s_{j-1}.  r_j + t_j is exact.  After a tie the parity is
//     known (even, plus t_j), hence the parity at every position follows from a SEGMENTED xor scan, here by
//     ballots and mbcnt; the tie corrections are added to the increments and ONE prefix sum -- exact: integers
//     below 2^53 -- yields all s_j.  (Sums near zero, |s| << |c|, are covered too: c is then on the grid
//     itself and nothing rounds.)
//   mode B, 2^(p+1) <= |s| < 2^(p+2) (the sums sit just above a power of two and the intermediates dip below
//     it, one block in 16): s_{j-1} is an EVEN multiple of q, so the first rounding is settled locally; the
//     second one, to multiples of 2, is exact for an even r_j + t_j and a state-dependent tie for an odd one.
// Anything else (intermediates changing binade, operands off the grid, non-finite values) produces
// candidates that do not verify (returns false).  tools/model/xysum_grid_model.cpp is the CPU model of this.
PSK_DEV void xysum_grid(int lane, double s_c, const double (&c)[kR], const double (&t)[kR], double (&xs)[kR], double &xs_prev)
{
    const double r_first = s_c - read_lane(c[0], 0);          // (wave-uniform, like s_c)
    const int eb = (__double2hiint(r_first) >> 20) & 0x7ff;   // biased exponent of the intermediates' binade
    const double q = pow2_biased(eb - 52), inv_q = pow2_biased(2098 - eb);
    const bool modeB = !(__builtin_fabs(s_c) < pow2_biased(eb + 1));
    const double S_c = s_c * inv_q;
    // Per position: T = a state-dependent tie; V = parity of s_j (mode B: of s_j / 2), absolute after a tie,
    // else relative to s_{j-1}; E (mode B) = which neighbour an even s_{j-1}/2 selects; inc = s_j - s_{j-1}
    // before the tie correction.  RN(c) = rint(c) is EVEN at a tie, so s_{j-1} - RN(c) has the parity of s_{j-1}:
    // mode A picks the neighbour by that parity alone, and mode B's first rounding never needs a correction.
    bool T[kR], V[kR], E[kR], dpos[kR];
    double inc[kR];
#pragma unroll
    for (int r = 0; r < kR; r++) {
        const double nc = c[r] * inv_q, nt = t[r] * inv_q;
        const double ch = __builtin_rint(nc);  // RN(c), ties to even
        inc[r] = nt - ch;
        if (!modeB) {
            const double d = nc - ch;
            T[r] = __builtin_fabs(d) == 0.5;
            dpos[r] = d > 0.0;  // the tie lies below RN(c): the other neighbour is one step down
            E[r] = false;
            V[r] = odd_f64(inc[r]);  // after a tie: even + t, and inc = t - even
        } else {
            T[r] = odd_f64(inc[r]);  // r_j + t_j odd: the rounding to multiples of 2 is a tie
            const bool hw = odd_f64((T[r] ? inc[r] - 1.0 : inc[r]) * 0.5);
            E[r] = hw;
            V[r] = !T[r] && hw;
            dpos[r] = false;
        }
    }
    // parity entering this lane: the lane's two positions composed, then segmented over the wave
    const bool Ac = T[0] || T[1];
    const bool Vc = T[1] ? V[1] : (V[0] != V[1]);
    const unsigned long long mA = vote_mask(Ac), mV = vote_mask(Vc);
    const bool G = (__builtin_amdgcn_mbcnt_hi((unsigned)(mV >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mV, 0u)) & 1u) != 0;
    const unsigned long long mF = vote_mask(Ac && G);
    const unsigned long long lt = lane ? (~0ull >> (64 - lane)) : 0ull;
    const unsigned long long X = mA & lt, Y = mF & lt;  // ties below this lane; ... those whose G is odd
    const bool P_c = modeB ? odd_f64(S_c * 0.5) : odd_f64(S_c);
    const bool P_in = (X ? (Y > (X >> 1)) : P_c) != G;  // G at the last tie below this lane, else the carry's parity
    const bool sel0 = P_in != E[0];
    const bool P0 = T[0] ? V[0] : (P_in != V[0]);
    const bool sel1 = P0 != E[1];
    double a0, a1;
    if (!modeB) {
        a0 = inc[0] + ((T[0] && sel0) ? (dpos[0] ? -1.0 : 1.0) : 0.0);
        a1 = inc[1] + ((T[1] && sel1) ? (dpos[1] ? -1.0 : 1.0) : 0.0);
    } else {
        a0 = inc[0] + (T[0] ? (sel0 ? 1.0 : -1.0) : 0.0);
        a1 = inc[1] + (T[1] ? (sel1 ? 1.0 : -1.0) : 0.0);
    }
    // (integers below 2^53: every sum here is exact, so the lane's second sum may be taken from the scan -- the same expression
    // the next lane forms for its predecessor: xs_prev IS the previous lane's xs[1], bit for bit, without a cross-lane move)
    const double incl = wave_scan_f64(a0 + a1);
    const double xp = S_c + wave_up1_zero(incl);
    xs_prev = xp * q;
    xs[0] = (xp + a0) * q;
    xs[1] = (S_c + incl) * q;
}

PSK_DEV void xysum_grid(int lane, double s_c, const double (&c)[kR], const double (&t)[kR], double (&xs)[kR])
{
    double xs_prev;
    xysum_grid(lane, s_c, c, t, xs, xs_prev);
}

// The certificate.  Every valid position re-runs the reference's four statements (cpp/psk_soft.cpp:70, :72, :77,
// :78) on its predecessor's candidate sums and compares bits: if all positions agree the candidates ARE the
// sequential sums, by induction from the carried sums, whatever produced them.  The operands are rebuilt here
// from the block's y and z (cheaper than keeping the candidates' own operands in registers across xysum_grid);
// where the window is still filling there is no pop (z = 0, and x - 0 is x).  A NaN never verifies (the chain
// then produces the reference's own).  Returns bit 0: ySum fails, bit 1: xySum fails.
// ys_prev / xs_prev: the candidates of the position in front of the lane's first one -- the previous lane's second sums, which
// the candidates' producers form by the same expression as that lane does (the carried sums for lane 0).
PSK_DEV int fit_sums_verify(const bool (&valid)[kR], const bool (&steady)[kR], float xd, const float (&sizef)[kR],
                            double ys_prev, double xs_prev, const float (&z)[kR], const float (&y)[kR],
                            const double (&ySum_l)[kR], const double (&xySum_l)[kR])
{
    const double p0 = ys_prev - (double)z[0], p1 = ySum_l[0] - (double)z[1];  // ySum after the pop, :70
    const bool y0 = p0 + (double)y[0] == ySum_l[0], y1 = p1 + (double)y[1] == ySum_l[1];
    const double c0 = steady[0] ? (double)xd * p0 : 0.0, c1 = steady[1] ? (double)xd * p1 : 0.0;  // :72
    float t0 = y[0] * sizef[0];  // :78, size before the push
    t0 = t0 * xd;
    float t1 = y[1] * sizef[1];
    t1 = t1 * xd;
    const bool x0 = (xs_prev - c0) + (double)t0 == xySum_l[0], x1 = (xySum_l[0] - c1) + (double)t1 == xySum_l[1];
    const bool y_ok = vote_all((y0 || !valid[0]) && (y1 || !valid[1]));
    const bool x_ok = vote_all((x0 || !valid[0]) && (x1 || !valid[1]));
    return (y_ok ? 0 : 1) | (x_ok ? 0 : 2);
}

// The recurrence itself (cpp/psk_soft.cpp:70-79), lane after lane: in step k every lane takes the sums
// its left neighbour holds and runs its own two symbols; lane k's neighbour is final from step k - 1 on, so
// after lane_last + 1 steps every valid position holds the reference's sums (a lane that is already final
// recomputes the same values).  with_y = false: ySum_l holds verified sums, only xySum is chained.  The two sums
// are chained one after the other (ySum does not depend on xySum): two short loops of 2 moves + 4 dependent
// additions a step instead of one long one.
template <bool WARM>
PSK_DEV void fit_sums_chain(bool with_y, int lane, int lane_last, uint32_t q0, uint32_t n, float xd, float sizef_steady,
                            double ySum_c, double xySum_c, const bool (&valid)[kR], const float *yring, uint32_t ymask,
                            const float (&y)[kR], double (&ySum_l)[kR], double (&xySum_l)[kR])
{
    const double xdd = (double)xd;
    // operands rebuilt from what the block left behind (the ring holds every y of the window).  Pops where the
    // window is still filling get zero operands (x - 0 is x); positions past the end run on whatever their lanes
    // hold: nothing valid comes after them.
    double zz[kR], yy[kR], tt[kR];
    bool steady[kR];
#pragma unroll
    for (int r = 0; r < kR; r++) {
        const uint32_t before = q0 + (uint32_t)(2 * lane + r);
        steady[r] = WARM ? before >= n : true;
        const float sizef = WARM ? (float)(steady[r] ? n - 1 : before) : sizef_steady;
        const float zf = yring[(before - n) & ymask];
        zz[r] = steady[r] ? (double)zf : 0.0;
        yy[r] = (double)y[r];
        float tf = y[r] * sizef;  // :78, size before the push
        tf = tf * xd;
        tt[r] = (double)tf;
    }
    // (the value a step takes from the lane below lands in the register that held it the step before: a DPP move leaves lane 0,
    // which has no lane below, as it was -- the carried sum, put there once.  Handing the constant in at every step instead cost
    // two register moves a step, a quarter of the chain's instructions.)
    if (with_y) {
        double ys = ySum_c, a = ySum_c;
#pragma unroll 1
        for (int k = 0; k <= lane_last; k++) {
            a = wave_up1(ys, a);
            ySum_l[0] = (a - zz[0]) + yy[0];  // ySum -= yvals.front(), :70;  ySum += yval, :77
            ySum_l[1] = (ySum_l[0] - zz[1]) + yy[1];
            ys = ySum_l[1];
        }
    }
    // xdelta*ySum after each pop, :72 (ySum_l holds the reference's sums by now)
    const double ys_prev = wave_up1(ySum_l[1], ySum_c);
    const double c0 = steady[0] ? xdd * (ys_prev - zz[0]) : 0.0;
    const double c1 = steady[1] ? xdd * (ySum_l[0] - zz[1]) : 0.0;
    double xs = xySum_c, b = xySum_c;
    const int steps = PSK_CHAIN_UNROLL * ((lane_last + PSK_CHAIN_UNROLL) / PSK_CHAIN_UNROLL);  // (extra steps recompute final values)
#pragma unroll 1
    for (int k = 0; k < steps; k += PSK_CHAIN_UNROLL) {
#pragma unroll
        for (int u = 0; u < PSK_CHAIN_UNROLL; u++) {
            b = wave_up1(xs, b);
            xs = ((b - c0) + tt[0] - c1) + tt[1];  // :72 and :78, twice
        }
    }
    const double b_fin = wave_up1(xs, b);
    xySum_l[0] = (b_fin - c0) + tt[0];
    xySum_l[1] = xs;
}

// One block (128 symbols) of the feedback unwrap + LinearFit::next recurrence
// (reference cpp/psk_soft.cpp:477-482, 48-87, 135-174).  The recurrence
//     est[i-1] -> numWraps[i] -> y[i] -> (ySum, xySum) -> est[i]
// is solved for all positions at once: numWraps is speculated by consecutive differences, the
// sums are two double prefix scans over y[i] - y[i-n] and term[i] - xdelta*ySum'[i] with the
// reference's float-rounded term[i] (quirk Q4), and every position re-derives numWraps from its
// predecessor's estimate exactly as :477 does; disagreements repeat the pass, each pass fixes at
// least one more position.  WARM = the fit window is still growing somewhere in the block.
// Returns the number of extra passes; den_last / xavg_last = LinearFit::denominator / xAvg after
// the block's last valid symbol.
// WIDE (the exact tier): numWraps kept as the 64-bit integer the reference has -- an estimate that is NaN or astronomically
// large (what a non-finite sample leaves behind) unwraps by counts far beyond 2^31, and (long)NaN is LONG_MIN on x86.
template <bool WARM, bool WIDE>
PSK_DEV int fit_block(int lane, uint32_t q0, uint32_t n, float xd, float den_s, float xavg_s, const FitKnown &fk,
                      const bool (&valid)[kR], const float (&raw)[kR], const FastCarry &cy,
                      float *yring, uint32_t ymask, float (&y)[kR], float (&est)[kR], double (&ySum_l)[kR], double (&xySum_l)[kR],
                      int lane_last, int r_last, float &den_last, float &xavg_last, bool cheap, int &rejected, float &m_last)
{
    uint32_t before[kR];
    bool steady[kR];
    float sizef[kR];  // (float)yvals.size() at cpp/psk_soft.cpp:78
    uint32_t pts[kR];
    float den_l[kR], xavg_l[kR];
#pragma unroll
    for (int r = 0; r < kR; r++) {
        before[r] = q0 + (uint32_t)(2 * lane + r);  // values pushed before this next()
        if (WARM) {
            steady[r] = before[r] >= n;              // cpp/psk_soft.cpp:54
            sizef[r] = (float)(steady[r] ? n - 1 : before[r]);
            pts[r] = steady[r] ? n : before[r] + 1;  // yvals.size() at calculateFit
            den_l[r] = den_s;
            xavg_l[r] = xavg_s;
            if (pts[r] > 1 && pts[r] < n)            // the window is still growing
                fit_denominator(xd, pts[r], den_l[r], xavg_l[r]);
        } else {
            steady[r] = true;
            sizef[r] = fk.sizef;
            pts[r] = n;
            den_l[r] = den_s;
            xavg_l[r] = xavg_s;
        }
    }
    // Speculate numWraps from a PREDICTION of the feedback: the estimate fed back at position s of
    // the block is, to first order, the carried estimate continued along the last fitted line
    // (cpp/psk_soft.cpp:477 with est[s-1] ~ est_carry + slope*s).  Every position guesses on its
    // own, so a noisy sample spoils one guess, not all that follow (guessing by consecutive raw
    // differences did: at 10 dB it needed 9 correction passes a block).  Position 0 is exact.
    typedef typename WrapInt<WIDE>::type WInt;
    WInt w[kR];
    {
        // (a guess only: float arithmetic is enough; every count is verified below)
        const float inv2pi = 0.15915494f;
#pragma unroll
        for (int r = 0; r < kR; r++) {
            const float pred = cy.est + cy.slope * (float)(2 * lane + r);
            if (WIDE)
                w[r] = (WInt)unwrap_count(pred, (double)raw[r], (int)q0);
            else
                w[r] = (WInt)(int)__builtin_rintf((pred - raw[r]) * inv2pi);
        }
    }
    const double two_pi = PSK_KD(kTwoPi, (int)q0);
    int pass = 0;
    for (;;) {
#pragma unroll
        for (int r = 0; r < kR; r++) {
            double yd = (double)raw[r] + (double)(long long)w[r] * two_pi;  // cpp/psk_soft.cpp:478 (thisPhase is a float widened, :474)
            y[r] = (float)yd;                                         // next(float yval), :481
            if (valid[r])
                yring[before[r] & ymask] = y[r];
        }
        wave_lds_fence();
        float z[kR];
#pragma unroll
        for (int r = 0; r < kR; r++)
            z[r] = steady[r] ? yring[(before[r] - n) & ymask] : 0.0f;  // yvals.front(), :70
        wave_lds_fence();
        // candidate sums, wave-parallel (positions past the end need no masking: a prefix sum never looks ahead)
        double ys_prev, xs_prev;
        {
            const double dy0 = (double)y[0] - (double)z[0], dy1 = (double)y[1] - (double)z[1];
            const double incl = wave_scan_f64(dy0 + dy1);
            const double base = cy.ySum + wave_up1_zero(incl);  // ySum after the previous lane's symbols
            ySum_l[0] = base + dy0;
            ySum_l[1] = cy.ySum + incl;  // (the expression the next lane calls `base`: the certificate's chain needs no move)
            ys_prev = base;
            // xdelta*ySum after the pop, :70 and :72
            const double c_d[kR] = {steady[0] ? (double)xd * (base - (double)z[0]) : 0.0,
                                    steady[1] ? (double)xd * (ySum_l[0] - (double)z[1]) : 0.0};
            float t0 = y[0] * sizef[0];  // :78, size before the push
            t0 = t0 * xd;
            float t1 = y[1] * sizef[1];
            t1 = t1 * xd;
            const double t_d[kR] = {(double)t0, (double)t1};
            if (WARM || cheap) {
                // (the caller runs the recurrence for xySum over this block anyway: any guess will do here)
                const double cc0 = t_d[0] - c_d[0], cc1 = t_d[1] - c_d[1];
                const double i2 = wave_scan_f64(cc0 + cc1);
                const double b2 = cy.xySum + wave_up1_zero(i2);
                xySum_l[0] = b2 + cc0;
                xySum_l[1] = xySum_l[0] + cc1;
                xs_prev = b2;
            } else {
                xysum_grid(lane, cy.xySum, c_d, t_d, xySum_l, xs_prev);
            }
        }
        int rej = WARM ? 3 : fit_sums_verify(valid, steady, xd, sizef, ys_prev, xs_prev, z, y, ySum_l, xySum_l);
        if (cheap)
            rej |= 2;
        rejected = rej;
        float m_l[kR];  // LinearFit::m of the fits (a by-product: the block's last one predicts the next block's estimates)
#pragma unroll
        for (int r = 0; r < kR; r++) {
            float m_ = 0.0f, b_;
            if (!WARM) {  // steady state: both divisors are wave-uniform
                est[r] = fit_value_known(ySum_l[r], xySum_l[r], fk, m_);
            } else if (pts[r] > 1) {
                est[r] = fit_value(ySum_l[r], xySum_l[r], xd, pts[r], den_l[r], xavg_l[r], m_, b_);
            } else {  // :164-171, a single point: b = yvals.back()
                est[r] = y[r];
            }
            m_l[r] = m_;
        }
        m_last = r_last ? m_l[1] : m_l[0];
        const float est_prev0 = wave_up1(est[1], cy.est);
        {
            // round((est_prev - raw)/2pi) == w  <=>  |est_prev - (raw + 2 pi w)| < pi; with the
            // unwrapped value y within 3.0 of the feedback (and small enough that its float
            // rounding is far below the 0.14 of slack) no exact quotient is needed
            const float g0 = __builtin_fabsf(est_prev0 - y[0]), g1 = __builtin_fabsf(est[0] - y[1]);
            const bool sure0 = !valid[0] || (g0 < 3.0f && __builtin_fabsf(y[0]) < 65536.0f);
            const bool sure1 = !valid[1] || (g1 < 3.0f && __builtin_fabsf(y[1]) < 65536.0f);
            if (__all(sure0 && sure1))
                break;
        }
        const long long c2_0 = unwrap_count(est_prev0, (double)raw[0], (int)q0);  // cpp/psk_soft.cpp:477 with the true feedback
        const long long c2_1 = unwrap_count(est[0], (double)raw[1], (int)q0);
        if constexpr (!WIDE) {
            // a count beyond 32 bits -- the feedback is NaN ((long)NaN is LONG_MIN on x86) or astronomically large, what a
            // non-finite sample leaves behind in a channel -- is the exact tier's, which keeps the counts in 64 bits: truncated,
            // LONG_MIN would pass for 0 and a wrong count could be taken for verified
            const bool wide0 = valid[0] && c2_0 != (long long)(int)c2_0, wide1 = valid[1] && c2_1 != (long long)(int)c2_1;
            if (__any(wide0 || wide1)) {
                pass = kMaxUnwrapPasses + 1;  // (the caller refuses the call)
                break;
            }
        }
        WInt w2_0 = (WInt)c2_0;
        WInt w2_1 = (WInt)c2_1;
        const bool bad = (valid[0] && w2_0 != w[0]) || (valid[1] && w2_1 != w[1]);
        if (!__any(bad))
            break;
        if constexpr (!WARM && !WIDE) {
            int w2[kR] = {w2_0, w2_1};
            refine_unwrap(lane, n, est_prev0, est, raw, valid, w, w2);
            w2_0 = w2[0];
            w2_1 = w2[1];
        }
        w[0] = w2_0;
        w[1] = w2_1;
        if (++pass > kMaxUnwrapPasses)
            break;
    }
    if (WARM) {
        den_last = read_lane(r_last ? den_l[1] : den_l[0], lane_last);
        xavg_last = read_lane(r_last ? xavg_l[1] : xavg_l[0], lane_last);
    }
    return pass;
}

// float window sum W_k at the LAST position of the newest kept block (= the carry into the next
// block), recomputed from the energies in registers: the A = u*kB + v symbols ending there are
// the u newest blocks whole plus the positions >= kB - v of the one before
template <int S, int H>
PSK_DEV float window_end_f32(const BlockKeep<S> (&hist)[H], uint32_t A, int lane, int k)
{
    const int u = (int)((A - 1) / kB), v = (int)A - u * kB;  // 1 <= v <= kB
    float acc = 0.0f;
#pragma unroll
    for (int j = 0; j < H; j++) {
        const bool whole = (H > 1) && j < u;
        const bool part = (H == 1) ? true : j == u;
#pragma unroll
        for (int r = 0; r < kR; r++)
            if (whole || (part && 2 * lane + r >= kB - v))
                acc += hist[j].e[r][k];
    }
    return read_lane(wave_scan_f32(acc), 63);
}

// the same without a history (H == 0): the window's symbols read again, newest block first like the register variant
template <int S>
PSK_DEV void window_end_reread_f32(const XView &X, int c, int c_begin, uint32_t A, long long tau_last, int lane, float (&W)[S])
{
    const int u = (int)((A - 1) / kB), v = (int)A - u * kB;  // 1 <= v <= kB
    float acc[S];
#pragma unroll
    for (int k = 0; k < S; k++) acc[k] = 0.0f;
    for (int j = 0; j <= u; j++) {
        float2 x[kR][S];
        load_block<S>(X, (long long)(c - j), A, (long long)c_begin * kB, tau_last, lane, x);
#pragma unroll
        for (int r = 0; r < kR; r++) {
            const bool in = j < u || 2 * lane + r >= kB - v;
#pragma unroll
            for (int k = 0; k < S; k++) {
                const float e = norm_f(x[r][k].x, x[r][k].y);
                acc[k] += in ? e : 0.0f;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < S; k++) W[k] = read_lane(wave_scan_f32(acc[k]), 63);
}

// EXACT window sums at the last position of the newest kept block, for the in-place redo of a block in the screened tier of the
// windows longer than a block: float-valued energies added in double (exact under the exponent-spread guard, which is fed
// with every energy that enters here; an energy that is not finite sets cy.refuse: the reference's RUNNING sums keep an inf
// while it is in the window and are NaN from the moment it leaves, which only the exact tier's running sums reproduce),
// from the history in registers ...
template <int S, int H>
PSK_DEV void window_end_f64(const BlockKeep<S> (&hist)[H], uint32_t A, int lane, FastCarry &cy, double (&W)[S])
{
    const int u = (int)((A - 1) / kB), v = (int)A - u * kB;  // 1 <= v <= kB
#pragma unroll
    for (int k = 0; k < S; k++) {
        double acc = 0.0;
#pragma unroll
        for (int j = H - 1; j >= 0; j--) {  // (oldest block first, like the call's prologue)
            const bool whole = j < u, part = j == u;
#pragma unroll
            for (int r = 0; r < kR; r++)
                if (whole || (part && 2 * lane + r >= kB - v)) {
                    guard_track<false>(cy, hist[j].e[r][k]);
                    acc += (double)hist[j].e[r][k];
                }
        }
        W[k] = wave_sum_f64(acc);
    }
}
// ... or, without a history (H == 0), from the window's symbols read again: the A symbols that end with block c_end's last
template <int S>
PSK_DEV void window_end_reread_f64(const XView &X, int c_end, int c_begin, uint32_t A, long long tau_last, int lane, FastCarry &cy, double (&W)[S])
{
    const int u = (int)((A - 1) / kB), v = (int)A - u * kB;
    double acc[S];
#pragma unroll
    for (int k = 0; k < S; k++) acc[k] = 0.0;
    for (int j = u; j >= 0; j--) {
        float2 x[kR][S];
        load_block<S>(X, (long long)(c_end - j), A, (long long)c_begin * kB, tau_last, lane, x);
#pragma unroll
        for (int r = 0; r < kR; r++) {
            const bool in = j < u || 2 * lane + r >= kB - v;
#pragma unroll
            for (int k = 0; k < S; k++) {
                const float e = in ? norm_f(x[r][k].x, x[r][k].y) : 0.0f;
                guard_track<false>(cy, e);  // (zero: neutral; inf / NaN: the call is the exact tier's, whose sums run like the reference's)
                acc[k] += (double)e;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < S; k++) W[k] = wave_sum_f64(acc[k]);
}

// the same for numAvg <= 128 from the LDS ring: the window is the positions >= kB - A of the block
// at `base`
template <bool DYN>
PSK_DEV float window_end_ring_f32(const ERingT<DYN> &er, int base, uint32_t A, int lane, int k)
{
    const float2 t = *reinterpret_cast<const float2 *>(er.row(k) + er.wrap(base + 2 * lane));
    float acc = 0.0f;
    if (2 * lane >= kB - (int)A)
        acc += t.x;
    if (2 * lane + 1 >= kB - (int)A)
        acc += t.y;
    return read_lane(wave_scan_f32(acc), 63);
}

// Exact-timing bookkeeping shared by the exact kernel and the in-place redo of the screened one.
// The window sums are float-valued energies accumulated in double.  When their exponent spread is
// small they are EXACT, in the reference as here, and even a tie resolves identically (first
// maximum).  When it is not, the reference's accumulators (rebuilt at the top of every call, quirk
// Q2, then updated twice per symbol, cpp/psk_soft.cpp:572-577) carry rounding errors of at most
// (updates so far) * 2^-53 * (largest sum so far), and so do ours; the argmax is still provably the
// reference's wherever best and runner-up differ by more than that.  `ArgTop` tracks both.
struct ArgTop {
    double best;
    float gap;  // best minus the largest other sum so far (rounded to float: only compared with a bound)
    int k;
};
PSK_DEV void argtop_first(ArgTop &t, double W)
{
    t.best = W;
    t.gap = __builtin_inff();
    t.k = 0;
}
PSK_DEV void argtop_next(ArgTop &t, double W, int k)
{
    const bool gt = t.best < W;  // std::max_element: first maximum, strict '<' (cpp/psk_soft.cpp:462)
    // A sum that is inf or NaN (a non-finite sample in the window, or one that left it: inf - inf) compares the way
    // IEEE says, in the reference as here -- a NaN never wins and is never beaten (phase 0 keeps the lead it starts
    // with, the others are passed over), the first inf beats everything finite: nothing there depends on rounding,
    // so such a pair is as unambiguous as a pair can be.
    const bool special = !(__builtin_fabs(W) < (double)__builtin_inff()) || !(__builtin_fabs(t.best) < (double)__builtin_inff());
    const float d = special ? __builtin_inff() : (float)(gt ? W - t.best : t.best - W);
    const float lo = d < t.gap ? d : t.gap;
    t.gap = gt ? d : lo;
    t.best = gt ? W : t.best;
    t.k = gt ? k : t.k;
}
// relative size of the rounding the two sides may have accumulated by output symbol i of the call
PSK_DEV float drift_bound(int symbols_so_far, uint32_t A)
{
    return (float)(4 * symbols_so_far + 2 * (int)A + 2048) * 2.220446e-16f;  // * 2^-52
}
PSK_DEV bool argtop_ambiguous(const ArgTop &t, float bound_abs)
{
    return !(t.gap > bound_abs);  // (NaN: ambiguous)
}

// The timing argmax of one block redone exactly from the LDS energy ring (numAvg <= 128), for the
// screened kernel when its margin test fails somewhere in the block: window sums of float-valued
// energies accumulated in double (exact under the exponent-spread guard, which is fed here and
// evaluated at the end of the kernel), the reference's first-maximum rule
// (cpp/psk_soft.cpp:445-466).  `base` = ring offset of the current block.
// Only the positions that failed the test are re-decided, and only the phases that contend there
// are summed exactly: the best and the runner-up of the screening at those positions, plus every
// other phase whose float sum -- recomputed here from the ring with the screening's own scan, hence
// within its error bound -- is not below best - thr at every failing position (thr covers twice the
// bound, see the screening pass).  On a shaped pulse two phases contend, so a redo costs two double scans
// and S - 2 float ones instead of S double ones.  (A NaN compares false: such a phase stays in.)
template <int S, bool FRONT = false>
PSK_DEV void exact_block_from_ring(const ERingT<ering_dynamic(S)> &er, int base, uint32_t A, int lane, FastCarry &cy, int (&bestK)[kR],
                                   float bound_abs, const bool (&fail)[kR], const float (&best_f)[kR],
                                   const int (&second_k)[kR], float thr, float bound_rel = 1.0f)
{
    const int i_new = er.wrap(base + 2 * lane);
    const int i_prev = er.wrap(base - kB + 2 * lane);
    const int i_old0 = er.wrap(base + 2 * lane - (int)A);
    const int i_old1 = er.wrap(i_old0 + 1);
    const bool in0 = 2 * lane >= kB - (int)A, in1 = 2 * lane + 1 >= kB - (int)A;
    // contenders at the failing positions, wave-wide
    unsigned mine = 0;
#pragma unroll
    for (int r = 0; r < kR; r++)
        if (fail[r])
            mine |= (1u << bestK[r]) | (1u << second_k[r]);
    unsigned cand = 0;
#pragma unroll
    for (int k = 0; k < S; k++)
        if (__any((mine >> k) & 1u))
            cand |= 1u << k;
    ArgTop top[kR];
    argtop_first(top[0], 0.0);
    argtop_first(top[1], 0.0);
    bool first = true;
#pragma unroll 1
    for (int k = 0; k < S; k++) {  // (a real loop: this path is rare, its registers and code size are not)
        const float *row = er.row(k);
        const float2 en = *reinterpret_cast<const float2 *>(row + i_new);
        const float eo0 = row[i_old0], eo1 = row[i_old1];
        if (!((cand >> k) & 1u)) {  // (wave-uniform) not a contender by the screening: make sure in float
            const float d0 = en.x - eo0, d1 = en.y - eo1;
            const float incl = wave_scan_f32(d0 + d1);
            // the float sum at the block's last position, re-summed from the ring (within 16 ulp)
            const float w_end = read_lane(wave_scan_f32((in0 ? en.x : 0.0f) + (in1 ? en.y : 0.0f)), 63);
            const float W1 = (w_end - read_lane(incl, 63)) + incl;
            const float W0 = W1 - d1;
            const bool below = (!fail[0] || W0 < best_f[0] - thr) && (!fail[1] || W1 < best_f[1] - thr);
            if (__all(below))
                continue;
        }
        const float2 ep = *reinterpret_cast<const float2 *>(row + i_prev);
        guard_track<false>(cy, en.x);
        guard_track<false>(cy, en.y);
        guard_track<false>(cy, ep.x);
        guard_track<false>(cy, ep.y);
        guard_track<false>(cy, eo0);
        guard_track<false>(cy, eo1);
        // the window ending at the previous block's last position: its positions >= kB - A
        const double Wc = wave_sum_f64((in0 ? (double)ep.x : 0.0) + (in1 ? (double)ep.y : 0.0));
        const double d0 = (double)en.x - (double)eo0, d1 = (double)en.y - (double)eo1;
        const double W1 = Wc + wave_scan_f64(d0 + d1);
        const double W0 = W1 - d1;
        if (first) {
            argtop_first(top[0], W0);
            argtop_first(top[1], W1);
            top[0].k = top[1].k = k;
            first = false;
        } else {
            argtop_next(top[0], W0, k);
            argtop_next(top[1], W1, k);
        }
    }
#pragma unroll
    for (int r = 0; r < kR; r++) {
        if (fail[r]) {
            bestK[r] = top[r].k;
            cy.ambiguous = cy.ambiguous || argtop_ambiguous(top[r], bound_abs);
            if constexpr (FRONT) {
                const float g = top[r].gap / bound_rel;
                cy.gap_rel = (g < cy.gap_rel) ? g : ((g == g) ? cy.gap_rel : 0.0f);  // (NaN: ambiguous whatever the scale)
            }
        }
    }
}

// Feedback unwrap + LinearFit::next over one block of 128 symbols (reference cpp/psk_soft.cpp:476-481, 48-87): the
// raw phases in, the phase estimates out, the carried sums / estimate / window bookkeeping of `cy` advanced to the
// block's last valid position.  `c` = index of the block in the call.
template <bool EXACT>
PSK_DEV void fit_stage(int c, int lane, uint32_t n, float xd, float den_s, float xavg_s, const FitKnown &fk, const bool (&valid)[kR],
                       const float (&raw)[kR], int nvalid, int lane_last, int r_last, float *yring, uint32_t ymask, FastCarry &cy,
                       float (&est)[kR])
{
    const uint32_t q0 = cy.q;
    float y[kR];
    double ySum_l[kR], xySum_l[kR];
    float den_last = den_s, xavg_last = xavg_s;
    int pass, rejected = 0;
    float m_lane = 0.0f;  // (per lane) slope of the lane's last fit
    // the fit window is still filling: the first phaseAvg symbols after a history clear.  phaseAvg == 1 takes that path for good: a
    // window of one point returns the point itself (cpp/psk_soft.cpp:164-171), whatever the sums have become -- after an infinite
    // phase they are NaN (inf - inf at :70) and the steady-state formula, which goes through them, would return NaN from then on
    const bool warm = !__builtin_expect(q0 >= n && n != 1u, 1);
    const bool cheap = cy.chain_run != 0;  // the recurrence ran on the last blocks: do not bother with candidates
    if (!warm) {
        pass = fit_block<false, EXACT>(lane, q0, n, xd, den_s, xavg_s, fk, valid, raw, cy, yring, ymask, y, est, ySum_l, xySum_l,
                                lane_last, r_last, den_last, xavg_last, cheap, rejected, m_lane);
    } else {
        pass = fit_block<true, EXACT>(lane, q0, n, xd, den_s, xavg_s, fk, valid, raw, cy, yring, ymask, y, est, ySum_l, xySum_l,
                               lane_last, r_last, den_last, xavg_last, cheap, rejected, m_lane);
    }
#ifdef PSK_ABL_NOCHAIN  /* (ablation builds only: what the chain costs) */
    rejected = 0;
#endif
    if (rejected) {
        // The candidates are not the reference's sums (or were not attempted): the recurrence itself, then the
        // estimates from ITS sums.  The unwrap counts were verified against the candidates' estimates, which
        // differ from these by an ulp here and there: a count that would change under them (a feedback within
        // an ulp of the half-way point of the unwrap) sends the call to the reference-order kernel.
        if (!warm)
            fit_sums_chain<false>((rejected & 1) != 0, lane, lane_last, q0, n, xd, fk.sizef, cy.ySum, cy.xySum, valid, yring, ymask, y, ySum_l, xySum_l);
        else
            fit_sums_chain<true>(true, lane, lane_last, q0, n, xd, fk.sizef, cy.ySum, cy.xySum, valid, yring, ymask, y, ySum_l, xySum_l);
        float est2[kR];
#pragma unroll
        for (int r = 0; r < kR; r++) {
            float m_, b_;
            if (!warm) {
                est2[r] = fit_value_known(ySum_l[r], xySum_l[r], fk, m_);
            } else {
                const uint32_t before = q0 + (uint32_t)(2 * lane + r);
                const uint32_t pts = before < n ? before + 1 : n;
                float den_r = den_s, xavg_r = xavg_s;
                if (pts > 1 && pts < n)
                    fit_denominator(xd, pts, den_r, xavg_r);
                est2[r] = pts > 1 ? fit_value(ySum_l[r], xySum_l[r], xd, pts, den_r, xavg_r, m_, b_) : y[r];
            }
        }
        const float fb0 = wave_up1(est2[1], cy.est), old_fb0 = wave_up1(est[1], cy.est);
        const bool diff0 = valid[0] && __float_as_uint(fb0) != __float_as_uint(old_fb0);
        const bool diff1 = valid[1] && __float_as_uint(est2[0]) != __float_as_uint(est[0]);
        if (__any(diff0 || diff1)) {  // (an ulp here and there: does any count see it?)
            const bool moved0 = diff0 && unwrap_count(fb0, (double)raw[0], c) != unwrap_count(old_fb0, (double)raw[0], c);
            const bool moved1 = diff1 && unwrap_count(est2[0], (double)raw[1], c) != unwrap_count(est[0], (double)raw[1], c);
            if (__any(moved0 || moved1))
                cy.refuse = true;
        }
        est[0] = est2[0];
        est[1] = est2[1];
        cy.stat_chain += 1;
        // a block whose candidates failed is usually followed by more of them (sums wandering around zero):
        // the next few blocks go straight to the recurrence, then the candidates get another try
        if (cheap)
            cy.chain_run -= 1;
        else if (!warm)
            cy.chain_run = (cy.chain_streak & 0xffffu) >= PSK_CHAIN_STREAK ? PSK_CHAIN_RUN : 0;
        cy.chain_streak = ((cy.chain_streak & 0xffffu) + 1u) | (16u << 16);
    } else {
        cy.chain_streak = (cy.chain_streak >> 16) ? ((cy.chain_streak >> 16) - 1u) << 16 : 0u;
    }
    if (pass > kMaxUnwrapPasses)
        cy.refuse = true;
    cy.stat_blocks += 1;
    cy.stat_extra += (uint32_t)pass;
    // ---- carries into the next block: the last valid position of this one ----
    {
        const double ys = r_last ? ySum_l[1] : ySum_l[0];
        const double xys = r_last ? xySum_l[1] : xySum_l[0];
        const float e_ = r_last ? est[1] : est[0];
        cy.ySum = read_lane(ys, lane_last);
        cy.xySum = read_lane(xys, lane_last);
        cy.est = read_lane(e_, lane_last);
        {
            // slope of the line just fitted, per symbol (LinearFit::m * xdelta, steady-state
            // constants): only a hint for the next block's speculation, so approximate is fine
            const float m_hint = read_lane(m_lane, lane_last) * xd;  // (of the wave-parallel candidates where the chain ran: a hint)
            cy.slope = is_fin(m_hint) ? m_hint : 0.0f;
        }
        const uint32_t pts_last = (q0 + (uint32_t)nvalid - 1 >= n) ? n : q0 + (uint32_t)nvalid;
        if (pts_last > 1) {  // calculateDenominator ran for the window size reached (cpp/psk_soft.cpp:81-83)
            cy.den = den_last;
            cy.xavg = xavg_last;
        }
    }
    cy.q = q0 + (uint32_t)nvalid;
}

// De-rotation and hard decisions of one block (reference cpp/psk_soft.cpp:484-566) and its soft / phase / bits output,
// two symbols per lane.  `last_c` = the sample output just before the block's first one (differential decoding);
// G: the output rows are addressed as global memory (see PSK_GLOBAL), EXACT: libgcc's complex-multiply recovery.
template <bool G, bool EXACT>
PSK_DEV void output_stage(const ChanPlan &p, int c, int i0, const bool (&valid)[kR], const cf32 (&s)[kR], const float (&est)[kR],
                          cf32 last_c, const AtanTabDev &atab, bool qpsk_sign_map, bool m_pow2, float inv_M)
{
    const uint32_t M = p.M;
    cf32 corr[kR];
#pragma unroll
    for (int r = 0; r < kR; r++) {
        float phaseCorrection = 0.0f;
        cf32 smp = s[r];
        if (p.diff) {
            cf32 last;
            if (r == 0) {
                last.re = wave_up1(s[1].re, last_c.re);
                last.im = wave_up1(s[1].im, last_c.im);
            } else {
                last = s[0];
            }
            smp = cdiv<true>(s[r], last);  // (__divsc3's recovery behind a wave-uniform test: a silent
                                           //  stream divides by zero on every symbol)
        } else {
            phaseCorrection = m_pow2 ? (-est[r]) * inv_M : -est[r] / (float)M;
        }
        if (M == 4)
            phaseCorrection = (float)((double)phaseCorrection + PSK_KD(kPi4, c));
        float sn, cs;
        sincosf_wave(phaseCorrection, &sn, &cs, c);
        cf32 ph;
        ph.re = 1.0f * cs;
        ph.im = 1.0f * sn;
        // (__mulsc3's recovery only ever changes a product with an infinite factor: impossible
        // without differential decoding, where smp is a sample whose M-th power was finite)
        corr[r] = (p.diff || EXACT) ? cmul<true>(smp, ph) : cmul<false>(smp, ph);
    }

    // ---- four output streams, two symbols per lane ----
    unsigned short sym8[kR] = {0, 0};
    if (p.bits && p.bpb == 3) {  // 8-PSK slicing with the whole wave active (the atan2f table lives in lanes 0-4)
#pragma unroll
        for (int r = 0; r < kR; r++) {
            sym8[r] = slice_8psk(corr[r].re, corr[r].im, atab);
        }
    }
    if (valid[1]) {
        if (p.soft) {
            f4u v = {corr[0].re, corr[0].im, corr[1].re, corr[1].im};
            store_f4u(mem_ptr<G>(p.soft) + 2 * (size_t)i0, v);
        }
        if (p.phase) {
            f2u v = {est[0], est[1]};
            store_f2u(mem_ptr<G>(p.phase) + i0, v);
        }
        if (!p.bits) {
        } else if (p.bpb == 1) {
            s2u v = {(short)(corr[0].re < 0), (short)(corr[1].re < 0)};
            store_s2u(mem_ptr<G>(p.bits) + i0, v);
        } else if (p.bpb == 2) {  // quirk Q1 (float -> bool is "!= 0") unless the sign map was asked for
            int a0, a1, b0, b1;
            qpsk_bits(corr[0].re, corr[0].im, qpsk_sign_map, a0, a1);
            qpsk_bits(corr[1].re, corr[1].im, qpsk_sign_map, b0, b1);
            s4u v = {(short)a0, (short)a1, (short)b0, (short)b1};
            store_s4u(mem_ptr<G>(p.bits) + 2 * i0, v);
        } else if (p.bpb == 3) {
            const unsigned short a = sym8[0], b = sym8[1];
            s2u v0 = {(short)(a & 1), (short)((a >> 1) & 1)};
            s2u v1 = {(short)((a >> 2) & 1), (short)(b & 1)};
            s2u v2 = {(short)((b >> 1) & 1), (short)((b >> 2) & 1)};
            store_s2u(mem_ptr<G>(p.bits) + 3 * i0, v0);
            store_s2u(mem_ptr<G>(p.bits) + 3 * i0 + 2, v1);
            store_s2u(mem_ptr<G>(p.bits) + 3 * i0 + 4, v2);
        }
    } else if (valid[0]) {  // an odd tail: one symbol
        if (p.soft) {
            mem_ptr<G>(p.soft)[2 * (size_t)i0] = corr[0].re;
            mem_ptr<G>(p.soft)[2 * (size_t)i0 + 1] = corr[0].im;
        }
        if (p.phase)
            mem_ptr<G>(p.phase)[i0] = est[0];
        if (!p.bits) {
        } else if (p.bpb == 1) {
            mem_ptr<G>(p.bits)[i0] = (int16_t)(corr[0].re < 0);
        } else if (p.bpb == 2) {
            int a0, a1;
            qpsk_bits(corr[0].re, corr[0].im, qpsk_sign_map, a0, a1);
            mem_ptr<G>(p.bits)[2 * i0] = (int16_t)a0;
            mem_ptr<G>(p.bits)[2 * i0 + 1] = (int16_t)a1;
        } else if (p.bpb == 3) {
            const unsigned short a = sym8[0];
            mem_ptr<G>(p.bits)[3 * i0] = (int16_t)(a & 1);
            mem_ptr<G>(p.bits)[3 * i0 + 1] = (int16_t)((a >> 1) & 1);
            mem_ptr<G>(p.bits)[3 * i0 + 2] = (int16_t)((a >> 2) & 1);
        }
    }
}

#ifndef PSK_SELECT_UNIFORM
#define PSK_SELECT_UNIFORM 1
#endif
// ---- paced priorities: see fast_main_loop ----
#ifndef PSK_PACE
#define PSK_PACE 1
#endif
#define PSK_PACE_ON(FRONT_, EXACT_) (PSK_PACE != 0 && !(FRONT_) && !(EXACT_))
#ifndef PSK_PACE_SHORT
#define PSK_PACE_SHORT 12
#endif
#ifndef PSK_PACE_SHORT_MODE
#define PSK_PACE_SHORT_MODE 2
#endif
#ifndef PSK_PACE_EVERY
#define PSK_PACE_EVERY 4  /* blocks between two looks at the table (a power of two): every block 2.57 ms, every second 2.39, every fourth 2.37, never 2.44 (headline, one box) */
#endif
constexpr int kPaceRows = 2048, kPaceSlots = 8;  // rows: XCC (3 bits) : SE (2) : CU (4) : SIMD (2); slots: WAVE_ID
// (one table per translation unit, i.e. per instantiation of the kernel: waves of different kernels that meet on a SIMD do
// not see one another -- they just keep the hardware's oldest-first order among themselves)
static __device__ uint32_t g_pace_table[kPaceRows * kPaceSlots];
// this wave's place in the table: word index of its SIMD's row, and its slot in the row (bits 16 ..)
PSK_DEV uint32_t pace_key()
{
    const uint32_t hw = (uint32_t)__builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID: WAVE_ID [3:0] SIMD_ID [5:4] CU_ID [11:8] SE_ID [15:13]
    const uint32_t xcc = (uint32_t)__builtin_amdgcn_s_getreg((3 << 11) | 20);   // HW_REG_XCC_ID [3:0]
    const uint32_t r = ((xcc & 7u) << 8) | (((hw >> 13) & 3u) << 6) | (((hw >> 8) & 15u) << 2) | ((hw >> 4) & 3u);
    return (r * (uint32_t)kPaceSlots) | ((hw & 7u) << 16);
}
// the row, past the vector cache (sc1: the other waves' stores sit in the XCD's L2): the lane's word (lanes 0 .. 7)
PSK_DEV uint32_t pace_fetch(uint32_t key, int lane)
{
    uint32_t v = 0u;
    if (lane < kPaceSlots)
        v = __hip_atomic_load(g_pace_table + (key & 0xffffu) + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return v;
}
// publishes the blocks this wave has left (0 = gone)
PSK_DEV void pace_post(uint32_t key, uint32_t left, int lane)
{
    if (lane == 0)
        __hip_atomic_store(g_pace_table + (key & 0xffffu) + (key >> 16), left, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// priority by rank: 3 for the wave with the most blocks left on its SIMD, one less for every neighbour that has more
PSK_DEV int pace_rank(uint32_t key, uint32_t row_word, uint32_t left, int lane)
{
    const unsigned long long more = __builtin_amdgcn_ballot_w64(lane < kPaceSlots && lane != (int)(key >> 16) && row_word > left);
    const int ahead = __builtin_popcountll(more);
    return ahead >= 3 ? 0 : 3 - ahead;
}

// FRONT = true is the first stage of the time-tiled kernels (psk_tile_kernel.h): the same loop over the blocks
// [c_begin, c_end) of the call -- its window rebuilt from the numAvg - 1 symbols in front of block c_begin exactly as
// the call's first window is rebuilt from the carried samples --, stopping after the raw phase: the picked samples
// and their raw phases go to scratch (t_s, t_raw: the channel's arrays, indexed by output symbol), sampleIndex to
// the caller.  Unwrap, fit and de-rotation happen in the later stages.
template <int S, int H, bool EXACT, bool FRONT = false>
PSK_DEV void fast_main_loop(const ChanPlan &p, const XView &X, float *yring, uint32_t ymask, const ERingT<ering_dynamic(S)> &er, FastCarry &cy,
                            int c_begin = 0, int c_end = -1, float *t_raw = nullptr, float2 *t_s = nullptr, float wmax_floor = 0.0f)
{

    const int lane = threadIdx.x & 63;
    const uint32_t A = p.A, M = p.M, n = p.lf_n;
    const int n_out = (int)p.n_out;  // <= 2^20 on this path
    // the caller's output rows (address space: see PSK_GLOBAL in psk_wave.h).  Not copied into locals:
    // four pointers held across the loop are eight more scalar registers under pressure (measured
    // +3 ... +6 % at samplesPerBaud 2 and 4); the plan is re-read where a store needs one.
#define g_soft mem_ptr<packet_global(S)>(p.soft)
#define g_phase mem_ptr<packet_global(S)>(p.phase)
#define g_bits mem_ptr<packet_global(S)>(p.bits)
#define g_sidx mem_ptr<packet_global(S)>(p.sidx)
    const float xd = p.lf_xdelta;
    const long long tau_last = (long long)n_out + (long long)A - 2;  // newest symbol any emitted window uses

    // ---- prologue: the first A-1 symbols (the carried window) as "blocks" -H .. -1:
    //      W_k(-1) = their energy sums (= resyncEnergy, reference cpp/psk_soft.cpp:619-636) ----
    //      H == 0 (REREAD): no history at all.  A window longer than a block costs 22 registers per block of history
    //      (numAvg 400: 88, two waves on a SIMD instead of four); this variant reads the samples of the symbols that LEAVE the
    //      window a second time instead -- they passed through numAvg symbols ago, L2 / Infinity Cache territory -- and
    //      computes their energies again (the same float operation on the same floats: the same energies).
    constexpr bool REREAD = (H == 0);
    BlockKeep<S> hist[H ? H : 1];
    double Wc[S];  // EXACT: the exact carried window sums
    float Wf[S];   // screened: their float shadow
    if constexpr (REREAD) {
        static_assert(!REREAD || !EXACT, "the exact tier keeps its history in registers");
        double acc[S];
#pragma unroll
        for (int k = 0; k < S; k++) acc[k] = 0.0;
        // (block -(h+1) holds the symbols tau = kB * (c_begin - h - 1) + s + A - 1 of the carried window [kB * c_begin, + A - 2]:
        // something for h + 1 <= (A + 126) / kB; the same order of additions as the register variant)
        for (int h = (int)((A + 126u) / (uint32_t)kB) - 1; h >= 0; h--) {
            float2 x[kR][S];
            load_block<S>(X, (long long)c_begin - (long long)(h + 1), A, (long long)c_begin * kB, (long long)c_begin * kB + (long long)A - 2, lane, x);
#pragma unroll
            for (int r = 0; r < kR; r++) {
#pragma unroll
                for (int k = 0; k < S; k++) acc[k] += (double)norm_f(x[r][k].x, x[r][k].y);
            }
        }
#pragma unroll
        for (int k = 0; k < S; k++) {
            Wc[k] = wave_sum_f64(acc[k]);
            Wf[k] = uni((float)Wc[k]);
        }
    } else {
        double acc[S];
#pragma unroll
        for (int k = 0; k < S; k++) acc[k] = 0.0;
#pragma unroll
        for (int h = H - 1; h >= 0; h--) {
            float2 x[kR][S];
            load_block<S>(X, (long long)c_begin - (long long)(h + 1), A, (long long)c_begin * kB, (long long)c_begin * kB + (long long)A - 2, lane, x);
#pragma unroll
            for (int r = 0; r < kR; r++) {
#pragma unroll
                for (int k = 0; k < S; k++) {
                    float e = norm_f(x[r][k].x, x[r][k].y);
                    if (EXACT)
                        guard_track<true>(cy, e);  // zero-filled (absent) symbols are neutral
                    hist[h].e[r][k] = e;
                    acc[k] += (double)e;
                }
                hist[h].kp[r] = (int)cy.last_k;
                hist[h].pk[r] = select_sample<S>(x[r], (int)cy.last_k);
            }
        }
#pragma unroll
        for (int k = 0; k < S; k++) {
            Wc[k] = wave_sum_f64(acc[k]);
            Wf[k] = uni((float)Wc[k]);
        }
    }
    // error bound of the float shadow (screened path)
    const float kU = 5.9604645e-08f;  // 2^-24
    float wmax_prev = 0.0f;
#pragma unroll
    for (int k = 0; k < S; k++) wmax_prev = __builtin_fmaxf(wmax_prev, Wf[k]);
    // (a tile starts from what the channel's last call says its window sums can reach: see ChanState::emax_hint; a piece of a
    // call cut in time from the largest sum of the pieces before it: PLAN_CARRY_DRIFT; zero everywhere else)
    wmax_prev = __builtin_fmaxf(wmax_prev, wmax_floor);
    if constexpr (EXACT)
        cy.wmax = __builtin_fmaxf(cy.wmax, wmax_floor);
    float err_c = 2.0f * kU * wmax_prev;
    // rounding-error budget of one screened block, relative to the largest window sum in play:
    // the local roundings of up to 2*(128/A+1) window-loads of energy pass through the scan, plus
    // the scan's own additions and the carry
    const float c_blk = uni(kU * (64.0f + 16.0f * ((float)kB / (float)A + 1.0f)));
    int since_refresh = 0;

    // cross-lane fetch parameters: energies leave the window A symbols after they entered it;
    // the picked-from symbol entered it A-1 symbols ago
    const RotParam rotE = rot_param(lane, A);
    const RotParam rotP = rot_param(lane, A - 1);
    // (numAvg <= 128: both fetches reach at most one block back, known at compile time)
    const int uE = (H <= 1) ? 0 : rotE.u, uP = (H <= 1) ? 0 : rotP.u;

    // steady-state fit constants
    float den_s = cy.den, xavg_s = cy.xavg;
    if (n > 1)
        fit_denominator(xd, n, den_s, xavg_s);
    const FitKnown fk = fit_known(xd, n, den_s, xavg_s);
    // -est/M (cpp/psk_soft.cpp:494): for a power-of-two M the division is an exact scaling
    const bool m_pow2 = M != 0 && (M & (M - 1)) == 0;
    const float inv_M = uni(1.0f / (float)(M ? M : 1));

    const int n_blocks = (FRONT && c_end >= 0) ? c_end : (n_out + kB - 1) / kB;
    const AtanTabDev atab = atan_tab_dev(lane);  // range table of the straight-line atan2f
    const bool qpsk_sign_map = (p.lf_flags & PLAN_QPSK_SIGN_MAP) != 0;
    int kpred[kR] = {(int)cy.last_k, (int)cy.last_k};  // timing index this lane chose one block ago

    int ring_base = 0;  // (H == 1) ring offset of the current block
    if constexpr (H == 1)
        ering_put<S>(er, er.length() - kB, lane, hist[0].e);  // block -1

    // ---- paced priorities (PSK_PACE) ----
    // The four waves that share a SIMD are issued oldest first: left to itself the first-dispatched wave of a SIMD runs a
    // fifth faster than the last (measured on the headline: the four dispatch quarters of a launch end at 1.95 / 2.06 /
    // 2.21 / 2.37 ms), and from the moment the oldest waves retire the SIMDs run with three, two, one wave -- a launch lasts
    // as long as its youngest waves while the machine idles behind the others.  Every wave therefore publishes the blocks it
    // has LEFT in a table in global memory, one row per SIMD (found from HW_ID / XCC_ID) and one word per wave slot, reads the
    // row once a block, and takes the priority of its rank: most work left, highest priority.  The waves of a SIMD then
    // advance together and end together, whatever slows one of them down (an older neighbour, blocks that need the
    // lane-after-lane chain, exact timing redos).  Priorities change no result; stale or missing table entries only cost speed.
    // A batch that mixes window classes: the waves of the long windows (two to a SIMD, or one beside two of the short class)
    // post their count with a bias that ranks them in front of every short-window wave.  They are the launch's long pole --
    // register-bound residency, so a second round of short-window waves has to wait for their registers -- and the sooner
    // they are through, the sooner the machine is back to four waves a SIMD.
    constexpr uint32_t kPaceBias = (PSK_WIDE_FIRST && H >= 2) ? (1u << 24) : 0u;
    uint32_t pace = 0u;
    if constexpr (PSK_PACE_ON(FRONT, EXACT))
        pace = pace_key();
    auto set_prio_dyn = [](int v) {
        switch (v) {
        case 0: __builtin_amdgcn_s_setprio(0); break;
        case 1: __builtin_amdgcn_s_setprio(1); break;
        case 2: __builtin_amdgcn_s_setprio(2); break;
        default: __builtin_amdgcn_s_setprio(3); break;
        }
    };
    // Windows longer than a block (H >= 2) run two waves to a SIMD, 172 ... 230 registers each: too few waves to hide the
    // latency of a block's loads behind each other's arithmetic, and registers to spare (256 a wave at that residency).  There
    // the samples of block c + 1 are requested before block c is worked on.  (At numAvg <= 128, four waves to a SIMD and not
    // a register free, the same was measured 1.5 % slower.)
    constexpr bool DBUF = PSK_DBUF && H >= 2 && !FRONT && !EXACT;
    float2 xnext[kR][S];
    if constexpr (DBUF) {
        if (c_begin < n_blocks)
            load_block<S>(X, (long long)c_begin, A, 0, tau_last, lane, xnext);
    }
#if PSK_PACE_SHORT_MODE == 2
    const int pace_mask = n_blocks <= PSK_PACE_SHORT ? 0 : PSK_PACE_EVERY - 1;
#endif
    for (int c = c_begin; c < n_blocks; c++) {
        // Without pacing: the memory-facing half of the block (loads, their use, the LDS ring) runs at raised wave
        // priority, the arithmetic half (pow, atan2f, fit, sincosf) at normal: requests go out early
        // (measured -1 %; requesting the samples a whole block ahead measured +1.5 %, touching the next block's
        // lines with a one-dword load per lane half a block ahead +9 %: the loads in flight are not the limit)
        // (the front stage of the time-tiled path stays below the serial fit stage, which in the pipelined mode runs beside it
        // on the same SIMDs at priority 3: psk_tile.hip)
        if constexpr (!PSK_PACE_ON(FRONT, EXACT))
            __builtin_amdgcn_s_setprio(FRONT ? 1 : 3);
        uint32_t pace_row = 0u;
        // (wave-uniform; a call of a few blocks looks every block: it is over before the fourth)
#if PSK_PACE_SHORT_MODE == 1
        const bool pace_now = PSK_PACE_ON(FRONT, EXACT) && ((c & (PSK_PACE_EVERY - 1)) == 0 || n_blocks <= PSK_PACE_SHORT);
#elif PSK_PACE_SHORT_MODE == 3
        const bool pace_now = PSK_PACE_ON(FRONT, EXACT) && ((c & (PSK_PACE_EVERY - 1)) == 0 || c < PSK_PACE_SHORT);
#elif PSK_PACE_SHORT_MODE == 2
        const bool pace_now = PSK_PACE_ON(FRONT, EXACT) && (c & pace_mask) == 0;
#else
        const bool pace_now = PSK_PACE_ON(FRONT, EXACT) && (c & (PSK_PACE_EVERY - 1)) == 0;
#endif
        if (pace_now)
            pace_row = pace_fetch(pace, lane);  // (in front of the block's loads: it is back before they are)
        float2 xn[kR][S];
        if constexpr (DBUF) {
#pragma unroll
            for (int r = 0; r < kR; r++) {
#pragma unroll
                for (int k = 0; k < S; k++) xn[r][k] = xnext[r][k];
            }
            if (c + 1 < n_blocks)  // (wave-uniform)
                load_block<S>(X, (long long)(c + 1), A, 0, tau_last, lane, xnext);
        } else {
            load_block<S>(X, (long long)c, A, 0, tau_last, lane, xn);
        }
        if (pace_now) {
            pace_post(pace, (uint32_t)(n_blocks - c) + kPaceBias, lane);  // (behind them: nothing waits for a store)
            set_prio_dyn(pace_rank(pace, pace_row, (uint32_t)(n_blocks - c) + kPaceBias, lane));
        }

        const int i0 = c * kB + 2 * lane;  // first output symbol of this lane
        bool valid[kR];
        valid[0] = i0 < n_out;
        valid[1] = i0 + 1 < n_out;
        const int rem = n_out - c * kB;
        const int nvalid = rem < kB ? rem : kB;  // valid positions of this block
        const int lane_last = (nvalid - 1) >> 1, r_last = (nvalid - 1) & 1;

        // REREAD: the symbols that leave the window at this block's positions -- output symbol i drops symbol i - 1 -- and,
        // for the symbols being output, the sample at the timing index their lane chose a block ago (verified below, like
        // the kept samples of the other variants); all of it requested here, behind the new symbols
        float e_old[kR][S];
        float px[kR], py[kR];
        int pkk[kR];
        if constexpr (REREAD) {
            float2 xo[kR][S];
            load_block<S>(X, (long long)c, 0u, (long long)c_begin * kB, tau_last, lane, xo);  // tau = kB * c + s - 1
#pragma unroll
            for (int r = 0; r < kR; r++) {
                const uint64_t j = (uint64_t)(c * kB + 2 * lane + r) * S + (uint64_t)kpred[r];  // (exists: A > kB)
                const f2g *q = j < X.L0 ? X.ring + j : X.in + (j - X.L0);
                const f2g g = *mem_ptr<packet_global(S)>(q);
                px[r] = g.x;
                py[r] = g.y;
                pkk[r] = kpred[r];
            }
#pragma unroll
            for (int r = 0; r < kR; r++) {
#pragma unroll
                for (int k = 0; k < S; k++) e_old[r][k] = norm_f(xo[r][k].x, xo[r][k].y);
            }
        }

        // ================= timing recovery =================
        // energies of the new symbols; of their samples keep the one at the predicted timing index
        BlockKeep<S> cur;
#pragma unroll
        for (int r = 0; r < kR; r++) {
#pragma unroll
            for (int k = 0; k < S; k++) {
                float e = norm_f(xn[r][k].x, xn[r][k].y);
                if (EXACT)
                    guard_track<true>(cy, e);
                if constexpr (FRONT)
                    cy.emax = __builtin_fmaxf(cy.emax, e);
                cur.e[r][k] = e;
            }
            cur.kp[r] = kpred[r];
        }
        if constexpr (!REREAD) {
            const int k0u = __builtin_amdgcn_readfirstlane(kpred[0]), k1u = __builtin_amdgcn_readfirstlane(kpred[1]);
            if (PSK_SELECT_UNIFORM && vote_all(kpred[0] == k0u && kpred[1] == k1u)) {
                cur.pk[0] = select_sample_uniform<S>(xn[0], k0u);
                cur.pk[1] = select_sample_uniform<S>(xn[1], k1u);
            } else {
                cur.pk[0] = select_sample<S>(xn[0], kpred[0]);
                cur.pk[1] = select_sample<S>(xn[1], kpred[1]);
            }
        }
        // energy of symbol i-1 for every phase (it entered the window A symbols before symbol
        // i+A-1 did): all cross-lane fetches issued back to back
        if constexpr (REREAD) {
        } else if constexpr (H == 1) {
            ering_put<S>(er, ring_base, lane, cur.e);
            wave_lds_fence();
            ering_get<S>(er, ring_base, lane, A, e_old);
        } else {
#pragma unroll
            for (int k = 0; k < S; k++) {
                float nw[kR], od[kR];
#pragma unroll
                for (int rr = 0; rr < kR; rr++) {
                    nw[rr] = block_back<S, H>(uE, cur, hist, [&](const BlockKeep<S> &b) { return b.e[rr][k]; });
                    od[rr] = block_back<S, H>(uE + 1, cur, hist, [&](const BlockKeep<S> &b) { return b.e[rr][k]; });
                }
#pragma unroll
                for (int r = 0; r < kR; r++) e_old[r][k] = rot_pull<float>(rotE, r, nw, od);
            }
        }
        // the sample kept A-1 symbols ago for the symbol now being output, and the index it was kept at
        if constexpr (!REREAD) {
            float nx[kR], ox[kR], ny[kR], oy[kR];
            int nk[kR], ok2[kR];
#pragma unroll
            for (int rr = 0; rr < kR; rr++) {
                nx[rr] = block_back<S, H>(uP, cur, hist, [&](const BlockKeep<S> &b) { return b.pk[rr].x; });
                ox[rr] = block_back<S, H>(uP + 1, cur, hist, [&](const BlockKeep<S> &b) { return b.pk[rr].x; });
                ny[rr] = block_back<S, H>(uP, cur, hist, [&](const BlockKeep<S> &b) { return b.pk[rr].y; });
                oy[rr] = block_back<S, H>(uP + 1, cur, hist, [&](const BlockKeep<S> &b) { return b.pk[rr].y; });
                nk[rr] = block_back<S, H>(uP, cur, hist, [&](const BlockKeep<S> &b) { return b.kp[rr]; });
                ok2[rr] = block_back<S, H>(uP + 1, cur, hist, [&](const BlockKeep<S> &b) { return b.kp[rr]; });
            }
#pragma unroll
            for (int r = 0; r < kR; r++) {
                px[r] = rot_pull<float>(rotP, r, nx, ox);
                py[r] = rot_pull<float>(rotP, r, ny, oy);
                pkk[r] = rot_pull<int>(rotP, r, nk, ok2);
            }
        }

        int bestK[kR] = {0, 0};
        if constexpr (!EXACT) {
            // ---- screening pass in float ----
            // All S scans run interleaved.  The argmax works on the float bit patterns as
            // integers (the sums are >= 0 up to their error bound; the sign bit is dropped): the
            // low IB bits carry the phase index, so one max chain yields best sum AND index, and
            // a median-of-three chain the runner-up.  Dropping the sign and overwriting IB low
            // bits moves a value by at most 2*E + 2^IB ulp; the acceptance threshold below
            // absorbs that.  A NaN anywhere has the largest pattern and fails the test.
            constexpr int IB = S <= 2 ? 1 : S <= 4 ? 2 : S <= 8 ? 3 : S <= 16 ? 4 : 5;
            constexpr int IMASK = (1 << IB) - 1;
            float inc[S], d1v[S];
#pragma unroll
            for (int k = 0; k < S; k++) {
                const float d0 = cur.e[0][k] - e_old[0][k];
                d1v[k] = cur.e[1][k] - e_old[1][k];
                inc[k] = d0 + d1v[k];
            }
            wave_scan_f32_multi(inc);
            int m1[kR], m2[kR];
            int nz = 0;  // (OR of the sums' bit patterns: is any of them not exactly zero?)
#pragma unroll
            for (int k = 0; k < S; k++) {
                const float W1 = Wf[k] + inc[k];
                const float W0 = W1 - d1v[k];
                nz |= __float_as_int(W0) | __float_as_int(W1);
                Wf[k] = read_lane(W1, 63);
                const int p0 = (__float_as_int(W0) & (0x7FFFFFFF & ~IMASK)) | (IMASK - k);
                const int p1 = (__float_as_int(W1) & (0x7FFFFFFF & ~IMASK)) | (IMASK - k);
                if (k == 0) {
                    m1[0] = p0;
                    m1[1] = p1;
                } else if (k == 1) {
                    m2[0] = p0 < m1[0] ? p0 : m1[0];
                    m2[1] = p1 < m1[1] ? p1 : m1[1];
                    m1[0] = p0 > m1[0] ? p0 : m1[0];
                    m1[1] = p1 > m1[1] ? p1 : m1[1];
                } else {
                    m2[0] = med3_i32(m1[0], m2[0], p0);
                    m2[1] = med3_i32(m1[1], m2[1], p1);
                    m1[0] = p0 > m1[0] ? p0 : m1[0];
                    m1[1] = p1 > m1[1] ? p1 : m1[1];
                }
            }
            bestK[0] = IMASK - (m1[0] & IMASK);
            bestK[1] = IMASK - (m1[1] & IMASK);
            const int mm = m1[0] > m1[1] ? m1[0] : m1[1];
            const float wmax = __builtin_fmaxf(wave_max_f32(__int_as_float(mm)), wmax_prev);
            const float e_blk = c_blk * wmax;
            // The bounds above are RELATIVE (2^-24 of the largest sum) and so is what overwriting the low IB bits costs -- in the
            // normal range.  Window sums down among the denormals (samples of 1e-20 and below) sit on a fixed grid of 2^-149: there
            // the index bits alone move a sum by up to 2^IB grid steps, far more than 2^-24 of it, and the products above underflow.
            // An absolute floor of a few such steps covers both (nothing to a sum of ordinary size); a block whose sums are ALL
            // exactly zero (a silent stretch: the reference's first phase wins, and so does pattern IMASK - 0) is exempt.
            // (Found by the randomised comparison at amplitudes of 1e-21, PSK_FUZZ_EXTREME.)
            const float thr_floor = vote_any((nz & 0x7FFFFFFF) != 0) ? (float)(16 << IB) * 1.4012985e-45f : 0.0f;
            const float thr = 6.0f * (err_c + e_blk) + (float)(4 << IB) * kU * wmax + thr_floor;
            // (NaN / inf anywhere makes the comparison false)
            const bool ok0 = !valid[0] || ((__int_as_float(m1[0]) - __int_as_float(m2[0])) > thr);
            const bool ok1 = !valid[1] || ((__int_as_float(m1[1]) - __int_as_float(m2[1])) > thr);
            if constexpr (FRONT) {
                // (a position the screening accepts beat its runner-up by more than thr: that decides it against the
                // reference's sums too while thr >= 2 * drift * (largest sum of the call) -- the single wave of the
                // wave-scan kernel knows that maximum as it goes, a tile does not)
                if (__any((valid[0] && ok0) || (valid[1] && ok1))) {
                    const float cap_blk = thr / (2.0f * drift_bound(c * kB + kB, A));
                    cy.cap = cap_blk < cy.cap ? cap_blk : cy.cap;
                }
            }
            err_c += e_blk;
            wmax_prev = wmax;
            since_refresh++;
            // a near-tie (or a non-finite energy) anywhere: the call goes to the exact kernel
            if (!__all(ok0 && ok1)) {
                if constexpr (H == 1) {
                    // settle this block exactly, here.  The float carries stay valid (still within
                    // their error bound); they are re-summed at the end of this block so that the
                    // bound, and with it the acceptance threshold, starts small again.
                    const bool fail[kR] = {!ok0, !ok1};
                    const float best_f[kR] = {__int_as_float(m1[0]), __int_as_float(m1[1])};
                    const int second_k[kR] = {IMASK - (m2[0] & IMASK), IMASK - (m2[1] & IMASK)};
                    // (symbols since the reference last rebuilt its sums: this call's, plus -- PLAN_CARRY_DRIFT -- the earlier pieces')
                    const int sym0 = (p.lf_flags & PLAN_CARRY_DRIFT) ? (int)p.count0 : 0;
                    exact_block_from_ring<S, FRONT>(er, ring_base, A, lane, cy, bestK,
                                                    2.0f * drift_bound(sym0 + c * kB + kB, A) * wmax_prev, fail, best_f, second_k, thr,
                                                    2.0f * drift_bound(sym0 + c * kB + kB, A));
                    since_refresh = kScreenRefresh;
                    cy.stat_exact_blocks += 1;
                } else if (PSK_TIES_IN_PLACE && !(p.lf_flags & PLAN_TIES_HANDOVER)) {
                    // windows longer than a block: this block settled exactly too, here -- the exact tier's pass over it (float-valued
                    // addends in double, the reference's first-maximum rule, the ambiguity left to the guard at the end of the kernel),
                    // started from window sums rebuilt from the history.  (Handing the whole call to the exact tier instead -- one or
                    // two waves to a SIMD -- cost a noisy batch of numAvg 400 80 % and 8-PSK at 10 samples per baud 160 %.)
                    double Wst[S];
                    if constexpr (REREAD)
                        window_end_reread_f64<S>(X, c - 1, c_begin, A, tau_last, lane, cy, Wst);
                    else
                        window_end_f64<S, H>(hist, A, lane, cy, Wst);
                    ArgTop top[kR];
#pragma unroll
                    for (int k = 0; k < S; k++) {
                        guard_track<false>(cy, cur.e[0][k]);  // (an energy that is not finite: cy.refuse, as in the numAvg <= 128 redo)
                        guard_track<false>(cy, cur.e[1][k]);
                        const double d0 = (double)cur.e[0][k] - (double)e_old[0][k];
                        const double d1 = (double)cur.e[1][k] - (double)e_old[1][k];
                        const double incl = wave_scan_f64(d0 + d1);
                        const double W1 = Wst[k] + incl;
                        const double W0 = (Wst[k] + wave_up1(incl, 0.0)) + d0;
                        if (k == 0) {
                            argtop_first(top[0], W0);
                            argtop_first(top[1], W1);
                        } else {
                            argtop_next(top[0], W0, k);
                            argtop_next(top[1], W1, k);
                        }
                    }
                    const int sym0 = (p.lf_flags & PLAN_CARRY_DRIFT) ? (int)p.count0 : 0;
                    const float bound_abs = 2.0f * drift_bound(sym0 + c * kB + kB, A) * wmax_prev * 1.00001f;
#pragma unroll
                    for (int r = 0; r < kR; r++) {
                        bestK[r] = top[r].k;
                        cy.ambiguous = cy.ambiguous || (valid[r] && argtop_ambiguous(top[r], bound_abs));
                    }
                    since_refresh = kScreenRefresh;
                    cy.stat_exact_blocks += 1;
                } else {
                    cy.refuse = true;
                    return;
                }
            }
        } else {
            // ---- exact pass: float-valued addends summed in double ----
            ArgTop top[kR];
#pragma unroll
            for (int k = 0; k < S; k++) {
                // (positions past the end of the call only pollute sums of later positions, which
                // are past the end too: no masking needed)
                double d0 = (double)cur.e[0][k] - (double)e_old[0][k];
                double d1 = (double)cur.e[1][k] - (double)e_old[1][k];
                double incl = wave_scan_f64(d0 + d1);  // exact under the guard (quirk Q8)
                double W1 = Wc[k] + incl;              // window sum of the lane's second symbol
                // ... and of its first: the sums before it plus its own difference (W1 - d1 is the same number
                // while d1 is finite; an inf or NaN entering or leaving the window at the second symbol must not
                // reach back to the first)
                double W0 = (Wc[k] + wave_up1(incl, 0.0)) + d0;
                Wc[k] = read_lane(W1, 63);
                if (k == 0) {
                    argtop_first(top[0], W0);
                    argtop_first(top[1], W1);
                } else {
                    argtop_next(top[0], W0, k);
                    argtop_next(top[1], W1, k);
                }
            }
            {
                // (the largest FINITE sum scales the rounding bound)
                const double b0 = (valid[0] && __builtin_fabs(top[0].best) < (double)__builtin_inff()) ? top[0].best : 0.0;
                const double b1 = (valid[1] && __builtin_fabs(top[1].best) < (double)__builtin_inff()) ? top[1].best : 0.0;
                const float mx = (float)__builtin_fmax(b0, b1);
                cy.wmax = __builtin_fmaxf(cy.wmax, wave_max_f32(__builtin_fmaxf(mx, 0.0f)) * 1.0000002f);
                const int sym0 = (p.lf_flags & PLAN_CARRY_DRIFT) ? (int)p.count0 : 0;
                const float bound_abs = 2.0f * drift_bound(sym0 + c * kB + kB, A) * cy.wmax;
#pragma unroll
                for (int r = 0; r < kR; r++) {
                    bestK[r] = top[r].k;
                    cy.ambiguous = cy.ambiguous || (valid[r] && argtop_ambiguous(top[r], bound_abs));
                }
            }
            cy.stat_exact_blocks += 1;
        }

        cy.last_k = (uint32_t)__builtin_amdgcn_readlane(r_last ? bestK[1] : bestK[0], lane_last);

        // the sample to output: kept at a predicted index -- verify, else re-read (rare, exact either way)
        cf32 s[kR];
#pragma unroll
        for (int r = 0; r < kR; r++) {
            const bool miss = valid[r] && (pkk[r] != bestK[r]);
            if (vote_any(miss)) {  // (wave-uniform, like every branch around loads here: see load_block)
                const uint64_t j = miss ? (uint64_t)(i0 + r) * S + (uint64_t)bestK[r] : 0ull;  // (sample 0 exists)
                const f2g *q = j < X.L0 ? X.ring + j : X.in + (j - X.L0);
                const f2g g = *mem_ptr<packet_global(S)>(q);
                px[r] = miss ? g.x : px[r];
                py[r] = miss ? g.y : py[r];
            }
            s[r].re = px[r];
            s[r].im = py[r];
            kpred[r] = valid[r] ? bestK[r] : kpred[r];
        }

        // sampleIndex_dataShort_out (reference cpp/psk_soft.cpp:466) --
        // after the re-read above, whose wait would otherwise also wait for this store
        if (p.sidx) {
            if (valid[1]) {
                s2u v = {(short)(unsigned short)bestK[0], (short)(unsigned short)bestK[1]};
                store_s2u(g_sidx + i0, v);
            } else if (valid[0]) {
                g_sidx[i0] = (int16_t)(unsigned short)bestK[0];
            }
        }

        // history for the next block
        if constexpr (!REREAD) {
#pragma unroll
            for (int h = H - 1; h > 0; h--) hist[h] = hist[h - 1];
            hist[0] = cur;  // (numAvg <= 128: only the kept samples are used from it)
        }

        // (a wave whose sums need the lane-after-lane chain -- a stationary carrier sitting at zero phase -- pays
        // ~2 us of pure latency per block there.  The launch waits for its slowest wave, so such a wave keeps a
        // raised priority through the arithmetic half too: its other work then runs ahead of its neighbours'
        // and the wave keeps pace with them.  There are a handful of them in thousands.)
        if constexpr (PSK_PACE_ON(FRONT, EXACT)) {
        } else if (cy.chain_streak >> 16)
            __builtin_amdgcn_s_setprio(PSK_CHAIN_PRIO);
        else
            __builtin_amdgcn_s_setprio(0);
        // ================= raw phase: arg(pow(sample, M)) (reference cpp/psk_soft.cpp:474) =================
        float raw[kR];  // arg(pow(sample, M)): a float (std::arg of complex<float>), widened where the reference widens it
#pragma unroll
        for (int r = 0; r < kR; r++) {
            // (the screened tier hands samples whose power is not finite to the exact tier, which carries
            // libgcc's __mulsc3 recovery: cmul<true>)
            cf32 pw = cpow_uint<EXACT>(s[r], M);
            if (!EXACT && valid[r] && !(is_fin(pw.re) && is_fin(pw.im)))
                cy.refuse = true;
            raw[r] = atan2f_wave(pw.im, pw.re, atab);
        }

        if constexpr (FRONT) {
            // (the scratch rows are padded to whole blocks: positions past the end of the call are written too)
            *reinterpret_cast<float2 *>(t_raw + i0) = make_float2(raw[0], raw[1]);
            *reinterpret_cast<float4 *>(t_s + i0) = make_float4(s[0].re, s[0].im, s[1].re, s[1].im);
            cy.stat_blocks += 1;
            if (__any(cy.refuse)) {  // the whole call goes to the wave-scan kernels
                cy.refuse = true;
                return;
            }
        } else {
        // ================= feedback unwrap + LinearFit::next, 128 symbols at a time =================
        float est[kR];
        fit_stage<EXACT>(c, lane, n, xd, den_s, xavg_s, fk, valid, raw, nvalid, lane_last, r_last, yring, ymask, cy, est);
        if constexpr (!EXACT) {
            // a call this tier cannot finish is the exact tier's from its first symbol on: stop here
            if (__any(cy.refuse)) {
                cy.refuse = true;
                return;
            }
        }

        // ================= de-rotation and hard decisions (reference cpp/psk_soft.cpp:484-566) =================
        {
            cf32 last_c;
            last_c.re = cy.last_re;
            last_c.im = cy.last_im;
            output_stage<packet_global(S), EXACT>(p, c, i0, valid, s, est, last_c, atab, qpsk_sign_map, m_pow2, inv_M);
        }
        if (p.diff) {  // `last` only moves while differentialDecoding is on (cpp/psk_soft.cpp:486-491)
            const float sre = r_last ? s[1].re : s[0].re;
            const float sim = r_last ? s[1].im : s[0].im;
            cy.last_re = read_lane(sre, lane_last);
            cy.last_im = read_lane(sim, lane_last);
        }
        }  // (!FRONT)

        if constexpr (!EXACT) {
            // every kScreenRefresh blocks recompute the float window sums from the energies in
            // registers, so that their error bound does not grow with the length of the call
            if (since_refresh >= kScreenRefresh) {
                float wm = 0.0f;
                float wre[S];
                if constexpr (REREAD)
                    window_end_reread_f32<S>(X, c, c_begin, A, tau_last, lane, wre);
#pragma unroll
                for (int k = 0; k < S; k++) {
                    if constexpr (REREAD)
                        Wf[k] = wre[k];
                    else if constexpr (H == 1)
                        Wf[k] = window_end_ring_f32(er, ring_base, A, lane, k);
                    else
                        Wf[k] = window_end_f32<S, H>(hist, A, lane, k);
                    wm = __builtin_fmaxf(wm, Wf[k]);
                }
                wmax_prev = __builtin_fmaxf(wmax_prev, wm);
                // (a 7-level float tree sum of non-negative terms; REREAD: two more additions per block of window in front of it)
                err_c = (REREAD ? (float)(8u + 2u * ((A + 127u) / (uint32_t)kB)) : 16.0f) * kU * wmax_prev;
                since_refresh = 0;
            }
        }
        if constexpr (H == 1)
            ring_base = er.wrap(ring_base + kB);
    }
    if constexpr (!EXACT)
        cy.wmax = wmax_prev;  // (the largest window sum of the call, within the float shadow's bound: TileInfo::wmax, ChanState::pad_state)
}

#undef g_soft
#undef g_phase
#undef g_bits
#undef g_sidx
}  // namespace psk
#endif
