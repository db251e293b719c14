// psk_fast_loop.h -- the symbol loop of the wave-scan kernel (included by psk_kernels.hip).
//
// One wave walks one channel in blocks of B = 128 output symbols; lane l owns the two
// consecutive symbols at block positions s = 2l and 2l+1 ("R = 2": every cross-lane scan,
// carry and fixed-point check is shared by two symbols, and the four output streams are
// written as 16/8/4/8-byte vectors).
//
// Input traffic is ONE pass: a symbol is loaded exactly once, when it becomes the newest
// symbol of a window (A-1 symbols before it is output).  What later steps need from it is kept
// in registers for H = ceil(A/128) blocks and fetched across lanes with ds_bpermute:
//   * its S energies  -- subtracted from the window sums A symbols later
//                        (symbolEnergy[k] -= energy[k], cpp/psk_soft.cpp:572-577);
//   * ONE of its samples, the one at the timing index the lane predicts (the index it has
//     just chosen for its own output symbol).  When the symbol is output and the true
//     argmax equals the prediction -- timing is stationary, so practically always -- the
//     sample is already there; otherwise the lane re-reads it from memory (exact either way).
#ifndef PSK_FAST_LOOP_H
#define PSK_FAST_LOOP_H

namespace psk {

#ifndef PSK_PREFETCH_NEXT_BLOCK
#define PSK_PREFETCH_NEXT_BLOCK 0  // 1: issue block c+1's loads before the phase part of block c (+32 VGPRs)
#endif
constexpr int kR = 2;             // symbols per lane per block
constexpr int kB = kWave * kR;    // symbols per block

struct FastCarry {
    double ySum, xySum;    // LinearFit sums after the last processed symbol
    float est;             // phaseEstimate
    float last_re, last_im;
    float den, xavg;       // LinearFit::denominator / xAvg
    float m, b;
    uint32_t q;            // number of values ever written to the LDS y ring (history included)
    uint32_t last_k;       // timing index of the last emitted symbol (prediction seed)
    unsigned umax, umin1;  // exactness guard: max energy bits, min (energy bits - 1)
    bool refuse;
    uint32_t stat_blocks, stat_extra;
};

// what a block keeps of the symbols it loaded (positions s = 2*lane + r)
template <int S>
struct BlockKeep {
    float e[kR][S];   // energies
    float2 pk[kR];    // the sample at the predicted timing index
    int kp[kR];       // the predicted timing index
};

PSK_DEV float bperm(int src_lane, float v)
{
    return __int_as_float(__builtin_amdgcn_ds_bpermute(src_lane << 2, __float_as_int(v)));
}
PSK_DEV int bperm(int src_lane, int v) { return __builtin_amdgcn_ds_bpermute(src_lane << 2, v); }

// Value at stream position (this block's position s) - v, 1 <= v < kB, where positions >= 0 lie
// in `newer` and negative ones in `older` (the block before it).  r-th result for this lane.
template <class T>
PSK_DEV T rot_fetch(int lane, int r, int v, const T (&newer)[kR], const T (&older)[kR])
{
    const bool odd = (v & 1) != 0;
    // what this lane OFFERS to the lane that will read slot (r ^ odd) of it; the array indices
    // stay compile-time constants (a runtime index would push the arrays to scratch)
    const int r_src = r ^ (int)odd;
    const T nsel = odd ? newer[r ^ 1] : newer[r];
    const T osel = odd ? older[r ^ 1] : older[r];
    const int p_self = 2 * lane + r_src;
    const T offered = (p_self >= kB - v) ? osel : nsel;
    const int src_lane = (((2 * lane + r - v) & (kB - 1)) >> 1);
    return bperm(src_lane, offered);
}

// x[k] for a per-lane k without a runtime-indexed array (which would live in scratch): a
// binary tree of selects over compile-time indices
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
    constexpr int P = S <= 2 ? 2 : S <= 4 ? 4 : S <= 8 ? 8 : 16;
    return make_float2(sel_tree<S, 0, P>(x, k, false), sel_tree<S, 0, P>(x, k, true));
}

// loads the two symbols at positions 2*lane, 2*lane+1 of "new-symbol block" cblk:
// tau = kB*cblk + s + A - 1; symbols outside [tau_lo, tau_hi] are zero-filled
template <int S>
PSK_DEV void load_block(const XView &X, long long cblk, uint32_t A, long long tau_lo, long long tau_hi, int lane,
                        float2 (&x)[kR][S], bool (&ok)[kR])
{
#pragma unroll
    for (int r = 0; r < kR; r++) {
        const long long tau = cblk * kB + 2 * lane + r + (long long)A - 1;
        ok[r] = tau >= tau_lo && tau <= tau_hi;
        load_symbol<S>(X, (uint64_t)(ok[r] ? tau : 0), ok[r], x[r]);
    }
}

template <int S, int H>
PSK_DEV void fast_main_loop(const ChanPlan &p, const XView &X, float *yring, FastCarry &cy)
{
    const int lane = threadIdx.x & 63;
    const uint32_t A = p.A, M = p.M, n = p.lf_n;
    const long long n_out = (long long)p.n_out;
    const float xd = p.lf_xdelta;
    const long long tau_last = n_out + (long long)A - 2;  // newest symbol any emitted window uses

    // ---- prologue: the first A-1 symbols (the carried window) as "blocks" -H .. -1:
    //      W_k(-1) = their energy sums (= resyncEnergy, cpp/psk_soft.cpp:619-636) ----
    BlockKeep<S> hist[H];
    double Wc[S];
    {
        double acc[S];
#pragma unroll
        for (int k = 0; k < S; k++) acc[k] = 0.0;
#pragma unroll
        for (int h = H - 1; h >= 0; h--) {
            float2 x[kR][S];
            bool ok[kR];
            load_block<S>(X, -(long long)(h + 1), A, 0, (long long)A - 2, lane, x, ok);
#pragma unroll
            for (int r = 0; r < kR; r++) {
#pragma unroll
                for (int k = 0; k < S; k++) {
                    float e = norm_f(x[r][k].x, x[r][k].y);
                    if (ok[r]) {
                        unsigned eb = __float_as_uint(e);
                        cy.umax = eb > cy.umax ? eb : cy.umax;
                        cy.umin1 = (eb - 1u) < cy.umin1 ? (eb - 1u) : cy.umin1;
                    }
                    hist[h].e[r][k] = e;
                    acc[k] += (double)e;
                }
                hist[h].kp[r] = (int)cy.last_k;
                hist[h].pk[r] = select_sample<S>(x[r], (int)cy.last_k);
            }
        }
#pragma unroll
        for (int k = 0; k < S; k++) Wc[k] = wave_sum_f64(acc[k]);
    }

    // rotation distances: energies leave the window A symbols after they entered it; the
    // picked-from symbol entered it A-1 symbols ago
    const int uE = (int)(A / kB), vE = (int)(A % kB);
    const int uP = (int)((A - 1) / kB), vP = (int)((A - 1) % kB);

    // steady-state fit constants
    float den_s = cy.den, xavg_s = cy.xavg;
    if (n > 1)
        fit_denominator(xd, n, den_s, xavg_s);
    const double rden_s = 1.0 / (double)den_s, rpts_s = 1.0 / (double)n;

    const long long n_blocks = (n_out + kB - 1) / kB;
    float2 xn[kR][S];
    bool okn[kR];
#if PSK_PREFETCH_NEXT_BLOCK
    load_block<S>(X, 0, A, 0, tau_last, lane, xn, okn);
#endif

    for (long long c = 0; c < n_blocks; c++) {
#if !PSK_PREFETCH_NEXT_BLOCK
        load_block<S>(X, c, A, 0, tau_last, lane, xn, okn);
#endif
        const long long i0 = c * kB + 2 * lane;  // first output symbol of this lane
        bool valid[kR];
        valid[0] = i0 < n_out;
        valid[1] = i0 + 1 < n_out;
        const long long rem = n_out - c * kB;
        const int nvalid = rem < (long long)kB ? (int)rem : kB;  // valid positions of this block
        const int lane_last = (nvalid - 1) >> 1, r_last = (nvalid - 1) & 1;

        // ================= timing recovery =================
        BlockKeep<S> cur;
#pragma unroll
        for (int r = 0; r < kR; r++)
#pragma unroll
            for (int k = 0; k < S; k++) {
                float e = norm_f(xn[r][k].x, xn[r][k].y);
                if (okn[r]) {
                    unsigned eb = __float_as_uint(e);
                    cy.umax = eb > cy.umax ? eb : cy.umax;
                    cy.umin1 = (eb - 1u) < cy.umin1 ? (eb - 1u) : cy.umin1;
                }
                cur.e[r][k] = e;
            }

        double bestW[kR] = {0.0, 0.0};
        int bestK[kR] = {0, 0};
#pragma unroll
        for (int k = 0; k < S; k++) {
            // energy of symbol i-1 (it entered the window A symbols before symbol i+A-1 did)
            float e_old[kR];
#pragma unroll
            for (int r = 0; r < kR; r++) {
                float nw[kR], od[kR];
                // blocks (c - uE) and (c - uE - 1); block 0 back = cur, j back = hist[j-1]
#pragma unroll
                for (int rr = 0; rr < kR; rr++) {
                    nw[rr] = cur.e[rr][k];
                    od[rr] = hist[0].e[rr][k];
                }
#pragma unroll
                for (int j = 1; j <= H; j++)
                    if (uE == j) {
#pragma unroll
                        for (int rr = 0; rr < kR; rr++) {
                            nw[rr] = hist[j - 1].e[rr][k];
                            od[rr] = hist[j < H ? j : H - 1].e[rr][k];
                        }
                    }
                e_old[r] = (vE == 0) ? nw[r] : rot_fetch<float>(lane, r, vE, nw, od);
            }
            double d0 = valid[0] ? (double)cur.e[0][k] - (double)e_old[0] : 0.0;
            double d1 = valid[1] ? (double)cur.e[1][k] - (double)e_old[1] : 0.0;
            double t1 = d0 + d1;                        // exact: float-valued addends (Q8)
            double incl = wave_scan_f64(t1);
            double W1 = Wc[k] + incl;                   // window sum of the lane's second symbol
            double W0 = W1 - d1;                        // ... and of its first (exact)
            Wc[k] = read_lane(W1, 63);
            // std::max_element: first maximum, strict '<' (cpp/psk_soft.cpp:462)
            if (k == 0) {
                bestW[0] = W0;
                bestW[1] = W1;
            } else {
                if (bestW[0] < W0) {
                    bestW[0] = W0;
                    bestK[0] = k;
                }
                if (bestW[1] < W1) {
                    bestW[1] = W1;
                    bestK[1] = k;
                }
            }
        }

        // what this block keeps of its new symbols: the sample at the index just chosen
#pragma unroll
        for (int r = 0; r < kR; r++) {
            cur.kp[r] = bestK[r];
            cur.pk[r] = select_sample<S>(xn[r], bestK[r]);
        }

        // the sample to output: kept (A-1 symbols ago) at a predicted index -- verify, else re-read
        cf32 s[kR];
        {
            bool miss_any = false;
#pragma unroll
            for (int r = 0; r < kR; r++) {
                float nx[kR], ox[kR], ny[kR], oy[kR];
                int nk[kR], ok2[kR];
#pragma unroll
                for (int rr = 0; rr < kR; rr++) {
                    nx[rr] = cur.pk[rr].x; ny[rr] = cur.pk[rr].y; nk[rr] = cur.kp[rr];
                    ox[rr] = hist[0].pk[rr].x; oy[rr] = hist[0].pk[rr].y; ok2[rr] = hist[0].kp[rr];
                }
#pragma unroll
                for (int j = 1; j <= H; j++)
                    if (uP == j) {
#pragma unroll
                        for (int rr = 0; rr < kR; rr++) {
                            nx[rr] = hist[j - 1].pk[rr].x; ny[rr] = hist[j - 1].pk[rr].y; nk[rr] = hist[j - 1].kp[rr];
                            const int jo = j < H ? j : H - 1;
                            ox[rr] = hist[jo].pk[rr].x; oy[rr] = hist[jo].pk[rr].y; ok2[rr] = hist[jo].kp[rr];
                        }
                    }
                float px, py;
                int pkk;
                if (vP == 0) {
                    px = nx[r]; py = ny[r]; pkk = nk[r];
                } else {
                    px = rot_fetch<float>(lane, r, vP, nx, ox);
                    py = rot_fetch<float>(lane, r, vP, ny, oy);
                    pkk = rot_fetch<int>(lane, r, vP, nk, ok2);
                }
                const bool miss = valid[r] && (pkk != bestK[r]);
                if (miss) {  // timing index moved since the prediction: re-read (rare)
                    float2 g = x_at(X, (uint64_t)(i0 + r) * S + (uint64_t)bestK[r]);
                    px = g.x;
                    py = g.y;
                }
                miss_any |= miss;
                s[r].re = px;
                s[r].im = py;
            }
            (void)miss_any;
        }

        // history for the next block; prefetch its new symbols (their latency hides under the
        // phase part below)
#pragma unroll
        for (int h = H - 1; h > 0; h--) hist[h] = hist[h - 1];
        hist[0] = cur;
#if PSK_PREFETCH_NEXT_BLOCK
        load_block<S>(X, c + 1, A, 0, tau_last, lane, xn, okn);
#endif

        // ================= raw phase: arg(pow(sample, M)) (cpp/psk_soft.cpp:474) =================
        double rawd[kR];
#pragma unroll
        for (int r = 0; r < kR; r++) {
            cf32 pw = cpow_uint<false>(s[r], M);
            if (valid[r] && !(is_fin(pw.re) && is_fin(pw.im)))
                cy.refuse = true;  // overflow / NaN: the reference-order kernel owns __mulsc3 semantics
            rawd[r] = (double)lm_atan2f(pw.im, pw.re);
        }

        // ================= feedback unwrap + LinearFit::next, 128 symbols at a time =================
        const uint32_t q0 = cy.q;
        uint32_t before[kR], size_b[kR], pts[kR];
        bool steady[kR];
        float den_l[kR], xavg_l[kR];
#pragma unroll
        for (int r = 0; r < kR; r++) {
            before[r] = q0 + (uint32_t)(2 * lane + r);   // values pushed before this next()
            steady[r] = before[r] >= n;                  // cpp/psk_soft.cpp:54
            size_b[r] = steady[r] ? n - 1 : before[r];   // yvals.size() at :78
            pts[r] = steady[r] ? n : before[r] + 1;      // yvals.size() at calculateFit
            den_l[r] = den_s;
            xavg_l[r] = xavg_s;
            if (q0 < n && pts[r] > 1 && pts[r] < n)      // warm-up: the window is still growing
                fit_denominator(xd, pts[r], den_l[r], xavg_l[r]);
        }

        // speculate numWraps by consecutive differences; position 0 is exact (carried estimate)
        int w[kR];
        {
            double raw_prev0 = wave_up1(rawd[1], rawd[1]);
            int dl0 = (lane == 0) ? (int)unwrap_count(cy.est, rawd[0])
                                  : (int)to_long_x86(__builtin_round((raw_prev0 - rawd[0]) * kInvTwoPi));
            int dl1 = (int)to_long_x86(__builtin_round((rawd[0] - rawd[1]) * kInvTwoPi));
            dl0 = valid[0] ? dl0 : 0;
            dl1 = valid[1] ? dl1 : 0;
            int incl = wave_scan_i32(dl0 + dl1);
            w[1] = incl;
            w[0] = incl - dl1;
        }
        float y[kR] = {0.0f, 0.0f}, est[kR] = {0.0f, 0.0f}, m_l[kR] = {0.0f, 0.0f}, b_l[kR] = {0.0f, 0.0f};
        double ySum_l[kR] = {0.0, 0.0}, xySum_l[kR] = {0.0, 0.0};
        int pass = 0;
        for (;;) {
            double y_d[kR];
#pragma unroll
            for (int r = 0; r < kR; r++) {
                double yd = rawd[r] + (double)(long long)w[r] * kTwoPi;  // cpp/psk_soft.cpp:478
                y[r] = (float)yd;                                         // next(float yval), :481
                if (valid[r])
                    yring[(q0 + 2 * lane + r) & kYMask] = y[r];
                y_d[r] = valid[r] ? (double)y[r] : 0.0;
            }
            wave_lds_fence();
            float z[kR];
#pragma unroll
            for (int r = 0; r < kR; r++)
                z[r] = (valid[r] && steady[r]) ? yring[(before[r] - n) & kYMask] : 0.0f;  // yvals.front(), :70
            wave_lds_fence();
            const double dy0 = y_d[0] - (double)z[0], dy1 = y_d[1] - (double)z[1];
            {
                double incl = wave_scan_f64(dy0 + dy1);
                double base = cy.ySum + wave_up1(incl, 0.0);  // ySum after the previous lane's symbols
                double ySumP0 = base - (double)z[0];           // ySum after the pop, :70
                ySum_l[0] = base + dy0;
                double ySumP1 = ySum_l[0] - (double)z[1];
                ySum_l[1] = ySum_l[0] + dy1;
                float t0 = y[0] * (float)size_b[0];            // :78, size before the push
                t0 = t0 * xd;
                float t1 = y[1] * (float)size_b[1];
                t1 = t1 * xd;
                double c0 = (double)t0 - (steady[0] ? (double)xd * ySumP0 : 0.0);  // :72 and :78
                double c1 = (double)t1 - (steady[1] ? (double)xd * ySumP1 : 0.0);
                c0 = valid[0] ? c0 : 0.0;
                c1 = valid[1] ? c1 : 0.0;
                double incl2 = wave_scan_f64(c0 + c1);
                double base2 = cy.xySum + wave_up1(incl2, 0.0);
                xySum_l[0] = base2 + c0;
                xySum_l[1] = xySum_l[0] + c1;
            }
#pragma unroll
            for (int r = 0; r < kR; r++) {
                if (q0 >= n) {  // steady state: both divisors are wave-uniform
                    est[r] = fit_value_known(ySum_l[r], xySum_l[r], xd, n, den_s, xavg_s, rden_s, rpts_s, m_l[r], b_l[r]);
                } else if (pts[r] > 1) {
                    est[r] = fit_value(ySum_l[r], xySum_l[r], xd, pts[r], den_l[r], xavg_l[r], m_l[r], b_l[r]);
                } else {  // :164-171, a single point: b = yvals.back()
                    m_l[r] = 0.0f;
                    b_l[r] = y[r];
                    est[r] = y[r];
                }
            }
            float est_prev0 = wave_up1(est[1], cy.est);
            int w2_0 = (int)unwrap_count(est_prev0, rawd[0]);  // cpp/psk_soft.cpp:477 with the true feedback
            int w2_1 = (int)unwrap_count(est[0], rawd[1]);
            bool bad = (valid[0] && w2_0 != w[0]) || (valid[1] && w2_1 != w[1]);
            if (!__any(bad))
                break;
            w[0] = w2_0;
            w[1] = w2_1;
            if (++pass > 2 * kMaxUnwrapPasses) {
                cy.refuse = true;
                break;
            }
        }
        cy.stat_blocks += 1;
        cy.stat_extra += (uint32_t)pass;

        // ================= de-rotation and hard decisions (cpp/psk_soft.cpp:484-566) =================
        cf32 corr[kR];
#pragma unroll
        for (int r = 0; r < kR; r++) {
            float phaseCorrection = 0.0f;
            cf32 smp = s[r];
            if (p.diff) {
                cf32 last;
                if (r == 0) {
                    last.re = wave_up1(s[1].re, cy.last_re);
                    last.im = wave_up1(s[1].im, cy.last_im);
                } else {
                    last = s[0];
                }
                smp = cdiv(s[r], last);
            } else {
                phaseCorrection = -est[r] / (float)M;
            }
            if (M == 4)
                phaseCorrection = (float)((double)phaseCorrection + kPi4);
            float sn, cs;
            lm_sincosf(phaseCorrection, &sn, &cs);
            cf32 ph;
            ph.re = 1.0f * cs;
            ph.im = 1.0f * sn;
            corr[r] = cmul<true>(smp, ph);
        }

        // ---- four output streams, two symbols per lane ----
        if (valid[1]) {
            typedef float f4u __attribute__((ext_vector_type(4), aligned(8)));
            typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
            typedef short s2u __attribute__((ext_vector_type(2), aligned(4)));
            typedef short s4u __attribute__((ext_vector_type(4), aligned(4)));
            if (p.soft) {
                f4u v = {corr[0].re, corr[0].im, corr[1].re, corr[1].im};
                *reinterpret_cast<f4u *>(p.soft + 2 * i0) = v;
            }
            if (p.phase) {
                f2u v = {est[0], est[1]};
                *reinterpret_cast<f2u *>(p.phase + i0) = v;
            }
            if (p.sidx) {
                s2u v = {(short)(unsigned short)bestK[0], (short)(unsigned short)bestK[1]};
                *reinterpret_cast<s2u *>(p.sidx + i0) = v;
            }
            if (!p.bits) {
            } else if (p.bpb == 1) {
                s2u v = {(short)(corr[0].re < 0), (short)(corr[1].re < 0)};
                *reinterpret_cast<s2u *>(p.bits + i0) = v;
            } else if (p.bpb == 2) {  // quirk Q1: float -> bool is "!= 0"
                int r0 = (corr[0].re != 0), m0 = (corr[0].im != 0), r1 = (corr[1].re != 0), m1 = (corr[1].im != 0);
                s4u v = {(short)(r0 ^ m0), (short)(!m0), (short)(r1 ^ m1), (short)(!m1)};
                *reinterpret_cast<s4u *>(p.bits + 2 * i0) = v;
            } else if (p.bpb == 3) {
                unsigned short a = slice_8psk(corr[0].re, corr[0].im), b = slice_8psk(corr[1].re, corr[1].im);
                s2u v0 = {(short)(a & 1), (short)((a >> 1) & 1)};
                s2u v1 = {(short)((a >> 2) & 1), (short)(b & 1)};
                s2u v2 = {(short)((b >> 1) & 1), (short)((b >> 2) & 1)};
                s2u *q = reinterpret_cast<s2u *>(p.bits + 3 * i0);
                q[0] = v0;
                q[1] = v1;
                q[2] = v2;
            }
        } else if (valid[0]) {  // an odd tail: one symbol
            if (p.soft)
                reinterpret_cast<float2 *>(p.soft)[i0] = make_float2(corr[0].re, corr[0].im);
            if (p.phase)
                p.phase[i0] = est[0];
            if (p.sidx)
                p.sidx[i0] = (int16_t)(unsigned short)bestK[0];
            if (!p.bits) {
            } else if (p.bpb == 1) {
                p.bits[i0] = (int16_t)(corr[0].re < 0);
            } else if (p.bpb == 2) {
                int r0 = (corr[0].re != 0), m0 = (corr[0].im != 0);
                p.bits[2 * i0] = (int16_t)(r0 ^ m0);
                p.bits[2 * i0 + 1] = (int16_t)(!m0);
            } else if (p.bpb == 3) {
                unsigned short a = slice_8psk(corr[0].re, corr[0].im);
                p.bits[3 * i0] = (int16_t)(a & 1);
                p.bits[3 * i0 + 1] = (int16_t)((a >> 1) & 1);
                p.bits[3 * i0 + 2] = (int16_t)((a >> 2) & 1);
            }
        }

        // ---- carries into the next block: the last valid position of this one ----
        {
            const double ys = r_last ? ySum_l[1] : ySum_l[0];
            const double xys = r_last ? xySum_l[1] : xySum_l[0];
            const float e_ = r_last ? est[1] : est[0];
            const float mm = r_last ? m_l[1] : m_l[0];
            const float bb = r_last ? b_l[1] : b_l[0];
            const float sre = r_last ? s[1].re : s[0].re;
            const float sim = r_last ? s[1].im : s[0].im;
            const int kk = r_last ? bestK[1] : bestK[0];
            const float dl = r_last ? den_l[1] : den_l[0];
            const float xl = r_last ? xavg_l[1] : xavg_l[0];
            cy.ySum = read_lane(ys, lane_last);
            cy.xySum = read_lane(xys, lane_last);
            cy.est = read_lane(e_, lane_last);
            cy.m = read_lane(mm, lane_last);
            cy.b = read_lane(bb, lane_last);
            cy.last_re = read_lane(sre, lane_last);
            cy.last_im = read_lane(sim, lane_last);
            cy.last_k = (uint32_t)__builtin_amdgcn_readlane(kk, lane_last);
            const uint32_t pts_last = (q0 + (uint32_t)nvalid - 1 >= n) ? n : q0 + (uint32_t)nvalid;
            if (pts_last > 1 && pts_last < n) {
                cy.den = read_lane(dl, lane_last);
                cy.xavg = read_lane(xl, lane_last);
            } else if (pts_last > 1) {
                cy.den = den_s;
                cy.xavg = xavg_s;
            }
        }
        cy.q = q0 + (uint32_t)nvalid;
    }
}

}  // namespace psk
#endif
