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

#ifndef PSK_PREFETCH
#define PSK_PREFETCH 0  // 1: issue block c+1's loads before the phase part of block c
#endif
#ifndef PSK_WAVES_H1
#define PSK_WAVES_H1 4  // waves per SIMD the H = 1 instantiations are register-limited to
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

// exactness guard bookkeeping: max of the energy bit patterns and min of (bits - 1); a zero
// energy gives bits 0 / 0xFFFFFFFF and so constrains neither
PSK_DEV void guard_track(FastCarry &cy, float e)
{
    const unsigned eb = __float_as_uint(e);
    cy.umax = eb > cy.umax ? eb : cy.umax;
    const unsigned em = eb - 1u;
    cy.umin1 = em < cy.umin1 ? em : cy.umin1;
}

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
PSK_DEV float bperm_addr(int addr, float v) { return __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(v))); }
PSK_DEV int bperm_addr(int addr, int v) { return __builtin_amdgcn_ds_bpermute(addr, v); }

// Cross-lane fetch of "the value that sat D symbol positions earlier in the stream of loaded
// symbols" (D = numAvg for the energies leaving the window, numAvg-1 for the picked-from symbol).
// D = u*kB + v with 1 <= v <= kB (D = 0: u = v = 0): positions s >= v of this block find it in
// block c-u ("newer"), the others in block c-u-1 ("older").  The per-lane parameters depend only
// on numAvg, so they are computed once per call; the fetch itself is branch-free.
struct RotParam {
    int u;            // how many blocks back "newer" is (wave-uniform)
    bool odd;         // v odd: the two slots of a lane swap roles
    int src_addr[kR]; // byte address (lane*4) this lane pulls slot r from
    bool offer_old[kR];  // what this lane offers for the puller of its slot r: older or newer block
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
        const int r_src = r ^ (int)p.odd;  // the slot of the source lane that is read for result r
        (void)r_src;
        // this lane's slot q is pulled by result r = q ^ odd of some lane; that puller needs the
        // older block iff its own position < v, i.e. iff this slot's position >= kB - v
        p.offer_old[r] = (2 * lane + r) >= kB - v;
    }
    return p;
}
template <class T>
PSK_DEV T rot_pull(const RotParam &p, int r, const T (&newer)[kR], const T (&older)[kR])
{
    // slot of the source lane read for result r (compile-time indices only: no scratch)
    const T off0 = p.offer_old[0] ? older[0] : newer[0];
    const T off1 = p.offer_old[1] ? older[1] : newer[1];
    const T offered = (r == 0) ? (p.odd ? off1 : off0) : (p.odd ? off0 : off1);
    return bperm_addr(p.src_addr[r], offered);
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

// One block (128 symbols) of the feedback unwrap + LinearFit::next recurrence
// (cpp/psk_soft.cpp:477-482, 48-87, 135-174).  WARM = the fit window is still growing somewhere in
// the block (first phaseAvg symbols after a history clear): per-lane window sizes and denominators.
// Returns the number of extra fixed-point passes; den_last / xavg_last = LinearFit::denominator /
// xAvg after the block's last valid symbol.
template <bool WARM>
PSK_DEV int fit_block(int lane, uint32_t q0, uint32_t n, float xd, float den_s, float xavg_s, double rden_s,
                      double rpts_s, const bool (&valid)[kR], const double (&rawd)[kR], const FastCarry &cy,
                      float *yring, float (&y)[kR], float (&est)[kR], double (&ySum_l)[kR], double (&xySum_l)[kR],
                      int lane_last, int r_last, float &den_last, float &xavg_last)
{
    uint32_t before[kR];
    bool steady[kR];
    float sizef[kR];   // (float)yvals.size() at cpp/psk_soft.cpp:78
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
            sizef[r] = (float)(n - 1);
            pts[r] = n;
            den_l[r] = den_s;
            xavg_l[r] = xavg_s;
        }
    }
    // speculate numWraps by consecutive differences; position 0 is exact (carried estimate)
    int w[kR];
    {
        double raw_prev0 = wave_up1(rawd[1], rawd[1]);
        int dl0 = (lane == 0) ? (int)unwrap_count(cy.est, rawd[0])
                              : (int)to_long_x86(__builtin_round((raw_prev0 - rawd[0]) * kInvTwoPi));
        int dl1 = (int)to_long_x86(__builtin_round((rawd[0] - rawd[1]) * kInvTwoPi));
        int incl = wave_scan_i32(dl0 + dl1);
        w[1] = incl;
        w[0] = incl - dl1;
    }
    int pass = 0;
    for (;;) {
        double y_d[kR];
#pragma unroll
        for (int r = 0; r < kR; r++) {
            double yd = rawd[r] + (double)(long long)w[r] * kTwoPi;  // cpp/psk_soft.cpp:478
            y[r] = (float)yd;                                         // next(float yval), :481
            if (valid[r])
                yring[(before[r]) & kYMask] = y[r];
            y_d[r] = (double)y[r];  // (positions past the end only feed sums past the end)
        }
        wave_lds_fence();
        float z[kR];
#pragma unroll
        for (int r = 0; r < kR; r++)
            z[r] = steady[r] ? yring[(before[r] - n) & kYMask] : 0.0f;  // yvals.front(), :70
        wave_lds_fence();
        const double dy0 = y_d[0] - (double)z[0], dy1 = y_d[1] - (double)z[1];
        double incl = wave_scan_f64(dy0 + dy1);
        double base = cy.ySum + wave_up1(incl, 0.0);  // ySum after the previous lane's symbols
        double ySumP0 = base - (double)z[0];           // ySum after the pop, :70
        ySum_l[0] = base + dy0;
        double ySumP1 = ySum_l[0] - (double)z[1];
        ySum_l[1] = ySum_l[0] + dy1;
        float t0 = y[0] * sizef[0];                    // :78, size before the push
        t0 = t0 * xd;
        float t1 = y[1] * sizef[1];
        t1 = t1 * xd;
        double c0 = (double)t0 - (steady[0] ? (double)xd * ySumP0 : 0.0);  // :72 and :78
        double c1 = (double)t1 - (steady[1] ? (double)xd * ySumP1 : 0.0);
        double incl2 = wave_scan_f64(c0 + c1);
        double base2 = cy.xySum + wave_up1(incl2, 0.0);
        xySum_l[0] = base2 + c0;
        xySum_l[1] = xySum_l[0] + c1;
#pragma unroll
        for (int r = 0; r < kR; r++) {
            float m_, b_;
            if (!WARM) {  // steady state: both divisors are wave-uniform
                est[r] = fit_value_known(ySum_l[r], xySum_l[r], xd, n, den_s, xavg_s, rden_s, rpts_s, m_, b_);
            } else if (pts[r] > 1) {
                est[r] = fit_value(ySum_l[r], xySum_l[r], xd, pts[r], den_l[r], xavg_l[r], m_, b_);
            } else {  // :164-171, a single point: b = yvals.back()
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
        if (++pass > 2 * kMaxUnwrapPasses)
            break;
    }
    if (WARM) {
        den_last = read_lane(r_last ? den_l[1] : den_l[0], lane_last);
        xavg_last = read_lane(r_last ? xavg_l[1] : xavg_l[0], lane_last);
    }
    return pass;
}

// block j back in time: 0 = the block being processed, j >= 1 = hist[j-1]; u is wave-uniform
template <int S, int H, class F>
PSK_DEV float pick_block_e(int u, const BlockKeep<S> &cur, const BlockKeep<S> (&hist)[H], int older, F field)
{
    // value of `field` in block (u + older) back
    float v = field(u + older == 0 ? cur : hist[0]);
#pragma unroll
    for (int j = 1; j <= H; j++)
        if (u + older == j)
            v = field(hist[j - 1 < H ? j - 1 : H - 1]);
    return v;
}

template <int S, int H>
PSK_DEV void fast_main_loop(const ChanPlan &p, const XView &X, float *yring, FastCarry &cy)
{
    const int lane = threadIdx.x & 63;
    const uint32_t A = p.A, M = p.M, n = p.lf_n;
    const int n_out = (int)p.n_out;  // <= 2^20 on this path
    const float xd = p.lf_xdelta;
    const long long tau_last = (long long)n_out + (long long)A - 2;  // newest symbol any emitted window uses

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
                    guard_track(cy, e);  // zero-filled (absent) symbols are neutral
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

    // cross-lane fetch parameters: energies leave the window A symbols after they entered it;
    // the picked-from symbol entered it A-1 symbols ago
    const RotParam rotE = rot_param(lane, A);
    const RotParam rotP = rot_param(lane, A - 1);

    // steady-state fit constants
    float den_s = cy.den, xavg_s = cy.xavg;
    if (n > 1)
        fit_denominator(xd, n, den_s, xavg_s);
    const double rden_s = 1.0 / (double)den_s, rpts_s = 1.0 / (double)n;
    // -est/M (cpp/psk_soft.cpp:494): for a power-of-two M the division is an exact scaling
    const bool m_pow2 = M != 0 && (M & (M - 1)) == 0;
    const float inv_M = 1.0f / (float)(M ? M : 1);

    const int n_blocks = (n_out + kB - 1) / kB;
    int kpred[kR] = {(int)cy.last_k, (int)cy.last_k};  // timing index this lane chose one block ago
    float2 xn[kR][S];
    bool okn[kR];
#if PSK_PREFETCH
    load_block<S>(X, 0, A, 0, tau_last, lane, xn, okn);
#endif

    for (int c = 0; c < n_blocks; c++) {
#if !PSK_PREFETCH
        load_block<S>(X, (long long)c, A, 0, tau_last, lane, xn, okn);
#endif
        const int i0 = c * kB + 2 * lane;  // first output symbol of this lane
        bool valid[kR];
        valid[0] = i0 < n_out;
        valid[1] = i0 + 1 < n_out;
        const int rem = n_out - c * kB;
        const int nvalid = rem < kB ? rem : kB;  // valid positions of this block
        const int lane_last = (nvalid - 1) >> 1, r_last = (nvalid - 1) & 1;

        // ================= timing recovery =================
        // energies of the new symbols; of their samples keep the one at the predicted timing index
        BlockKeep<S> cur;
#pragma unroll
        for (int r = 0; r < kR; r++) {
#pragma unroll
            for (int k = 0; k < S; k++) {
                float e = norm_f(xn[r][k].x, xn[r][k].y);
                guard_track(cy, e);
                cur.e[r][k] = e;
            }
            cur.kp[r] = kpred[r];
            cur.pk[r] = select_sample<S>(xn[r], kpred[r]);
        }
        // energy of symbol i-1 for every phase (it entered the window A symbols before symbol
        // i+A-1 did): all cross-lane fetches issued back to back
        float e_old[kR][S];
#pragma unroll
        for (int k = 0; k < S; k++) {
            float nw[kR], od[kR];
#pragma unroll
            for (int rr = 0; rr < kR; rr++) {
                nw[rr] = pick_block_e<S, H>(rotE.u, cur, hist, 0, [&](const BlockKeep<S> &b) { return b.e[rr][k]; });
                od[rr] = pick_block_e<S, H>(rotE.u, cur, hist, 1, [&](const BlockKeep<S> &b) { return b.e[rr][k]; });
            }
#pragma unroll
            for (int r = 0; r < kR; r++) e_old[r][k] = rot_pull<float>(rotE, r, nw, od);
        }
        // the sample kept A-1 symbols ago for the symbol now being output, and the index it was kept at
        float px[kR], py[kR];
        int pkk[kR];
        {
            float nx[kR], ox[kR], ny[kR], oy[kR];
            int nk[kR], ok2[kR];
#pragma unroll
            for (int rr = 0; rr < kR; rr++) {
                nx[rr] = pick_block_e<S, H>(rotP.u, cur, hist, 0, [&](const BlockKeep<S> &b) { return b.pk[rr].x; });
                ox[rr] = pick_block_e<S, H>(rotP.u, cur, hist, 1, [&](const BlockKeep<S> &b) { return b.pk[rr].x; });
                ny[rr] = pick_block_e<S, H>(rotP.u, cur, hist, 0, [&](const BlockKeep<S> &b) { return b.pk[rr].y; });
                oy[rr] = pick_block_e<S, H>(rotP.u, cur, hist, 1, [&](const BlockKeep<S> &b) { return b.pk[rr].y; });
                nk[rr] = __float_as_int(pick_block_e<S, H>(rotP.u, cur, hist, 0, [&](const BlockKeep<S> &b) { return __int_as_float(b.kp[rr]); }));
                ok2[rr] = __float_as_int(pick_block_e<S, H>(rotP.u, cur, hist, 1, [&](const BlockKeep<S> &b) { return __int_as_float(b.kp[rr]); }));
            }
#pragma unroll
            for (int r = 0; r < kR; r++) {
                px[r] = rot_pull<float>(rotP, r, nx, ox);
                py[r] = rot_pull<float>(rotP, r, ny, oy);
                pkk[r] = rot_pull<int>(rotP, r, nk, ok2);
            }
        }

        double bestW[kR] = {0.0, 0.0};
        int bestK[kR] = {0, 0};
#pragma unroll
        for (int k = 0; k < S; k++) {
            // positions past the end of the call only pollute sums of later positions, which are
            // past the end too: no masking needed
            double d0 = (double)cur.e[0][k] - (double)e_old[0][k];
            double d1 = (double)cur.e[1][k] - (double)e_old[1][k];
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

        // the sample to output: kept at a predicted index -- verify, else re-read (rare, exact either way)
        cf32 s[kR];
#pragma unroll
        for (int r = 0; r < kR; r++) {
            const bool miss = valid[r] && (pkk[r] != bestK[r]);
            if (miss) {
                float2 g = x_at(X, (uint64_t)(i0 + r) * S + (uint64_t)bestK[r]);
                px[r] = g.x;
                py[r] = g.y;
            }
            s[r].re = px[r];
            s[r].im = py[r];
            kpred[r] = valid[r] ? bestK[r] : kpred[r];
        }

        // history for the next block
#pragma unroll
        for (int h = H - 1; h > 0; h--) hist[h] = hist[h - 1];
        hist[0] = cur;
#if PSK_PREFETCH
        // prefetch the next block's symbols: their latency hides under the phase part below
        load_block<S>(X, (long long)c + 1, A, 0, tau_last, lane, xn, okn);
#endif

        // ================= raw phase: arg(pow(sample, M)) (cpp/psk_soft.cpp:474) =================
        double rawd[kR];
#pragma unroll
        for (int r = 0; r < kR; r++) {
            cf32 pw = cpow_uint<false>(s[r], M);
            if (valid[r] && !(is_fin(pw.re) && is_fin(pw.im)))
                cy.refuse = true;  // overflow / NaN: the reference-order kernel owns __mulsc3 semantics
            rawd[r] = (double)atan2f_wave(pw.im, pw.re);
        }

        // ================= feedback unwrap + LinearFit::next, 128 symbols at a time =================
        const uint32_t q0 = cy.q;
        float y[kR], est[kR];
        double ySum_l[kR], xySum_l[kR];
        float den_last = den_s, xavg_last = xavg_s;
        int pass;
        if (__builtin_expect(q0 >= n, 1)) {
            pass = fit_block<false>(lane, q0, n, xd, den_s, xavg_s, rden_s, rpts_s, valid, rawd, cy, yring, y, est, ySum_l,
                                    xySum_l, lane_last, r_last, den_last, xavg_last);
        } else {
            pass = fit_block<true>(lane, q0, n, xd, den_s, xavg_s, rden_s, rpts_s, valid, rawd, cy, yring, y, est, ySum_l,
                                   xySum_l, lane_last, r_last, den_last, xavg_last);
        }
        if (pass > 2 * kMaxUnwrapPasses)
            cy.refuse = true;
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
                phaseCorrection = m_pow2 ? (-est[r]) * inv_M : -est[r] / (float)M;
            }
            if (M == 4)
                phaseCorrection = (float)((double)phaseCorrection + kPi4);
            float sn, cs;
            sincosf_wave(phaseCorrection, &sn, &cs);
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
                *reinterpret_cast<f4u *>(p.soft + 2 * (size_t)i0) = v;
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
            const float sre = r_last ? s[1].re : s[0].re;
            const float sim = r_last ? s[1].im : s[0].im;
            const int kk = r_last ? bestK[1] : bestK[0];
            cy.ySum = read_lane(ys, lane_last);
            cy.xySum = read_lane(xys, lane_last);
            cy.est = read_lane(e_, lane_last);
            cy.last_re = read_lane(sre, lane_last);
            cy.last_im = read_lane(sim, lane_last);
            cy.last_k = (uint32_t)__builtin_amdgcn_readlane(kk, lane_last);
            const uint32_t pts_last = (q0 + (uint32_t)nvalid - 1 >= n) ? n : q0 + (uint32_t)nvalid;
            if (pts_last > 1) {  // calculateDenominator ran for the window size reached (cpp/psk_soft.cpp:81-83)
                cy.den = den_last;
                cy.xavg = xavg_last;
            }
        }
        cy.q = q0 + (uint32_t)nvalid;
    }
}

}  // namespace psk
#endif
