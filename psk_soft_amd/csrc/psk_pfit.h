// psk_pfit.h -- the feedback unwrap + LinearFit::next of a whole call, in parallel along time.
//
// The fit kernel of the time-tiled path (psk_tile.hip) walks a channel block by block: 1.3 us a block, 1.3 ms for the
// 2^20 samples of one channel, however idle the machine.  The recurrence (reference cpp/psk_soft.cpp:476-481, 48-87)
//     est[t-1] -> numWraps[t] -> y[t] -> (ySum, xySum) -> est[t]
// is serial, but on a clean signal every link of it can be GUESSED without its predecessor and CHECKED with it:
//
//   pf_begin    the call's prologue (one wave per channel): the sums the first next() starts from -- LinearFit::reset()
//               rebuilds them at the top of practically every call (quirks Q2 / Q3)
//   pf_unwrap   numWraps towards a smooth trajectory of the carrier (phasor averages of 16 symbols, unwrapped against each
//               other by an integer prefix sum: per tile, tile totals) -- see there
//   pf_y        y[t] = (float)(raw[t] + 2 pi numWraps[t])  (the tile's offset = the totals of the tiles before it)
//   pf_ydiff    ySum: the running sum of float-valued terms is exact while the exponent range is small, hence equal to
//               ySum(carried) + prefix sum of y[t] - y[t-n], whatever the order (per tile; tile totals)
//   pf_ysum     ... offsets added; statement :77 re-run on every position's predecessor and compared bit for bit (the
//               certificate for ySum); the two xySum operands of the position, c = fl(xdelta*ySum') and the float term
//   pf_xblock   xySum, r = fl(s - c), s' = fl(r + t): in units of the ulp q of the binade the sums move in, a block
//               of 128 positions adds an INTEGER to s that depends on s only through its parity (xysum_grid,
//               psk_fast_loop.h) -- two integers per block, computed without knowing s (its binade predicted from a
//               plain double prefix sum), plus the range of s for which that holds
//   pf_xwalk    one wave per channel walks the blocks with the true s: two additions a block where the prediction
//               holds, the block's recurrence itself (candidates + certificate, else the lane-after-lane chain) where
//               it does not (sums crossing a binade: a block in a hundred)
//   pf_verify   per block, with its true carried s: the 128 sums, statements :72 / :78 re-run on every position's
//               predecessor and compared bit for bit; the estimates; numWraps re-derived from the predecessor's
//               estimate exactly as :477 does and compared with the guess
//   (commit)    nothing failed anywhere: by induction from the carried state every value IS the reference's; the fit kernel
//               of the time-tiled path, launched behind in any case, ends the call from these results (pf_commit in
//               psk_tile.hip: end-of-call wrap and state commit as in the other kernels) instead of walking it.
//
// A second round (pf_retry, then pf_y ... pf_verify again) takes the calls whose unwrap counts, and nothing else, failed the
// first: its guess is what the first round's estimates -- off by a few hundredths of a radian around the wrong counts --
// say the counts are.  The host enqueues it for a while after a call reported such a failure (a word in page-locked
// memory, PfScratch::hint), so that clean streams do not pay for seven idle launches.
// A call that fails any check in the end (a noisy unwrap, a sum that rounds) is left untouched and the block-by-block fit
// kernel right behind redoes it.  Calls that start with the fit window still filling go there directly (PLAN_PFIT is not set).
#ifndef PSK_PFIT_H
#define PSK_PFIT_H

#include "psk_fast_kernel.h"

namespace psk {

// why a call's parallel fit did not verify (PfChan::fail; reported as psk_soft_stats_t::parallel_fit_refusals)
constexpr uint32_t kPfFailYSum = 1u, kPfFailXySum = 2u, kPfFailUnwrap = 4u;

PSK_DEV bool pf_mine(const ChanPlan &p) { return p.mode == PLAN_FAST && (p.lf_flags & PLAN_PFIT) && p.n_out != 0; }
// round 0: every call planned for the parallel fit; round 1: the calls whose unwrap counts, and nothing else, failed round 0
PSK_DEV bool pf_in_round(const ChanPlan &p, const PfScratch &sc, uint32_t bi, int round) { return pf_mine(p) && (round == 0 || sc.chan[bi].retry != 0u); }

// geometry of the wave's tile
struct PfGeo {
    int lane, n_out, n_blocks, c_begin, c_end, tile;
    uint64_t off;    // first symbol of the channel in the per-symbol arrays
    uint32_t tbase;  // first tile
    uint32_t bbase;  // first block
};
PSK_DEV bool pf_geo(const ChanPlan &p, PfGeo &g)
{
    g.lane = threadIdx.x & 63;
    g.n_out = (int)p.n_out;
    g.n_blocks = (g.n_out + kB - 1) / kB;
    g.tile = (int)blockIdx.x;
    g.c_begin = g.tile * (int)p.tile_blocks;
    g.c_end = g.c_begin + (int)p.tile_blocks < g.n_blocks ? g.c_begin + (int)p.tile_blocks : g.n_blocks;
    g.off = p.tile_off;
    g.tbase = p.tile_base;
    g.bbase = (uint32_t)(p.tile_off / kB);
    return g.c_begin < g.n_blocks;
}
// value leaving the fit window at symbol t: y[t - n], from the call itself or from the carried history
PSK_DEV float pf_z(const ChanPlan &p, const float *y_row, const float *yv, uint32_t fit_cap, int t)
{
    const int n = (int)p.lf_n;
    return t >= n ? y_row[t - n] : yv[(p.lf_head + (uint32_t)t) % fit_cap];
}

// ---- pf_unwrap: grid (tiles, channels) ----
// The guess of the unwrap counts.  The reference unwraps every raw phase towards the estimate fed back from the fit,
// a smooth line through the last phaseAvg values.  Unwrapping towards the previous RAW phase instead follows the noise:
// at 20 dB it slips by 2 pi somewhere in most calls, and everything behind a slip is off.  So the guess is anchored to
// a smooth trajectory of its own: the carrier's phasor averaged over sub-blocks of 16 symbols (noise / 4), the sub-block
// angles unwrapped against each other (an integer prefix sum: per tile, tile totals), and every symbol unwrapped
// towards the line through its sub-block's angle with the slope its neighbours give.  What is left are symbols whose
// own noise comes within a few tenths of pi: single wrong counts, no slips -- the second round's business.
// (The sub-block angles follow a carrier of up to ~0.15 rad per symbol; a channel whose last fit ran steeper than
// kPfSteep keeps the plain guess by consecutive raw phases, good to pi per symbol on a clean signal.)
constexpr float kPfSteep = 0.12f;
__global__ __launch_bounds__(64) void pf_unwrap_kernel(const ChanPlan *__restrict__ plans, const uint32_t *__restrict__ list, uint32_t ch0,
                                                       const ChanState *__restrict__ states, const float *__restrict__ t_raw, PfScratch sc)
{
    const uint32_t bi = list[blockIdx.y];
    const ChanPlan &p = plans[bi];
    PfGeo g;
    if (!pf_mine(p) || !pf_geo(p, g))
        return;
    const float *raw_row = t_raw + g.off;
    int *k_row = sc.k + g.off;
    const float inv2pi = 0.15915494f, two_pi = 6.2831853f;
    const ChanState &st = states[ch0 + bi];
    const bool steep = !(__builtin_fabsf(st.lf_m) < kPfSteep);  // (wave-uniform; NaN: steep)
    int run = 0;
    if (steep) {
        for (int c = g.c_begin; c < g.c_end; c++) {
            const int i0 = c * kB + 2 * g.lane;
            const float2 rw = *reinterpret_cast<const float2 *>(raw_row + i0);
            const float before = c ? raw_row[c * kB - 1] : rw.x;
            const float prev = wave_up1(rw.y, before);
            int j0 = (int)__builtin_rintf((prev - rw.x) * inv2pi);
            const int j1 = (int)__builtin_rintf((rw.x - rw.y) * inv2pi);
            if (i0 == 0)  // (the call's first symbol: its count comes from the carried estimate, pf_y)
                j0 = 0;
            const int incl = wave_scan_i32(j0 + j1);
            const int k0 = run + wave_up1(incl, 0) + j0;
            *reinterpret_cast<int2 *>(k_row + i0) = make_int2(k0, k0 + j1);
            run += __builtin_amdgcn_readlane(incl, 63);
        }
    } else {
        // angle of the sub-block in front of the tile (the call's first: the carried estimate, already unwrapped)
        float a_last = st.phaseEstimate;
        if (g.c_begin > 0) {
            const int t = g.c_begin * kB - 16 + 2 * (g.lane & 7);
            const float2 rw = *reinterpret_cast<const float2 *>(raw_row + t);
            float ph[2] = {__cosf(rw.x) + __cosf(rw.y), __sinf(rw.x) + __sinf(rw.y)};
            wave_scan_f32_multi(ph);
            a_last = atan2f(read_lane(ph[1], 7), read_lane(ph[0], 7));
        }
        const int grp = g.lane >> 3;
        for (int c = g.c_begin; c < g.c_end; c++) {
            const int i0 = c * kB + 2 * g.lane;
            const float2 rw = *reinterpret_cast<const float2 *>(raw_row + i0);
            // phasor sums of the eight sub-blocks (symbols past the end of the call left out): prefix sums over the wave,
            // differenced at the sub-block ends
            const bool v0 = i0 < g.n_out, v1 = i0 + 1 < g.n_out;
            float ph[2] = {(v0 ? __cosf(rw.x) : 0.0f) + (v1 ? __cosf(rw.y) : 0.0f), (v0 ? __sinf(rw.x) : 0.0f) + (v1 ? __sinf(rw.y) : 0.0f)};
            wave_scan_f32_multi(ph);
            const int hi = (g.lane | 7) << 2, lo = ((g.lane & ~7) - 1) << 2;
            const float re = bperm_addr(hi, ph[0]) - (grp ? bperm_addr(lo, ph[0]) : 0.0f);
            const float im = bperm_addr(hi, ph[1]) - (grp ? bperm_addr(lo, ph[1]) : 0.0f);
            const float a = atan2f(im, re);
            // sub-block angles unwrapped against their predecessors
            const float a_left = bperm_addr(lo, a);
            const float a_prev = grp ? a_left : a_last;
            const int jb = (g.lane & 7) == 0 ? (int)__builtin_rintf((a_prev - a) * inv2pi) : 0;
            const int K = run + wave_scan_i32(jb);  // (inclusive: the lane's own sub-block counted)
            const float U = a + two_pi * (float)(K - run);  // unwrapped relative to the block's start
            // slope from the neighbouring sub-blocks (one-sided at the block's ends)
            const float U_l = bperm_addr(((g.lane & ~7) - 8) << 2, U), U_r = bperm_addr(((g.lane & ~7) + 8) << 2, U);
            const float slope = grp == 0 ? (U_r - U) * (1.0f / 16.0f) : grp == 7 ? (U - U_l) * (1.0f / 16.0f) : (U_r - U_l) * (1.0f / 32.0f);
            const float dt0 = (float)(2 * (g.lane & 7)) - 7.5f;
            const float line0 = a + slope * dt0, line1 = line0 + slope;
            const int k0 = K + (int)__builtin_rintf((line0 - rw.x) * inv2pi), k1 = K + (int)__builtin_rintf((line1 - rw.y) * inv2pi);
            *reinterpret_cast<int2 *>(k_row + i0) = make_int2(k0, k1);
            run = __builtin_amdgcn_readlane(K, 63);
            a_last = read_lane(a, 63);
        }
    }
    if (g.lane == 0)
        sc.tile[g.tbase + g.tile].jsum = run;
}

// ---- pf_y: grid (tiles, channels) ----
__global__ __launch_bounds__(64) void pf_y_kernel(const ChanPlan *__restrict__ plans, const uint32_t *__restrict__ list, uint32_t ch0,
                                                  const ChanState *__restrict__ states, const float *__restrict__ t_raw, PfScratch sc, int round)
{
    const uint32_t bi = list[blockIdx.y];
    const ChanPlan &p = plans[bi];
    PfGeo g;
    if (!pf_in_round(p, sc, bi, round) || !pf_geo(p, g))
        return;
    const float *raw_row = t_raw + g.off;
    int *k_row = sc.k + g.off;
    float *y_row = sc.y + g.off;
    // numWraps of the call's first symbol from the carried estimate (cpp/psk_soft.cpp:477), then the tiles before this one
    int koff = 0;
    if (round == 0) {
        for (int j = g.lane; j < g.tile; j += kWave) koff += sc.tile[g.tbase + j].jsum;
        koff = __builtin_amdgcn_readlane(wave_scan_i32(koff), 63);
        // (plain guess: the first count from the carried estimate, cpp/psk_soft.cpp:477; the anchored guess starts from it)
        if (!(__builtin_fabsf(states[ch0 + bi].lf_m) < kPfSteep))
            koff += (int)unwrap_count(states[ch0 + bi].phaseEstimate, (double)raw_row[0]);
    }  // (round 1: the counts are absolute already, corrected by pf_verify where the first guess was wrong)
    // (a tile's blocks are independent here: the next block's operands are asked for before this one is worked on -- a wave
    // otherwise pays a round trip to memory per block, and four such waves do not fill a SIMD's time)
    float2 rw_n = *reinterpret_cast<const float2 *>(raw_row + g.c_begin * kB + 2 * g.lane);
    int2 k_n = *reinterpret_cast<const int2 *>(k_row + g.c_begin * kB + 2 * g.lane);
    for (int c = g.c_begin; c < g.c_end; c++) {
        const int i0 = c * kB + 2 * g.lane;
        const float2 rw = rw_n;
        int2 k = k_n;
        if (c + 1 < g.c_end) {
            rw_n = *reinterpret_cast<const float2 *>(raw_row + i0 + kB);
            k_n = *reinterpret_cast<const int2 *>(k_row + i0 + kB);
        }
        k.x += koff;
        k.y += koff;
        *reinterpret_cast<int2 *>(k_row + i0) = k;
        const double two_pi = PSK_KD(kTwoPi, c);
        const float y0 = (float)((double)rw.x + (double)(long long)k.x * two_pi);  // cpp/psk_soft.cpp:478, :481
        const float y1 = (float)((double)rw.y + (double)(long long)k.y * two_pi);
        *reinterpret_cast<float2 *>(y_row + i0) = make_float2(y0, y1);
    }
}

// ---- pf_ydiff: grid (tiles, channels) ----
__global__ __launch_bounds__(64) void pf_ydiff_kernel(const ChanPlan *__restrict__ plans, const uint32_t *__restrict__ list, uint32_t ch0,
                                                      const float *__restrict__ yvs, uint32_t fit_cap, PfScratch sc, int round)
{
    const uint32_t bi = list[blockIdx.y];
    const ChanPlan &p = plans[bi];
    PfGeo g;
    if (!pf_in_round(p, sc, bi, round) || !pf_geo(p, g))
        return;
    const float *y_row = sc.y + g.off;
    const float *yv = yvs + (size_t)(ch0 + bi) * fit_cap;
    double *S_row = sc.S + g.off;
    double run = 0.0, dlast = 0.0;
    // (the next block's operands asked for ahead, as in pf_y; positions past the end of the call read the padding of the row)
    const int j0 = g.c_begin * kB + 2 * g.lane;
    float2 y_n = *reinterpret_cast<const float2 *>(y_row + j0);
    float z0_n = pf_z(p, y_row, yv, fit_cap, j0), z1_n = pf_z(p, y_row, yv, fit_cap, j0 + 1);
    for (int c = g.c_begin; c < g.c_end; c++) {
        const int i0 = c * kB + 2 * g.lane;
        const float2 y = y_n;
        const float z0 = z0_n, z1 = z1_n;
        if (c + 1 < g.c_end) {
            y_n = *reinterpret_cast<const float2 *>(y_row + i0 + kB);
            z0_n = pf_z(p, y_row, yv, fit_cap, i0 + kB);
            z1_n = pf_z(p, y_row, yv, fit_cap, i0 + kB + 1);
        }
        const bool v0 = i0 < g.n_out, v1 = i0 + 1 < g.n_out;
        const double d0 = v0 ? (double)y.x - (double)z0 : 0.0;
        const double d1 = v1 ? (double)y.y - (double)z1 : 0.0;
        const double incl = wave_scan_f64(d0 + d1);
        const double a0 = (run + wave_up1(incl, 0.0)) + d0;
        *reinterpret_cast<double2 *>(S_row + i0) = make_double2(a0, a0 + d1);
        run += read_lane(incl, 63);
        const int rem = g.n_out - c * kB;
        const int last = (rem < kB ? rem : kB) - 1;
        dlast = read_lane((last & 1) ? a0 + d1 : a0, last >> 1);
    }
    if (g.lane == 0) {
        sc.tile[g.tbase + g.tile].dsum = run;
        sc.tile[g.tbase + g.tile].dlast = dlast;
    }
}

// ---- pf_ysum: grid (tiles, channels) ----
__global__ __launch_bounds__(64) void pf_ysum_kernel(const ChanPlan *__restrict__ plans, const uint32_t *__restrict__ list, uint32_t ch0,
                                                     const ChanState *__restrict__ states, const float *__restrict__ yvs, uint32_t fit_cap,
                                                     PfScratch sc, int round)
{
    const uint32_t bi = list[blockIdx.y];
    const ChanPlan &p = plans[bi];
    PfGeo g;
    if (!pf_in_round(p, sc, bi, round) || !pf_geo(p, g))
        return;
    const float *y_row = sc.y + g.off;
    const float *yv = yvs + (size_t)(ch0 + bi) * fit_cap;
    double *S_row = sc.S + g.off;
    double *c_row = sc.c + g.off;
    float *t_row = sc.tt + g.off;
    double soff = 0.0;
    for (int j = g.lane; j < g.tile; j += kWave) soff += sc.tile[g.tbase + j].dsum;
    soff = sc.chan[bi].ySum_c + wave_sum_f64(soff);
    bool bad = false;
    if (g.tile > 0) {
        // The certificate runs from position to position; where it passes from one tile's wave to the next it must be
        // the same number on both sides: the ySum this tile starts from has to be, bit for bit, the one the tile before
        // ends with (its own offset, summed the same way, plus its last prefix sum).  Equal whenever the sums are exact.
        double soff_b = 0.0;
        for (int j = g.lane; j < g.tile - 1; j += kWave) soff_b += sc.tile[g.tbase + j].dsum;
        soff_b = sc.chan[bi].ySum_c + wave_sum_f64(soff_b);
        bad = !same_bits(soff_b + sc.tile[g.tbase + g.tile - 1].dlast, soff);
    }
    const float xd = p.lf_xdelta;
    const float sizef = (float)(p.lf_n - 1u);  // (float)yvals.size() before the push, :78
    double S_before = soff;  // ySum after the symbol in front of the block
    double xrun = 0.0;
    // (the next block's operands asked for ahead, as in pf_y)
    const int j0 = g.c_begin * kB + 2 * g.lane;
    float2 y_n = *reinterpret_cast<const float2 *>(y_row + j0);
    double2 dl_n = *reinterpret_cast<const double2 *>(S_row + j0);
    float z0_n = pf_z(p, y_row, yv, fit_cap, j0), z1_n = pf_z(p, y_row, yv, fit_cap, j0 + 1);
    for (int c = g.c_begin; c < g.c_end; c++) {
        const int i0 = c * kB + 2 * g.lane;
        const bool v0 = i0 < g.n_out, v1 = i0 + 1 < g.n_out;
        const float2 y = y_n;
        const double2 dl = dl_n;
        const float z0f = z0_n, z1f = z1_n;
        if (c + 1 < g.c_end) {
            y_n = *reinterpret_cast<const float2 *>(y_row + i0 + kB);
            dl_n = *reinterpret_cast<const double2 *>(S_row + i0 + kB);
            z0_n = pf_z(p, y_row, yv, fit_cap, i0 + kB);
            z1_n = pf_z(p, y_row, yv, fit_cap, i0 + kB + 1);
        }
        const double S0 = soff + dl.x, S1 = soff + dl.y;
        const double z0 = v0 ? (double)z0f : 0.0, z1 = v1 ? (double)z1f : 0.0;
        const double Sp = wave_up1(S1, S_before);
        const double a0 = Sp - z0, a1 = S0 - z1;  // ySum after the pop, :70
        bad = bad || (v0 && !same_bits(a0 + (double)y.x, S0)) || (v1 && !same_bits(a1 + (double)y.y, S1));  // :77
        const double c0 = (double)xd * a0, c1 = (double)xd * a1;  // :72
        float t0 = y.x * sizef;  // :78
        t0 = t0 * xd;
        float t1 = y.y * sizef;
        t1 = t1 * xd;
        *reinterpret_cast<double2 *>(S_row + i0) = make_double2(S0, S1);
        *reinterpret_cast<double2 *>(c_row + i0) = make_double2(c0, c1);
        *reinterpret_cast<float2 *>(t_row + i0) = make_float2(t0, t1);
        const double bs = wave_sum_f64((v0 ? (double)t0 - c0 : 0.0) + (v1 ? (double)t1 - c1 : 0.0));
        if (g.lane == 0)
            sc.blk[g.bbase + c].bsum = bs;
        xrun += bs;
        const int rem = g.n_out - c * kB;
        const int last = (rem < kB ? rem : kB) - 1;
        S_before = read_lane((last & 1) ? S1 : S0, last >> 1);
    }
    if (g.lane == 0)
        sc.tile[g.tbase + g.tile].xsum = xrun;
    if (vote_any(bad) && g.lane == 0)
        atomicOr(&sc.chan[bi].fail, kPfFailYSum);
}

// The per-position classification of xysum_grid (psk_fast_loop.h) for a GIVEN binade and mode, and its resolution for
// a given parity of the carried sum: the same arithmetic, split so that a block can be prepared before its carried
// sum is known.
struct PfGrid {
    bool T[kR], V[kR], E[kR], dpos[kR];
    double inc[kR], ch[kR];
};
PSK_DEV void pf_classify(double inv_q, bool modeB, const double (&c)[kR], const double (&t)[kR], PfGrid &gr)
{
#pragma unroll
    for (int r = 0; r < kR; r++) {
        const double nc = c[r] * inv_q, nt = t[r] * inv_q;
        const double ch = __builtin_rint(nc);
        gr.ch[r] = ch;
        gr.inc[r] = nt - ch;
        if (!modeB) {
            const double d = nc - ch;
            gr.T[r] = __builtin_fabs(d) == 0.5;
            gr.dpos[r] = d > 0.0;
            gr.E[r] = false;
            gr.V[r] = odd_f64(gr.inc[r]);
        } else {
            gr.T[r] = odd_f64(gr.inc[r]);
            const bool hw = odd_f64((gr.T[r] ? gr.inc[r] - 1.0 : gr.inc[r]) * 0.5);
            gr.E[r] = hw;
            gr.V[r] = !gr.T[r] && hw;
            gr.dpos[r] = false;
        }
    }
}
PSK_DEV void pf_resolve(int lane, bool modeB, bool P_c, const PfGrid &gr, double &a0, double &a1)
{
    const bool Ac = gr.T[0] || gr.T[1];
    const bool Vc = gr.T[1] ? gr.V[1] : (gr.V[0] != gr.V[1]);
    const unsigned long long mA = vote_mask(Ac), mV = vote_mask(Vc);
    const bool G = (__builtin_amdgcn_mbcnt_hi((unsigned)(mV >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mV, 0u)) & 1u) != 0;
    const unsigned long long mF = vote_mask(Ac && G);
    const unsigned long long lt = lane ? (~0ull >> (64 - lane)) : 0ull;
    const unsigned long long X = mA & lt, Y = mF & lt;
    const bool P_in = (X ? (Y > (X >> 1)) : P_c) != G;
    const bool sel0 = P_in != gr.E[0];
    const bool P0 = gr.T[0] ? gr.V[0] : (P_in != gr.V[0]);
    const bool sel1 = P0 != gr.E[1];
    if (!modeB) {
        a0 = gr.inc[0] + ((gr.T[0] && sel0) ? (gr.dpos[0] ? -1.0 : 1.0) : 0.0);
        a1 = gr.inc[1] + ((gr.T[1] && sel1) ? (gr.dpos[1] ? -1.0 : 1.0) : 0.0);
    } else {
        a0 = gr.inc[0] + (gr.T[0] ? (sel0 ? 1.0 : -1.0) : 0.0);
        a1 = gr.inc[1] + (gr.T[1] ? (sel1 ? 1.0 : -1.0) : 0.0);
    }
}

// ---- pf_xblock: grid (tiles, channels) ----
__global__ __launch_bounds__(64) void pf_xblock_kernel(const ChanPlan *__restrict__ plans, const uint32_t *__restrict__ list, uint32_t ch0,
                                                       const ChanState *__restrict__ states, PfScratch sc, int round)
{
    const uint32_t bi = list[blockIdx.y];
    const ChanPlan &p = plans[bi];
    PfGeo g;
    if (!pf_in_round(p, sc, bi, round) || !pf_geo(p, g))
        return;
    const double *c_row = sc.c + g.off;
    const float *t_row = sc.tt + g.off;
    double s_pred = 0.0;
    for (int j = g.lane; j < g.tile; j += kWave) s_pred += sc.tile[g.tbase + j].xsum;
    s_pred = sc.chan[bi].xySum_c + wave_sum_f64(s_pred);
    // (binade, mode) predicted for the block in front of the tile, as that tile's wave predicts it
    int prev_scale = -1;
    if (g.c_begin > 0) {
        const double s_before = s_pred - sc.blk[g.bbase + g.c_begin - 1].bsum;
        const int eb_b = (__double2hiint(s_before - c_row[(g.c_begin - 1) * kB]) >> 20) & 0x7ff;
        if (eb_b >= 60 && eb_b <= 2040)
            prev_scale = 2 * eb_b + (!(__builtin_fabs(s_before) < pow2_biased(eb_b + 1)) ? 1 : 0);
    }
    for (int c = g.c_begin; c < g.c_end; c++) {
        const int i0 = c * kB + 2 * g.lane;
        const bool v0 = i0 < g.n_out, v1 = i0 + 1 < g.n_out;
        const double2 cv = *reinterpret_cast<const double2 *>(c_row + i0);
        const float2 tv = *reinterpret_cast<const float2 *>(t_row + i0);
        // (positions past the end add nothing and never tie)
        const double cc[kR] = {v0 ? cv.x : 0.0, v1 ? cv.y : 0.0};
        const double tt[kR] = {v0 ? (double)tv.x : 0.0, v1 ? (double)tv.y : 0.0};
        const double c_first = read_lane(cc[0], 0);
        const double r_first = s_pred - c_first;
        const int eb = (__double2hiint(r_first) >> 20) & 0x7ff;
        PfBlock rec;
        rec.bsum = sc.blk[g.bbase + c].bsum;
        rec.a = __builtin_inf();  // (empty range)
        rec.b = -__builtin_inf();
        rec.kmul = rec.D0s = rec.D1s = 0.0;
        rec.flags = rec.pad = 0;
        int scale_id = -1;  // (binade, mode) the record is made for
        if (eb >= 60 && eb <= 2040) {  // (else zero, denormal, inf, NaN: the walker runs the block itself)
            const double inv_q = pow2_biased(2098 - eb);
            const bool modeB = !(__builtin_fabs(s_pred) < pow2_biased(eb + 1));
            const bool neg = s_pred < 0.0;
            PfGrid gr;
            pf_classify(inv_q, modeB, cc, tt, gr);
            double a0, a1, b0, b1;
            pf_resolve(g.lane, modeB, false, gr, a0, a1);
            pf_resolve(g.lane, modeB, true, gr, b0, b1);
            const double incl = wave_scan_f64(a0 + a1);
            const double D0 = read_lane(incl, 63), D1 = wave_sum_f64(b0 + b1);
            // Range of the carried sum for which D0 / D1 hold.  With sigma = its sign and everything in units of q:
            //   sigma*s_j = sigma*S_c + e_j,  e_j = sigma * (prefix sum of the increments up to j), e_{-1} = 0
            //   sigma*r_j = sigma*s_{j-1} - u_j = sigma*S_c + w_j,  u_j = sigma*RN(c_j),  w_j = e_{j-1} - u_j.
            // Mode A wants every r_j in [2^52, 2^53) and every s_j below 2^53; mode B every s_j in [2^53, 2^54) and every
            // r_j in [2^52, 2^53).  The extremes are taken in float, rounded outwards, from the prefix sums of an EVEN carried
            // sum; an odd one moves them by a unit per tie: `slack` covers both.  A block whose carried sum falls outside
            // is run by the walker itself.
            const double sg = neg ? -1.0 : 1.0;
            const double e_before = sg * wave_up1(incl, 0.0);
            const double e0 = e_before + sg * a0, e1 = e0 + sg * a1;
            const double w0 = e_before - sg * gr.ch[0], w1 = e0 - sg * gr.ch[1];
            const float kUp = 1.0000002f;
            const float big = 3.0e38f;
            // (positions past the end: their increments are 0, so e repeats the last valid value; their w is left out)
            const float e_hi = wave_max_f32(__builtin_fmaxf(__builtin_fmaxf((float)e0, (float)e1) * kUp, 0.0f));
            const float e_lo = wave_max_f32(__builtin_fmaxf(__builtin_fmaxf(-(float)e0, -(float)e1) * kUp, 0.0f));  // = -(min e), >= 0
            const float w0f = v0 ? (float)w0 : -big, w1f = v1 ? (float)w1 : -big;
            const float n0f = v0 ? -(float)w0 : -big, n1f = v1 ? -(float)w1 : -big;
            // max w and max -w, either sign: shift by the block's own scale so that the non-negative wave maximum applies
            const float scale = wave_max_f32(__builtin_fmaxf(__builtin_fabsf(v0 ? (float)w0 : 0.0f), __builtin_fabsf(v1 ? (float)w1 : 0.0f))) * kUp + 1.0f;
            const float w_hi = wave_max_f32(__builtin_fmaxf(__builtin_fmaxf(w0f, w1f) + scale, 0.0f)) * kUp - scale * 0.999999f;   // >= max w
            const float w_lo = wave_max_f32(__builtin_fmaxf(__builtin_fmaxf(n0f, n1f) + scale, 0.0f)) * kUp - scale * 0.999999f;   // >= max -w
            const double slack = 300.0;
            const double two52 = 4503599627370496.0, two53 = 9007199254740992.0;
            // sigma*S_c + w_j >= 2^52 for all j  <=  sigma*S_c >= 2^52 + max(-w);   sigma*S_c + w_j < 2^53  <=  sigma*S_c < 2^53 - max(w)
            const double r_lo = two52 + (double)w_lo + slack, r_hi = two53 - (double)w_hi - slack;
            double lo, hi;
            if (!modeB) {
                lo = r_lo;
                const double s_hi = two53 - (double)e_hi - slack;
                hi = s_hi < r_hi ? s_hi : r_hi;
            } else {
                const double s_lo = two53 + (double)e_lo + slack;
                lo = s_lo > r_lo ? s_lo : r_lo;
                hi = r_hi;
            }
            // (inside that range the binade of the intermediates and the mode are the predicted ones: nothing else to test)
            if (lo < hi) {
                const double q = pow2_biased(eb - 52);
                rec.a = neg ? -(hi * q) : lo * q;
                rec.b = neg ? -(lo * q) : hi * q;
                rec.kmul = inv_q * (modeB ? 0.25 : 0.5);
                rec.D0s = D0 * q;
                rec.D1s = D1 * q;
                scale_id = 2 * eb + (modeB ? 1 : 0);
                // parity of the sum after the block: mode A odd(S + D) = P ^ odd(D); mode B (S and D even) odd(S/2 + D/2)
                const bool f0 = modeB ? odd_f64(D0 * 0.5) : odd_f64(D0), f1 = modeB ? odd_f64(D1 * 0.5) : odd_f64(D1);
                rec.flags = (f0 ? 1 : 0) | (f1 ? 2 : 0) | ((scale_id == prev_scale) ? 4 : 0);
            }
        }
        prev_scale = scale_id;
        if (g.lane == 0)
            sc.blk[g.bbase + c] = rec;
        s_pred += rec.bsum;
    }
}

// the two xySum statements on every position's predecessor (:72, :78), bit for bit; c and t are the operands pf_ysum made
PSK_DEV bool pf_x_ok(double s_c, const bool (&valid)[kR], const double (&c)[kR], const double (&t)[kR], const double (&xs)[kR])
{
    const double xp = wave_up1(xs[1], s_c);
    const bool x0 = same_bits((xp - c[0]) + t[0], xs[0]), x1 = same_bits((xs[0] - c[1]) + t[1], xs[1]);
    return vote_all((x0 || !valid[0]) && (x1 || !valid[1]));
}

// ---- pf_xwalk: one wave per channel ----
// the block's recurrence itself, as the block-by-block kernels run it: candidates from the true carried sum and their
// certificate (try_grid), else the recurrence lane after lane; the 128 sums go to xs_row, the last valid one is returned
struct PfOps {  // the two xySum operands of a lane's two positions (cpp/psk_soft.cpp:72, :78), as pf_ysum left them
    double2 cv;
    float2 tv;
};
PSK_DEV PfOps pf_load_ops(int lane, int b, const double *c_row, const float *t_row)
{
    const int i0 = b * kB + 2 * lane;  // (the per-symbol arrays are padded to whole blocks)
    PfOps o;
    o.cv = *reinterpret_cast<const double2 *>(c_row + i0);
    o.tv = *reinterpret_cast<const float2 *>(t_row + i0);
    return o;
}
PSK_DEV double pf_walk_block(int lane, int b, int n_out, double s_c, const PfOps &ops, double *xs_row, bool try_grid, bool &grid_failed)
{
    const int i0 = b * kB + 2 * lane;
    const bool valid[kR] = {i0 < n_out, i0 + 1 < n_out};
#ifdef PSK_DIAG_WALK_NOLOAD  // (timing experiments only: wrong sums, which the certificate of pf_verify turns into a refusal)
    const double2 cv = make_double2(s_c * 0.25, s_c * 0.125);
    const float2 tv = make_float2((float)lane, 1.0f);
#else
    const double2 cv = ops.cv;
    const float2 tv = ops.tv;
#endif
    const double cc[kR] = {valid[0] ? cv.x : 0.0, valid[1] ? cv.y : 0.0};
    const double tt[kR] = {valid[0] ? (double)tv.x : 0.0, valid[1] ? (double)tv.y : 0.0};
    double xs[kR];
    bool done = false;
#ifdef PSK_DIAG_WALK_NOCHAIN
    xs[0] = s_c - cc[0], xs[1] = s_c - tt[1];
    done = true;
#endif
    if (try_grid && !done) {  // (a block whose prepared range merely missed the sum is crossing a binade: the candidates cannot hold)
        xysum_grid(lane, s_c, cc, tt, xs);
        done = pf_x_ok(s_c, valid, cc, tt, xs);
        grid_failed = !done;
    }
    if (!done) {
        double x = s_c;
#pragma unroll 1
        for (int k = 0; k < kWave; k += PSK_CHAIN_UNROLL) {
#pragma unroll
            for (int u = 0; u < PSK_CHAIN_UNROLL; u++) {
                const double bb = wave_up1(x, s_c);
                x = ((bb - cc[0]) + tt[0] - cc[1]) + tt[1];  // :72 and :78, twice
            }
        }
        const double b_fin = wave_up1(x, s_c);
        xs[0] = (b_fin - cc[0]) + tt[0];
        xs[1] = x;
    }
    *reinterpret_cast<double2 *>(xs_row + i0) = make_double2(xs[0], xs[1]);
    const int rem = n_out - b * kB;
    const int last = (rem < kB ? rem : kB) - 1;
    return read_lane((last & 1) ? xs[1] : xs[0], last >> 1);
}

// value of `v` in lane `src` (any lane; ds_bpermute)
PSK_DEV double pf_from_lane(double v, int src)
{
    const int lo = bperm_addr(src << 2, __double2loint(v)), hi = bperm_addr(src << 2, __double2hiint(v));
    return __hiloint2double(hi, lo);
}
// The walk over 64 blocks at a time.  While the scale (binade, mode) does not change, a block is a map on (parity, sum):
//     sum += D[parity],  parity ^= F[parity],
// and such maps compose associatively: for either incoming parity, the total added and the outgoing parity.  A wave scan
// over the blocks' maps (six steps) gives every block the sum and parity it is entered with; every block then checks its
// range by itself.  The first block that cannot be entered that way -- its range does not hold the sum, or its scale is
// not its predecessor's -- stops the scan: everything in front of it stands, it is dealt with on its own, the scan resumes
// behind it.
__global__ __launch_bounds__(64) void pf_xwalk_kernel(const ChanPlan *__restrict__ plans, const uint32_t *__restrict__ list,
                                                      const PfBlock *__restrict__ blk_all, PfScratch sc, int round)
{
    const uint32_t bi = list[blockIdx.x];
    const ChanPlan &p = plans[bi];
    if (!pf_in_round(p, sc, bi, round))
        return;
    const int lane = threadIdx.x & 63;
    const int n_out = (int)p.n_out;
    const int n_blocks = (n_out + kB - 1) / kB;
    const PfBlock *const blk = blk_all + (uint32_t)(p.tile_off / kB);
    PfWalk *const walk = sc.walk + (uint32_t)(p.tile_off / kB);
    const double *c_row = sc.c + p.tile_off;
    const float *t_row = sc.tt + p.tile_off;
    double *xs_row = sc.xs + p.tile_off;
    double s_c = sc.chan[bi].xySum_c;
    uint32_t slow_blocks = 0;
#ifdef PSK_DIAG_WALK_TIME
    const unsigned long long t_begin = wall_clock64();  // (100 MHz)
#endif
    int par = -1;  // parity of the carried sum in the scale of the block in front (wave-uniform; -1: not known)
    for (int b0 = 0; b0 < n_blocks; b0 += kWave) {
        const bool have = b0 + lane < n_blocks;
        const PfBlock rec = blk[have ? b0 + lane : n_blocks - 1];
        const int nb = n_blocks - b0 < kWave ? n_blocks - b0 : kWave;
        const bool usable = have && rec.a < rec.b;
        double s_in_mine = 0.0;
        int slow_mine = 0;
        int start = 0;
        while (start < nb) {
#ifdef PSK_DIAG_WALK_SCANS  // (diagnostic build: the statistic counts scan passes, a thousand each, next to the slow blocks)
            slow_blocks += 1000u;
#endif
            // parity the run starts with: carried, or from the sum where the scale is new
            const int fl_s = __builtin_amdgcn_readlane(rec.flags, start);
            if (par < 0 || !(fl_s & 4))
                par = (__builtin_amdgcn_fract(s_c * read_lane(rec.kmul, start)) != 0.0) ? 1 : 0;
            par = __builtin_amdgcn_readfirstlane(par);
            // this lane's map, the identity in front of the run; behind a stop nothing is used
            const bool in_run = lane >= start && usable;
            double dS0 = in_run ? rec.D0s : 0.0, dS1 = in_run ? rec.D1s : 0.0;
            int out = in_run ? ((rec.flags & 1) ? 1 : 0) | ((rec.flags & 2) ? 0 : 2) : 2;  // bit p: parity that leaves for parity p entering
#pragma unroll
            for (int d = 1; d < kWave; d <<= 1) {
                const int src = lane - d;
                const bool take = src >= start;
                const double l0 = pf_from_lane(dS0, src), l1 = pf_from_lane(dS1, src);
                const int lout = bperm_addr(src << 2, out);
                if (take) {  // left = lanes up to src, right = what this lane holds
                    const int o0 = lout & 1, o1 = (lout >> 1) & 1;
                    const double r0 = dS0, r1 = dS1;
                    dS0 = l0 + (o0 ? r1 : r0);
                    dS1 = l1 + (o1 ? r1 : r0);
                    out = ((out >> o0) & 1) | (((out >> o1) & 1) << 1);
                }
            }
            // sum and parity after this lane's block, and before it
            const double s_after = s_c + (par ? dS1 : dS0);
            const int par_after = (out >> par) & 1;
            const double s_left = pf_from_lane(s_after, lane - 1);  // (every lane takes part: a lane that sits out supplies nothing)
            const double s_before = lane > start ? s_left : s_c;
            const bool inside = lane >= start && lane < nb;
            const bool enter_ok = usable && rec.a < s_before && s_before < rec.b && (lane == start || (rec.flags & 4));
            const unsigned long long stops = vote_mask(inside && !enter_ok);
            const int f = stops ? (int)__builtin_ctzll(stops) : nb;  // first block that cannot be entered in this run
            if (lane >= start && lane < f)
                s_in_mine = s_before;
            if (f > start) {
                s_c = read_lane(s_after, f - 1);
                par = __builtin_amdgcn_readlane(par_after, f - 1);
            }
            if (f >= nb)
                break;
            // block f: its scale differs from its predecessor's (resume there, the parity taken from the sum), or its range
            // does not hold the sum (the block's recurrence itself; the sums are crossing a binade)
            const double a_f = read_lane(rec.a, f), b_f = read_lane(rec.b, f);
            if (a_f < s_c && s_c < b_f && f > start) {
                par = -1;
                start = f;
                continue;
            }
            // Blocks the walker runs itself, one after the other for as long as the next one cannot be entered either (a
            // channel whose sums hover around zero is such a run from end to end): no scan in between, and the next block's
            // operands are asked for before this block's recurrence starts -- a run costs its recurrences and little else.
            int fs = f;
            PfOps ops = pf_load_ops(lane, b0 + fs, c_row, t_row);
            bool grid_first = !(a_f < b_f);
            bool grid_failed = false;  // (once the candidates of a block of the run have not verified, the rest of it goes straight to the recurrence)
            for (;;) {
                const bool more = fs + 1 < nb;
                PfOps ahead = ops;
                if (more)
                    ahead = pf_load_ops(lane, b0 + fs + 1, c_row, t_row);
                if (lane == fs) {
                    s_in_mine = s_c;
                    slow_mine = 1;
                }
                s_c = pf_walk_block(lane, b0 + fs, n_out, s_c, ops, xs_row, grid_first, grid_failed);
                slow_blocks++;
                start = fs + 1;
                if (!more)
                    break;
                const double a_n = read_lane(rec.a, fs + 1), b_n = read_lane(rec.b, fs + 1);
                if (a_n < s_c && s_c < b_n)
                    break;  // (enterable: the scan resumes there)
                fs++;
                ops = ahead;
                grid_first = !(a_n < b_n) && !grid_failed;
            }
            par = -1;
        }
        if (have) {
            PfWalk w;
            w.s_in = s_in_mine;
            w.slow = slow_mine;
            w.pad = 0;
            walk[b0 + lane] = w;
        }
    }
#ifdef PSK_DIAG_WALK_TIME  // (diagnostic build: the statistic is the walker's own duration, in units of 10 ns)
    slow_blocks = (uint32_t)((wall_clock64() - t_begin) / 1u);
#endif
    if (lane == 0)
        sc.chan[bi].slow_blocks = slow_blocks;
}

// ---- pf_verify: grid (tiles, channels) ----
__global__ __launch_bounds__(64) void pf_verify_kernel(const ChanPlan *__restrict__ plans, const uint32_t *__restrict__ list, uint32_t ch0,
                                                       const ChanState *__restrict__ states, const float *__restrict__ yvs, uint32_t fit_cap,
                                                       const float *__restrict__ t_raw, float *__restrict__ t_est, PfScratch sc, int round)
{
    const uint32_t bi = list[blockIdx.y];
    const ChanPlan &p = plans[bi];
    PfGeo g;
    if (!pf_in_round(p, sc, bi, round) || !pf_geo(p, g))
        return;
    const ChanState &st = states[ch0 + bi];
    const float *raw_row = t_raw + g.off;
    const int *k_row = sc.k + g.off;
    const float *y_row = sc.y + g.off;
    const double *S_row = sc.S + g.off;
    const float *yv = yvs + (size_t)(ch0 + bi) * fit_cap;
    double *xs_row = sc.xs + g.off;
    float *est_row = t_est + g.off;
    const uint32_t n = p.lf_n;
    const float xd = p.lf_xdelta;
    const float sizef = (float)(p.lf_n - 1u);  // (float)yvals.size() before the push, :78
    float den_s = st.lf_den, xavg_s = st.lf_xavg;
    fit_denominator(xd, n, den_s, xavg_s);
    const FitKnown fk = fit_known(xd, n, den_s, xavg_s);
    bool bad = false, bad_k = false;
    for (int c = g.c_begin; c < g.c_end; c++) {
        const int i0 = c * kB + 2 * g.lane;
        const bool valid[kR] = {i0 < g.n_out, i0 + 1 < g.n_out};
        const PfWalk *rec = sc.walk + g.bbase + c;
        const double s_c = rec->s_in;
        // The two xySum operands of every position, formed again from ySum and y exactly as pf_ysum formed them (:70, :72, :78) --
        // 12 bytes a symbol less to read than taking them from scratch, in a launch that is bound by its traffic; y[t - n] sits
        // in the cache lines the block's neighbours have just read.
        const double2 Sv = *reinterpret_cast<const double2 *>(S_row + i0);
        const float2 y = *reinterpret_cast<const float2 *>(y_row + i0);
        const double S_front = c ? S_row[c * kB - 1] : sc.chan[bi].ySum_c + 0.0;  // ySum after the symbol in front of the block
        const double Sp = wave_up1(Sv.y, S_front);
        const double z0 = valid[0] ? (double)pf_z(p, y_row, yv, fit_cap, i0) : 0.0, z1 = valid[1] ? (double)pf_z(p, y_row, yv, fit_cap, i0 + 1) : 0.0;
        const double a0 = Sp - z0, a1 = Sv.x - z1;  // ySum after the pop, :70
        float t0 = y.x * sizef;  // :78
        t0 = t0 * xd;
        float t1 = y.y * sizef;
        t1 = t1 * xd;
        const double cc[kR] = {valid[0] ? (double)xd * a0 : 0.0, valid[1] ? (double)xd * a1 : 0.0};  // :72
        const double tt[kR] = {valid[0] ? (double)t0 : 0.0, valid[1] ? (double)t1 : 0.0};
        double xs[kR];
        if (rec->slow) {
            const double2 xv = *reinterpret_cast<const double2 *>(xs_row + i0);
            xs[0] = xv.x, xs[1] = xv.y;
        } else {
            xysum_grid(g.lane, s_c, cc, tt, xs);
            *reinterpret_cast<double2 *>(xs_row + i0) = make_double2(xs[0], xs[1]);
        }
        if (!pf_x_ok(s_c, valid, cc, tt, xs))
            bad = true;
        // (... and from block to block: the sum the walker carried into the next block is this block's last one, bit for bit)
        if (c + 1 < g.n_blocks && !same_bits(read_lane(xs[1], 63), rec[1].s_in))
            bad = true;
        float m_;
        const float est[kR] = {fit_value_known(Sv.x, xs[0], fk, m_), fit_value_known(Sv.y, xs[1], fk, m_)};
        // the estimate fed back into the block's first symbol: the carried one, or the fit at the symbol in front of it
        const float est_before = c ? fit_value_known(S_front, s_c, fk, m_) : st.phaseEstimate;
        const float est_prev0 = wave_up1(est[1], est_before);
        const float2 rw = *reinterpret_cast<const float2 *>(raw_row + i0);
        const int2 k = *reinterpret_cast<const int2 *>(k_row + i0);
        // round((est_prev - raw)/2pi) == k  <=>  |est_prev - (raw + 2 pi k)| < pi: see fit_block
        const bool sure0 = __builtin_fabsf(est_prev0 - y.x) < 3.0f && __builtin_fabsf(y.x) < 65536.0f;
        const bool sure1 = __builtin_fabsf(est[0] - y.y) < 3.0f && __builtin_fabsf(y.y) < 65536.0f;
        if (!vote_all((sure0 || !valid[0]) && (sure1 || !valid[1]))) {
            const long long w0 = unwrap_count(est_prev0, (double)rw.x, c), w1 = unwrap_count(est[0], (double)rw.y, c);
            if ((valid[0] && w0 != (long long)k.x) || (valid[1] && w1 != (long long)k.y)) {
                bad_k = true;
                // what the (slightly off) estimates of this round say the counts are: the second round's guess
                if (round == 0 && w0 == (long long)(int)w0 && w1 == (long long)(int)w1)
                    *reinterpret_cast<int2 *>(sc.k + g.off + i0) = make_int2((int)w0, (int)w1);
            }
        }
        *reinterpret_cast<float2 *>(est_row + i0) = make_float2(est[0], est[1]);
    }
    if (vote_any(bad) && g.lane == 0)
        atomicOr(&sc.chan[bi].fail, kPfFailXySum);
    if (vote_any(bad_k) && g.lane == 0)
        atomicOr(&sc.chan[bi].fail, kPfFailUnwrap);
}

// ---- pf_retry: between the rounds, one wave per channel ----
__global__ __launch_bounds__(64) void pf_retry_kernel(const ChanPlan *__restrict__ plans, const uint32_t *__restrict__ list, PfScratch sc)
{
    const uint32_t bi = list[blockIdx.x];
    if (!pf_mine(plans[bi]) || (threadIdx.x & 63) != 0)
        return;
    const uint32_t f = sc.chan[bi].fail;
    if (f & kPfFailUnwrap)
        *sc.hint = 1u;
    const bool again = f == kPfFailUnwrap;  // (a sum that rounds would round again)
    sc.chan[bi].retry = again ? 1u : 0u;
    if (again)
        sc.chan[bi].fail = 0u;
}

}  // namespace psk
#endif
