// psk_tile.hip -- the time-tiled kernels (psk_tile_kernel.h): the fit and back kernels, which do not depend on
// samplesPerBaud, and the dispatch to the per-samplesPerBaud instantiations of the front kernel (psk_tile_inst.hip).
#include "psk_tile_kernel.h"
#include "psk_pfit.h"

namespace psk {

// What the tiles of the front kernel found, folded over the call: true = the call is not theirs to carry.
PSK_DEV bool tile_fold(const ChanPlan &p, const TileInfo *ti, int n_tiles, int lane, int &exact_blocks_out, float &emax_out)
{
    float emax = 0.0f;
    unsigned umax = 0u, umin1 = 0xFFFFFFFFu, refuse_b = 0u, gap_b = 0x7F800000u, cap_b = 0x7F800000u, wmax_b = 0u;
    int exact_blocks = 0;
    for (int j = lane; j < n_tiles; j += kWave) {
        const TileInfo t = ti[j];
        umax = t.umax > umax ? t.umax : umax;
        umin1 = t.umin1 < umin1 ? t.umin1 : umin1;
        refuse_b |= t.refuse;
        const unsigned g = __float_as_uint(t.gap_rel), cp = __float_as_uint(t.cap), w = __float_as_uint(t.wmax);
        gap_b = g < gap_b ? g : gap_b;
        cap_b = cp < cap_b ? cp : cap_b;
        // (a NaN maximum -- bit pattern above inf's -- stays on top and fails both comparisons below)
        wmax_b = w > wmax_b ? w : wmax_b;
        exact_blocks += (int)t.stat_exact;
        emax = __builtin_fmaxf(emax, t.emax);
    }
    emax_out = wave_max_f32(__builtin_fmaxf(emax, 0.0f));
    umax = wave_max_u32(umax);
    umin1 = wave_min_u32(umin1);
    gap_b = wave_min_u32(gap_b);
    cap_b = wave_min_u32(cap_b);
    wmax_b = wave_max_u32(wmax_b);
    exact_blocks = __builtin_amdgcn_readlane(wave_scan_i32(exact_blocks), 63);
    bool refuse = __any(refuse_b != 0u);
    {
        const float wmax = __uint_as_float(wmax_b);
        // the screening thresholds of every tile must cover the drift of the reference's sums at the scale of the call
        // (a silent call: every window sum is exactly zero, `wmax` is the denormal the phase-index bits of the screening's
        // argmax leave behind, thresholds and `cap` are zero -- and the first phase wins there as in the reference)
        if (!(wmax <= __uint_as_float(cap_b)) && !(wmax < 1.0e-37f))
            refuse = true;
        // an exact re-decision closer than that drift: only the exactness guard (quirk Q8) can vouch for it
        const bool ambiguous = !(__uint_as_float(gap_b) > wmax);
        if (umin1 != 0xFFFFFFFFu && ambiguous) {
            int emax = (int)(umax >> 23), emin = (int)((umin1 + 1u) >> 23);
            emax = emax < 1 ? 1 : emax;
            emin = emin < 1 ? 1 : emin;
            const int terms_log2 = 32 - __builtin_clz((unsigned)(p.A + 2u * kB));
            if (24 + (emax - emin) + terms_log2 > 52)
                refuse = true;
        }
    }
    exact_blocks_out = exact_blocks;
    return refuse;
}

// ---- front, any samplesPerBaud and numAvg: grid (tiles, channels of the launch) ----
// The instantiated front kernel keeps a symbol's samplesPerBaud energies in one lane and scans along the symbols; this
// one lays the timing PHASES across the lanes (phase k in lane k mod 64, up to 16 a lane) and walks the tile's symbols
// one after the other, every lane updating its own window sums in the reference's order (cpp/psk_soft.cpp:445-452: a
// symbol's energies are added as its samples arrive; :572-577: the leaving symbol's are subtracted after the pick):
// coalesced loads whatever the samplesPerBaud, no window history to keep whatever the numAvg.  The pick is a reduction
// over the wave per symbol (first maximum, :462); picked samples wait in lane (symbol mod 64) until 64 are there and take
// the M-th power and atan2f together.  A tile's first window is a fresh sum of the numAvg - 1 symbols in front of it --
// the call's first one IS resyncEnergy's sum (:619-636), the others equal the reference's running sums while those are
// exact: the tile reports the exponent range of every energy it saw and the smallest gap between best and runner-up,
// the fit kernel's fold decides as for the instantiated front kernel.  ~25 us per block of 128 symbols and tile
// (the reference-order kernel: 600).
constexpr int kAnyPhases = 16;  // phases per lane: samplesPerBaud <= 1024
struct AnyTop {
    double best, second;
    int k;
};
PSK_DEV AnyTop any_merge(const AnyTop &a, const AnyTop &b)  // first maximum: the larger sum, the lower phase on a tie
{
    const bool b_wins = b.best > a.best || (b.best == a.best && b.k < a.k);
    AnyTop r;
    r.best = b_wins ? b.best : a.best;
    r.k = b_wins ? b.k : a.k;
    const double loser = b_wins ? a.best : b.best;
    const double s2 = a.second > b.second ? a.second : b.second;
    r.second = loser > s2 ? loser : s2;
    return r;
}
// wave-wide maximum of a double / minimum of an unsigned: the scan pattern of wave_scan_f64 (row_shr 1, 2, 4, 8, row_bcast 15
// and 31) with the extremum in place of the addition, lanes without a source taking the identity; the result sits in lane 63
template <int CTRL, int ROW_MASK>
PSK_DEV double any_f64_from(double v)
{
    return __hiloint2double(__builtin_amdgcn_update_dpp((int)0xFFF00000u, __double2hiint(v), CTRL, ROW_MASK, 0xF, false),
                            __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xF, false));
}
PSK_DEV double any_max_f64(double v)
{
    v = __builtin_fmax(v, any_f64_from<0x111, 0xF>(v));
    v = __builtin_fmax(v, any_f64_from<0x112, 0xF>(v));
    v = __builtin_fmax(v, any_f64_from<0x114, 0xF>(v));
    v = __builtin_fmax(v, any_f64_from<0x118, 0xF>(v));
    v = __builtin_fmax(v, any_f64_from<0x142, 0xA>(v));
    v = __builtin_fmax(v, any_f64_from<0x143, 0xC>(v));
    return read_lane(v, 63);
}
// the same for U independent values, level by level: U chains that do not wait for one another
template <int CTRL, int ROW_MASK, int U>
PSK_DEV void any_max_f64_level(double (&v)[U])
{
    double o[U];
#pragma unroll
    for (int u = 0; u < U; u++) o[u] = any_f64_from<CTRL, ROW_MASK>(v[u]);
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = __builtin_fmax(v[u], o[u]);
}
template <int U>
PSK_DEV void any_max_f64_multi(double (&v)[U])
{
    any_max_f64_level<0x111, 0xF>(v);
    any_max_f64_level<0x112, 0xF>(v);
    any_max_f64_level<0x114, 0xF>(v);
    any_max_f64_level<0x118, 0xF>(v);
    any_max_f64_level<0x142, 0xA>(v);
    any_max_f64_level<0x143, 0xC>(v);
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = read_lane(v[u], 63);
}
template <int CTRL, int ROW_MASK>
PSK_DEV unsigned any_u32_from(unsigned v)
{
    return (unsigned)__builtin_amdgcn_update_dpp(-1, (int)v, CTRL, ROW_MASK, 0xF, false);
}
PSK_DEV unsigned any_min_u32(unsigned v)
{
    unsigned o;
    o = any_u32_from<0x111, 0xF>(v), v = o < v ? o : v;
    o = any_u32_from<0x112, 0xF>(v), v = o < v ? o : v;
    o = any_u32_from<0x114, 0xF>(v), v = o < v ? o : v;
    o = any_u32_from<0x118, 0xF>(v), v = o < v ? o : v;
    o = any_u32_from<0x142, 0xA>(v), v = o < v ? o : v;
    o = any_u32_from<0x143, 0xC>(v), v = o < v ? o : v;
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
template <int CTRL, int ROW_MASK, int U>
PSK_DEV void any_min_u32_level(unsigned (&v)[U])
{
    unsigned o[U];
#pragma unroll
    for (int u = 0; u < U; u++) o[u] = any_u32_from<CTRL, ROW_MASK>(v[u]);
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = o[u] < v[u] ? o[u] : v[u];
}
template <int U>
PSK_DEV void any_min_u32_multi(unsigned (&v)[U])
{
    any_min_u32_level<0x111, 0xF>(v);
    any_min_u32_level<0x112, 0xF>(v);
    any_min_u32_level<0x114, 0xF>(v);
    any_min_u32_level<0x118, 0xF>(v);
    any_min_u32_level<0x142, 0xA>(v);
    any_min_u32_level<0x143, 0xC>(v);
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = (unsigned)__builtin_amdgcn_readlane((int)v[u], 63);
}
// NP: phases a lane holds at most (samplesPerBaud <= 64 * NP for every channel of the launch).  With one phase a lane (NP = 1,
// samplesPerBaud <= 64) the energies entering and leaving the window are asked for eight symbols at a time: the walk is one
// dependent step per symbol, and a step that waits for its own two loads is all latency.
template <int NP>
__global__ __launch_bounds__(64) void psk_tile_front_any_kernel(const ChanPlan *__restrict__ plans, const uint32_t *__restrict__ list, uint32_t ch0,
                                                                const ChanState *__restrict__ states, const float2 *__restrict__ rings,
                                                                uint32_t ring_cap, TileInfo *__restrict__ tiles, float *__restrict__ t_raw,
                                                                float2 *__restrict__ t_s, PfChan *__restrict__ pf_chan)
{
    const int lane = threadIdx.x & 63;
    const uint32_t bi = list[blockIdx.y];
    const ChanPlan &p = plans[bi];
    if (!tile_plan_mine(p) || !(p.lf_flags & PLAN_ANYFRONT))
        return;
    const int n_out = (int)p.n_out;
    const int n_blocks = (n_out + kB - 1) / kB;
    const int c_begin = (int)(blockIdx.x * p.tile_blocks);
    if (c_begin >= n_blocks)
        return;
    if (blockIdx.x == 0 && lane == 0)
        pf_chan[bi].fail = pf_chan[bi].done = pf_chan[bi].slow_blocks = pf_chan[bi].retry = 0u;
    const int c_end = c_begin + (int)p.tile_blocks < n_blocks ? c_begin + (int)p.tile_blocks : n_blocks;
    const int i_begin = c_begin * kB, i_end = c_end * kB < n_out ? c_end * kB : n_out;
    const uint32_t ch = ch0 + bi;
    XView X;
    X.ring = reinterpret_cast<const f2g *>(rings + ((size_t)ch * 2u + p.ring_src) * ring_cap);
    X.in = reinterpret_cast<const f2g *>(p.in);
    X.L0 = p.ring_len0;
    const int S = (int)p.S, A = (int)p.A;
    const uint32_t M = p.M;
    const int nk = NP == 1 ? 1 : (S + kWave - 1) / kWave;  // (a compile-time 1 where it can be: no branches inside the symbol step)
    const bool timing = S > 1;  // (samplesPerBaud == 1: a symbol per sample, nothing to pick, cpp/psk_soft.cpp:468-469)
    const AtanTabDev atab = atan_tab_dev(lane);
    float *raw_row = t_raw + p.tile_off;
    float2 *s_row = t_s + p.tile_off;

    double W[NP];
#pragma unroll
    for (int j = 0; j < NP; j++) W[j] = 0.0;
    unsigned umax = 0u, umin1 = 0xFFFFFFFFu;
    bool refuse = false;
    float emax = 0.0f;
    // energy of sample `k` of symbol `tau` (0 past the symbol's end), with the guard's bookkeeping
    auto energy_at = [&](uint64_t j) -> float {  // of sample j of the call's stream
        const float2 v = x_at(X, j);
        const float e = norm_f(v.x, v.y);
        const unsigned eb = __float_as_uint(e);
        if (eb >= 0x7F800000u)
            refuse = true;  // (inf / NaN: the reference-order kernel's)
        umax = eb > umax ? eb : umax;
        umin1 = (eb - 1u) < umin1 ? (eb - 1u) : umin1;
        emax = __builtin_fmaxf(emax, e);
        return e;
    };
    auto energy = [&](long long tau, int k) -> float {
        return k < S ? energy_at((uint64_t)tau * (uint64_t)S + (uint64_t)k) : 0.0f;
    };
    // the window in front of the tile's first symbol: symbols i_begin .. i_begin + numAvg - 2, in order
    // (several symbols' loads at a time here too: with numAvg in the thousands this loop is most of a tile)
    constexpr int UW = NP == 1 ? 8 : NP == 4 ? 4 : 1;
    const long long tau_end = (long long)i_begin + A - 1;
    // One phase a lane: the window's samples come in through LDS, 512 at a time with every lane loading (a lane a phase would
    // leave most of the wave idle for narrow symbols, and wait for memory once per eight symbols); the phases then add their
    // energies up in order.
    __shared__ float e_lds[kWave * 8], e_lds_out[kWave * 8];
    const bool staged = NP == 1 && timing;
    if (staged) {
        const uint64_t j1 = (uint64_t)tau_end * (uint64_t)S;
        const uint32_t per = (uint32_t)((kWave * 8) / S) * (uint32_t)S;  // whole symbols
        for (uint64_t jb = (uint64_t)i_begin * (uint64_t)S; jb < j1; jb += per) {
            const uint32_t len = j1 - jb < per ? (uint32_t)(j1 - jb) : per;
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const uint32_t o = (uint32_t)(lane + kWave * r);
                e_lds[o] = o < len ? energy_at(jb + o) : 0.0f;
            }
            wave_lds_fence();
            if (lane < S) {
                const int ns = (int)(len / (uint32_t)S);
                for (int t = 0; t < ns; t++) W[0] += (double)e_lds[t * S + lane];
            }
            wave_lds_fence();
        }
    }
    for (long long tau = i_begin; timing && !staged && tau < tau_end; tau += UW) {
        float e[UW][NP];
#pragma unroll
        for (int u = 0; u < UW; u++)
#pragma unroll
            for (int j = 0; j < NP; j++)
                e[u][j] = (tau + u < tau_end && j < nk) ? energy(tau + u, lane + kWave * j) : 0.0f;
#pragma unroll
        for (int u = 0; u < UW; u++)
#pragma unroll
            for (int j = 0; j < NP; j++)
                if (tau + u < tau_end && j < nk)
                    W[j] += (double)e[u][j];
    }
    float gap_rel = __builtin_inff(), wmax = 0.0f;
    int kb = 0, k_last = 0;
    // (the picked samples are only fetched when 64 picks are known: one load per lane in place of one per symbol)
    constexpr int U = NP == 1 ? 8 : 1;  // symbols whose energies are loaded together
    for (int i0 = i_begin; i0 < i_end; i0 += U) {
        float e_in[U][NP], e_out[U][NP];
        if constexpr (U > 1) {
            // the chunk's symbols entering and leaving the window: two runs of U * samplesPerBaud consecutive samples, loaded by all
            // the lanes into LDS and read back by the phases (the leaving ones have been through the bookkeeping when they entered)
            if (timing) {
                const int nu = i_end - i0 < U ? i_end - i0 : U;
                const uint32_t len = (uint32_t)(nu * S);
                const uint64_t j_in = (uint64_t)((long long)i0 + A - 1) * (uint64_t)S, j_out = (uint64_t)i0 * (uint64_t)S;
#pragma unroll
                for (int r = 0; r < 8; r++) {
                    if ((uint32_t)(kWave * r) >= len)  // (wave-uniform)
                        break;
                    const uint32_t o = (uint32_t)(lane + kWave * r);
                    float ei = 0.0f, eo = 0.0f;
                    if (o < len) {
                        ei = energy_at(j_in + o);
                        const float2 v = x_at(X, j_out + o);
                        eo = norm_f(v.x, v.y);
                    }
                    e_lds[o] = ei;
                    e_lds_out[o] = eo;
                }
                wave_lds_fence();
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const bool on = u < nu && lane < S;
                    e_in[u][0] = on ? e_lds[u * S + lane] : 0.0f;
                    e_out[u][0] = on ? e_lds_out[u * S + lane] : 0.0f;
                }
                wave_lds_fence();
            }
        }
        // (no way out of the chunk half-way: one straight run of code lets the steps of its symbols overlap -- each is a long chain of
        // dependent reductions, and only the two additions to the sums tie a step to the one before.  Symbols past the end of
        // the tile add zeros and leave nothing behind.)
        if constexpr (NP == 1) {
            if (timing) {
                // one phase a lane: the sums of the chunk's symbols first (two additions a symbol, in order), then the three
                // reductions for all of them level by level, then the bookkeeping in order
                double best[U], second[U];
                unsigned k_win[U];
#pragma unroll
                for (int u = 0; u < U; u++) {
                    W[0] += (double)e_in[u][0];  // the newest symbol of the window arrives
                    best[u] = lane < S ? W[0] : -__builtin_inf();
                    W[0] -= (double)e_out[u][0];  // the oldest leaves
                    second[u] = best[u];
                }
                any_max_f64_multi(best);
#pragma unroll
                for (int u = 0; u < U; u++) k_win[u] = second[u] == best[u] ? (unsigned)lane : 0xFFFFFFFFu;  // (std::max_element, cpp/psk_soft.cpp:462: the first)
                any_min_u32_multi(k_win);
#pragma unroll
                for (int u = 0; u < U; u++) second[u] = (unsigned)lane == k_win[u] ? -__builtin_inf() : second[u];
                any_max_f64_multi(second);
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const int i = i0 + u;
                    const bool live = i < i_end;
                    // (every sum NaN -- a window of overflowing energies: inf arrives, inf - inf leaves -- and no lane holds the maximum.
                    // The tile refuses, energy_at has seen to that; the pick still has to be a sample of the symbol, it is fetched below)
                    const int kbest = k_win[u] < (unsigned)S ? (int)k_win[u] : 0;
                    k_last = live ? kbest : k_last;
                    wmax = live ? __builtin_fmaxf(wmax, (float)best[u] * 1.0000002f) : wmax;
                    const float g = (float)(best[u] - second[u]) / (2.0f * drift_bound(i + 1 + kB, (uint32_t)A));
                    const float gap_new = (g < gap_rel) ? g : ((g == g) ? gap_rel : 0.0f);
                    gap_rel = live ? gap_new : gap_rel;
                    kb = (live && lane == (i & (kWave - 1))) ? kbest : kb;  // the pick (cpp/psk_soft.cpp:465) waits in lane (i mod 64)
                }
            }
        } else if (timing) {
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int i = i0 + u;
                const bool live = U == 1 || i < i_end;
                // the newest symbol of the window arrives
#pragma unroll
                for (int j = 0; j < NP; j++)
                    if (j < nk)
                        W[j] += (double)(U > 1 ? e_in[u][j] : energy((long long)i + A - 1, lane + kWave * j));
                // first maximum over the phases, and the runner-up (std::max_element, cpp/psk_soft.cpp:462, keeps the first of equal sums)
                AnyTop top;
                top.best = -__builtin_inf();
                top.second = -__builtin_inf();
                top.k = 0x7fffffff;
#pragma unroll
                for (int j = 0; j < NP; j++) {
                    const int k = lane + kWave * j;
                    if (j < nk && k < S) {
                        AnyTop one;
                        one.best = W[j];
                        one.second = -__builtin_inf();
                        one.k = k;
                        top = any_merge(top, one);
                    }
                }
                // (`top`: this lane's own phases.  The wave's largest sum; the lowest phase that holds it; the largest of all the others --
                // the winner's lane puts up the runner-up among its own.  The sums are finite, or the tile refuses -- but runs to its end.)
                const double best = any_max_f64(top.best);
                const unsigned k_win = any_min_u32(top.best == best ? (unsigned)top.k : 0xFFFFFFFFu);
                const double second = any_max_f64((unsigned)top.k == k_win ? top.second : top.best);
                const int kbest = k_win < (unsigned)S ? (int)k_win : 0;  // (every sum NaN: see above -- k_win is AnyTop's sentinel then)
                k_last = live ? kbest : k_last;
                const float best_f = (float)best;
                wmax = live ? __builtin_fmaxf(wmax, best_f * 1.0000002f) : wmax;
                const float g = (float)(best - second) / (2.0f * drift_bound(i + 1 + kB, (uint32_t)A));
                const float gap_new = (g < gap_rel) ? g : ((g == g) ? gap_rel : 0.0f);
                gap_rel = live ? gap_new : gap_rel;
                // the oldest symbol of the window leaves
#pragma unroll
                for (int j = 0; j < NP; j++)
                    if (j < nk)
                        W[j] -= (double)(U > 1 ? e_out[u][j] : energy(i, lane + kWave * j));
                // the pick (cpp/psk_soft.cpp:465) waits in lane (i mod 64)
                kb = (live && lane == (i & (kWave - 1))) ? kbest : kb;
            }
        }
        // (tiles start on block boundaries and U divides 64: a group of 64 picks ends with a chunk)
        const int i = (i0 + U < i_end ? i0 + U : i_end) - 1;
        const int slot = i & (kWave - 1);
        if (slot == kWave - 1 || i == i_end - 1) {
            const int mine = i - slot + lane;
            const bool have = lane <= slot;
            // (samplesPerBaud == 1 takes the packet's sample as it comes, :468-469: the deque may still hold stale samples of
            // an earlier configuration, which that mode never touches)
            const uint64_t j_pick = !have ? (uint64_t)X.L0 : timing ? (uint64_t)mine * (uint64_t)S + (uint64_t)kb : (uint64_t)X.L0 + (uint64_t)mine;
            const float2 pk = x_at(X, j_pick);
            cf32 sv;
            sv.re = pk.x, sv.im = pk.y;
            const cf32 pw = cpow_uint<false>(sv, M);
            const float raw = atan2f_wave(pw.im, pw.re, atab);
            if (have) {
                if (!(is_fin(pw.re) && is_fin(pw.im)))
                    refuse = true;
                raw_row[mine] = raw;
                s_row[mine] = pk;
                if (p.sidx && timing)
                    p.sidx[mine] = (int16_t)(unsigned short)kb;
            }
        }
    }
    const unsigned umax_w = wave_max_u32(umax), umin1_w = wave_min_u32(umin1);
    const bool refuse_w = vote_any(refuse);
    const float emax_w = wave_max_f32(__builtin_fmaxf(emax, 0.0f));
    if (lane == 0) {
        TileInfo &t = tiles[p.tile_base + blockIdx.x];
        t.umax = umax_w;
        t.umin1 = umin1_w;
        t.refuse = refuse_w ? 1u : 0u;
        t.gap_rel = gap_rel;
        t.wmax = wmax;
        t.stat_exact = (uint32_t)(c_end - c_begin);
        t.last_k = (uint32_t)k_last;
        t.cap = __builtin_inff();
        t.emax = emax_w;
    }
}

// ---- the parallel fit verified (psk_pfit.h): end of the call from its results ----
PSK_DEV void pf_commit(const ChanPlan &p, uint32_t bi, uint32_t ch, ChanState *st, float2 *rings, uint32_t ring_cap, float *yvs, uint32_t fit_cap,
                       float *yring, uint32_t ymask, TileInfo *ti, int n_tiles, int exact_blocks, const float2 *t_s, const float *t_est,
                       const PfScratch &sc, int lane)
{
    const int n_out = (int)p.n_out;
    const int n_blocks = (n_out + kB - 1) / kB;
    float2 *ring_base = rings + (size_t)ch * 2u * ring_cap;
    const float2 *ring_src = ring_base + (size_t)p.ring_src * ring_cap;
    float2 *ring_dst = ring_base + (size_t)(p.ring_src ^ 1u) * ring_cap;
    float *yv = yvs + (size_t)ch * fit_cap;
    XView X;
    X.ring = reinterpret_cast<const f2g *>(ring_src);
    X.in = reinterpret_cast<const f2g *>(p.in);
    X.L0 = p.ring_len0;

    // the fit window at the end of the call, where the epilogue looks for it: position j of the window at ring index
    // q - n + j, q = values pushed so far
    const int n = (int)p.lf_n;
    const float *y_row = sc.y + p.tile_off;
    const uint32_t q_end = p.lf_len0 + (uint32_t)n_out;
    for (int j = lane; j < n; j += kWave) {
        const int t = n_out - n + j;
        yring[(q_end - (uint32_t)n + (uint32_t)j) & ymask] = t >= 0 ? y_row[t] : yv[(p.lf_head + (uint32_t)(t + n)) % fit_cap];
    }
    wave_lds_fence();
    FastCarry cy;
    cy.ySum = sc.S[p.tile_off + (uint64_t)(n_out - 1)];
    cy.xySum = sc.xs[p.tile_off + (uint64_t)(n_out - 1)];
    cy.est = t_est[p.tile_off + (uint64_t)(n_out - 1)];
    cy.last_re = st->last_re;
    cy.last_im = st->last_im;
    const float last0_re = cy.last_re, last0_im = cy.last_im;
    cy.den = st->lf_den;
    cy.xavg = st->lf_xavg;
    fit_denominator(p.lf_xdelta, p.lf_n, cy.den, cy.xavg);
    {
        const FitKnown fk = fit_known(p.lf_xdelta, p.lf_n, cy.den, cy.xavg);
        float m_hint;
        (void)fit_value_known(cy.ySum, cy.xySum, fk, m_hint);
        m_hint *= p.lf_xdelta;
        cy.slope = is_fin(m_hint) ? m_hint : 0.0f;
    }
    cy.m = cy.b = 0.0f;
    cy.wmax = __builtin_inff();  // (no piece of a call cut in time follows a time-tiled one: see PLAN_CARRY_DRIFT)
    cy.q = q_end;
    cy.last_k = ti[n_tiles - 1].last_k;
    cy.stat_blocks = (uint32_t)n_blocks;
    cy.stat_extra = 0;
    cy.stat_exact_blocks = (uint32_t)exact_blocks;
    cy.stat_chain = sc.chan[bi].slow_blocks;
    if (p.diff) {
        const float2 l = t_s[p.tile_off + (uint64_t)(n_out - 1)];
        cy.last_re = l.x;
        cy.last_im = l.y;
    }
    if (lane == 0) {
        ti[0].last0_re = last0_re;
        ti[0].last0_im = last0_im;
        sc.chan[bi].done = 1u;
    }
    call_epilogue(p, st, yv, fit_cap, yring, ymask, X, ring_dst, lane, cy, kGuardTiled);
    if (lane == 0)
        st->stat_pfit = 1u | (sc.chan[bi].retry ? 0x100u : 0u);
}

// ---- fit: one wave per channel ----
__global__ __launch_bounds__(64) void psk_tile_fit_kernel(const ChanPlan *__restrict__ plans, const uint32_t *__restrict__ list, uint32_t ch0,
                                                          ChanState *__restrict__ states, float2 *__restrict__ rings, uint32_t ring_cap,
                                                          float *__restrict__ yvs, uint32_t fit_cap, uint32_t y_len,
                                                          TileInfo *__restrict__ tiles, const float *__restrict__ t_raw,
                                                          const float2 *__restrict__ t_s, float *__restrict__ t_est, PfScratch sc)
{
    extern __shared__ __attribute__((aligned(16))) float lds_dyn[];
    float *const yring = lds_dyn;
    const uint32_t ymask = y_len - 1u;
    const int lane = threadIdx.x & 63;
    const uint32_t bi = list[blockIdx.x];
    const ChanPlan &p = plans[bi];
    if (!tile_plan_mine(p))
        return;
    const uint32_t ch = ch0 + bi;
    ChanState *st = &states[ch];
    const int n_out = (int)p.n_out;
    const int n_blocks = (n_out + kB - 1) / kB;
    const int n_tiles = (n_blocks + (int)p.tile_blocks - 1) / (int)p.tile_blocks;
    TileInfo *const ti = tiles + p.tile_base;

    int exact_blocks = 0;
    float emax = 0.0f;
    const bool refuse = tile_fold(p, ti, n_tiles, lane, exact_blocks, emax);
    if (lane == 0)
        st->emax_hint = emax;  // (not part of the reference's state: kept whether or not the call stays here)
    if (refuse) {
        if (lane == 0) {
            st->guard = 1u;
            atomicAdd(p.handed_over, 1u);
        }
        return;
    }
    if ((p.lf_flags & PLAN_PFIT) && sc.chan[bi].fail == 0u) {
        // the parallel fit (psk_pfit.h) carried this call and every position verified: commit from its results
        pf_commit(p, bi, ch, st, rings, ring_cap, yvs, fit_cap, yring, ymask, ti, n_tiles, exact_blocks, t_s, t_est, sc, lane);
        return;
    }

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
    const float last0_re = cy.last_re, last0_im = cy.last_im;

    const uint32_t n = p.lf_n;
    const float xd = p.lf_xdelta;
    float den_s = cy.den, xavg_s = cy.xavg;
    if (n > 1)
        fit_denominator(xd, n, den_s, xavg_s);
    const FitKnown fk = fit_known(xd, n, den_s, xavg_s);
    const float *raw_row = t_raw + p.tile_off;
    float *est_row = t_est + p.tile_off;
    float2 nxt = *reinterpret_cast<const float2 *>(raw_row + 2 * lane);
    for (int c = 0; c < n_blocks; c++) {
        const int i0 = c * kB + 2 * lane;
        const float raw[kR] = {nxt.x, nxt.y};
        if (c + 1 < n_blocks)  // (the next block's raw phases are on their way while this one is fitted)
            nxt = *reinterpret_cast<const float2 *>(raw_row + i0 + kB);
        const bool valid[kR] = {i0 < n_out, i0 + 1 < n_out};
        const int rem = n_out - c * kB;
        const int nvalid = rem < kB ? rem : kB;
        const int lane_last = (nvalid - 1) >> 1, r_last = (nvalid - 1) & 1;
        if (cy.chain_streak >> 16)
            __builtin_amdgcn_s_setprio(PSK_CHAIN_PRIO);
        else
            __builtin_amdgcn_s_setprio(0);
        float est[kR];
        fit_stage<false>(c, lane, n, xd, den_s, xavg_s, fk, valid, raw, nvalid, lane_last, r_last, yring, ymask, cy, est);
        if (__any(cy.refuse)) {
            if (lane == 0) {
                st->guard = 1u;
                atomicAdd(p.handed_over, 1u);
            }
            return;
        }
        *reinterpret_cast<float2 *>(est_row + i0) = make_float2(est[0], est[1]);  // (rows padded to whole blocks)
    }

    cy.last_k = ti[n_tiles - 1].last_k;
    cy.stat_exact_blocks = (uint32_t)exact_blocks;
    if (p.diff) {  // psk_soft_i::last = the last sample output (cpp/psk_soft.cpp:486-491)
        const float2 l = t_s[p.tile_off + (uint64_t)(n_out - 1)];
        cy.last_re = l.x;
        cy.last_im = l.y;
    }
    if (lane == 0) {  // the back kernel starts from the old one
        ti[0].last0_re = last0_re;
        ti[0].last0_im = last0_im;
    }
    call_epilogue(p, st, yv, fit_cap, yring, ymask, X, ring_dst, lane, cy, kGuardTiled);
    if ((p.lf_flags & PLAN_PFIT) && lane == 0) {
        st->stat_pfit = sc.chan[bi].fail << 1;  // (statistics: why the parallel fit left the call to this kernel)
        if (sc.chan[bi].fail & kPfFailUnwrap)
            *sc.hint = 1u;
    }
}

// ---- fit, a range of tiles at a time: the pipelined mode (psk_capi.cpp) ----
// For a few hundred to a couple of thousand channels the serial fit (1.5 us a block and channel, whatever the company) and the
// machine-filling front stage take about as long as each other; one after the other they add up.  The host therefore cuts the
// call's tiles into ranges of about one machine-load and launches  front(range j)  on one stream,  fit(range j)  on a second
// behind it,  back(range j)  on a third behind that: the fit of range j runs under the front stage of range j + 1.  Between
// two launches a channel's fit state waits in scratch (PipeCarry: the FastCarry of every lane, what the fold over the tiles
// has seen so far, and the LDS ring of unwrapped phases); the launch that reaches the channel's last tile ends the call exactly
// as the one-launch fit kernel does.
struct PipeCarry {
    FastCarry cy[kWave];
    unsigned umax, umin1, refuse, gap_b, cap_b, wmax_b;
    int exact_blocks;
    float emax;
    float last0_re, last0_im;
    uint32_t dead;  // a launch of this call has handed the channel over: the later ones leave it alone
    uint32_t pad;
};
__global__ __launch_bounds__(64) void psk_tile_fit_range_kernel(const ChanPlan *__restrict__ plans, const uint32_t *__restrict__ list, uint32_t ch0,
                                                                ChanState *__restrict__ states, float2 *__restrict__ rings, uint32_t ring_cap,
                                                                float *__restrict__ yvs, uint32_t fit_cap, uint32_t y_len,
                                                                TileInfo *__restrict__ tiles, const float *__restrict__ t_raw,
                                                                const float2 *__restrict__ t_s, float *__restrict__ t_est,
                                                                PipeCarry *__restrict__ carry, float *__restrict__ carry_y, uint32_t tile0,
                                                                uint32_t ntiles)
{
    extern __shared__ __attribute__((aligned(16))) float lds_dyn[];
    float *const yring = lds_dyn;
    const uint32_t ymask = y_len - 1u;
    const int lane = threadIdx.x & 63;
    const uint32_t bi = list[blockIdx.x];
    const ChanPlan &p = plans[bi];
    if (!tile_plan_mine(p))
        return;
    const uint32_t ch = ch0 + bi;
    ChanState *st = &states[ch];
    const int n_out = (int)p.n_out;
    const int n_blocks = (n_out + kB - 1) / kB;
    const int n_tiles = (n_blocks + (int)p.tile_blocks - 1) / (int)p.tile_blocks;
    if ((int)tile0 >= n_tiles)
        return;  // (the channel's call ended in an earlier range)
    const int t_end = (int)(tile0 + ntiles) < n_tiles ? (int)(tile0 + ntiles) : n_tiles;
    const bool first = tile0 == 0u, last = t_end == n_tiles;
    TileInfo *const ti = tiles + p.tile_base;
    PipeCarry &pc = carry[blockIdx.x];
    float *const ysave = carry_y + (size_t)blockIdx.x * y_len;
    if (!first && pc.dead)
        return;

    // what the front stage reports about this range's tiles, folded into what the earlier ranges left (tile_fold, range by range)
    float emax = first ? 0.0f : pc.emax;
    unsigned umax = first ? 0u : pc.umax, umin1 = first ? 0xFFFFFFFFu : pc.umin1, refuse_b = first ? 0u : pc.refuse;
    unsigned gap_b = first ? 0x7F800000u : pc.gap_b, cap_b = first ? 0x7F800000u : pc.cap_b, wmax_b = first ? 0u : pc.wmax_b;
    int exact_blocks = 0;
    for (int j = (int)tile0 + lane; j < t_end; j += kWave) {
        const TileInfo t = ti[j];
        umax = t.umax > umax ? t.umax : umax;
        umin1 = t.umin1 < umin1 ? t.umin1 : umin1;
        refuse_b |= t.refuse;
        const unsigned g = __float_as_uint(t.gap_rel), cp = __float_as_uint(t.cap), w = __float_as_uint(t.wmax);
        gap_b = g < gap_b ? g : gap_b;
        cap_b = cp < cap_b ? cp : cap_b;
        wmax_b = w > wmax_b ? w : wmax_b;
        exact_blocks += (int)t.stat_exact;
        emax = __builtin_fmaxf(emax, t.emax);
    }
    emax = wave_max_f32(__builtin_fmaxf(emax, 0.0f));
    umax = wave_max_u32(umax);
    umin1 = wave_min_u32(umin1);
    gap_b = wave_min_u32(gap_b);
    cap_b = wave_min_u32(cap_b);
    wmax_b = wave_max_u32(wmax_b);
    exact_blocks = __builtin_amdgcn_readlane(wave_scan_i32(exact_blocks), 63) + (first ? 0 : pc.exact_blocks);
    bool refuse = __any(refuse_b != 0u);
    {   // (the verdict of tile_fold on everything seen so far: a later range can only make it stricter)
        const float wmax = __uint_as_float(wmax_b);
        if (!(wmax <= __uint_as_float(cap_b)) && !(wmax < 1.0e-37f))
            refuse = true;
        const bool ambiguous = !(__uint_as_float(gap_b) > wmax);
        if (umin1 != 0xFFFFFFFFu && ambiguous) {
            int eh = (int)(umax >> 23), el = (int)((umin1 + 1u) >> 23);
            eh = eh < 1 ? 1 : eh;
            el = el < 1 ? 1 : el;
            const int terms_log2 = 32 - __builtin_clz((unsigned)(p.A + 2u * kB));
            if (24 + (eh - el) + terms_log2 > 52)
                refuse = true;
        }
    }
    auto hand_over = [&]() {
        if (lane == 0) {
            st->emax_hint = emax;
            st->guard = 1u;
            atomicAdd(p.handed_over, 1u);
            pc.dead = 1u;
        }
    };
    if (refuse) {
        hand_over();
        return;
    }

    float2 *ring_base = rings + (size_t)ch * 2u * ring_cap;
    const float2 *ring_src = ring_base + (size_t)p.ring_src * ring_cap;
    float2 *ring_dst = ring_base + (size_t)(p.ring_src ^ 1u) * ring_cap;
    float *yv = yvs + (size_t)ch * fit_cap;
    XView X;
    X.ring = reinterpret_cast<const f2g *>(ring_src);
    X.in = reinterpret_cast<const f2g *>(p.in);
    X.L0 = p.ring_len0;

    FastCarry cy;
    float last0_re, last0_im;
    if (first) {
        call_prologue(p, st, yv, fit_cap, yring, ymask, lane, cy);
        last0_re = cy.last_re, last0_im = cy.last_im;
        if (lane == 0) {  // the back stage starts from the old `last`
            ti[0].last0_re = last0_re;
            ti[0].last0_im = last0_im;
        }
    } else {
        cy = pc.cy[lane];
        last0_re = pc.last0_re, last0_im = pc.last0_im;
        for (uint32_t j = lane; j < y_len; j += kWave) yring[j] = ysave[j];
        wave_lds_fence();
    }

    const uint32_t n = p.lf_n;
    const float xd = p.lf_xdelta;
    float den_s = cy.den, xavg_s = cy.xavg;
    if (n > 1)
        fit_denominator(xd, n, den_s, xavg_s);
    const FitKnown fk = fit_known(xd, n, den_s, xavg_s);
    const float *raw_row = t_raw + p.tile_off;
    float *est_row = t_est + p.tile_off;
    const int c0 = (int)tile0 * (int)p.tile_blocks;
    const int c1 = last ? n_blocks : t_end * (int)p.tile_blocks;
    __builtin_amdgcn_s_setprio(3);  // (the serial stage: the front stage's waves of the next range fill the gaps it leaves)
    float2 nxt = *reinterpret_cast<const float2 *>(raw_row + c0 * kB + 2 * lane);
    for (int c = c0; c < c1; c++) {
        const int i0 = c * kB + 2 * lane;
        const float raw[kR] = {nxt.x, nxt.y};
        if (c + 1 < c1)
            nxt = *reinterpret_cast<const float2 *>(raw_row + i0 + kB);
        const bool valid[kR] = {i0 < n_out, i0 + 1 < n_out};
        const int rem = n_out - c * kB;
        const int nvalid = rem < kB ? rem : kB;
        const int lane_last = (nvalid - 1) >> 1, r_last = (nvalid - 1) & 1;
        float est[kR];
        fit_stage<false>(c, lane, n, xd, den_s, xavg_s, fk, valid, raw, nvalid, lane_last, r_last, yring, ymask, cy, est);
        if (__any(cy.refuse)) {
            hand_over();
            return;
        }
        *reinterpret_cast<float2 *>(est_row + i0) = make_float2(est[0], est[1]);  // (rows padded to whole blocks)
    }

    if (!last) {
        pc.cy[lane] = cy;
        wave_lds_fence();
        for (uint32_t j = lane; j < y_len; j += kWave) ysave[j] = yring[j];
        if (lane == 0) {
            pc.umax = umax, pc.umin1 = umin1, pc.refuse = refuse_b ? 1u : 0u, pc.gap_b = gap_b, pc.cap_b = cap_b, pc.wmax_b = wmax_b;
            pc.exact_blocks = exact_blocks;
            pc.emax = emax;
            pc.last0_re = last0_re, pc.last0_im = last0_im;
            pc.dead = 0u;
        }
        return;
    }
    if (lane == 0)
        st->emax_hint = emax;
    cy.last_k = ti[n_tiles - 1].last_k;
    cy.stat_exact_blocks = (uint32_t)exact_blocks;
    if (p.diff) {  // psk_soft_i::last = the last sample output (cpp/psk_soft.cpp:486-491)
        const float2 l = t_s[p.tile_off + (uint64_t)(n_out - 1)];
        cy.last_re = l.x;
        cy.last_im = l.y;
    }
    call_epilogue(p, st, yv, fit_cap, yring, ymask, X, ring_dst, lane, cy, kGuardTiled);
}

hipError_t launch_tile_fit_range(const ChanPlan *plans, const uint32_t *list, uint32_t ch0, uint32_t nch, ChanState *states, float2 *rings,
                                 uint32_t ring_cap, float *yvs, uint32_t fit_cap, uint32_t y_len, TileInfo *tiles, const float *t_raw,
                                 const float2 *t_s, float *t_est, void *carry, float *carry_y, uint32_t tile0, uint32_t ntiles,
                                 hipStream_t stream)
{
    if (!nch || !ntiles)
        return hipSuccess;
    hipLaunchKernelGGL(psk_tile_fit_range_kernel, dim3(nch), dim3(kWave), sizeof(float) * (size_t)y_len, stream, plans, list, ch0, states, rings,
                       ring_cap, yvs, fit_cap, y_len, tiles, t_raw, t_s, t_est, static_cast<PipeCarry *>(carry), carry_y, tile0, ntiles);
    return hipGetLastError();
}
size_t pipe_carry_bytes() { return sizeof(PipeCarry); }

// ---- pf_begin: one wave per channel (psk_pfit.h) ----
__global__ __launch_bounds__(64) void pf_begin_kernel(const ChanPlan *__restrict__ plans, const uint32_t *__restrict__ list, uint32_t ch0,
                                                      const ChanState *__restrict__ states, const float *__restrict__ yvs, uint32_t fit_cap,
                                                      uint32_t y_len, PfScratch sc)
{
    extern __shared__ __attribute__((aligned(16))) float lds_dyn[];
    const int lane = threadIdx.x & 63;
    const uint32_t bi = list[blockIdx.x];
    const ChanPlan &p = plans[bi];
    if (!pf_mine(p))
        return;
    FastCarry cy;
    call_prologue(p, &states[ch0 + bi], yvs + (size_t)(ch0 + bi) * fit_cap, fit_cap, lds_dyn, y_len - 1u, lane, cy);
    if (lane == 0) {
        sc.chan[bi].ySum_c = cy.ySum;
        sc.chan[bi].xySum_c = cy.xySum;
    }
}

// ---- back: grid (tiles, channels of the launch) ----
__global__ __launch_bounds__(64) void psk_tile_back_kernel(const ChanPlan *__restrict__ plans, const uint32_t *__restrict__ list, uint32_t ch0,
                                                           const ChanState *__restrict__ states, const TileInfo *__restrict__ tiles,
                                                           const float2 *__restrict__ t_s, const float *__restrict__ t_est,
                                                           uint32_t tile0, uint32_t early)
{
    const int lane = threadIdx.x & 63;
    const uint32_t bi = list[blockIdx.y];
    const ChanPlan &p = plans[bi];
    // (early: the pipelined mode writes a range's outputs as soon as its estimates are there, before the call is committed --
    // a call that is handed over later on is redone from its first symbol, outputs included)
    if (!tile_plan_mine(p) || (!early && states[ch0 + bi].guard != kGuardTiled))
        return;
    const int n_out = (int)p.n_out;
    const int n_blocks = (n_out + kB - 1) / kB;
    const int c_begin = (int)((blockIdx.x + tile0) * p.tile_blocks);
    if (c_begin >= n_blocks)
        return;
    const int c_end = c_begin + (int)p.tile_blocks < n_blocks ? c_begin + (int)p.tile_blocks : n_blocks;
    const uint32_t M = p.M;
    const bool m_pow2 = M != 0 && (M & (M - 1)) == 0;
    const float inv_M = uni(1.0f / (float)(M ? M : 1));
    const AtanTabDev atab = atan_tab_dev(lane);
    const bool qpsk_sign_map = (p.lf_flags & PLAN_QPSK_SIGN_MAP) != 0;
    const float2 *s_row = t_s + p.tile_off;
    const float *est_row = t_est + p.tile_off;
    for (int c = c_begin; c < c_end; c++) {
        const int i0 = c * kB + 2 * lane;
        const bool valid[kR] = {i0 < n_out, i0 + 1 < n_out};
        const float4 sv = *reinterpret_cast<const float4 *>(s_row + i0);
        const float2 ev = *reinterpret_cast<const float2 *>(est_row + i0);
        cf32 s[kR];
        s[0].re = sv.x, s[0].im = sv.y, s[1].re = sv.z, s[1].im = sv.w;
        const float est[kR] = {ev.x, ev.y};
        cf32 last_c;
        last_c.re = last_c.im = 0.0f;
        if (p.diff) {
            if (c == 0) {
                last_c.re = tiles[p.tile_base].last0_re;
                last_c.im = tiles[p.tile_base].last0_im;
            } else {
                const float2 l = s_row[c * kB - 1];
                last_c.re = l.x, last_c.im = l.y;
            }
        }
        // (complex products with libgcc's recovery throughout: a sample that needs it was refused by the front kernel
        // unless differential decoding divides by a silent one)
        output_stage<true, true>(p, c, i0, valid, s, est, last_c, atab, qpsk_sign_map, m_pow2, inv_M);
    }
}

#define PSK_TDECL(S) hipError_t launch_tile_front_S##S##_H1(PSK_TILE_FRONT_ARGS);
PSK_TDECL(2) PSK_TDECL(3) PSK_TDECL(4) PSK_TDECL(5) PSK_TDECL(6) PSK_TDECL(7) PSK_TDECL(8) PSK_TDECL(9)
PSK_TDECL(10) PSK_TDECL(11) PSK_TDECL(12) PSK_TDECL(13) PSK_TDECL(14) PSK_TDECL(15) PSK_TDECL(16)

// window classes the front kernel is built for: numAvg <= 128 (one block of history, exact re-decisions from the LDS
// energy ring), samplesPerBaud 2 .. 16
bool tile_front_has(int S, int H) { return H == 1 && S >= 2 && S <= 16; }

hipError_t launch_tile_front(int S, int H, PSK_TILE_FRONT_ARGS)
{
#define PSK_TCASE(Sv)       \
    if (S == Sv && H == 1) \
        return launch_tile_front_S##Sv##_H1(plans, list, ch0, nch, max_tiles, states, rings, ring_cap, r_len, tiles, t_raw, t_s, pf_chan, tile0, stream);
    PSK_TCASE(2) PSK_TCASE(3) PSK_TCASE(4) PSK_TCASE(5) PSK_TCASE(6) PSK_TCASE(7) PSK_TCASE(8) PSK_TCASE(9)
    PSK_TCASE(10) PSK_TCASE(11) PSK_TCASE(12) PSK_TCASE(13) PSK_TCASE(14) PSK_TCASE(15) PSK_TCASE(16)
    return hipErrorInvalidValue;
}

hipError_t launch_tile_front_any(const ChanPlan *plans, const uint32_t *list, uint32_t ch0, uint32_t nch, uint32_t max_tiles, uint32_t max_S,
                                 const ChanState *states, const float2 *rings, uint32_t ring_cap, TileInfo *tiles, float *t_raw, float2 *t_s,
                                 PfChan *pf_chan, hipStream_t stream)
{
    if (!nch || !max_tiles)
        return hipSuccess;
    // (the channels of a launch are the grid's y dimension, 65535 at most: a larger class goes out in slices of its list)
    for (uint32_t off = 0; off < nch; off += kGridYMax) {
        const uint32_t n = nch - off < kGridYMax ? nch - off : kGridYMax;
        if (max_S <= 64u)
            hipLaunchKernelGGL(psk_tile_front_any_kernel<1>, dim3(max_tiles, n), dim3(kWave), 0, stream, plans, list + off, ch0, states, rings,
                               ring_cap, tiles, t_raw, t_s, pf_chan);
        else if (max_S <= 256u)
            hipLaunchKernelGGL(psk_tile_front_any_kernel<4>, dim3(max_tiles, n), dim3(kWave), 0, stream, plans, list + off, ch0, states, rings,
                               ring_cap, tiles, t_raw, t_s, pf_chan);
        else
            hipLaunchKernelGGL(psk_tile_front_any_kernel<kAnyPhases>, dim3(max_tiles, n), dim3(kWave), 0, stream, plans, list + off, ch0, states,
                               rings, ring_cap, tiles, t_raw, t_s, pf_chan);
    }
    return hipGetLastError();
}

hipError_t launch_tile_fit(const ChanPlan *plans, const uint32_t *list, uint32_t ch0, uint32_t nch, ChanState *states, float2 *rings,
                           uint32_t ring_cap, float *yvs, uint32_t fit_cap, uint32_t y_len, TileInfo *tiles, const float *t_raw,
                           const float2 *t_s, float *t_est, const PfScratch &sc, hipStream_t stream)
{
    if (!nch)
        return hipSuccess;
    static LdsGrant granted;
    if (const hipError_t e = lds_grant(reinterpret_cast<const void *>(&psk_tile_fit_kernel), sizeof(float) * (size_t)y_len, granted))
        return e;
    hipLaunchKernelGGL(psk_tile_fit_kernel, dim3(nch), dim3(kWave), sizeof(float) * (size_t)y_len, stream, plans, list, ch0, states, rings,
                       ring_cap, yvs, fit_cap, y_len, tiles, t_raw, t_s, t_est, sc);
    return hipGetLastError();
}

// the parallel fit of a class (psk_pfit.h): nine launches between the front kernel and the block-by-block fit kernel, seven more
// with a second round
hipError_t launch_pfit(const ChanPlan *plans, const uint32_t *list, uint32_t ch0, uint32_t nch, uint32_t max_tiles, ChanState *states,
                       float2 *rings, uint32_t ring_cap, float *yvs, uint32_t fit_cap, uint32_t y_len, TileInfo *tiles, const float *t_raw,
                       const float2 *t_s, float *t_est, const PfScratch &sc, bool second_round, hipStream_t stream)
{
    if (!nch || !max_tiles)
        return hipSuccess;
    const dim3 wave(kWave);
    static LdsGrant granted_b;
    if (const hipError_t e = lds_grant(reinterpret_cast<const void *>(&pf_begin_kernel), sizeof(float) * (size_t)y_len, granted_b))
        return e;
    // (channels are the grid's y dimension: slices of 65535; the channels of different slices have nothing to do with one another)
    for (uint32_t off = 0; off < nch; off += kGridYMax) {
        const uint32_t n = nch - off < kGridYMax ? nch - off : kGridYMax;
        const uint32_t *const l = list + off;
        const dim3 grid(max_tiles, n);
        hipLaunchKernelGGL(pf_begin_kernel, dim3(n), wave, sizeof(float) * (size_t)y_len, stream, plans, l, ch0, states, yvs, fit_cap, y_len, sc);
        hipLaunchKernelGGL(pf_unwrap_kernel, grid, wave, 0, stream, plans, l, ch0, states, t_raw, sc);
        for (int round = 0; round <= (second_round ? 1 : 0); round++) {
            if (round)
                hipLaunchKernelGGL(pf_retry_kernel, dim3(n), wave, 0, stream, plans, l, sc);
            hipLaunchKernelGGL(pf_y_kernel, grid, wave, 0, stream, plans, l, ch0, states, t_raw, sc, round);
            hipLaunchKernelGGL(pf_ydiff_kernel, grid, wave, 0, stream, plans, l, ch0, yvs, fit_cap, sc, round);
            hipLaunchKernelGGL(pf_ysum_kernel, grid, wave, 0, stream, plans, l, ch0, states, yvs, fit_cap, sc, round);
            hipLaunchKernelGGL(pf_xblock_kernel, grid, wave, 0, stream, plans, l, ch0, states, sc, round);
            hipLaunchKernelGGL(pf_xwalk_kernel, dim3(n), wave, 0, stream, plans, l, sc.blk, sc, round);
            hipLaunchKernelGGL(pf_verify_kernel, grid, wave, 0, stream, plans, l, ch0, states, yvs, fit_cap, t_raw, t_est, sc, round);
        }
    }
    return hipGetLastError();
}

hipError_t launch_tile_back(const ChanPlan *plans, const uint32_t *list, uint32_t ch0, uint32_t nch, uint32_t max_tiles,
                            const ChanState *states, const TileInfo *tiles, const float2 *t_s, const float *t_est, uint32_t tile0,
                            uint32_t early, hipStream_t stream)
{
    if (!nch || !max_tiles)
        return hipSuccess;
    for (uint32_t off = 0; off < nch; off += kGridYMax) {
        const uint32_t n = nch - off < kGridYMax ? nch - off : kGridYMax;
        hipLaunchKernelGGL(psk_tile_back_kernel, dim3(max_tiles, n), dim3(kWave), 0, stream, plans, list + off, ch0, states, tiles, t_s, t_est, tile0, early);
    }
    return hipGetLastError();
}

}  // namespace psk
