// psk_tile_kernel.h -- the time-tiled kernels: a call of FEW channels and MANY symbols.
//
// The wave-scan kernel gives one wave to a channel for the whole call: a batch of a few dozen channels leaves the
// machine idle however long the packets are (one channel: 0.2 Gsamples/s).  Most of the work per symbol does not
// depend on the symbols before it -- what the reference carries from symbol to symbol (cpp/psk_soft.h:66-86) is the
// energy window (numAvg symbols back) and the unwrap / LinearFit feedback:
//
//   front  (tiles x channels)   timing recovery, the picked sample, its M-th power and raw phase.  A tile of K blocks
//                               of 128 symbols rebuilds its energy window from the numAvg - 1 symbols in front of it
//                               (the halo: (numAvg-1)*samplesPerBaud samples read twice), exactly as a call rebuilds
//                               it from the carried samples (resyncEnergy, cpp/psk_soft.cpp:619-636).  It is the loop
//                               of the screened wave-scan kernel stopped after the raw phase (fast_main_loop<FRONT>).
//                               Out: sampleIndex to the caller, picked samples (8 B) and raw phases (4 B) to scratch.
//   fit    (one wave a channel) feedback unwrap + LinearFit::next over the raw phases, block after block: the serial
//                               part, identical to the wave-scan kernel's (fit_stage); call prologue and epilogue
//                               (LinearFit history, end-of-call wrap, state commit) around it.  Out: the phase
//                               estimates (4 B a symbol) to scratch.
//   back   (tiles x channels)   de-rotation, slicing, the soft / phase / bits streams (output_stage).
//
// Exactness is argued as in the wave-scan kernel; two bounds there scale with the largest window sum met SO FAR in the
// call, which a tile does not know.  Each tile reports what it can vouch for (TileInfo: gap_rel, cap, wmax, the
// exponent range of the energies it summed exactly) and the fit kernel, which sees all tiles of its channel, decides
// with the maximum over the whole call (at least as strict).  A call the tiles cannot carry -- a near-tie the
// exactness guard does not cover, a non-finite sample, an unwrap that does not settle -- is handed to the wave-scan
// kernels launched behind (ChanState::guard = 1; nothing committed), like the screened one hands over to the exact one.
#ifndef PSK_TILE_KERNEL_H
#define PSK_TILE_KERNEL_H

#include "psk_fast_kernel.h"

namespace psk {

constexpr uint32_t kGuardTiled = 4u;  // ChanState::guard: the time-tiled kernels finished this call
constexpr uint32_t kGridYMax = 65535u;  // channels of one launch of the kernels whose grid is (tiles, channels)

PSK_DEV bool tile_plan_mine(const ChanPlan &p) { return p.mode == PLAN_FAST && (p.lf_flags & PLAN_TILED) && p.n_out != 0; }

// ---- front: grid (tiles, channels of the launch) ----
template <int SV, int HV>
__global__ __launch_bounds__(64) void psk_tile_front_kernel(const ChanPlan *__restrict__ plans, const uint32_t *__restrict__ list, uint32_t ch0,
                                                            const ChanState *__restrict__ states, const float2 *__restrict__ rings,
                                                            uint32_t ring_cap, uint32_t r_len, TileInfo *__restrict__ tiles,
                                                            float *__restrict__ t_raw, float2 *__restrict__ t_s, PfChan *__restrict__ pf_chan,
                                                            uint32_t tile0)
{
    extern __shared__ __attribute__((aligned(16))) float lds_dyn[];
    constexpr bool kDyn = ering_dynamic(SV);
    ERingT<kDyn> er;
    if constexpr (kDyn) {
        er.mem = lds_dyn;
        er.set_len((int)r_len);
    } else {
        __shared__ __attribute__((aligned(16))) float ering_s[HV == 1 ? SV * kERing : 4];
        er.mem = ering_s;
    }
    const int lane = threadIdx.x & 63;
    const uint32_t bi = list[blockIdx.y];
    const ChanPlan &p = plans[bi];
    if (!tile_plan_mine(p) || p.S != (uint32_t)SV || hist_blocks_for(p.A) != HV)
        return;
    const int n_blocks = (int)((p.n_out + kB - 1) / kB);
    const uint32_t tile = blockIdx.x + tile0;  // (tile0: the launch covers a range of tiles -- the pipelined mode, psk_capi.cpp)
    const int c_begin = (int)(tile * p.tile_blocks);
    if (c_begin >= n_blocks)
        return;
    if (tile == 0 && lane == 0)  // the call's entry in the parallel fit's bookkeeping (psk_pfit.h: PfChan) starts clean
        pf_chan[bi].fail = pf_chan[bi].done = pf_chan[bi].slow_blocks = pf_chan[bi].retry = 0u;
    const int c_end = c_begin + (int)p.tile_blocks < n_blocks ? c_begin + (int)p.tile_blocks : n_blocks;
    const uint32_t ch = ch0 + bi;
    const float2 *ring_src = rings + ((size_t)ch * 2u + p.ring_src) * ring_cap;

    XView X;
    X.ring = reinterpret_cast<const f2g *>(ring_src);
    X.in = reinterpret_cast<const f2g *>(p.in);
    X.L0 = p.ring_len0;

    FastCarry cy;
    cy.last_k = states[ch].last_k < p.S ? states[ch].last_k : 0u;  // (a prediction seed only)
    cy.umax = 0u;
    cy.umin1 = 0xFFFFFFFFu;
    cy.wmax = 0.0f;
    cy.ambiguous = false;
    cy.refuse = false;
    cy.stat_blocks = 0;
    cy.stat_exact_blocks = 0;
    cy.gap_rel = __builtin_inff();
    cy.cap = __builtin_inff();
    cy.emax = 0.0f;
    // a window sum of the call can reach numAvg times its largest sample energy; the last tiled call's stands in for this one's
    // (a call that outgrows it by four decades is handed over by the fit kernel's fold, TileInfo::cap)
    const float hint = states[ch].emax_hint;
    const float wmax_floor = (hint > 0.0f && hint < 3.0e38f) ? hint * (float)p.A * 1.000001f : 0.0f;
    fast_main_loop<SV, HV, false, true>(p, X, nullptr, 0u, er, cy, c_begin, c_end, t_raw + p.tile_off, t_s + p.tile_off, wmax_floor);

    const unsigned umax = wave_max_u32(cy.umax), umin1 = wave_min_u32(cy.umin1);
    // (gap_rel and cap are non-negative or +inf: their bit patterns order like the values)
    const unsigned gap_b = wave_min_u32(__float_as_uint(cy.gap_rel)), cap_b = wave_min_u32(__float_as_uint(cy.cap));
    const bool refuse = __any(cy.refuse);
    const float emax = wave_max_f32(__builtin_fmaxf(cy.emax, 0.0f));
    if (lane == 0) {
        TileInfo &t = tiles[p.tile_base + tile];
        t.umax = umax;
        t.umin1 = umin1;
        t.refuse = refuse ? 1u : 0u;
        t.gap_rel = __uint_as_float(gap_b);
        t.wmax = cy.wmax;
        t.stat_exact = cy.stat_exact_blocks;
        t.last_k = cy.last_k;
        t.cap = __uint_as_float(cap_b);
        t.emax = emax;
    }
}

// (the fit and back kernels do not depend on samplesPerBaud: psk_tile.hip)

#define PSK_TILE_FRONT_ARGS                                                                                                    \
    const ChanPlan *plans, const uint32_t *list, uint32_t ch0, uint32_t nch, uint32_t max_tiles, const ChanState *states,       \
        const float2 *rings, uint32_t ring_cap, uint32_t r_len, TileInfo *tiles, float *t_raw, float2 *t_s, PfChan *pf_chan,          \
        uint32_t tile0, hipStream_t stream

template <int SV, int HV>
hipError_t launch_tile_front_inst(PSK_TILE_FRONT_ARGS)
{
    if (!nch || !max_tiles)
        return hipSuccess;
    const size_t lds_bytes = sizeof(float) * (ering_dynamic(SV) ? (size_t)SV * r_len : 0);
    for (uint32_t off = 0; off < nch; off += kGridYMax) {  // (channels are the grid's y dimension: slices of 65535)
        const uint32_t n = nch - off < kGridYMax ? nch - off : kGridYMax;
        hipLaunchKernelGGL((psk_tile_front_kernel<SV, HV>), dim3(max_tiles, n), dim3(kWave), lds_bytes, stream, plans, list + off, ch0, states,
                           rings, ring_cap, r_len, tiles, t_raw, t_s, pf_chan, tile0);
    }
    return hipGetLastError();
}

}  // namespace psk
#endif
