// psk_plan.h -- the call plan handed from the host control plane to the device kernels.
//
// Everything in serviceFunction() that steers the loop but does not depend on sample
// VALUES -- property snapshot, reset flags, deque lengths, symbol clock, how many symbols
// the call emits (reference cpp/psk_soft.cpp:365-426, 454-457, 568-590) -- is resolved
// on the host (psk_ctl.h) and frozen into one ChanPlan per channel per call.  The
// kernels (psk_kernels.hip) do the arithmetic.  Plain PODs shared by host and device.
#ifndef PSK_PLAN_H
#define PSK_PLAN_H

#include <stdint.h>

namespace psk {

enum PlanMode : uint32_t {
    PLAN_SKIP = 0,    // no packet / real data / nothing to do on the device
    PLAN_FAST = 1,    // wave-scan kernel
    PLAN_SEQ = 2,     // reference-order kernel, regular window mode (samplesPerBaud > 1)
    PLAN_SEQ_S1 = 3,  // reference-order kernel, samplesPerBaud == 1 emitting (numAvg == 0)
};

enum LfFlags : uint32_t {
    LF_RECOMPUTE = 1u,  // LinearFit::reset() ran: sums rebuilt from yvals, count = 0 (cpp/psk_soft.cpp:110-122)
    // (not a LinearFit flag; it rides in the same word) opt-in QPSK bit map by the SIGNS of the
    // de-rotated symbol, as the diagram at cpp/psk_soft.cpp:516-521 describes, instead of the
    // float->bool conversions of :523-526 that make every bit 0 (quirk Q1)
    PLAN_QPSK_SIGN_MAP = 2u,
    // (not a LinearFit flag either) the call goes through the time-tiled kernels first (psk_tile_kernel.h): the
    // wave-scan kernels behind them only pick it up if those hand it over (ChanState::guard == 1)
    PLAN_TILED = 4u,
    // ... and, of those, the call's feedback unwrap and fit are attempted in parallel along time (psk_pfit.h): the fit
    // window is full at the start of the call
    PLAN_PFIT = 8u,
    // a window class the wave-scan kernels have no instantiation for (samplesPerBaud > 32, numAvg > 1024): always through
    // the time-tiled kernels, behind the front stage that takes samplesPerBaud and numAvg at run time (psk_tile.hip:
    // psk_tile_front_any_kernel); what that hands over is the reference-order kernel's
    PLAN_ANYFRONT = 16u,
    // a piece of a call the library has cut (more than 2^20 symbols: psk_capi.cpp) that is not the call's last: the end-of-call
    // wrap of the phase estimate (cpp/psk_soft.cpp:592-603) belongs to the end of the CALL
    PLAN_NO_WRAP = 32u,
    // a piece of a call the library has cut where the reference does NOT rebuild its energy sums (psk_capi.cpp: the classes of a
    // mixed batch in pieces): the rounding the reference's running sums have gathered since the call began -- ChanPlan::count0
    // symbols ago, at the scale of the largest window sum met since, carried in ChanState::pad_state -- counts for the bounds
    // of this piece too (drift_bound, psk_fast_loop.h)
    PLAN_CARRY_DRIFT = 64u,
    // (tests, A/B runs: PSK_SOFT_TIES_IN_PLACE=0) windows longer than a block hand a call with a near-tie to the exact tier, as
    // before round 3, instead of settling the block in place
    PLAN_TIES_HANDOVER = 128u,
};

constexpr uint32_t kResyncCount = 1048576u;  // cpp/psk_soft.cpp:51, 582

// wave-scan kernel, numAvg <= 128: the instantiations whose LDS energy ring is sized by the host per
// launch (psk_fast_loop.h) instead of the fixed 256 positions -- the host sizes their phase ring tighter too
constexpr bool ering_dynamic(int S) { return S == 9 || S == 10; }

struct ChanPlan {
    // data (device pointers)
    const float *in;   // packet: interleaved I,Q
    float *soft;
    int16_t *bits;
    float *phase;
    int16_t *sidx;
    uint64_t n_in;     // complex samples in the packet
    uint64_t n_out;    // symbols this call emits
    // geometry
    uint32_t mode;     // PlanMode
    uint32_t S;        // samplesPerBaud snapshot       (cpp/psk_soft.cpp:376)
    uint32_t A;        // numAvg snapshot               (:377)
    uint32_t M;        // constelationSize snapshot     (:378)
    uint32_t bpb;      // bitsPerBaud                   (:384-390)
    uint32_t diff;     // differentialDecoding
    uint32_t ring_len0;  // samples.size() after the prologue (after resyncEnergy's trim), device-resident
    uint32_t ring_len1;  // samples.size() at return (clipped to the ring capacity)
    uint32_t ring_src;   // which of the two ring buffers holds the samples now
    // LinearFit control state (host-mirrored)
    uint32_t lf_flags;
    uint32_t lf_n;       // LinearFit::n after the prologue
    uint32_t lf_head;    // circular yvals buffer: index of the oldest value
    uint32_t lf_len0;    // yvals.size() after the prologue
    uint32_t lf_count0;  // LinearFit::count at the first next()
    uint32_t count0;     // psk_soft_i::count at loop start
    float lf_xdelta;     // LinearFit::xdelta after the prologue
    // time-tiled kernels (PLAN_TILED): where this channel's symbols and tiles sit in the scratch of the call
    uint32_t tile_blocks;  // 128-symbol blocks per tile
    uint32_t tile_base;    // index of the channel's first TileInfo
    uint32_t tile_pad;
    uint64_t tile_off;     // offset of its first symbol in the raw-phase / picked-sample / estimate arrays
    // word 0 of the header in front of the call's plans in device memory (plan_header() below): the kernel that hands this
    // channel's call over counts it there.  (In every plan, although every kernel could find the header from its `plans`
    // argument: holding that argument across the symbol loop for the one refusal in thousands of calls cost the headline
    // instantiation eleven more spilled scalar registers and 10 % of its speed.)
    uint32_t *handed_over;
};

// The plans of a call sit behind a small header in device memory (uploaded with them, one copy): word 0 counts the channels
// of the call that a kernel has handed over so far (ChanState::guard = 1) -- zero at upload, so the exact-timing and the
// reference-order launches of a call in which nothing was handed over end at their first instruction instead of reading plan
// and state of every channel to find that out (4.8 + 4.2 us of a 4096-channel call) --, word 1 says whether the host planned any
// channel for the reference-order kernel.  The kernels find the header in front of `plans`.
constexpr unsigned kPlanHeaderBytes = 128u;
#if defined(__HIPCC__) || defined(__CUDACC__)
__host__ __device__
#endif
inline uint32_t *plan_header(const ChanPlan *plans)
{
    return reinterpret_cast<uint32_t *>(const_cast<char *>(reinterpret_cast<const char *>(plans)) - kPlanHeaderBytes);
}

// What one tile of the time-tiled front kernel reports (psk_tile_kernel.h): the fit kernel, one wave per channel,
// folds the tiles of its channel together and decides for the whole call.
struct TileInfo {
    uint32_t umax, umin1;  // exactness guard over the energies the tile summed exactly (FastCarry)
    uint32_t refuse;       // the tile met something the screened timing does not carry
    float gap_rel;         // smallest (best - runner-up) / (relative rounding bound) among its exact re-decisions
    float wmax;            // largest window sum it saw
    uint32_t stat_exact;   // blocks re-decided exactly
    uint32_t last_k;       // timing index of its last symbol (the last tile's is the channel's)
    float cap;             // largest sum of the whole call its screening thresholds allow for (FastCarry::cap)
    float last0_re, last0_im;  // (tile 0, written by the fit kernel) psk_soft_i::last at the start of the call
    float emax;                // largest sample energy the tile loaded
    uint32_t pad2;
};

// Bookkeeping of the parallel fit (psk_pfit.h), in the scratch of the call next to the TileInfo records.
struct PfTile {  // per tile
    int jsum;     // numWraps increments of the tile
    int pad;
    double dsum;  // y[t] - y[t-n] over the tile
    double dlast; // the tile's prefix sum of those at its last valid position (what pf_ysum turns into the tile's last ySum)
    double xsum;  // term - c over the tile (plain double: only predicts the binade)
};
struct PfBlock {  // per 128-symbol block: what pf_xblock prepares for the walker
    double bsum;      // term - c over the block (plain double: predicts the binade)
    double a, b;      // the carried xySum s for which the block's transfer holds: a < s < b (empty: the walker runs the block)
    double kmul;      // s * kmul has a fractional part iff s, in units of the block's q (mode B: of 2q), is odd
    double D0s, D1s;  // what the block adds to s for an even / odd carried sum
    int flags;        // bit 0 / 1: adding D0s / D1s flips the parity; bit 2: same q and mode as the block before, so the
    int pad;          // parity carries over from it and need not be taken from s again
};
struct PfWalk {  // per 128-symbol block: what the walker found
    double s_in;  // xySum carried into the block
    int slow;     // the block's positions were produced by the walker itself: PfScratch::xs holds them
    int pad;
};
struct PfChan {  // per channel of the call
    uint32_t fail, done, slow_blocks;
    uint32_t retry;  // the first guess of the unwrap counts failed and nothing else did: a second round runs on corrected counts
    double ySum_c, xySum_c;  // LinearFit's sums at the call's first next(): carried, or rebuilt where reset() ran (pf_begin)
};
struct PfScratch {
    int *k;       // numWraps (tile-local prefix, then absolute)
    float *y;     // unwrapped phases
    double *S;    // ySum after each symbol (tile-local prefix of the differences first)
    double *c;    // fl(xdelta * ySum') per symbol
    float *tt;    // the float term of :78
    double *xs;   // xySum after each symbol
    PfTile *tile;
    PfBlock *blk;
    PfWalk *walk;
    PfChan *chan;
    uint32_t *hint;  // page-locked host word: set when a call's first guess of the unwrap counts failed (the host then enqueues second rounds)
};

// Data-dependent per-channel state that lives in HBM between calls.
struct ChanState {
    double lf_ySum;       // LinearFit::ySum
    double lf_xySum;      // LinearFit::xySum
    float last_re;        // psk_soft_i::last
    float last_im;
    float phaseEstimate;  // psk_soft_i::phaseEstimate
    float lf_den;         // LinearFit::denominator
    float lf_xavg;        // LinearFit::xAvg
    float lf_m;           // LinearFit::m, b (write-only in the reference; kept for completeness)
    float lf_b;
    uint32_t guard;       // set by the wave-scan kernel when it hands the call to the reference-order kernel
    // statistics of the last call
    uint32_t stat_blocks;
    uint32_t stat_extra;
    uint32_t last_k;      // timing index of the last emitted symbol (prediction seed of the wave-scan kernel)
    uint32_t stat_exact;  // blocks whose timing argmax needed the exact double-precision pass
    uint32_t stat_chain;  // blocks whose LinearFit sums were redone by the reference-order chain (psk_fast_loop.h)
    float emax_hint;      // (time-tiled kernels) largest sample energy of the channel's last tiled call: scales the screening
                          // thresholds of the next one's tiles, which do not know the largest window sum of their call
    uint32_t pad_state;
    uint32_t stat_pfit;   // bit 0: the call's unwrap and fit were done in parallel along time (psk_pfit.h), bit 8: in the second
                          // round; else 2 * (why not: PfChan::fail)
};

}  // namespace psk
#endif
