// psk_capi.cpp -- the C ABI of libpsk_soft_hip.so (include/psk_soft_hip.h).
//
// Host side of the drop-in boundary: per-channel control plane (psk_ctl.h, mirrors the
// non-data state of psk_soft_i, reference cpp/psk_soft.cpp:353-426), HBM-resident channel
// state, plan upload and kernel launches.  There is no CPU compute path here: without a
// usable GPU psk_soft_create() fails (PSK_SOFT_ERR_NO_DEVICE) unless the caller explicitly
// asks for a control-plane-only handle (PSK_SOFT_DEVICE_NONE), which never touches data.
#include <hip/hip_runtime_api.h>
#include <hip/hip_vector_types.h>

#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "psk_ctl.h"
#include "psk_plan.h"
#include "psk_soft_hip.h"

namespace psk {
hipError_t launch_fast(int S, int H, int exact, const ChanPlan *plans, const uint32_t *list, uint32_t ch0, uint32_t nch, ChanState *states,
                       float2 *rings, uint32_t ring_cap, float *yvs, uint32_t fit_cap, uint32_t y_len, uint32_t r_len,
                       hipStream_t stream);
hipError_t launch_seq(const ChanPlan *plans, const uint32_t *list, uint32_t ch0, uint32_t nch, ChanState *states, float2 *rings,
                      uint32_t ring_cap, float *yvs, uint32_t fit_cap, hipStream_t stream);
hipError_t launch_read_probe(const void *src, uint64_t bytes, float *sink, hipStream_t stream);
// time-tiled kernels (psk_tile.hip)
bool tile_front_has(int S, int H);
hipError_t launch_tile_front(int S, int H, const ChanPlan *plans, const uint32_t *list, uint32_t ch0, uint32_t nch, uint32_t max_tiles,
                             const ChanState *states, const float2 *rings, uint32_t ring_cap, uint32_t r_len, TileInfo *tiles, float *t_raw,
                             float2 *t_s, PfChan *pf_chan, uint32_t tile0, hipStream_t stream);
hipError_t launch_tile_front_any(const ChanPlan *plans, const uint32_t *list, uint32_t ch0, uint32_t nch, uint32_t max_tiles, uint32_t max_S,
                                 const ChanState *states, const float2 *rings, uint32_t ring_cap, TileInfo *tiles, float *t_raw, float2 *t_s,
                                 PfChan *pf_chan, hipStream_t stream);
hipError_t launch_tile_fit(const ChanPlan *plans, const uint32_t *list, uint32_t ch0, uint32_t nch, ChanState *states, float2 *rings,
                           uint32_t ring_cap, float *yvs, uint32_t fit_cap, uint32_t y_len, TileInfo *tiles, const float *t_raw,
                           const float2 *t_s, float *t_est, const PfScratch &sc, hipStream_t stream);
hipError_t launch_pfit(const ChanPlan *plans, const uint32_t *list, uint32_t ch0, uint32_t nch, uint32_t max_tiles, ChanState *states,
                       float2 *rings, uint32_t ring_cap, float *yvs, uint32_t fit_cap, uint32_t y_len, TileInfo *tiles, const float *t_raw,
                       const float2 *t_s, float *t_est, const PfScratch &sc, bool second_round, hipStream_t stream);
hipError_t launch_tile_fit_range(const ChanPlan *plans, const uint32_t *list, uint32_t ch0, uint32_t nch, ChanState *states, float2 *rings,
                                 uint32_t ring_cap, float *yvs, uint32_t fit_cap, uint32_t y_len, TileInfo *tiles, const float *t_raw,
                                 const float2 *t_s, float *t_est, void *carry, float *carry_y, uint32_t tile0, uint32_t ntiles,
                                 hipStream_t stream);
size_t pipe_carry_bytes();
hipError_t launch_tile_back(const ChanPlan *plans, const uint32_t *list, uint32_t ch0, uint32_t nch, uint32_t max_tiles,
                            const ChanState *states, const TileInfo *tiles, const float2 *t_s, const float *t_est, uint32_t tile0,
                            uint32_t early, hipStream_t stream);
}  // namespace psk

namespace {

thread_local std::string g_last_error;
thread_local bool g_long_call = false;  // process_round's note to psk_soft_process_device: plan the call in pieces
// ... and: the batch mixes window classes that cannot be resident together -- cut every channel's call into g_split_pieces pieces in
// time and let the classes run through them on their own streams, joined once at the end of the call (see process_device)
thread_local int g_split_pieces = 0;
thread_local bool g_split_mode = false;  // the rounds of such a call: their side streams are not joined in between

psk_soft_status fail(psk_soft_status st, const std::string &msg)
{
    g_last_error = msg;
    return st;
}

#define PSK_HIP(call)                                                                             \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return fail(PSK_SOFT_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));     \
    } while (0)

constexpr int kPlanSlots = 4;
constexpr int kAuxStreams = 3;  // side streams for the launches of a batch that mixes window classes (see psk_soft_process_device)
constexpr int kStageSlots = 3;  // chunks of the host-buffer path in flight (< kPlanSlots)

// Minimal fork-join pool for the host-buffer path: packing packets into pinned memory and
// unpacking results are plain memcpy work that one thread cannot do at PCIe rate.
class CopyPool {
public:
    explicit CopyPool(int n_threads)
    {
        for (int t = 0; t < n_threads; t++) workers_.emplace_back([this] { loop(); });
    }
    ~CopyPool()
    {
        {
            std::lock_guard<std::mutex> g(m_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto &w : workers_) w.join();
    }
    // fn(i) for i in [0, n); the caller takes part; returns when all are done
    void run(uint32_t n, const std::function<void(uint32_t)> &fn)
    {
        if (!n)
            return;
        if (workers_.empty() || n == 1) {
            for (uint32_t i = 0; i < n; i++) fn(i);
            return;
        }
        {
            std::lock_guard<std::mutex> g(m_);
            fn_ = &fn;
            n_ = n;
            next_.store(0);
            busy_ = (int)workers_.size();
            gen_++;
        }
        cv_.notify_all();
        work();
        std::unique_lock<std::mutex> g(m_);
        done_.wait(g, [this] { return busy_ == 0; });
        fn_ = nullptr;
    }

private:
    void work()
    {
        for (;;) {
            uint32_t i = next_.fetch_add(1);
            if (i >= n_)
                break;
            (*fn_)(i);
        }
    }
    void loop()
    {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> g(m_);
                cv_.wait(g, [&] { return stop_ || gen_ != seen; });
                if (stop_)
                    return;
                seen = gen_;
            }
            work();
            {
                std::lock_guard<std::mutex> g(m_);
                if (--busy_ == 0)
                    done_.notify_all();
            }
        }
    }
    std::vector<std::thread> workers_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    const std::function<void(uint32_t)> *fn_ = nullptr;
    uint32_t n_ = 0;
    std::atomic<uint32_t> next_{0};
    int busy_ = 0;
    uint64_t gen_ = 0;
    bool stop_ = false;
};

// What the control-plane pass over a batch found out: the first refusal, if any, and which kernels
// the accepted plans need.
struct PlanSummary {
    psk_soft_status st = PSK_SOFT_OK;
    uint32_t bad = 0;  // first refused channel (index into the batch)
    int why = 0;       // 0: status of plan_call, 1: samplesPerBaud > 1024, 2: alignment
    bool any = false, any_emit = false, any_seq = false, any_quiet = false;
    bool long_call = false;  // some channel's call is planned for the reference-order kernel only because of its length
    bool need_SH[33][17] = {};
    uint32_t cnt_SH[33][17] = {}, cnt_quiet = 0;  // channels per launch: each launch gets a compact list of its own
    // LDS rings of a launch are sized for the largest phaseAvg / numAvg among its channels: a ring of
    // y_len unwrapped phases (a power of two >= phaseAvg + 128) and, for numAvg <= 128, an energy
    // ring of r_len positions (even, >= numAvg + 128)
    uint32_t max_n[33][17] = {}, max_A[33][17] = {};
    uint32_t max_n_quiet = 0;  // ... and of the channels that emit nothing this call
    // time-tiled kernels: 128-symbol blocks of the class, in all and of its longest call
    uint64_t blocks_SH[33][17] = {};
    uint32_t max_blocks_SH[33][17] = {};
    // the window classes without an instantiation (PLAN_ANYFRONT), one launch set for all of them
    uint32_t cnt_any = 0, max_n_any = 0, max_A_any = 0, max_blocks_any = 0, max_S_any = 0;
    uint64_t blocks_any = 0;
};

// One chunk of channels of the host-buffer path in flight: pinned and device buffers for the packed
// packets and the packed four output streams, its own stream (so that the upload of one chunk, the
// kernels of another and the download of a third overlap), and an event for "outputs are in host
// memory".
struct StageSlot {
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;
    uint8_t *h_buf = nullptr;  // pinned: [in | soft | phase | bits | sidx]
    uint8_t *d_buf = nullptr;
    size_t in_cap = 0;         // bytes of the input region; the others follow as IN, IN/2, IN, IN/4
    bool busy = false;
    // what the chunk in flight needs for unpacking
    uint32_t ch0 = 0, nch = 0;
    std::vector<size_t> off_soft, off_phase, off_bits, off_sidx;
    size_t soft_bytes = 0, phase_bytes = 0, bits_bytes = 0, sidx_bytes = 0;
};
inline size_t region_soft(size_t in_cap) { return in_cap; }
inline size_t region_phase(size_t in_cap) { return in_cap + in_cap; }
inline size_t region_bits(size_t in_cap) { return in_cap + in_cap + in_cap / 2; }
inline size_t region_sidx(size_t in_cap) { return in_cap + in_cap + in_cap / 2 + in_cap; }
inline size_t region_total(size_t in_cap) { return in_cap + in_cap + in_cap / 2 + in_cap + in_cap / 4; }
inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
// Largest phaseAvg of the wave-scan kernels: their LDS ring of unwrapped phases holds phaseAvg + 128 values in a power
// of two; 32768 floats (128 KiB) leave room for the energy ring next to it.  Channels with phaseAvg > kDeepFit are
// launched apart from the others of their window class ("deep" classes, index H + 8): a ring that size allows one wave
// per CU, and sized for the whole launch it would take the residency of thousands of ordinary channels with it.
constexpr uint32_t kFastFitMax = 32768 - 128;
constexpr uint32_t kDeepFit = 2048 - 128;
const int kClassH[] = {1, 2, 4, 8, 9, 10, 12, 16};  // second index of the per-class tables: history blocks (+ 8: deep fit window)
inline int class_H(int Hi) { return Hi > 8 ? Hi - 8 : Hi; }
// time-tiled kernels, automatic choice (measured, tools/tiled_sweep2.sh: QPSK, samplesPerBaud 8): a class of at most 64
// channels whose longest call has at least 16 blocks of 128 symbols, or of at most 512 channels and 192 blocks (at 128
// blocks the two paths are level there; above 512 channels the wave-scan kernels fill the machine by themselves);
// tiles of 2 .. 16 blocks, as many as make kTiledTargetTiles tiles
constexpr uint32_t kTiledFewChannels = 64, kTiledMinBlocksFew = 16, kTiledMaxChannels = 512, kTiledMinBlocks = 192;
constexpr uint64_t kTiledTargetTiles = 4096;
// the pipelined mode of the time-tiled path (front / fit / back of consecutive ranges of tiles on three streams, psk_tile.hip:
// psk_tile_fit_range_kernel): window classes of a few hundred to a few thousand channels with long calls
constexpr uint32_t kPipeMinChannels = 288, kPipeMaxChannels = 1408, kPipeMinBlocks = 256, kPipeMaxRanges = 32, kPipeMaxYLen = 1024;
constexpr size_t kPipeMaxSymbols = (size_t)1 << 29;  // (16 bytes of scratch a symbol: 8 GiB)
constexpr int kPipeEvents = 2 * (int)kPipeMaxRanges + 2;
// a batch of several window classes whose calls are at least this long is cut in time (psk_soft_process_device)
constexpr uint32_t kSplitMinBlocks = 128;
constexpr uint32_t kSeqMaxS = 1024;    // symbolEnergy[] of the reference-order kernel lives in LDS
const int kFastS[] = {2,  3,  4,  5,  6,  7,  8,  9,  10, 11, 12, 13, 14, 15, 16, 17,
                      18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32};

}  // namespace

struct psk_soft_handle {
    int device = PSK_SOFT_DEVICE_NONE;
    bool dry = true;
    uint32_t nch = 0;
    psk::Limits lim{};
    psk_soft_limits_t user{};
    std::vector<psk::ChanCtl> ctl;
    std::vector<psk::ChanCtl> ctl_next;     // scratch of one call: planned on copies, committed on success
    std::vector<psk::ChanPlan> plans_dry;   // plans of a control-plane-only handle (no pinned slots)
    std::vector<uint32_t> last_mode;  // PlanMode of the last call, per channel (statistics)
    // Uniform run of the control plane (the stamped path of process_round): the channels [uni_lo, uni_hi) are known to hold
    // IDENTICAL control state.  While `lazy` is set that state lives in the run's ctl / mode alone and ctl[] / last_mode[] of
    // the range are stale: every reader goes through ctl_sync() first, every writer through ctl_touch().
    // A few such runs are kept side by side (disjoint): a caller that feeds a handle in two or four slices on as many streams --
    // so that the tail of one slice's launch overlaps the body of the next's -- keeps all of them stamped.
    struct UniRun {
        uint32_t lo = 0, hi = 0;
        bool lazy = false;
        psk::ChanCtl ctl;
        uint32_t mode = psk::PLAN_SKIP;
    };
    static constexpr int kUniRuns = 8;
    UniRun uni[kUniRuns];
    uint32_t mixed_lo = 0, mixed_hi = 0;  // a range that was compared and found mixed ...
    int mixed_ttl = 0;                    // ... is not compared again for this many calls (or until something is configured)
    int opt_stamp = 1;                    // PSK_SOFT_STAMP=0 (environment): every channel planned on its own (A/B runs, tests)
    // device memory
    psk::ChanState *d_state = nullptr;
    float2 *d_ring = nullptr;
    float *d_yv = nullptr;
    psk::ChanPlan *h_plans[kPlanSlots] = {};
    psk::ChanPlan *d_plans[kPlanSlots] = {};
    hipEvent_t ev[kPlanSlots] = {};
    // the plans of a call are uploaded on a stream of their own, so that the copy runs under the kernels of the call before
    // instead of behind them (20 us of a 4096-channel call); the caller's stream waits for ev_up of the slot
    hipStream_t up_stream = nullptr;
    hipEvent_t ev_up[kPlanSlots] = {};
    int opt_up_stream = 1;  // PSK_SOFT_PLAN_STREAM=0 (environment): upload on the caller's stream (A/B runs)
    bool ev_used[kPlanSlots] = {};
    int slot = 0;
    bool opt_qpsk_sign_map = false;  // PSK_SOFT_OPT_QPSK_SIGN_BITMAP
    hipStream_t stream = nullptr;
    hipStream_t aux[kAuxStreams] = {};     // created on first use
    hipEvent_t aux_fork = nullptr, aux_join[kAuxStreams] = {};
    int opt_fork = 1;                       // PSK_SOFT_OPT_CONCURRENT_CLASSES
    // PSK_SOFT_OPT_DEFERRED_JOIN: the side streams of a batch that mixes window classes are NOT joined into the caller's stream at
    // the end of the call -- every class ends its calls on its own stream (exact tier and reference-order hand-over included) and
    // the next call's launches of the class queue behind them there, so that a short class runs ahead into the next calls while a
    // long one is still busy.  The caller's stream sees the results after psk_soft_join() / psk_soft_synchronize().  Safe as long
    // as every channel keeps going to the same stream: the channel -> stream assignment of a call is hashed (deferred_sig), and
    // a call whose assignment differs from the pending one joins everything first.
    int opt_deferred = 0;
    bool deferred_pending = false;
    uint64_t deferred_sig = 0;
    uint32_t deferred_ch0 = 0, deferred_nch = 0;
    hipStream_t deferred_stream = nullptr;
    hipEvent_t slot_aux_ev[kPlanSlots][kAuxStreams] = {};  // end of a deferred call on each side stream (a plan slot is reused after them too)
    bool slot_aux_used[kPlanSlots][kAuxStreams] = {};
    // what the call that last used each plan slot worked on, and on which stream (its end is the slot's event)
    hipStream_t slot_stream[kPlanSlots] = {};
    uint32_t slot_ch0[kPlanSlots] = {}, slot_nch[kPlanSlots] = {};
    // time-tiled kernels (psk_tile_kernel.h): scratch of one call -- per-tile reports, and per symbol the raw phase, the
    // picked sample and the phase estimate -- grown on demand; calls that use it on different streams are ordered by tile_ev
    int opt_tiled = 1;  // PSK_SOFT_OPT_TIME_TILED
    psk::TileInfo *d_tiles = nullptr;
    float *d_traw = nullptr, *d_test = nullptr;
    float2 *d_ts = nullptr;
    size_t tile_cap = 0, tile_sym_cap = 0;
    size_t pf_cap = 0, pf_sym_cap = 0;  // ... of the parallel fit's arrays (the classes that use it come first in the scratch)
    psk::PfScratch pf{};   // ... and of the parallel fit (psk_pfit.h), same capacities; PfChan: one per channel of the handle
    int opt_pfit = 1;      // PSK_SOFT_PARALLEL_FIT (environment): 0 = time-tiled calls keep the block-by-block fit (A/B runs),
                           // 2 = the second round of the parallel fit is always enqueued (tests), 1 = for a while after
                           // a call reported a first guess that failed (pf.hint, a word the kernels write into page-locked memory)
    int pf_second_ttl = 0;  // tiled calls left with the second round enqueued
    int opt_ties_in_place = 1;              // PSK_SOFT_TIES_IN_PLACE=0 (environment): PLAN_TIES_HANDOVER in every plan (tests, A/B runs)
    int opt_trace = 0;                      // PSK_SOFT_TRACE_LAUNCHES=1 (environment, debugging): see `mark` in process_round
    int opt_validate = 0;                   // PSK_SOFT_VALIDATE=1 (environment, tests): see `validate` in process_round
    int opt_split = 2;                      // PSK_SOFT_SPLIT_CLASSES=n (environment): pieces a mixed batch's calls are cut into (0 / 1: never)
    int opt_pipe = 1;                       // PSK_SOFT_PIPELINED=0 (environment): never the pipelined mode (A/B runs)
    hipStream_t pipe_st[2] = {};            // its fit and back streams (the front stage stays on the class's stream)
    hipEvent_t pipe_ev[kPipeEvents] = {};
    void *d_pipe_carry = nullptr;           // per channel of a pipelined launch: the fit state between two ranges (PipeCarry)
    float *d_pipe_y = nullptr;              // ... and its ring of unwrapped phases
    size_t pipe_cap = 0;
    hipEvent_t tile_ev = nullptr;
    hipStream_t tile_stream = nullptr;  // stream of the last call that used the scratch
    bool tile_ev_used = false;
    bool poisoned = false;  // a HIP call failed after kernels of a call were enqueued: host mirror and device state may disagree
    // ingest pipeline of the host-buffer entry point (psk_soft_process_host)
    StageSlot stage[kStageSlots];
    CopyPool *pool = nullptr;
    size_t stage_bytes = 32u << 20;  // input bytes per chunk (PSK_SOFT_STAGE_MB)
};

namespace {
// the per-channel mirror of one run brought up to date (see psk_soft_handle::UniRun)
void run_sync(psk_soft_handle *h, psk_soft_handle::UniRun &r)
{
    if (!r.lazy)
        return;
    for (uint32_t i = r.lo; i < r.hi; i++) {
        h->ctl[i] = r.ctl;
        h->last_mode[i] = r.mode;
    }
    r.lazy = false;
}
// ... of the whole handle
void ctl_sync(const psk_soft_handle *hc)
{
    psk_soft_handle *h = const_cast<psk_soft_handle *>(hc);
    for (auto &r : h->uni) run_sync(h, r);
}
// ... and about to be changed channel by channel: nothing is known to be uniform any more
void ctl_touch(psk_soft_handle *h)
{
    for (auto &r : h->uni) {
        run_sync(h, r);
        r.lo = r.hi = 0;
    }
    h->mixed_lo = h->mixed_hi = 0;
    h->mixed_ttl = 0;
}
// ... of the channels [lo, hi), which are about to be planned one by one: the runs that overlap them end
void ctl_touch_range(psk_soft_handle *h, uint32_t lo, uint32_t hi)
{
    for (auto &r : h->uni)
        if (r.hi > r.lo && r.lo < hi && lo < r.hi) {
            run_sync(h, r);
            r.lo = r.hi = 0;
        }
}
// the run that is exactly [lo, hi), or none
psk_soft_handle::UniRun *run_find(psk_soft_handle *h, uint32_t lo, uint32_t hi)
{
    for (auto &r : h->uni)
        if (r.hi > r.lo && r.lo == lo && r.hi == hi)
            return &r;
    return nullptr;
}
// PSK_SOFT_OPT_DEFERRED_JOIN: `stream` waits for everything the side streams still carry
hipError_t deferred_join(psk_soft_handle *h, hipStream_t stream)
{
    if (!h->deferred_pending)
        return hipSuccess;
    for (int a = 0; a < kAuxStreams; a++) {
        if (!h->aux[a])
            continue;
        if (const hipError_t e = hipEventRecord(h->aux_join[a], h->aux[a]))
            return e;
        if (const hipError_t e = hipStreamWaitEvent(stream, h->aux_join[a], 0))
            return e;
    }
    h->deferred_pending = false;
    return hipSuccess;
}
bool ctl_equal(const psk::ChanCtl &a, const psk::ChanCtl &b)
{
    return a.props.samplesPerBaud == b.props.samplesPerBaud && a.props.constelationSize == b.props.constelationSize &&
           a.props.numAvg == b.props.numAvg && a.props.phaseAvg == b.props.phaseAvg &&
           a.props.differentialDecoding == b.props.differentialDecoding && a.props.resetState == b.props.resetState &&
           a.resetSamplesPerBaud == b.resetSamplesPerBaud && a.resetNumSymbols == b.resetNumSymbols && a.resetPhaseAvg == b.resetPhaseAvg &&
           a.ring_len == b.ring_len && a.symEnergySize == b.symEnergySize && a.index == b.index && a.count == b.count &&
           std::memcmp(&a.sampleRate, &b.sampleRate, sizeof(float)) == 0 && a.lf_n == b.lf_n &&
           std::memcmp(&a.lf_xdelta, &b.lf_xdelta, sizeof(float)) == 0 && a.lf_len == b.lf_len && a.lf_count == b.lf_count &&
           a.lf_head == b.lf_head && a.ring_src == b.ring_src && a.lf_recompute_pending == b.lf_recompute_pending;
}
}  // namespace

extern "C" {

uint32_t psk_soft_abi_version(void) { return PSK_SOFT_ABI_VERSION; }
const char *psk_soft_last_error(void) { return g_last_error.c_str(); }

psk_soft_status psk_soft_create(int device, uint32_t n_channels, const psk_soft_limits_t *limits,
                                psk_soft_handle_t **out)
{
    if (!out || !n_channels)
        return fail(PSK_SOFT_ERR_INVALID_ARG, "psk_soft_create: null output or zero channels");
    *out = nullptr;
    psk_soft_limits_t lim;
    lim.max_window_samples = 16384;
    lim.max_phase_avg = 512;
    lim.max_packet_complex = 1u << 20;
    if (limits)
        lim = *limits;
    if (lim.max_window_samples < 16 || lim.max_phase_avg < 1)
        return fail(PSK_SOFT_ERR_INVALID_ARG, "psk_soft_create: limits too small");
    psk_soft_handle *h = new (std::nothrow) psk_soft_handle();
    if (!h)
        return fail(PSK_SOFT_ERR_INVALID_ARG, "out of host memory");
    h->nch = n_channels;
    h->user = lim;
    h->lim.ring_cap = lim.max_window_samples;
    h->lim.fit_cap = lim.max_phase_avg + 1;  // circular yvals buffer
    h->lim.fast_fit_max = kFastFitMax;
    h->lim.force_seq = false;
    h->ctl.resize(n_channels);
    h->ctl_next.resize(n_channels);
    h->last_mode.assign(n_channels, psk::PLAN_SKIP);
    h->device = device;
    h->dry = (device == PSK_SOFT_DEVICE_NONE);
    if (const char *e = std::getenv("PSK_SOFT_TIME_TILED"))
        h->opt_tiled = std::atoi(e) < 0 ? 0 : std::atoi(e) > 2 ? 2 : std::atoi(e);
    if (const char *e = std::getenv("PSK_SOFT_TIES_IN_PLACE"))
        h->opt_ties_in_place = std::atoi(e) != 0;
    if (const char *e = std::getenv("PSK_SOFT_TRACE_LAUNCHES"))
        h->opt_trace = std::atoi(e);
    if (const char *e = std::getenv("PSK_SOFT_VALIDATE"))
        h->opt_validate = std::atoi(e);
    if (const char *e = std::getenv("PSK_SOFT_SPLIT_CLASSES"))
        h->opt_split = std::atoi(e) < 0 ? 0 : std::atoi(e) > 16 ? 16 : std::atoi(e);
    if (const char *e = std::getenv("PSK_SOFT_PIPELINED"))
        h->opt_pipe = std::atoi(e) < 0 ? 0 : std::atoi(e) > 2 ? 2 : std::atoi(e);
    if (const char *e = std::getenv("PSK_SOFT_DEFERRED_JOIN"))  // (as psk_soft_set_option(PSK_SOFT_OPT_DEFERRED_JOIN))
        h->opt_deferred = std::atoi(e) != 0;
    if (const char *e = std::getenv("PSK_SOFT_STAMP"))
        h->opt_stamp = std::atoi(e) != 0;
    if (const char *e = std::getenv("PSK_SOFT_PARALLEL_FIT"))
        h->opt_pfit = std::atoi(e) < 0 ? 0 : std::atoi(e) > 2 ? 2 : std::atoi(e);
    if (!h->dry) {
        int ndev = 0;
        hipError_t e = hipGetDeviceCount(&ndev);
        if (e != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
            delete h;
            return fail(PSK_SOFT_ERR_NO_DEVICE,
                        std::string("psk_soft_create: no usable HIP device (") +
                            (e != hipSuccess ? hipGetErrorString(e) : "device index out of range") + ")");
        }
        hipDeviceProp_t prop;
        if (hipSetDevice(device) != hipSuccess || hipGetDeviceProperties(&prop, device) != hipSuccess) {
            delete h;
            return fail(PSK_SOFT_ERR_NO_DEVICE, "psk_soft_create: hipSetDevice failed");
        }
        if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
            delete h;
            return fail(PSK_SOFT_ERR_NO_DEVICE,
                        std::string("psk_soft_create: kernels are built for gfx950 only, device is ") + prop.gcnArchName);
        }
        auto bail = [&](const char *what, hipError_t e2) {
            std::string msg = std::string("psk_soft_create: ") + what + ": " + hipGetErrorString(e2);
            psk_soft_destroy(h);
            return fail(PSK_SOFT_ERR_HIP, msg);
        };
        hipError_t e2;
        if ((e2 = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking)) != hipSuccess)
            return bail("hipStreamCreate", e2);
        size_t state_b = sizeof(psk::ChanState) * (size_t)n_channels;
        size_t ring_b = sizeof(float2) * 2u * (size_t)h->lim.ring_cap * n_channels;
        size_t yv_b = sizeof(float) * (size_t)h->lim.fit_cap * n_channels;
        if ((e2 = hipMalloc((void **)&h->d_state, state_b)) != hipSuccess) return bail("hipMalloc state", e2);
        if ((e2 = hipMalloc((void **)&h->d_ring, ring_b)) != hipSuccess) return bail("hipMalloc ring", e2);
        if ((e2 = hipMalloc((void **)&h->d_yv, yv_b)) != hipSuccess) return bail("hipMalloc yvals", e2);
        // psk_soft_i constructor state: phaseEstimate 0, last (0,0), LinearFit denominator 1, xAvg 0
        // (cpp/psk_soft.cpp:35-46, 187-199)
        std::vector<psk::ChanState> init(n_channels);
        std::memset(init.data(), 0, state_b);
        for (auto &s : init) s.lf_den = 1.0f;
        if ((e2 = hipMemcpy(h->d_state, init.data(), state_b, hipMemcpyHostToDevice)) != hipSuccess)
            return bail("hipMemcpy state", e2);
        if ((e2 = hipMemset(h->d_ring, 0, ring_b)) != hipSuccess) return bail("hipMemset", e2);
        if ((e2 = hipMemset(h->d_yv, 0, yv_b)) != hipSuccess) return bail("hipMemset", e2);
        for (int s = 0; s < kPlanSlots; s++) {
            // (a slot = the plans of a call followed by the compact channel lists of its launches: one upload)
            // (... behind the header the kernels find in front of the plans: psk_plan.h)
            char *hb = nullptr, *db = nullptr;
            if ((e2 = hipHostMalloc((void **)&hb, psk::kPlanHeaderBytes + (sizeof(psk::ChanPlan) + sizeof(uint32_t)) * n_channels)) != hipSuccess)
                return bail("hipHostMalloc plans", e2);
            std::memset(hb, 0, psk::kPlanHeaderBytes);
            h->h_plans[s] = reinterpret_cast<psk::ChanPlan *>(hb + psk::kPlanHeaderBytes);
            if ((e2 = hipMalloc((void **)&db, psk::kPlanHeaderBytes + (sizeof(psk::ChanPlan) + sizeof(uint32_t)) * n_channels)) != hipSuccess)
                return bail("hipMalloc plans", e2);
            h->d_plans[s] = reinterpret_cast<psk::ChanPlan *>(db + psk::kPlanHeaderBytes);
            if ((e2 = hipEventCreateWithFlags(&h->ev[s], hipEventDisableTiming)) != hipSuccess)
                return bail("hipEventCreate", e2);
            if ((e2 = hipEventCreateWithFlags(&h->ev_up[s], hipEventDisableTiming)) != hipSuccess)
                return bail("hipEventCreate", e2);
        }
        if ((e2 = hipStreamCreateWithFlags(&h->up_stream, hipStreamNonBlocking)) != hipSuccess)
            return bail("hipStreamCreate", e2);
        if (const char *e = std::getenv("PSK_SOFT_PLAN_STREAM"))
            h->opt_up_stream = std::atoi(e) != 0;
    }
    *out = h;
    return PSK_SOFT_OK;
}

psk_soft_status psk_soft_destroy(psk_soft_handle_t *h)
{
    if (!h)
        return PSK_SOFT_OK;
    if (!h->dry) {
        (void)hipSetDevice(h->device);
        if (h->stream)
            (void)hipStreamSynchronize(h->stream);
        for (int s = 0; s < kPlanSlots; s++) {
            if (h->h_plans[s]) (void)hipHostFree(psk::plan_header(h->h_plans[s]));
            if (h->d_plans[s]) (void)hipFree(psk::plan_header(h->d_plans[s]));
            if (h->ev[s]) (void)hipEventDestroy(h->ev[s]);
            if (h->ev_up[s]) (void)hipEventDestroy(h->ev_up[s]);
        }
        if (h->d_state) (void)hipFree(h->d_state);
        if (h->d_ring) (void)hipFree(h->d_ring);
        if (h->d_yv) (void)hipFree(h->d_yv);
        if (h->d_tiles) (void)hipFree(h->d_tiles);
        if (h->d_traw) (void)hipFree(h->d_traw);
        if (h->d_test) (void)hipFree(h->d_test);
        if (h->d_ts) (void)hipFree(h->d_ts);
        for (void *q : {(void *)h->pf.k, (void *)h->pf.y, (void *)h->pf.S, (void *)h->pf.c, (void *)h->pf.tt, (void *)h->pf.xs,
                        (void *)h->pf.tile, (void *)h->pf.blk, (void *)h->pf.walk, (void *)h->pf.chan})
            if (q) (void)hipFree(q);
        if (h->pf.hint) (void)hipHostFree(h->pf.hint);
        if (h->tile_ev) (void)hipEventDestroy(h->tile_ev);
        for (auto &sl : h->stage) {
            if (sl.stream) (void)hipStreamSynchronize(sl.stream);
            if (sl.h_buf) (void)hipHostFree(sl.h_buf);
            if (sl.d_buf) (void)hipFree(sl.d_buf);
            if (sl.done) (void)hipEventDestroy(sl.done);
            if (sl.stream) (void)hipStreamDestroy(sl.stream);
        }
        delete h->pool;
        for (int k = 0; k < kAuxStreams; k++) {
            if (h->aux[k]) (void)hipStreamSynchronize(h->aux[k]);
            if (h->aux_join[k]) (void)hipEventDestroy(h->aux_join[k]);
            if (h->aux[k]) (void)hipStreamDestroy(h->aux[k]);
        }
        for (hipStream_t &q : h->pipe_st)
            if (q) {
                (void)hipStreamSynchronize(q);
                (void)hipStreamDestroy(q);
            }
        for (hipEvent_t &e : h->pipe_ev)
            if (e) (void)hipEventDestroy(e);
        if (h->d_pipe_carry) (void)hipFree(h->d_pipe_carry);
        if (h->d_pipe_y) (void)hipFree(h->d_pipe_y);
        if (h->aux_fork) (void)hipEventDestroy(h->aux_fork);
        for (auto &row : h->slot_aux_ev)
            for (hipEvent_t &e : row)
                if (e) (void)hipEventDestroy(e);
        if (h->up_stream) {
            (void)hipStreamSynchronize(h->up_stream);
            (void)hipStreamDestroy(h->up_stream);
        }
        if (h->stream) (void)hipStreamDestroy(h->stream);
    }
    delete h;
    return PSK_SOFT_OK;
}

psk_soft_status psk_soft_configure(psk_soft_handle_t *h, uint32_t ch0, uint32_t nch, const psk_soft_props_t *props)
{
    if (!h || !props || (uint64_t)ch0 + nch > h->nch)
        return fail(PSK_SOFT_ERR_INVALID_ARG, "psk_soft_configure: bad channel range");
    for (uint32_t i = 0; i < nch; i++) {
        const psk_soft_props_t &p = props[i];
        if ((uint64_t)p.samplesPerBaud * p.numAvg > h->lim.ring_cap || p.phaseAvg >= h->lim.fit_cap ||
            p.samplesPerBaud > kSeqMaxS)
            return fail(PSK_SOFT_ERR_LIMIT, "psk_soft_configure: property exceeds the limits given at create "
                                            "(samplesPerBaud*numAvg, phaseAvg) or samplesPerBaud > 1024");
    }
    ctl_touch(h);
    for (uint32_t i = 0; i < nch; i++) h->ctl[ch0 + i].configure(props[i]);
    return PSK_SOFT_OK;
}

psk_soft_status psk_soft_query(const psk_soft_handle_t *h, uint32_t ch, psk_soft_props_t *props)
{
    if (!h || !props || ch >= h->nch)
        return fail(PSK_SOFT_ERR_INVALID_ARG, "psk_soft_query: bad channel");
    ctl_sync(h);
    *props = h->ctl[ch].props;
    return PSK_SOFT_OK;
}

psk_soft_status psk_soft_fire_listener(psk_soft_handle_t *h, uint32_t ch, int which)
{
    if (!h || ch >= h->nch)
        return fail(PSK_SOFT_ERR_INVALID_ARG, "psk_soft_fire_listener: bad channel");
    ctl_touch(h);
    switch (which) {
    case 0: h->ctl[ch].samplesPerBaudChanged(); break;
    case 1: h->ctl[ch].constelationSizeChanged(); break;
    case 2: h->ctl[ch].phaseAvgChanged(); break;
    default: return fail(PSK_SOFT_ERR_INVALID_ARG, "psk_soft_fire_listener: which must be 0..2");
    }
    return PSK_SOFT_OK;
}

uint64_t psk_soft_output_capacity(const psk_soft_handle_t *h, uint32_t ch, uint64_t n_complex)
{
    if (!h || ch >= h->nch)
        return 0;
    ctl_sync(h);
    uint64_t S = h->ctl[ch].props.samplesPerBaud ? h->ctl[ch].props.samplesPerBaud : 1;
    return (n_complex + S - 1) / S + 1;
}

// One pass of the control plane over a batch and the launches it asks for.  cont[i], pieces of a call the library
// has cut (process_device below): bit 0 = packet i continues the call of the packet before it (plan_call's `cont`), bit 1 = more
// pieces of the call follow (no end-of-call wrap yet).
static psk_soft_status process_round(psk_soft_handle_t *h, uint32_t ch0, uint32_t nch, const psk_soft_packet_t *pkts,
                                     psk_soft_output_t *outs, void *stream_v, const uint8_t *cont)
{
    if (!h || !pkts || !outs || !nch || (uint64_t)ch0 + nch > h->nch)
        return fail(PSK_SOFT_ERR_INVALID_ARG, "psk_soft_process: bad arguments");
    if (h->poisoned)
        return fail(PSK_SOFT_ERR_HIP, "psk_soft_process: an earlier call failed inside HIP after its kernels were enqueued; the "
                                      "channel states are undefined -- destroy the handle (or import saved states into a new one)");
    // The plans are written straight into the pinned upload slot of this call: wait until the launch
    // that last used the slot has consumed it.  (A refused call does not advance the slot.)
    const int slot = h->slot;
    psk::ChanPlan *plans;
    if (h->dry) {
        if (h->plans_dry.size() < nch)
            h->plans_dry.resize(nch);
        plans = h->plans_dry.data();
    } else {
        PSK_HIP(hipSetDevice(h->device));
        if (h->ev_used[slot])
            PSK_HIP(hipEventSynchronize(h->ev[slot]));
        for (int a = 0; a < kAuxStreams; a++)
            if (h->slot_aux_used[slot][a]) {  // (a deferred call's classes end on the side streams)
                PSK_HIP(hipEventSynchronize(h->slot_aux_ev[slot][a]));
                h->slot_aux_used[slot][a] = false;
            }
        plans = h->h_plans[slot];
    }
    // plan on copies (ctl_next); commit only if every channel of the batch is accepted
    const bool dry = h->dry;
    const uint32_t extra_flags = (h->opt_qpsk_sign_map ? (uint32_t)psk::PLAN_QPSK_SIGN_MAP : 0u) |
                                 (h->opt_ties_in_place ? 0u : (uint32_t)psk::PLAN_TIES_HANDOVER);
    const psk::Limits lim = h->lim;
    // what a planned channel asks of the launches (`mult` channels with this very plan)
    auto account = [&](const psk::ChanPlan &p, PlanSummary &r, uint32_t mult) {
        r.any = true;
        if (p.mode == psk::PLAN_FAST) {
            if (p.n_out && (p.lf_flags & psk::PLAN_ANYFRONT)) {
                r.any_emit = true;
                r.cnt_any += mult;
                const uint32_t nb = (uint32_t)((p.n_out + 127u) / 128u);
                r.blocks_any += (uint64_t)nb * mult;
                if (nb > r.max_blocks_any) r.max_blocks_any = nb;
                if (p.lf_n > r.max_n_any) r.max_n_any = p.lf_n;
                if (p.A > r.max_A_any) r.max_A_any = p.A;
                if (p.S > r.max_S_any) r.max_S_any = p.S;
            } else if (p.n_out) {
                r.any_emit = true;
                const int Hh = psk::fast_hist_blocks(p.A) + (p.lf_n > kDeepFit ? 8 : 0);
                r.need_SH[p.S][Hh] = true;
                r.cnt_SH[p.S][Hh] += mult;
                if (p.lf_n > r.max_n[p.S][Hh]) r.max_n[p.S][Hh] = p.lf_n;
                if (p.A > r.max_A[p.S][Hh]) r.max_A[p.S][Hh] = p.A;
                const uint32_t nb = (uint32_t)((p.n_out + 127u) / 128u);
                r.blocks_SH[p.S][Hh] += (uint64_t)nb * mult;
                if (nb > r.max_blocks_SH[p.S][Hh]) r.max_blocks_SH[p.S][Hh] = nb;
            } else {
                r.any_quiet = true;
                r.cnt_quiet += mult;
                if (p.lf_n > r.max_n_quiet) r.max_n_quiet = p.lf_n;
            }
        } else {
            r.any_seq = true;
            if (!lim.force_seq && p.lf_n <= lim.fast_fit_max &&
                (p.n_out > psk::kResyncCount || (uint64_t)p.lf_count0 + p.n_out > psk::kResyncCount))
                r.long_call = true;
        }
    };
    uint32_t *const handed_over = dry ? nullptr : psk::plan_header(h->d_plans[slot]);  // (psk_plan.h)
    auto misaligned = [&](const psk::ChanPlan &p) {
        return !dry && ((p.n_in && !p.in) || ((uintptr_t)p.in & 7u) || ((uintptr_t)p.soft & 7u) || ((uintptr_t)p.bits & 3u) ||
                        ((uintptr_t)p.phase & 3u) || ((uintptr_t)p.sidx & 3u));
    };

    // ---- the stamped path ----
    // Channels that were configured alike and have been fed packets of the same length ever since hold IDENTICAL control state
    // -- it depends on nothing else (psk_ctl.h) -- and a batch of equal packets then gets the same plan in every channel but for
    // its five pointers.  Such a batch is planned ONCE: plan_call on channel ch0, the plan stamped into the other slots with
    // the pointers patched, the result fields of outs[] copied, and the new control state kept in ONE copy (uni_ctl) that
    // stands for the whole range until somebody looks at a single channel (ctl_sync).  What the loop below still does per
    // channel is read the packet and the output descriptor (the equal-packets test) and write the plan: ~3 ns against ~14 ns.
    // Anything unusual -- a refused call, a packet that differs, a buffer too small or misaligned -- leaves the stamped path for
    // the ordinary one below, which plans every channel on its own and reports the error as it always did.
    bool stamped = false;
    psk::ChanCtl stamp_ctl;
    psk_soft_handle::UniRun *run = nullptr;
    PlanSummary res;
    if (h->opt_stamp && !cont && nch >= 16u) {
        bool uniform = false;
        run = run_find(h, ch0, ch0 + nch);
        if (run) {
            uniform = true;
        } else {
            ctl_touch_range(h, ch0, ch0 + nch);  // (runs that overlap the range without being it)
            const bool known_mixed = h->mixed_ttl > 0 && h->mixed_lo == ch0 && h->mixed_hi == ch0 + nch;
            if (known_mixed) {
                h->mixed_ttl--;
            } else {
                uniform = true;
                const psk::ChanCtl &c0 = h->ctl[ch0];
                for (uint32_t i = 1; i < nch && uniform; i++) uniform = ctl_equal(c0, h->ctl[ch0 + i]);
                if (uniform) {
                    for (auto &r : h->uni)
                        if (r.hi == r.lo) {
                            run = &r;
                            break;
                        }
                    if (!run) {  // (every slot taken: the first one makes room)
                        run = &h->uni[0];
                        run_sync(h, *run);
                    }
                    run->lo = ch0, run->hi = ch0 + nch, run->lazy = false;
                } else {
                    h->mixed_lo = ch0, h->mixed_hi = ch0 + nch, h->mixed_ttl = 256;
                }
            }
        }
        if (uniform) {
            stamp_ctl = run->lazy ? run->ctl : h->ctl[ch0];
            psk::ChanPlan &p0 = plans[0];
            const psk_soft_packet_t &k0 = pkts[0];
            bool ok = stamp_ctl.props.samplesPerBaud <= kSeqMaxS &&
                      psk::plan_call(stamp_ctl, lim, k0, outs[0], p0, false) == PSK_SOFT_OK && p0.mode != psk::PLAN_SKIP &&
                      !misaligned(p0);
            if (ok) {
                p0.lf_flags |= extra_flags;
                p0.handed_over = handed_over;
                account(p0, res, nch);
                ok = !res.long_call;
            }
            if (ok) {
                const psk_soft_output_t o0 = outs[0];
                const uint64_t n_out = p0.n_out;
                uint32_t i = 1;
                for (; i < nch; i++) {
                    const psk_soft_packet_t &k = pkts[i];
                    psk_soft_output_t &o = outs[i];
                    if (k.n_floats != k0.n_floats || k.present != k0.present || k.sri_mode != k0.sri_mode || k.sriChanged != k0.sriChanged ||
                        k.inputQueueFlushed != k0.inputQueueFlushed || std::memcmp(&k.sri_xdelta, &k0.sri_xdelta, sizeof(double)) != 0)
                        break;
                    if (n_out > o.cap_symbols && (o.soft || o.phase))
                        break;
                    psk::ChanPlan &p = plans[i];
                    p = p0;
                    p.in = k.data, p.soft = o.soft, p.bits = o.bits, p.phase = o.phase, p.sidx = o.sampleIndex;
                    if (misaligned(p))
                        break;
                    o.ret = o0.ret, o.n_symbols = o0.n_symbols, o.n_bits = o0.n_bits, o.n_sampleIndex = o0.n_sampleIndex;
                    o.sri_pushed = o0.sri_pushed, o.sri_soft_xdelta = o0.sri_soft_xdelta, o.sri_bits_xdelta = o0.sri_bits_xdelta;
                    o.n_warn = o0.n_warn;
                }
                ok = i == nch;
            }
            stamped = ok;
            if (!stamped)
                res = PlanSummary();
        }
    }
    if (!stamped)
        ctl_touch_range(h, ch0, ch0 + nch);  // (the ordinary path writes the channels one by one)
    psk::ChanCtl *const next = h->ctl_next.data() + ch0;
    const psk::ChanCtl *const cur = h->ctl.data() + ch0;
    auto plan_range = [&](uint32_t lo, uint32_t hi, PlanSummary &r) {
        for (uint32_t i = lo; i < hi; i++) {
            next[i] = cur[i];
            if (next[i].props.samplesPerBaud > kSeqMaxS) {
                r.st = PSK_SOFT_ERR_LIMIT, r.bad = i, r.why = 1;
                return;
            }
            psk::ChanPlan &p = plans[i];
            psk_soft_status st = psk::plan_call(next[i], lim, pkts[i], outs[i], p, cont && (cont[i] & 1u));
            if (st != PSK_SOFT_OK) {
                r.st = st, r.bad = i, r.why = 0;
                return;
            }
            if (p.mode == psk::PLAN_SKIP)
                continue;
            if (misaligned(p)) {
                r.st = PSK_SOFT_ERR_INVALID_ARG, r.bad = i, r.why = 2;
                return;
            }
            p.lf_flags |= extra_flags;
            p.handed_over = handed_over;
            if (cont && (cont[i] & 2u))
                p.lf_flags |= psk::PLAN_NO_WRAP;  // (more pieces of this call follow)
            if (cont && (cont[i] & 4u))
                p.lf_flags |= psk::PLAN_CARRY_DRIFT;  // (a piece cut where the reference does not rebuild its sums)
            account(p, r, 1u);
        }
    };
    // One thread: the ordinary pass is ~15 ns and ~0.5 KB of cache traffic per channel (60 us for 4096
    // channels).  Splitting it over a thread pool was measured and dropped: the workers' share is
    // done in 10-20 us, after which they sleep until the next call, and waking them costs more than
    // the whole pass (spinning instead would burn cores between packets).
    if (!stamped)
        plan_range(0, nch, res);
    if (res.st == PSK_SOFT_OK && res.long_call && !cont) {
        g_long_call = true;  // (nothing committed, nothing enqueued: the caller cuts the call into pieces)
        return PSK_SOFT_OK;
    }
    if (res.st != PSK_SOFT_OK) {
        if (res.why == 1)
            return fail(PSK_SOFT_ERR_LIMIT, "samplesPerBaud > 1024");
        if (res.why == 2)
            return fail(PSK_SOFT_ERR_INVALID_ARG,
                        "psk_soft_process: packet data must be 8-byte aligned, soft 8, bits 4, phase 4, sampleIndex 4");
        char buf[160];
        std::snprintf(buf, sizeof buf, "psk_soft_process: channel %u refused (status %d)", ch0 + res.bad, (int)res.st);
        return fail(res.st, buf);
    }
    // The host mirror (ctl) is committed only once everything the call needs has been enqueued: a HIP error
    // before the first kernel launch leaves the channels untouched; one after it poisons the handle (device
    // state half advanced, nothing to roll it back with).
    auto commit = [&]() {
        if (stamped) {  // (one copy stands for the range: see ctl_sync)
            run->ctl = stamp_ctl;
            run->mode = plans[0].mode;
            run->lazy = true;
            return;
        }
        if (ch0 == 0 && nch == h->nch)
            h->ctl.swap(h->ctl_next);
        else
            std::memcpy(static_cast<void *>(h->ctl.data() + ch0), next, sizeof(psk::ChanCtl) * nch);
        for (uint32_t i = 0; i < nch; i++) h->last_mode[ch0 + i] = plans[i].mode;
    };
    if (h->dry || !res.any) {
        commit();
        return PSK_SOFT_OK;
    }
    const bool any_emit = res.any_emit, any_seq = res.any_seq, any_quiet = res.any_quiet;
    const auto &need_SH = res.need_SH;
    const auto &max_n = res.max_n;
    const auto &max_A = res.max_A;

    hipStream_t stream = stream_v ? (hipStream_t)stream_v : h->stream;
    // Calls that touch the same channels must run in order: a call waits for the earlier calls on OTHER streams
    // whose channel range overlaps its own (same stream: the stream orders them; older calls than the plan slots
    // remember have completed -- a slot is only reused after its event).
    for (int k = 0; k < kPlanSlots; k++)
        if (k != slot && h->ev_used[k] && h->slot_stream[k] != stream && h->slot_ch0[k] < ch0 + nch && ch0 < h->slot_ch0[k] + h->slot_nch[k]) {
            PSK_HIP(hipStreamWaitEvent(stream, h->ev[k], 0));
            for (int a = 0; a < kAuxStreams; a++)
                if (h->slot_aux_used[k][a])
                    PSK_HIP(hipStreamWaitEvent(stream, h->slot_aux_ev[k][a], 0));
        }
    // compact lists, one per launch, behind the plans: first the channels that emit nothing, then every (S, H)
    // class in launch order
    uint32_t *const h_list = reinterpret_cast<uint32_t *>(h->h_plans[slot] + nch);
    const uint32_t *const d_list = reinterpret_cast<const uint32_t *>(h->d_plans[slot] + nch);
    uint32_t off_SH[33][17] = {}, off_quiet = 0;
    const uint32_t off_any = res.cnt_quiet;
    {
        uint32_t run = res.cnt_quiet + res.cnt_any;
        for (int S : kFastS)
            for (int H : kClassH) {
                off_SH[S][H] = run;
                run += res.cnt_SH[S][H];
            }
        uint32_t fill_SH[33][17] = {}, fill_quiet = 0, fill_any = 0;
        if (stamped)  // (one class holds every channel, in order; the offsets of the others are equal to its end)
            for (uint32_t i = 0; i < nch; i++) h_list[i] = i;
        for (uint32_t i = 0; i < nch && !stamped; i++) {
            const psk::ChanPlan &p = plans[i];
            if (p.mode != psk::PLAN_FAST)
                continue;
            if (p.n_out && (p.lf_flags & psk::PLAN_ANYFRONT)) {
                h_list[off_any + fill_any++] = i;
            } else if (p.n_out) {
                const int Hh = psk::fast_hist_blocks(p.A) + (p.lf_n > kDeepFit ? 8 : 0);
                h_list[off_SH[p.S][Hh] + fill_SH[p.S][Hh]++] = i;
            } else {
                h_list[off_quiet + fill_quiet++] = i;
            }
        }
    }
    // Window classes of few channels and long calls go through the time-tiled kernels first (psk_tile_kernel.h): their
    // channels get a place in the scratch of the call -- symbols padded to whole blocks, K blocks to a tile, K chosen so
    // that the class makes a few thousand tiles.
    bool tiled_SH[33][17] = {};
    bool pf_second = false;
    uint32_t tiles_max_any = 0;
    uint32_t tiles_max_SH[33][17] = {};
    size_t tile_syms = 0, tile_count = 0;
    if (res.cnt_any) {
        // (window classes without a wave-scan instantiation: always tiled, whatever the option says -- the alternative is the
        // reference-order kernel; a tile at least as long as the longest window, so that rebuilding it stays a fraction)
        uint64_t K = res.blocks_any / kTiledTargetTiles;
        K = K < 2 ? 2 : K > 16 ? 16 : K;
        // (a tile rebuilds the window in front of it: tiles at least as long as the longest window keep that a fraction of the
        // work -- where there are tiles enough to fill the machine anyway; a few channels finish sooner on many short tiles)
        const uint64_t k_win = (res.max_A_any + 127u) / 128u;
        if (K < k_win) {
            const uint64_t k_fill = res.blocks_any / (kTiledTargetTiles / 2);
            const uint64_t k_long = k_win > 64 ? 64 : k_win;
            K = k_fill > k_long ? k_long : k_fill > K ? k_fill : K;
        }
        tiles_max_any = (uint32_t)((res.max_blocks_any + K - 1) / K);
        for (uint32_t i = 0; i < res.cnt_any; i++) {
            psk::ChanPlan &p = plans[h_list[off_any + i]];
            const uint64_t nb = (p.n_out + 127u) / 128u;
            p.lf_flags |= psk::PLAN_TILED;
            if (h->opt_pfit && p.lf_len0 == p.lf_n && p.lf_n >= 2)
                p.lf_flags |= psk::PLAN_PFIT;
            p.tile_blocks = (uint32_t)K;
            p.tile_base = (uint32_t)tile_count;
            p.tile_off = tile_syms;
            tile_count += (size_t)((nb + K - 1) / K);
            tile_syms += (size_t)nb * 128u;
        }
    }
    bool piped_SH[33][17] = {};
    uint32_t pipe_tiles_SH[33][17] = {};  // tiles of a range
    size_t pipe_need = 0;
    // (the parallel fit's scratch, 36 of the 52 bytes a symbol, is only needed by the classes that are not pipelined: those get
    // their places first, so that it need not cover the others)
    size_t pf_syms = tile_syms, pf_count = tile_count;
    for (int pass = 0; pass < (h->opt_tiled ? 2 : 0); pass++) {
        for (int S : kFastS)
            for (int H : kClassH) {
                if (!res.need_SH[S][H] || !psk::tile_front_has(S, class_H(H)))
                    continue;
                // pipelined: the serial fit of a range under the front stage of the next (see kPipeMinChannels)
                // (PSK_SOFT_PIPELINED=2, tests: wherever the kernels allow it, a few blocks to a range)
                const bool pipe = (h->opt_pipe == 2 ? res.max_blocks_SH[S][H] >= 4u
                                                    : h->opt_pipe && h->opt_tiled == 1 && res.cnt_SH[S][H] >= kPipeMinChannels &&
                                                          res.cnt_SH[S][H] <= kPipeMaxChannels && res.max_blocks_SH[S][H] >= kPipeMinBlocks) &&
                                  !cont && res.max_n[S][H] + 128u <= kPipeMaxYLen && (size_t)res.blocks_SH[S][H] * 128u <= kPipeMaxSymbols;
                if (pipe != (pass == 1))
                    continue;
                if (!pipe && h->opt_tiled == 1 &&
                    !((res.cnt_SH[S][H] <= kTiledFewChannels && res.max_blocks_SH[S][H] >= kTiledMinBlocksFew) ||
                      (res.cnt_SH[S][H] <= kTiledMaxChannels && res.max_blocks_SH[S][H] >= kTiledMinBlocks)))
                    continue;
                uint64_t K = res.blocks_SH[S][H] / kTiledTargetTiles;
                K = K < 2 ? 2 : K > 16 ? 16 : K;
                tiled_SH[S][H] = true;
                if (pipe) {
                    piped_SH[S][H] = true;
                    const uint32_t T = (uint32_t)((res.max_blocks_SH[S][H] + K - 1) / K);
                    uint32_t tr = (uint32_t)(kTiledTargetTiles / res.cnt_SH[S][H]);  // (a range = about one machine-load of tiles)
                    tr = tr < 1u ? 1u : tr;
                    if (h->opt_pipe == 2 && tr > 3u)
                        tr = 3u;
                    if (const char *e = std::getenv("PSK_SOFT_PIPE_RANGE_TILES"))  // (A/B runs)
                        tr = std::atoi(e) > 0 ? (uint32_t)std::atoi(e) : tr;
                    if ((T + tr - 1) / tr > kPipeMaxRanges)
                        tr = (T + kPipeMaxRanges - 1) / kPipeMaxRanges;
                    pipe_tiles_SH[S][H] = tr;
                    if (off_SH[S][H] + res.cnt_SH[S][H] > pipe_need)
                        pipe_need = off_SH[S][H] + res.cnt_SH[S][H];
                }
                tiles_max_SH[S][H] = (uint32_t)((res.max_blocks_SH[S][H] + K - 1) / K);
                for (uint32_t i = 0; i < res.cnt_SH[S][H]; i++) {
                    psk::ChanPlan &p = plans[h_list[off_SH[S][H] + i]];
                    const uint64_t nb = (p.n_out + 127u) / 128u;
                    p.lf_flags |= psk::PLAN_TILED;
                    if (!pipe && h->opt_pfit && p.lf_len0 == p.lf_n && p.lf_n >= 2)
                        p.lf_flags |= psk::PLAN_PFIT;
                    p.tile_blocks = (uint32_t)K;
                    p.tile_base = (uint32_t)tile_count;
                    p.tile_off = tile_syms;
                    tile_count += (size_t)((nb + K - 1) / K);
                    tile_syms += (size_t)nb * 128u;
                }
            }
        if (pass == 0)
            pf_syms = tile_syms, pf_count = tile_count;
    }
    if (tile_syms) {
        if (tile_syms > h->tile_sym_cap || tile_count > h->tile_cap || pf_syms > h->pf_sym_cap || pf_count > h->pf_cap) {
            // (rare: the scratch grows to the largest call seen, plus a quarter)
            PSK_HIP(hipDeviceSynchronize());
            if (h->d_tiles) (void)hipFree(h->d_tiles);
            if (h->d_traw) (void)hipFree(h->d_traw);
            if (h->d_test) (void)hipFree(h->d_test);
            if (h->d_ts) (void)hipFree(h->d_ts);
            for (void *q : {(void *)h->pf.k, (void *)h->pf.y, (void *)h->pf.S, (void *)h->pf.c, (void *)h->pf.tt, (void *)h->pf.xs,
                            (void *)h->pf.tile, (void *)h->pf.blk, (void *)h->pf.walk})
                if (q) (void)hipFree(q);
            h->d_tiles = nullptr, h->d_traw = h->d_test = nullptr, h->d_ts = nullptr;
            h->pf.k = nullptr, h->pf.y = nullptr, h->pf.S = h->pf.c = h->pf.xs = nullptr, h->pf.tt = nullptr, h->pf.tile = nullptr,
            h->pf.blk = nullptr, h->pf.walk = nullptr;
            h->tile_cap = h->tile_sym_cap = h->pf_cap = h->pf_sym_cap = 0;
            const size_t syms = tile_syms + tile_syms / 4, cnt = tile_count + tile_count / 4;
            const size_t psyms = pf_syms + pf_syms / 4 + 128u, pcnt = pf_count + pf_count / 4 + 1u;
            // (out of device memory: the call does without -- the wave-scan kernels carry everything on their own)
            auto grab = [](auto **q, size_t bytes) { return hipMalloc((void **)q, bytes) == hipSuccess; };
            bool got = grab(&h->d_tiles, sizeof(psk::TileInfo) * cnt) && grab(&h->d_traw, sizeof(float) * syms) &&
                       grab(&h->d_test, sizeof(float) * syms) && grab(&h->d_ts, sizeof(float2) * syms) &&
                       grab(&h->pf.k, sizeof(int) * psyms) && grab(&h->pf.y, sizeof(float) * psyms) && grab(&h->pf.S, sizeof(double) * psyms) &&
                       grab(&h->pf.c, sizeof(double) * psyms) && grab(&h->pf.tt, sizeof(float) * psyms) && grab(&h->pf.xs, sizeof(double) * psyms) &&
                       grab(&h->pf.tile, sizeof(psk::PfTile) * pcnt) && grab(&h->pf.blk, sizeof(psk::PfBlock) * (psyms / 128u + 1u)) &&
                       grab(&h->pf.walk, sizeof(psk::PfWalk) * (psyms / 128u + 1u));
            if (got && !h->pf.chan) {
                got = grab(&h->pf.chan, sizeof(psk::PfChan) * h->nch) && hipMemset(h->pf.chan, 0, sizeof(psk::PfChan) * h->nch) == hipSuccess &&
                      hipHostMalloc((void **)&h->pf.hint, 64) == hipSuccess;
                if (got)
                    *h->pf.hint = 0u;
            }
            if (!got) {
                (void)hipGetLastError();
                for (uint32_t i = 0; i < nch; i++) {
                    if (plans[i].lf_flags & psk::PLAN_ANYFRONT) {  // (no kernel but the reference-order one is left for these)
                        plans[i].mode = plans[i].S == 1u ? psk::PLAN_SEQ_S1 : psk::PLAN_SEQ;
                        res.any_seq = true;
                    }
                    plans[i].lf_flags &= ~(uint32_t)(psk::PLAN_TILED | psk::PLAN_PFIT | psk::PLAN_ANYFRONT);
                }
                res.cnt_any = 0;
                for (auto &row : tiled_SH)
                    for (bool &t : row) t = false;
                for (auto &row : piped_SH)
                    for (bool &t : row) t = false;
                tile_syms = 0;
            } else {
                h->tile_cap = cnt;
                h->tile_sym_cap = syms;
                h->pf_cap = pcnt;
                h->pf_sym_cap = psyms;
            }
        }
    }
    if (tile_syms) {
        if (!h->tile_ev)
            PSK_HIP(hipEventCreateWithFlags(&h->tile_ev, hipEventDisableTiming));
        // second round of the parallel fit: for the next 16 tiled calls after one whose first guess of the unwrap counts
        // failed somewhere (the kernels leave a note in page-locked memory; a late or lost note costs a call or two)
        if (h->pf.hint && *static_cast<volatile uint32_t *>(h->pf.hint)) {
            *static_cast<volatile uint32_t *>(h->pf.hint) = 0u;
            h->pf_second_ttl = 16;
        } else if (h->pf_second_ttl) {
            h->pf_second_ttl--;
        }
        pf_second = h->opt_pfit == 2 || h->pf_second_ttl > 0;
        if (h->tile_ev_used && h->tile_stream != stream)  // the scratch is one per handle
            PSK_HIP(hipStreamWaitEvent(stream, h->tile_ev, 0));
        if (pipe_need) {
            // the pipelined mode's own scratch (one PipeCarry and one ring of kPipeMaxYLen floats per channel of a launch) and its
            // two streams -- of another priority than the caller's, so that they get hardware queues of their own
            if (pipe_need > h->pipe_cap) {
                PSK_HIP(hipDeviceSynchronize());
                if (h->d_pipe_carry) (void)hipFree(h->d_pipe_carry);
                if (h->d_pipe_y) (void)hipFree(h->d_pipe_y);
                h->d_pipe_carry = nullptr, h->d_pipe_y = nullptr, h->pipe_cap = 0;
                const size_t cap = pipe_need + pipe_need / 4;
                if (hipMalloc(&h->d_pipe_carry, psk::pipe_carry_bytes() * cap) == hipSuccess &&
                    hipMalloc((void **)&h->d_pipe_y, sizeof(float) * kPipeMaxYLen * cap) == hipSuccess) {
                    h->pipe_cap = cap;
                } else {  // (out of device memory: the one-launch kernels do without)
                    (void)hipGetLastError();
                    if (h->d_pipe_carry) (void)hipFree(h->d_pipe_carry);
                    h->d_pipe_carry = nullptr;
                    for (auto &row : piped_SH)
                        for (bool &t : row) t = false;
                }
            }
            if (!h->pipe_st[0]) {
                int prio_lo = 0, prio_hi = 0;
                PSK_HIP(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
                // (CU-masked streams for the stages -- the serial ones on every n-th CU, the front stage on the others -- were measured:
                // no gain, 2.77 ... 2.84 against 2.75 ms at 512 channels)
                for (hipStream_t &q : h->pipe_st) PSK_HIP(hipStreamCreateWithPriority(&q, hipStreamNonBlocking, prio_hi));
                for (hipEvent_t &e : h->pipe_ev) PSK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            }
        }
    }
    // PSK_SOFT_VALIDATE=1 (tests, the randomised comparison): what the kernels take for granted about a plan -- the samples a call
    // reads exist, what it leaves behind fits the rings, its place in the scratch of the time-tiled kernels lies inside it -- is
    // checked here, on the host, in front of the first launch; a violation refuses the call (nothing enqueued, nothing committed)
    // instead of sending a kernel out of bounds.
    if (h->opt_validate) {
        const char *why = nullptr;
        uint32_t bad = 0;
        for (uint32_t i = 0; i < nch && !why; i++) {
            const psk::ChanPlan &p = plans[i];
            if (p.mode == psk::PLAN_SKIP)
                continue;
            bad = i;
            const uint64_t S = p.S ? p.S : 1u, have = (uint64_t)p.ring_len0 + p.n_in;
            const uint64_t nb = (p.n_out + 127u) / 128u;
            if (p.n_out && p.S > 1u && (p.n_out + p.A - 1u) * S > have)
                why = "the call reads samples behind the packet's end";
            else if (p.n_out && p.S <= 1u && p.mode != psk::PLAN_SEQ_S1 && p.n_out > p.n_in)
                why = "more symbols than samples at one sample per symbol";
            else if (p.ring_len0 > h->lim.ring_cap || p.ring_len1 > h->lim.ring_cap || p.ring_src > 1u)
                why = "carried samples beyond the ring";
            else if (p.ring_len1 > have)
                why = "more samples carried out of the call than it holds";
            else if (p.lf_n >= h->lim.fit_cap || p.lf_len0 > p.lf_n || p.lf_head >= h->lim.fit_cap)
                why = "LinearFit window beyond its ring";
            else if ((p.lf_flags & psk::PLAN_TILED) && p.mode == psk::PLAN_FAST && p.n_out &&
                     (!p.tile_blocks || p.tile_off + nb * 128u > h->tile_sym_cap ||
                      (uint64_t)p.tile_base + (nb + p.tile_blocks - 1u) / p.tile_blocks > h->tile_cap))
                why = "place in the time-tiled scratch outside it";
            else if ((p.lf_flags & psk::PLAN_PFIT) && p.mode == psk::PLAN_FAST && p.n_out &&
                     (p.tile_off + nb * 128u > h->pf_sym_cap || (uint64_t)p.tile_base + (nb + p.tile_blocks - 1u) / p.tile_blocks > h->pf_cap))
                why = "place in the parallel fit's scratch outside it";
            else if (p.mode == psk::PLAN_FAST && p.n_out && !(p.lf_flags & psk::PLAN_ANYFRONT) &&
                     (p.S < 2u || p.S > 32u || p.A > 1024u || (p.S > 16u && p.A > 512u)))
                why = "window class without a wave-scan instantiation planned for one";
            else if (p.mode == psk::PLAN_FAST && p.n_out > psk::kResyncCount)
                why = "a piece longer than 2^20 symbols on the wave-scan kernels";
        }
        if (why) {
            char buf[200];
            std::snprintf(buf, sizeof buf, "psk_soft_process: plan of channel %u fails validation: %s", ch0 + bad, why);
            return fail(PSK_SOFT_ERR_LIMIT, buf);
        }
    }
    // window classes of the call in launch order (deepest history first); class 0 stays on the caller's stream, the others take
    // the side streams in turn
    struct Cls {
        int S, H;
    };
    Cls cls[33 * 8];
    int n_cls = 0;
    for (int k = 7; k >= 0; k--)
        for (int S : kFastS)
            if (need_SH[S][kClassH[k]])
                cls[n_cls++] = Cls{S, kClassH[k]};
    const bool fork = n_cls > 1 && h->opt_fork;
    // deferred join (see psk_soft_handle::opt_deferred): only calls whose every channel runs on wave-scan launches
    bool deferred = fork && (h->opt_deferred || g_split_mode) && (!cont || g_split_mode) && !tile_syms && !res.cnt_any && !res.any_seq;
    if (fork && !cont && !h->opt_deferred && h->opt_split > 1 && !tile_syms && !res.cnt_any && !res.any_seq) {
        // Classes that cannot be resident together (a SIMD's registers hold four waves of the short windows or two of the long
        // ones) take two rounds of waves, and a wave that starts late still needs the whole call's time at the lone-wave rate.
        // Cut in time, the pieces of the short class that start late are short too, and the class runs its last pieces with the
        // machine to itself: configs[4] 3.4 -> 2.6 ms for ONE joined call.  The pieces are continuations of the one
        // serviceFunction() call (plan_call's `cont`), the bounds that count the reference's rounding since the call began carry
        // over (PLAN_CARRY_DRIFT).  Nothing is committed or enqueued here: the caller plans the pieces.
        uint32_t listed = res.cnt_quiet, longest = 0;
        for (int i = 0; i < n_cls; i++) {
            listed += res.cnt_SH[cls[i].S][cls[i].H];
            longest = res.max_blocks_SH[cls[i].S][cls[i].H] > longest ? res.max_blocks_SH[cls[i].S][cls[i].H] : longest;
        }
        static const uint32_t min_blocks = [] {  // (PSK_SOFT_SPLIT_MIN_BLOCKS: tests cut short calls too)
            const char *e = std::getenv("PSK_SOFT_SPLIT_MIN_BLOCKS");
            return e && std::atoi(e) > 0 ? (uint32_t)std::atoi(e) : kSplitMinBlocks;
        }();
        if (listed == nch && longest >= min_blocks) {
            g_split_pieces = h->opt_split;
            return PSK_SOFT_OK;
        }
    }
    uint64_t sig = 1469598103934665603ull;
    if (deferred) {
        auto mix = [&](uint64_t v) { sig = (sig ^ v) * 1099511628211ull; };
        mix(ch0), mix(nch), mix((uint64_t)(uintptr_t)stream), mix((uint64_t)n_cls), mix(res.cnt_quiet);
        for (int i = 0; i < n_cls; i++) mix(((uint64_t)cls[i].S << 40) | ((uint64_t)cls[i].H << 32) | res.cnt_SH[cls[i].S][cls[i].H]);
        for (uint32_t i = 0; i < nch; i++) mix(h_list[i]);  // (no planned SKIP / SEQ channels here: the lists hold all nch)
        uint32_t listed = res.cnt_quiet;
        for (int i = 0; i < n_cls; i++) listed += res.cnt_SH[cls[i].S][cls[i].H];
        if (listed != nch)  // (channels without a packet this call: they belong to no stream -- the joined way)
            deferred = false;
    }
    if (h->deferred_pending && !(deferred && sig == h->deferred_sig)) {
        // a channel may be about to change streams: everything the side streams carry first
        // (its own earlier calls on another stream: the range logic above has made this stream wait for them and their side streams)
        PSK_HIP(deferred_join(h, stream));
    }
    {
        uint32_t *hdr = psk::plan_header(h->h_plans[slot]);
        hdr[0] = 0u;                       // channels handed over: counted by the kernels
        hdr[1] = res.any_seq ? 1u : 0u;    // channels planned for the reference-order kernel
    }
    const size_t up_bytes = psk::kPlanHeaderBytes + (sizeof(psk::ChanPlan) + sizeof(uint32_t)) * nch;
    if (h->opt_up_stream) {
        // (the slot's previous user has finished -- waited for above --, nothing else reads or writes d_plans[slot])
        PSK_HIP(hipMemcpyAsync(psk::plan_header(h->d_plans[slot]), psk::plan_header(h->h_plans[slot]), up_bytes,
                               hipMemcpyHostToDevice, h->up_stream));
        PSK_HIP(hipEventRecord(h->ev_up[slot], h->up_stream));
        PSK_HIP(hipStreamWaitEvent(stream, h->ev_up[slot], 0));
    } else {
        PSK_HIP(hipMemcpyAsync(psk::plan_header(h->d_plans[slot]), psk::plan_header(h->h_plans[slot]), up_bytes,
                               hipMemcpyHostToDevice, stream));
    }
    // phase ring of a launch: a power of two >= phaseAvg + 128 for its channels, at least 512 floats (256 where
    // the energy ring is dynamic too and every byte of LDS counts towards residency)
    auto ring_floats = [](uint32_t n_max, uint32_t at_least) {
        uint32_t y = at_least;
        while (y < n_max + 128u) y <<= 1;
        return y;
    };
    // PSK_SOFT_TRACE_LAUNCHES=1 (debugging a faulting kernel): in front of every launch the host waits for everything enqueued so
    // far and writes one line for the launch and one per channel of its list to stderr -- the last launch named in the log of a
    // run that died is the one that did it, with the shapes it was given.  (=2: the launch lines only.)
    auto mark = [&](const char *what, int S, int H, uint32_t off, uint32_t cnt, uint32_t tiles, uint32_t y_len, uint32_t r_len) -> hipError_t {
        if (!h->opt_trace)
            return hipSuccess;
        if (const hipError_t e = hipDeviceSynchronize())
            return e;
        std::fprintf(stderr, "[psk_soft] ok; next: %s S=%d H=%d ch0=%u cnt=%u tiles=%u y_len=%u r_len=%u slot=%d stream=%p\n", what, S, H, ch0, cnt,
                     tiles, y_len, r_len, slot, (void *)stream);
        for (uint32_t i = 0; i < cnt && h->opt_trace == 1; i++) {
            const uint32_t bi = stamped || off == ~0u ? i : h_list[off + i];
            const psk::ChanPlan &p = plans[bi];
            if (off == ~0u && p.mode == psk::PLAN_SKIP)
                continue;
            std::fprintf(stderr, "[psk_soft]   ch %u mode=%u S=%u A=%u M=%u n=%u len0=%u n_out=%llu n_in=%llu L0=%u L1=%u flags=0x%x K=%u tbase=%u toff=%llu in=%p\n",
                         ch0 + bi, p.mode, p.S, p.A, p.M, p.lf_n, p.lf_len0, (unsigned long long)p.n_out, (unsigned long long)p.n_in, p.ring_len0,
                         p.ring_len1, p.lf_flags, p.tile_blocks, p.tile_base, (unsigned long long)p.tile_off, (const void *)p.in);
        }
        std::fflush(stderr);
        return hipSuccess;
    };
    // (timing experiments only -- PSK_SOFT_DIAG_NO_TAIL=1: the exact-timing and reference-order launches behind the screened tier are
    // left out, which is wrong as soon as a call is handed over; what the two launches cost a small call is measured that way)
    static const bool diag_no_tail = std::getenv("PSK_SOFT_DIAG_NO_TAIL") && std::atoi(std::getenv("PSK_SOFT_DIAG_NO_TAIL")) != 0;
    auto enqueue = [&]() -> psk_soft_status {
        if (any_quiet)
            PSK_HIP(mark("fast<0,1> (calls that emit nothing)", 0, 1, off_quiet, res.cnt_quiet, 0, ring_floats(res.max_n_quiet, 512u), 0));
        if (any_quiet)
            PSK_HIP(psk::launch_fast(0, 1, 0, h->d_plans[slot], d_list + off_quiet, ch0, res.cnt_quiet, h->d_state, h->d_ring, h->lim.ring_cap, h->d_yv,
                                     h->lim.fit_cap, ring_floats(res.max_n_quiet, 512u), 0u, stream));
        // Per window class (samplesPerBaud, history depth): screened timing first, then the exact-timing
        // instantiation, which picks up the calls the screening refused; the reference-order kernel (below) takes
        // the calls both refused.  A batch that mixes classes runs them SIDE BY SIDE: each class is its own
        // instantiation with its own register and LDS appetite (numAvg <= 128: 16 waves per CU; numAvg 400: 8), and
        // one after the other each would leave part of the machine idle.  The classes go to side streams forked off
        // the caller's stream behind the plan upload and joined again in front of the reference-order kernel; the
        // deepest histories (fewest waves per CU, longest tails) are launched first.
        if (res.cnt_any) {
            const uint32_t y_len = ring_floats(res.max_n_any, 512u);
            PSK_HIP(mark("tile_front_any", (int)res.max_S_any, 0, off_any, res.cnt_any, tiles_max_any, y_len, 0));
            PSK_HIP(psk::launch_tile_front_any(h->d_plans[slot], d_list + off_any, ch0, res.cnt_any, tiles_max_any, res.max_S_any, h->d_state, h->d_ring,
                                               h->lim.ring_cap, h->d_tiles, h->d_traw, h->d_ts, h->pf.chan, stream));
            if (h->opt_pfit)
                PSK_HIP(mark("pfit (any)", (int)res.max_S_any, 0, off_any, res.cnt_any, tiles_max_any, y_len, pf_second));
            if (h->opt_pfit)
                PSK_HIP(psk::launch_pfit(h->d_plans[slot], d_list + off_any, ch0, res.cnt_any, tiles_max_any, h->d_state, h->d_ring,
                                         h->lim.ring_cap, h->d_yv, h->lim.fit_cap, y_len, h->d_tiles, h->d_traw, h->d_ts, h->d_test, h->pf,
                                         pf_second, stream));
            PSK_HIP(mark("tile_fit (any)", (int)res.max_S_any, 0, off_any, res.cnt_any, tiles_max_any, y_len, 0));
            PSK_HIP(psk::launch_tile_fit(h->d_plans[slot], d_list + off_any, ch0, res.cnt_any, h->d_state, h->d_ring, h->lim.ring_cap,
                                         h->d_yv, h->lim.fit_cap, y_len, h->d_tiles, h->d_traw, h->d_ts, h->d_test, h->pf, stream));
            PSK_HIP(mark("tile_back (any)", (int)res.max_S_any, 0, off_any, res.cnt_any, tiles_max_any, y_len, 0));
            PSK_HIP(psk::launch_tile_back(h->d_plans[slot], d_list + off_any, ch0, res.cnt_any, tiles_max_any, h->d_state, h->d_tiles,
                                          h->d_ts, h->d_test, 0u, 0u, stream));
        }
        if (any_quiet && deferred)  // (every launch set ends its own calls: the quiet channels' on the caller's stream)
            PSK_HIP(psk::launch_seq(h->d_plans[slot], d_list + off_quiet, ch0, res.cnt_quiet, h->d_state, h->d_ring, h->lim.ring_cap, h->d_yv,
                                    h->lim.fit_cap, stream));
        if (fork) {
            if (!h->aux_fork) {
                PSK_HIP(hipEventCreateWithFlags(&h->aux_fork, hipEventDisableTiming));
                // (streams of one priority share a few hardware queues -- four unless GPU_MAX_HW_QUEUES says otherwise -- and two
                // streams that land on the same one run their kernels one after the other: measured, the classes of the mixed
                // batch did, 1.75 + 1.78 ms.  A stream of another priority gets a queue of its own.)
                int prio_lo = 0, prio_hi = 0;
                PSK_HIP(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
                for (int k = 0; k < kAuxStreams; k++) {
                    const char *pe = getenv("PSK_SOFT_AUX_PRIO");
                    const int mode = pe ? atoi(pe) : 1;
                    if (mode == 0 || prio_lo == prio_hi)
                        PSK_HIP(hipStreamCreateWithFlags(&h->aux[k], hipStreamNonBlocking));
                    else
                        PSK_HIP(hipStreamCreateWithPriority(&h->aux[k], hipStreamNonBlocking, mode == 1 ? prio_hi : prio_lo));
                    PSK_HIP(hipEventCreateWithFlags(&h->aux_join[k], hipEventDisableTiming));
                }
            }
            PSK_HIP(hipEventRecord(h->aux_fork, stream));
        }
        int used_aux = 0;
        for (int i = 0; i < n_cls; i++) {
            const int S = cls[i].S, H = cls[i].H;
            // class 0 stays on the caller's stream, the others take the side streams in turn
            hipStream_t st = stream;
            if (fork && i > 0) {
                const int a = (i - 1) % kAuxStreams;
                st = h->aux[a];
                if (i - 1 < kAuxStreams) {
                    PSK_HIP(hipStreamWaitEvent(st, h->aux_fork, 0));
                    used_aux = i;
                }
            }
            const uint32_t y_len = ring_floats(max_n[S][H], psk::ering_dynamic(S) ? 256u : 512u);
            const uint32_t r_len = class_H(H) == 1 ? ((max_A[S][H] + 128u + 1u) & ~1u) : 0u;
            if (tiled_SH[S][H] && piped_SH[S][H]) {
                // pipelined: front(range j) here, fit(range j) on a second stream behind it -- under front(range j + 1) --,
                // back(range j) on a third behind that (psk_tile.hip: psk_tile_fit_range_kernel)
                const uint32_t cnt = res.cnt_SH[S][H], T = tiles_max_SH[S][H], tr = pipe_tiles_SH[S][H];
                const uint32_t *const l = d_list + off_SH[S][H];
                char *const carry = static_cast<char *>(h->d_pipe_carry) + psk::pipe_carry_bytes() * off_SH[S][H];
                float *const carry_y = h->d_pipe_y + (size_t)kPipeMaxYLen * off_SH[S][H];
                const uint32_t y_pipe = ring_floats(max_n[S][H], 512u);
                int e = 0;
                PSK_HIP(hipEventRecord(h->pipe_ev[e], st));  // (the side streams start behind everything this one carries)
                PSK_HIP(hipStreamWaitEvent(h->pipe_st[0], h->pipe_ev[e], 0));
                PSK_HIP(hipStreamWaitEvent(h->pipe_st[1], h->pipe_ev[e], 0));
                e++;
                for (uint32_t t0 = 0; t0 < T; t0 += tr) {
                    const uint32_t nt = T - t0 < tr ? T - t0 : tr;
                    PSK_HIP(psk::launch_tile_front(S, class_H(H), h->d_plans[slot], l, ch0, cnt, nt, h->d_state, h->d_ring, h->lim.ring_cap, r_len,
                                                   h->d_tiles, h->d_traw, h->d_ts, h->pf.chan, t0, st));
                    PSK_HIP(hipEventRecord(h->pipe_ev[e], st));
                    PSK_HIP(hipStreamWaitEvent(h->pipe_st[0], h->pipe_ev[e], 0));
                    e++;
                    PSK_HIP(psk::launch_tile_fit_range(h->d_plans[slot], l, ch0, cnt, h->d_state, h->d_ring, h->lim.ring_cap, h->d_yv,
                                                       h->lim.fit_cap, y_pipe, h->d_tiles, h->d_traw, h->d_ts, h->d_test, carry, carry_y, t0, nt,
                                                       h->pipe_st[0]));
                    PSK_HIP(hipEventRecord(h->pipe_ev[e], h->pipe_st[0]));
                    PSK_HIP(hipStreamWaitEvent(h->pipe_st[1], h->pipe_ev[e], 0));
                    e++;
                    PSK_HIP(psk::launch_tile_back(h->d_plans[slot], l, ch0, cnt, nt, h->d_state, h->d_tiles, h->d_ts, h->d_test, t0, 1u,
                                                  h->pipe_st[1]));
                }
                PSK_HIP(hipEventRecord(h->pipe_ev[e], h->pipe_st[1]));  // (behind the last fit too: the last back waited for it)
                PSK_HIP(hipStreamWaitEvent(st, h->pipe_ev[e], 0));
            } else if (tiled_SH[S][H]) {
                // (a call these cannot carry comes out with guard 1 and nothing committed: the launches below redo it)
                PSK_HIP(mark("tile_front", S, H, off_SH[S][H], res.cnt_SH[S][H], tiles_max_SH[S][H], y_len, r_len));
                PSK_HIP(psk::launch_tile_front(S, class_H(H), h->d_plans[slot], d_list + off_SH[S][H], ch0, res.cnt_SH[S][H], tiles_max_SH[S][H],
                                               h->d_state, h->d_ring, h->lim.ring_cap, r_len, h->d_tiles, h->d_traw, h->d_ts,
                                               h->pf.chan, 0u, st));
                if (h->opt_pfit)
                    PSK_HIP(mark("pfit", S, H, off_SH[S][H], res.cnt_SH[S][H], tiles_max_SH[S][H], y_len, pf_second));
                if (h->opt_pfit)
                    PSK_HIP(psk::launch_pfit(h->d_plans[slot], d_list + off_SH[S][H], ch0, res.cnt_SH[S][H], tiles_max_SH[S][H], h->d_state,
                                             h->d_ring, h->lim.ring_cap, h->d_yv, h->lim.fit_cap, y_len, h->d_tiles, h->d_traw, h->d_ts,
                                             h->d_test, h->pf, pf_second, st));
                PSK_HIP(mark("tile_fit", S, H, off_SH[S][H], res.cnt_SH[S][H], tiles_max_SH[S][H], y_len, r_len));
                PSK_HIP(psk::launch_tile_fit(h->d_plans[slot], d_list + off_SH[S][H], ch0, res.cnt_SH[S][H], h->d_state, h->d_ring,
                                             h->lim.ring_cap, h->d_yv, h->lim.fit_cap, y_len, h->d_tiles, h->d_traw, h->d_ts, h->d_test,
                                             h->pf, st));
                PSK_HIP(mark("tile_back", S, H, off_SH[S][H], res.cnt_SH[S][H], tiles_max_SH[S][H], y_len, r_len));
                PSK_HIP(psk::launch_tile_back(h->d_plans[slot], d_list + off_SH[S][H], ch0, res.cnt_SH[S][H], tiles_max_SH[S][H], h->d_state,
                                              h->d_tiles, h->d_ts, h->d_test, 0u, 0u, st));
            }
            // (the exact tier only works on the calls the tier in front of it left; behind the time-tiled kernels, whose front
            // stage IS the screened timing, it is the exact tier that picks up what they hand over)
            for (int exact = tiled_SH[S][H] ? 1 : 0; exact <= 1; exact++) {
                if (exact && diag_no_tail)
                    break;
                PSK_HIP(mark(exact ? "fast (exact tier)" : "fast (screened tier)", S, H, off_SH[S][H], res.cnt_SH[S][H], 0, y_len, r_len));
                PSK_HIP(psk::launch_fast(S, class_H(H), exact, h->d_plans[slot], d_list + off_SH[S][H], ch0, res.cnt_SH[S][H], h->d_state,
                                         h->d_ring, h->lim.ring_cap, h->d_yv, h->lim.fit_cap, y_len, r_len, st));
            }
            if (deferred)  // (the class's hand-overs are redone on its own stream, in front of its next call)
                PSK_HIP(psk::launch_seq(h->d_plans[slot], d_list + off_SH[S][H], ch0, res.cnt_SH[S][H], h->d_state, h->d_ring, h->lim.ring_cap,
                                        h->d_yv, h->lim.fit_cap, st));
        }
        if (deferred) {
            for (int a = 0; a < used_aux && a < kAuxStreams; a++) {
                if (!h->slot_aux_ev[slot][a])
                    PSK_HIP(hipEventCreateWithFlags(&h->slot_aux_ev[slot][a], hipEventDisableTiming));
                PSK_HIP(hipEventRecord(h->slot_aux_ev[slot][a], h->aux[a]));
                h->slot_aux_used[slot][a] = true;
            }
            h->deferred_pending = true;
            h->deferred_sig = sig;
            h->deferred_stream = stream;
        } else {
            for (int a = 0; a < used_aux && a < kAuxStreams; a++) {
                PSK_HIP(hipEventRecord(h->aux_join[a], h->aux[a]));
                PSK_HIP(hipStreamWaitEvent(stream, h->aux_join[a], 0));
            }
            if (any_seq || any_emit)
                PSK_HIP(mark("seq (reference order)", 0, 0, ~0u, nch, 0, 0, 0));
            if ((any_seq || any_emit) && !diag_no_tail)  // any_emit: the exactness guard may hand calls over at run time
                PSK_HIP(psk::launch_seq(h->d_plans[slot], nullptr, ch0, nch, h->d_state, h->d_ring, h->lim.ring_cap, h->d_yv,
                                        h->lim.fit_cap, stream));
        }
        if (tile_syms)
            PSK_HIP(hipEventRecord(h->tile_ev, stream));
        PSK_HIP(hipEventRecord(h->ev[slot], stream));
        return PSK_SOFT_OK;
    };
    const psk_soft_status est = enqueue();
    if (est != PSK_SOFT_OK) {
        h->poisoned = true;
        return est;
    }
    commit();
    if (tile_syms) {
        h->tile_ev_used = true;
        h->tile_stream = stream;
    }
    h->slot = (h->slot + 1) % kPlanSlots;
    h->ev_used[slot] = true;
    h->slot_stream[slot] = stream;
    h->slot_ch0[slot] = ch0;
    h->slot_nch[slot] = nch;
    return PSK_SOFT_OK;
}

// The public entry.  A call that would emit more than 2^20 symbols in one channel, or run LinearFit::count past 2^20 in the
// middle (the reference then rebuilds the fit's sums at that symbol, cpp/psk_soft.cpp:51-52, and its energy sums after it,
// :582-583), is cut at those boundaries inside the library: the pieces are planned as continuations of ONE serviceFunction() call
// (no prologue between them) and run on the wave-scan / time-tiled kernels like any other call -- round 2 handed such calls to the
// reference-order kernel, 4.7 us per symbol on one lane.  Everything else goes straight through.
psk_soft_status psk_soft_process_device(psk_soft_handle_t *h, uint32_t ch0, uint32_t nch,
                                        const psk_soft_packet_t *pkts, psk_soft_output_t *outs, void *stream_v)
{
    if (!h || !pkts || !outs || !nch || (uint64_t)ch0 + nch > h->nch)
        return fail(PSK_SOFT_ERR_INVALID_ARG, "psk_soft_process: bad arguments");
    // (the ordinary call is planned as it is; the plan pass itself says when a channel's call is too long for one piece --
    // nothing is committed or enqueued then)
    g_long_call = false;
    g_split_pieces = 0;
    {
        const psk_soft_status st = process_round(h, ch0, nch, pkts, outs, stream_v, nullptr);
        if (st != PSK_SOFT_OK || (!g_long_call && g_split_pieces < 2))
            return st;
        g_long_call = false;
    }
    // (a mixed batch cut in time, see process_round: every channel's call in `split` pieces of whole blocks; the side streams of
    // the rounds are joined once, behind the last)
    const uint64_t split = g_split_pieces > 1 ? (uint64_t)g_split_pieces : 0;
    g_split_pieces = 0;
    struct SplitScope {
        bool on;
        explicit SplitScope(bool v) : on(v) { g_split_mode = v; }
        ~SplitScope() { g_split_mode = false; }
    } split_scope(split != 0);
    std::vector<uint64_t> piece_cap(split ? nch : 0, ~0ull);  // symbols a piece of the channel's call may emit

    // ---- pieces ----
    ctl_sync(h);
    std::vector<psk_soft_packet_t> pk(pkts, pkts + nch);
    std::vector<psk_soft_output_t> ou(outs, outs + nch), total(outs, outs + nch);
    std::vector<uint64_t> left(nch);   // floats of the packet not yet handed over
    std::vector<uint8_t> cont(nch, 0);
    std::vector<uint32_t> mode_of(nch, psk::PLAN_SKIP);  // (statistics: the kernel of each channel's last piece)
    for (uint32_t i = 0; i < nch; i++) left[i] = pkts[i].present ? pkts[i].n_floats : 0;
    for (int round = 0;; round++) {
        bool more = false;
        for (uint32_t i = 0; i < nch; i++) {
            psk_soft_packet_t &q = pk[i];
            if (round > 0) {
                q.present = left[i] ? 1 : 0;  // (a channel whose packet is used up sits the later rounds out)
                q.sriChanged = 0;
                q.inputQueueFlushed = 0;
            }
            cont[i] = round > 0 ? (split ? 5 : 1) : 0;  // (bit 2: the bounds of the call so far carry over, PLAN_CARRY_DRIFT)
            if (!q.present || q.sri_mode != 1) {
                left[i] = 0;
                continue;
            }
            const psk::ChanCtl &c = h->ctl[ch0 + i];
            const uint64_t S = c.props.samplesPerBaud ? c.props.samplesPerBaud : 1, A = c.props.numAvg;
            // the first round runs the call's prologue, which (quirk Q2) resets LinearFit::count in practically every call: plan on a
            // copy to learn what this piece would emit, and shorten it to an even number of symbols within the limit (the output
            // rows of the next piece must stay 4-byte aligned: bits and sampleIndex are 2 bytes a symbol)
            uint64_t n_fl = left[i];
            for (int attempt = 0; attempt < 4; attempt++) {
                psk::ChanCtl probe = c;
                psk::ChanPlan pl;
                psk_soft_packet_t qq = q;
                qq.n_floats = n_fl;
                psk_soft_output_t oo = ou[i];
                oo.cap_symbols = ~0ull;
                if (psk::plan_call(probe, h->lim, qq, oo, pl, (cont[i] & 1u) != 0) != PSK_SOFT_OK)
                    break;  // (the real pass reports it)
                uint64_t lim_sym = psk::kResyncCount - pl.lf_count0 < psk::kResyncCount ? psk::kResyncCount - pl.lf_count0 : psk::kResyncCount;
                if (split) {
                    if (round == 0 && attempt == 0)  // (the whole call's symbols: pieces of whole blocks, `split` of them)
                        piece_cap[i] = ((oo.n_symbols + split - 1) / split + 127ull) & ~127ull;
                    lim_sym = piece_cap[i] < lim_sym ? piece_cap[i] : lim_sym;
                }
                const bool last = n_fl == left[i];
                if (oo.n_symbols <= lim_sym && (last || (oo.n_symbols & 1ull) == 0))
                    break;
                // too many (or an odd number of) symbols: give the piece fewer samples
                uint64_t want = oo.n_symbols > lim_sym ? lim_sym : oo.n_symbols - 1;
                want &= ~1ull;
                const uint64_t excess = oo.n_symbols - want;
                const uint64_t cut = 2ull * S * excess;
                n_fl = n_fl > cut ? n_fl - cut : 2ull * S * (A + 2);
                (void)A;
            }
            q.n_floats = n_fl;
            left[i] -= n_fl < left[i] ? n_fl : left[i];
            if (left[i])
                cont[i] |= 2u;
            more = more || left[i] != 0;
        }
        const psk_soft_status st = process_round(h, ch0, nch, pk.data(), ou.data(), stream_v, cont.data());
        if (st != PSK_SOFT_OK) {
            if (round > 0)
                h->poisoned = true;  // (pieces of the call have run: the channels are in the middle of it)
            return st;
        }
        ctl_sync(h);
        for (uint32_t i = 0; i < nch; i++) {
            const psk_soft_output_t &o = ou[i];
            psk_soft_output_t &t = total[i];
            if (round == 0) {
                t = o;
                t.soft = outs[i].soft, t.bits = outs[i].bits, t.phase = outs[i].phase, t.sampleIndex = outs[i].sampleIndex;
                t.cap_symbols = outs[i].cap_symbols;
            } else if (pk[i].present) {
                t.n_symbols += o.n_symbols;
                t.n_bits += o.n_bits;
                t.n_sampleIndex += o.n_sampleIndex;
                t.n_warn += o.n_warn;
            }
            // the next piece reads and writes behind this one
            if (pk[i].present) {
                mode_of[i] = h->last_mode[ch0 + i];
                pk[i].data = pk[i].data ? pk[i].data + pk[i].n_floats : nullptr;
                pk[i].n_floats = left[i];
                if (ou[i].soft) ou[i].soft += 2 * o.n_symbols;
                if (ou[i].bits) ou[i].bits += o.n_bits;
                if (ou[i].phase) ou[i].phase += o.n_symbols;
                if (ou[i].sampleIndex) ou[i].sampleIndex += o.n_sampleIndex;
                ou[i].cap_symbols = ou[i].cap_symbols > o.n_symbols ? ou[i].cap_symbols - o.n_symbols : 0;
            }
        }
        if (!more)
            break;
    }
    ctl_sync(h);
    for (uint32_t i = 0; i < nch; i++) {
        outs[i] = total[i];
        h->last_mode[ch0 + i] = mode_of[i];
    }
    if (split && !h->dry && !h->opt_deferred) {  // (the call ends like any other: the caller's stream behind all of it)
        PSK_HIP(hipSetDevice(h->device));
        PSK_HIP(deferred_join(h, stream_v ? (hipStream_t)stream_v : h->stream));
    }
    return PSK_SOFT_OK;
}

// ---- host-buffer path: the ingest pipeline --------------------------
// The batch is cut into chunks of channels of about `stage_bytes` of input each.  Per chunk: the
// packets are packed into pinned memory (CopyPool), uploaded with ONE copy, processed, and the four
// output streams -- packed tightly, their sizes are known from the plan -- come back with one copy
// each and are scattered to the caller's buffers.  kStageSlots chunks are in flight on separate
// streams, so upload, kernels, download and the two host-side copies of different chunks overlap.
static psk_soft_status stage_ensure(psk_soft_handle *h, StageSlot &sl, size_t in_bytes)
{
    if (!sl.stream) {
        PSK_HIP(hipStreamCreateWithFlags(&sl.stream, hipStreamNonBlocking));
        PSK_HIP(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
    }
    if (sl.in_cap >= in_bytes)
        return PSK_SOFT_OK;
    size_t cap = align_up(in_bytes > h->stage_bytes ? in_bytes : h->stage_bytes, 4096);
    if (sl.h_buf) (void)hipHostFree(sl.h_buf);
    if (sl.d_buf) (void)hipFree(sl.d_buf);
    sl.h_buf = nullptr;
    sl.d_buf = nullptr;
    sl.in_cap = 0;
    PSK_HIP(hipHostMalloc((void **)&sl.h_buf, region_total(cap)));
    PSK_HIP(hipMalloc((void **)&sl.d_buf, region_total(cap)));
    sl.in_cap = cap;
    return PSK_SOFT_OK;
}

// results of the chunk in `sl` -> the caller's buffers
static psk_soft_status stage_retire(psk_soft_handle *h, StageSlot &sl, psk_soft_output_t *outs, uint32_t batch_ch0)
{
    if (!sl.busy)
        return PSK_SOFT_OK;
    PSK_HIP(hipEventSynchronize(sl.done));
    const uint8_t *hs = sl.h_buf + region_soft(sl.in_cap), *hp = sl.h_buf + region_phase(sl.in_cap);
    const uint8_t *hb = sl.h_buf + region_bits(sl.in_cap), *hx = sl.h_buf + region_sidx(sl.in_cap);
    psk_soft_output_t *o0 = outs + (sl.ch0 - batch_ch0);
    h->pool->run(sl.nch, [&](uint32_t i) {
        const psk_soft_output_t &o = o0[i];
        if (!o.n_symbols)
            return;
        if (o.soft) std::memcpy(o.soft, hs + sl.off_soft[i], sizeof(float) * 2 * o.n_symbols);
        if (o.phase) std::memcpy(o.phase, hp + sl.off_phase[i], sizeof(float) * o.n_symbols);
        if (o.bits && o.n_bits) std::memcpy(o.bits, hb + sl.off_bits[i], sizeof(int16_t) * o.n_bits);
        if (o.sampleIndex && o.n_sampleIndex) std::memcpy(o.sampleIndex, hx + sl.off_sidx[i], sizeof(int16_t) * o.n_sampleIndex);
    });
    sl.busy = false;
    return PSK_SOFT_OK;
}

psk_soft_status psk_soft_process_host(psk_soft_handle_t *h, uint32_t ch0, uint32_t nch, const psk_soft_packet_t *pkts,
                                      psk_soft_output_t *outs)
{
    if (!h || !pkts || !outs || !nch || (uint64_t)ch0 + nch > h->nch)
        return fail(PSK_SOFT_ERR_INVALID_ARG, "psk_soft_process_host: bad arguments");
    if (h->dry) {
        std::vector<psk_soft_packet_t> pk(pkts, pkts + nch);
        return psk_soft_process_device(h, ch0, nch, pk.data(), outs, nullptr);
    }
    PSK_HIP(hipSetDevice(h->device));
    if (!h->pool) {
        int nt = 8;
        if (const char *e = std::getenv("PSK_SOFT_HOST_THREADS")) nt = std::atoi(e);
        unsigned hw = std::thread::hardware_concurrency();
        if (hw && (unsigned)nt > hw) nt = (int)hw;
        h->pool = new CopyPool(nt > 1 ? nt - 1 : 0);  // the calling thread works too
        if (const char *e = std::getenv("PSK_SOFT_STAGE_MB")) {
            long mb = std::atol(e);
            if (mb >= 1 && mb <= 4096) h->stage_bytes = (size_t)mb << 20;
        }
    }
    // validate the whole batch first (all or nothing, as psk_soft_process_device), and learn the
    // output sizes: they depend only on packet sizes and properties
    struct Need {
        size_t in, soft, phase, bits, sidx;
    };
    std::vector<Need> need(nch);
    ctl_sync(h);
    for (uint32_t i = 0; i < nch; i++) {
        if (pkts[i].present && pkts[i].n_floats / 2 > h->user.max_packet_complex)
            return fail(PSK_SOFT_ERR_LIMIT, "psk_soft_process_host: packet longer than max_packet_complex");
        if (pkts[i].present && pkts[i].n_floats && !pkts[i].data)
            return fail(PSK_SOFT_ERR_INVALID_ARG, "psk_soft_process_host: null packet data");
        psk::ChanCtl probe = h->ctl[ch0 + i];
        psk::ChanPlan pl;
        psk_soft_output_t o = outs[i];
        o.cap_symbols = ~0ull;
        psk_soft_status st = psk::plan_call(probe, h->lim, pkts[i], o, pl);
        if (st != PSK_SOFT_OK) {
            char buf[160];
            std::snprintf(buf, sizeof buf, "psk_soft_process_host: channel %u refused (status %d)", ch0 + i, (int)st);
            return fail(st, buf);
        }
        if (o.n_symbols > outs[i].cap_symbols)
            return fail(PSK_SOFT_ERR_CAPACITY, "psk_soft_process_host: output buffer too small");
        // every channel's rows start on a cache line: rows that straddle lines cost 6-8 % of the
        // kernel's streaming rate (tools/micro/placement_probe.hip)
        need[i].in = pkts[i].present ? align_up(sizeof(float) * (pkts[i].n_floats & ~1ull), 128) : 0;
        need[i].soft = align_up(sizeof(float) * 2 * o.n_symbols, 128);
        need[i].phase = align_up(sizeof(float) * o.n_symbols, 128);
        need[i].bits = align_up(sizeof(int16_t) * o.n_bits, 128);
        need[i].sidx = align_up(sizeof(int16_t) * o.n_sampleIndex, 128);
    }
    std::vector<psk_soft_packet_t> dp;
    std::vector<psk_soft_output_t> dout;
    uint32_t first = 0;
    int turn = 0;
    psk_soft_status status = PSK_SOFT_OK;
    while (first < nch && status == PSK_SOFT_OK) {
        // chunk [first, last): as many channels as fit the five regions of a slot
        Need sum = {0, 0, 0, 0, 0};
        const size_t cap = h->stage_bytes;
        uint32_t last = first;
        while (last < nch) {
            const Need &n = need[last];
            if (last > first && (sum.in + n.in > cap || sum.soft + n.soft > cap || sum.phase + n.phase > cap / 2 ||
                                 sum.bits + n.bits > cap || sum.sidx + n.sidx > cap / 4))
                break;
            sum.in += n.in;
            sum.soft += n.soft;
            sum.phase += n.phase;
            sum.bits += n.bits;
            sum.sidx += n.sidx;
            last++;
        }
        // a single oversized packet gets a slot grown to fit it
        size_t want = sum.in;
        if (sum.soft > want) want = sum.soft;
        if (2 * sum.phase > want) want = 2 * sum.phase;
        if (sum.bits > want) want = sum.bits;
        if (4 * sum.sidx > want) want = 4 * sum.sidx;
        StageSlot &sl = h->stage[turn];
        turn = (turn + 1) % kStageSlots;
        if ((status = stage_retire(h, sl, outs, ch0)) != PSK_SOFT_OK)
            break;
        if ((status = stage_ensure(h, sl, want)) != PSK_SOFT_OK)
            break;
        const uint32_t n = last - first;
        sl.ch0 = ch0 + first;
        sl.nch = n;
        sl.off_soft.resize(n);
        sl.off_phase.resize(n);
        sl.off_bits.resize(n);
        sl.off_sidx.resize(n);
        dp.assign(pkts + first, pkts + last);
        dout.assign(outs + first, outs + last);
        std::vector<size_t> off_in(n);
        size_t oi = 0, os = 0, op = 0, ob = 0, ox = 0;
        uint8_t *d_in = sl.d_buf, *d_soft = sl.d_buf + region_soft(sl.in_cap), *d_phase = sl.d_buf + region_phase(sl.in_cap);
        uint8_t *d_bits = sl.d_buf + region_bits(sl.in_cap), *d_sidx = sl.d_buf + region_sidx(sl.in_cap);
        for (uint32_t i = 0; i < n; i++) {
            const Need &nd = need[first + i];
            off_in[i] = oi;
            sl.off_soft[i] = os;
            sl.off_phase[i] = op;
            sl.off_bits[i] = ob;
            sl.off_sidx[i] = ox;
            dp[i].data = (const float *)(d_in + oi);
            dout[i].soft = (float *)(d_soft + os);
            dout[i].phase = (float *)(d_phase + op);
            dout[i].bits = (int16_t *)(d_bits + ob);
            dout[i].sampleIndex = (int16_t *)(d_sidx + ox);
            dout[i].cap_symbols = ~0ull;
            oi += nd.in;
            os += nd.soft;
            op += nd.phase;
            ob += nd.bits;
            ox += nd.sidx;
        }
        sl.soft_bytes = os;
        sl.phase_bytes = op;
        sl.bits_bytes = ob;
        sl.sidx_bytes = ox;
        // pack, upload, process, download
        const psk_soft_packet_t *pk0 = pkts + first;
        uint8_t *h_in = sl.h_buf;
        h->pool->run(n, [&](uint32_t i) {
            if (pk0[i].present && pk0[i].n_floats)
                std::memcpy(h_in + off_in[i], pk0[i].data, sizeof(float) * (pk0[i].n_floats & ~1ull));
        });
        if (oi)
            PSK_HIP(hipMemcpyAsync(d_in, h_in, oi, hipMemcpyHostToDevice, sl.stream));
        status = psk_soft_process_device(h, ch0 + first, n, dp.data(), dout.data(), sl.stream);
        if (status == PSK_SOFT_OK)  // (deferred join: the downloads below read what the side streams write)
            PSK_HIP(deferred_join(h, sl.stream));
        if (status != PSK_SOFT_OK)
            break;
        for (uint32_t i = 0; i < n; i++) {
            psk_soft_output_t &o = outs[first + i];
            const psk_soft_output_t &d = dout[i];
            o.ret = d.ret;
            o.n_symbols = d.n_symbols;
            o.n_bits = d.n_bits;
            o.n_sampleIndex = d.n_sampleIndex;
            o.sri_pushed = d.sri_pushed;
            o.sri_soft_xdelta = d.sri_soft_xdelta;
            o.sri_bits_xdelta = d.sri_bits_xdelta;
            o.n_warn = d.n_warn;
        }
        if (os) PSK_HIP(hipMemcpyAsync(sl.h_buf + region_soft(sl.in_cap), d_soft, os, hipMemcpyDeviceToHost, sl.stream));
        if (op) PSK_HIP(hipMemcpyAsync(sl.h_buf + region_phase(sl.in_cap), d_phase, op, hipMemcpyDeviceToHost, sl.stream));
        if (ob) PSK_HIP(hipMemcpyAsync(sl.h_buf + region_bits(sl.in_cap), d_bits, ob, hipMemcpyDeviceToHost, sl.stream));
        if (ox) PSK_HIP(hipMemcpyAsync(sl.h_buf + region_sidx(sl.in_cap), d_sidx, ox, hipMemcpyDeviceToHost, sl.stream));
        PSK_HIP(hipEventRecord(sl.done, sl.stream));
        sl.busy = true;
        first = last;
    }
    // drain, oldest first
    for (int k = 0; k < kStageSlots; k++) {
        psk_soft_status st = stage_retire(h, h->stage[(turn + k) % kStageSlots], outs, ch0);
        if (status == PSK_SOFT_OK)
            status = st;
    }
    return status;
}

psk_soft_status psk_soft_synchronize(psk_soft_handle_t *h)
{
    if (!h)
        return fail(PSK_SOFT_ERR_INVALID_ARG, "null handle");
    if (h->dry)
        return PSK_SOFT_OK;
    PSK_HIP(hipSetDevice(h->device));
    for (int s = 0; s < kPlanSlots; s++)
        if (h->ev_used[s])
            PSK_HIP(hipEventSynchronize(h->ev[s]));
    PSK_HIP(hipStreamSynchronize(h->stream));
    for (int a = 0; a < kAuxStreams; a++)
        if (h->aux[a])
            PSK_HIP(hipStreamSynchronize(h->aux[a]));
    if (h->deferred_pending && h->deferred_stream)
        PSK_HIP(hipStreamSynchronize(h->deferred_stream));
    h->deferred_pending = false;
    for (auto &row : h->slot_aux_used)
        for (bool &u : row) u = false;
    for (auto &sl : h->stage)
        if (sl.stream)
            PSK_HIP(hipStreamSynchronize(sl.stream));
    return PSK_SOFT_OK;
}

psk_soft_status psk_soft_join(psk_soft_handle_t *h, void *stream_v)
{
    if (!h)
        return fail(PSK_SOFT_ERR_INVALID_ARG, "null handle");
    if (h->dry || !h->deferred_pending)
        return PSK_SOFT_OK;
    PSK_HIP(hipSetDevice(h->device));
    hipStream_t stream = stream_v ? (hipStream_t)stream_v : h->stream;
    // (the caller's stream of the deferred calls carries one class itself: a third stream waits for it too)
    if (h->deferred_stream && h->deferred_stream != stream) {
        if (!h->aux_fork)
            return PSK_SOFT_OK;
        PSK_HIP(hipEventRecord(h->aux_fork, h->deferred_stream));
        PSK_HIP(hipStreamWaitEvent(stream, h->aux_fork, 0));
    }
    h->deferred_pending = true;  // (deferred_join() clears it)
    PSK_HIP(deferred_join(h, stream));
    return PSK_SOFT_OK;
}

static void stats_add(psk_soft_stats_t *stats, uint32_t mode, const psk::ChanState &s)
{
    switch (mode) {
    case psk::PLAN_FAST:
        if (s.guard == 2u) {
            stats->channels_sequential++;
            stats->channels_guard++;
        } else {
            stats->channels_fast++;
            if (s.guard == 3u)
                stats->channels_exact_timing++;
            if (s.guard == 4u) {
                stats->channels_tiled++;
                if (s.stat_pfit & 1u) {
                    stats->channels_parallel_fit++;
                    if (s.stat_pfit & 0x100u)
                        stats->channels_parallel_fit_second_round++;
                } else {
                    stats->parallel_fit_refusals |= s.stat_pfit >> 1;
                }
            }
            stats->unwrap_blocks += s.stat_blocks;
            stats->unwrap_extra_passes += s.stat_extra;
            stats->timing_exact_blocks += s.stat_exact;
            stats->fit_chain_blocks += s.stat_chain;
        }
        break;
    case psk::PLAN_SEQ:
    case psk::PLAN_SEQ_S1: stats->channels_sequential++; break;
    default: break;
    }
}

static psk_soft_status stats_range(psk_soft_handle_t *h, uint32_t ch0, uint32_t nch, psk_soft_stats_t *stats, bool per_channel)
{
    if (!h || !stats || (uint64_t)ch0 + nch > h->nch)
        return fail(PSK_SOFT_ERR_INVALID_ARG, "psk_soft_get_stats: bad arguments");
    ctl_sync(h);
    std::memset(stats, 0, sizeof *stats * (per_channel ? nch : 1));
    if (h->dry) {  // control plane only: which kernel the last call was PLANNED for, per channel
        for (uint32_t c = 0; c < nch; c++) {
            psk_soft_stats_t *o = per_channel ? stats + c : stats;
            if (h->last_mode[ch0 + c] == psk::PLAN_FAST) o->channels_fast++;
            if (h->last_mode[ch0 + c] == psk::PLAN_SEQ || h->last_mode[ch0 + c] == psk::PLAN_SEQ_S1) o->channels_sequential++;
        }
        return PSK_SOFT_OK;
    }
    psk_soft_status st = psk_soft_synchronize(h);
    if (st != PSK_SOFT_OK)
        return st;
    std::vector<psk::ChanState> s(nch);
    PSK_HIP(hipMemcpy(s.data(), h->d_state + ch0, sizeof(psk::ChanState) * nch, hipMemcpyDeviceToHost));
    for (uint32_t c = 0; c < nch; c++) stats_add(per_channel ? stats + c : stats, h->last_mode[ch0 + c], s[c]);
    return PSK_SOFT_OK;
}

psk_soft_status psk_soft_get_stats(psk_soft_handle_t *h, psk_soft_stats_t *stats)
{
    return stats_range(h, 0, h ? h->nch : 0, stats, false);
}

psk_soft_status psk_soft_get_channel_stats(psk_soft_handle_t *h, uint32_t ch0, uint32_t nch, psk_soft_stats_t *stats)
{
    return stats_range(h, ch0, nch, stats, true);
}

psk_soft_status psk_soft_set_option(psk_soft_handle_t *h, int option, int value)
{
    if (!h)
        return fail(PSK_SOFT_ERR_INVALID_ARG, "null handle");
    switch (option) {
    case PSK_SOFT_OPT_QPSK_SIGN_BITMAP: h->opt_qpsk_sign_map = value != 0; return PSK_SOFT_OK;
    case PSK_SOFT_OPT_CONCURRENT_CLASSES: h->opt_fork = value != 0; return PSK_SOFT_OK;
    case PSK_SOFT_OPT_DEFERRED_JOIN:
        if (!value && h->deferred_pending && !h->dry) {
            const psk_soft_status st = psk_soft_synchronize(h);
            if (st != PSK_SOFT_OK)
                return st;
        }
        h->opt_deferred = value != 0;
        return PSK_SOFT_OK;
    case PSK_SOFT_OPT_TIME_TILED:
        if (value < 0 || value > 2)
            return fail(PSK_SOFT_ERR_INVALID_ARG, "psk_soft_set_option: PSK_SOFT_OPT_TIME_TILED takes 0, 1 or 2");
        h->opt_tiled = value;
        return PSK_SOFT_OK;
    case PSK_SOFT_OPT_PARALLEL_FIT:
        if (value < 0 || value > 2)
            return fail(PSK_SOFT_ERR_INVALID_ARG, "psk_soft_set_option: PSK_SOFT_OPT_PARALLEL_FIT takes 0, 1 or 2");
        h->opt_pfit = value;
        return PSK_SOFT_OK;
    default: return fail(PSK_SOFT_ERR_INVALID_ARG, "psk_soft_set_option: unknown option");
    }
}

psk_soft_status psk_soft_set_force_sequential(psk_soft_handle_t *h, int on)
{
    if (!h)
        return fail(PSK_SOFT_ERR_INVALID_ARG, "null handle");
    h->lim.force_seq = on != 0;
    return PSK_SOFT_OK;
}

// A saved channel state: header, control-plane mirror, device state, sample ring (the current buffer), phase history.
struct StateHeader {
    uint32_t magic, version;      // "PSKS", PSK_SOFT_ABI_VERSION
    uint32_t ring_cap, fit_cap;   // the limits the blob was written under: they size its two arrays
    uint32_t ctl_bytes, state_bytes;
};
constexpr uint32_t kStateMagic = 0x534B5350u;  // 'P' 'S' 'K' 'S'

uint64_t psk_soft_state_bytes(const psk_soft_handle_t *h)
{
    if (!h)
        return 0;
    return sizeof(StateHeader) + sizeof(psk::ChanCtl) + sizeof(psk::ChanState) + sizeof(float2) * (uint64_t)h->lim.ring_cap +
           sizeof(float) * (uint64_t)h->lim.fit_cap;
}

psk_soft_status psk_soft_export_state(psk_soft_handle_t *h, uint32_t ch, void *dst, uint64_t cap)
{
    if (!h || !dst || ch >= h->nch || cap < psk_soft_state_bytes(h))
        return fail(PSK_SOFT_ERR_INVALID_ARG, "psk_soft_export_state: bad arguments");
    ctl_sync(h);
    uint8_t *p = (uint8_t *)dst;
    std::memset(p, 0, psk_soft_state_bytes(h));
    StateHeader hd = {kStateMagic, PSK_SOFT_ABI_VERSION, h->lim.ring_cap, h->lim.fit_cap, (uint32_t)sizeof(psk::ChanCtl),
                      (uint32_t)sizeof(psk::ChanState)};
    std::memcpy(p, &hd, sizeof hd);
    p += sizeof hd;
    std::memcpy(p, &h->ctl[ch], sizeof(psk::ChanCtl));
    p += sizeof(psk::ChanCtl);
    if (h->dry)
        return PSK_SOFT_OK;
    psk_soft_status st = psk_soft_synchronize(h);
    if (st != PSK_SOFT_OK)
        return st;
    PSK_HIP(hipMemcpy(p, h->d_state + ch, sizeof(psk::ChanState), hipMemcpyDeviceToHost));
    p += sizeof(psk::ChanState);
    const float2 *ring = h->d_ring + ((size_t)ch * 2 + h->ctl[ch].ring_src) * h->lim.ring_cap;
    PSK_HIP(hipMemcpy(p, ring, sizeof(float2) * h->lim.ring_cap, hipMemcpyDeviceToHost));
    p += sizeof(float2) * h->lim.ring_cap;
    PSK_HIP(hipMemcpy(p, h->d_yv + (size_t)ch * h->lim.fit_cap, sizeof(float) * h->lim.fit_cap, hipMemcpyDeviceToHost));
    return PSK_SOFT_OK;
}

psk_soft_status psk_soft_import_state(psk_soft_handle_t *h, uint32_t ch, const void *src, uint64_t bytes)
{
    if (!h || !src || ch >= h->nch || bytes != psk_soft_state_bytes(h))
        return fail(PSK_SOFT_ERR_INVALID_ARG, "psk_soft_import_state: bad arguments (the blob must be exactly psk_soft_state_bytes long)");
    const uint8_t *p = (const uint8_t *)src;
    StateHeader hd;
    std::memcpy(&hd, p, sizeof hd);
    p += sizeof hd;
    if (hd.magic != kStateMagic || hd.version != PSK_SOFT_ABI_VERSION || hd.ring_cap != h->lim.ring_cap ||
        hd.fit_cap != h->lim.fit_cap || hd.ctl_bytes != sizeof(psk::ChanCtl) || hd.state_bytes != sizeof(psk::ChanState))
        return fail(PSK_SOFT_ERR_INVALID_ARG, "psk_soft_import_state: not a state blob of this library version and these limits");
    // everything the kernels use as an index is checked before anything is overwritten
    psk::ChanCtl c;
    std::memcpy(static_cast<void *>(&c), p, sizeof c);
    p += sizeof c;
    if (c.ring_src > 1u || c.lf_head >= h->lim.fit_cap || c.lf_len > c.lf_n || c.lf_n >= h->lim.fit_cap || c.lf_len >= h->lim.fit_cap ||
        c.lf_count > psk::kResyncCount || c.count > psk::kResyncCount ||
        (uint64_t)c.props.samplesPerBaud * c.props.numAvg > h->lim.ring_cap || c.props.phaseAvg >= h->lim.fit_cap)
        return fail(PSK_SOFT_ERR_INVALID_ARG, "psk_soft_import_state: control state out of range for this handle");
    if (!h->dry) {
        psk_soft_status st = psk_soft_synchronize(h);
        if (st != PSK_SOFT_OK)
            return st;
        PSK_HIP(hipSetDevice(h->device));
        PSK_HIP(hipMemcpy(h->d_state + ch, p, sizeof(psk::ChanState), hipMemcpyHostToDevice));
        p += sizeof(psk::ChanState);
        float2 *ring = h->d_ring + ((size_t)ch * 2 + c.ring_src) * h->lim.ring_cap;
        PSK_HIP(hipMemcpy(ring, p, sizeof(float2) * h->lim.ring_cap, hipMemcpyHostToDevice));
        p += sizeof(float2) * h->lim.ring_cap;
        PSK_HIP(hipMemcpy(h->d_yv + (size_t)ch * h->lim.fit_cap, p, sizeof(float) * h->lim.fit_cap, hipMemcpyHostToDevice));
    }
    ctl_touch(h);
    h->ctl[ch] = c;  // (last: a failed copy above leaves the host mirror as it was)
    return PSK_SOFT_OK;
}

psk_soft_status psk_soft_probe_read_ms(psk_soft_handle_t *h, const void *dev_ptr, uint64_t bytes, int reps,
                                       double *ms_per_pass)
{
    if (!h || h->dry || !dev_ptr || bytes < 16 || ((uintptr_t)dev_ptr & 15u) || reps < 1 || !ms_per_pass)
        return fail(PSK_SOFT_ERR_INVALID_ARG, "psk_soft_probe_read_ms: needs a device handle, a 16-byte aligned device "
                                              "pointer, at least 16 bytes and one repetition");
    PSK_HIP(hipSetDevice(h->device));
    hipEvent_t e0, e1;
    PSK_HIP(hipEventCreate(&e0));
    PSK_HIP(hipEventCreate(&e1));
    float *sink = reinterpret_cast<float *>(h->d_state);  // (never written: see the kernel)
    PSK_HIP(psk::launch_read_probe(dev_ptr, bytes, sink, h->stream));  // untimed first pass
    PSK_HIP(hipEventRecord(e0, h->stream));
    for (int r = 0; r < reps; r++) PSK_HIP(psk::launch_read_probe(dev_ptr, bytes, sink, h->stream));
    PSK_HIP(hipEventRecord(e1, h->stream));
    PSK_HIP(hipEventSynchronize(e1));
    float ms = 0.0f;
    PSK_HIP(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *ms_per_pass = (double)ms / reps;
    return PSK_SOFT_OK;
}

void *psk_soft_host_alloc(size_t bytes)
{
    void *p = nullptr;
    if (!bytes || hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) {
        g_last_error = "psk_soft_host_alloc: hipHostMalloc failed";
        return nullptr;
    }
    return p;
}

void psk_soft_host_free(void *p)
{
    if (p)
        (void)hipHostFree(p);
}

void *psk_soft_device_alloc(psk_soft_handle_t *h, size_t bytes)
{
    void *p = nullptr;
    if (!h || h->dry || !bytes || hipSetDevice(h->device) != hipSuccess || hipMalloc(&p, bytes) != hipSuccess) {
        g_last_error = "psk_soft_device_alloc: needs a device handle and a size; or hipMalloc failed";
        return nullptr;
    }
    return p;
}

void psk_soft_device_free(psk_soft_handle_t *h, void *p)
{
    if (h && !h->dry && p && hipSetDevice(h->device) == hipSuccess)
        (void)hipFree(p);
}

psk_soft_status psk_soft_device_upload(psk_soft_handle_t *h, void *dev_dst, const void *host_src, size_t bytes)
{
    if (!h || h->dry || !dev_dst || !host_src)
        return fail(PSK_SOFT_ERR_INVALID_ARG, "psk_soft_device_upload: bad arguments");
    PSK_HIP(hipSetDevice(h->device));
    PSK_HIP(hipMemcpy(dev_dst, host_src, bytes, hipMemcpyHostToDevice));
    return PSK_SOFT_OK;
}

psk_soft_status psk_soft_device_download(psk_soft_handle_t *h, void *host_dst, const void *dev_src, size_t bytes)
{
    if (!h || h->dry || !host_dst || !dev_src)
        return fail(PSK_SOFT_ERR_INVALID_ARG, "psk_soft_device_download: bad arguments");
    PSK_HIP(hipSetDevice(h->device));
    PSK_HIP(hipMemcpy(host_dst, dev_src, bytes, hipMemcpyDeviceToHost));
    return PSK_SOFT_OK;
}

psk_soft_status psk_soft_peek(const psk_soft_handle_t *h, uint32_t ch, uint64_t *ring_len, uint64_t *index,
                              uint64_t *fit_len)
{
    if (!h || ch >= h->nch)
        return fail(PSK_SOFT_ERR_INVALID_ARG, "psk_soft_peek: bad channel");
    ctl_sync(h);
    if (ring_len) *ring_len = h->ctl[ch].ring_len;
    if (index) *index = h->ctl[ch].index;
    if (fit_len) *fit_len = h->ctl[ch].lf_len;
    return PSK_SOFT_OK;
}

}  // extern "C"
