// psk_capi.cpp -- the C ABI of libpsk_soft_hip.so (include/psk_soft_hip.h).
//
// Host side of the drop-in boundary: per-channel control plane (psk_ctl.h, mirrors the
// non-data state of psk_soft_i, reference cpp/psk_soft.cpp:353-426), HBM-resident channel
// state, plan upload and kernel launches.  There is no CPU compute path here: without a
// usable GPU psk_soft_create() fails (PSK_SOFT_ERR_NO_DEVICE) unless the caller explicitly
// asks for a control-plane-only handle (PSK_SOFT_DEVICE_NONE), which never touches data.
#include <hip/hip_runtime_api.h>
#include <hip/hip_vector_types.h>

#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "psk_ctl.h"
#include "psk_plan.h"
#include "psk_soft_hip.h"

namespace psk {
hipError_t launch_fast(int S, int H, int exact, const ChanPlan *plans, uint32_t ch0, uint32_t nch, ChanState *states,
                       float2 *rings, uint32_t ring_cap, float *yvs, uint32_t fit_cap, hipStream_t stream);
hipError_t launch_seq(const ChanPlan *plans, uint32_t ch0, uint32_t nch, ChanState *states, float2 *rings,
                      uint32_t ring_cap, float *yvs, uint32_t fit_cap, hipStream_t stream);
}  // namespace psk

namespace {

thread_local std::string g_last_error;

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
constexpr uint32_t kFastFitMax = 384;  // LDS y ring of the wave-scan kernel: 512 - 128
constexpr uint32_t kSeqMaxS = 1024;    // symbolEnergy[] of the reference-order kernel lives in LDS
const int kFastS[] = {2, 4, 5, 8, 10, 16};

}  // namespace

struct psk_soft_handle {
    int device = PSK_SOFT_DEVICE_NONE;
    bool dry = true;
    uint32_t nch = 0;
    psk::Limits lim{};
    psk_soft_limits_t user{};
    std::vector<psk::ChanCtl> ctl;
    std::vector<uint32_t> last_mode;  // PlanMode of the last call, per channel (statistics)
    // device memory
    psk::ChanState *d_state = nullptr;
    float2 *d_ring = nullptr;
    float *d_yv = nullptr;
    psk::ChanPlan *h_plans[kPlanSlots] = {};
    psk::ChanPlan *d_plans[kPlanSlots] = {};
    hipEvent_t ev[kPlanSlots] = {};
    bool ev_used[kPlanSlots] = {};
    int slot = 0;
    hipStream_t stream = nullptr;
    // staging for the host-buffer entry point
    float *d_in = nullptr;
    float *d_soft = nullptr;
    float *d_phase = nullptr;
    int16_t *d_bits = nullptr;
    int16_t *d_sidx = nullptr;
    uint64_t stage_cap = 0;  // symbols per channel
};

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
    h->last_mode.assign(n_channels, psk::PLAN_SKIP);
    h->device = device;
    h->dry = (device == PSK_SOFT_DEVICE_NONE);
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
            if ((e2 = hipHostMalloc((void **)&h->h_plans[s], sizeof(psk::ChanPlan) * n_channels)) != hipSuccess)
                return bail("hipHostMalloc plans", e2);
            if ((e2 = hipMalloc((void **)&h->d_plans[s], sizeof(psk::ChanPlan) * n_channels)) != hipSuccess)
                return bail("hipMalloc plans", e2);
            if ((e2 = hipEventCreateWithFlags(&h->ev[s], hipEventDisableTiming)) != hipSuccess)
                return bail("hipEventCreate", e2);
        }
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
            if (h->h_plans[s]) (void)hipHostFree(h->h_plans[s]);
            if (h->d_plans[s]) (void)hipFree(h->d_plans[s]);
            if (h->ev[s]) (void)hipEventDestroy(h->ev[s]);
        }
        if (h->d_state) (void)hipFree(h->d_state);
        if (h->d_ring) (void)hipFree(h->d_ring);
        if (h->d_yv) (void)hipFree(h->d_yv);
        if (h->d_in) (void)hipFree(h->d_in);
        if (h->d_soft) (void)hipFree(h->d_soft);
        if (h->d_phase) (void)hipFree(h->d_phase);
        if (h->d_bits) (void)hipFree(h->d_bits);
        if (h->d_sidx) (void)hipFree(h->d_sidx);
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
    for (uint32_t i = 0; i < nch; i++) h->ctl[ch0 + i].configure(props[i]);
    return PSK_SOFT_OK;
}

psk_soft_status psk_soft_query(const psk_soft_handle_t *h, uint32_t ch, psk_soft_props_t *props)
{
    if (!h || !props || ch >= h->nch)
        return fail(PSK_SOFT_ERR_INVALID_ARG, "psk_soft_query: bad channel");
    *props = h->ctl[ch].props;
    return PSK_SOFT_OK;
}

psk_soft_status psk_soft_fire_listener(psk_soft_handle_t *h, uint32_t ch, int which)
{
    if (!h || ch >= h->nch)
        return fail(PSK_SOFT_ERR_INVALID_ARG, "psk_soft_fire_listener: bad channel");
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
    uint64_t S = h->ctl[ch].props.samplesPerBaud ? h->ctl[ch].props.samplesPerBaud : 1;
    return (n_complex + S - 1) / S + 1;
}

psk_soft_status psk_soft_process_device(psk_soft_handle_t *h, uint32_t ch0, uint32_t nch,
                                        const psk_soft_packet_t *pkts, psk_soft_output_t *outs, void *stream_v)
{
    if (!h || !pkts || !outs || !nch || (uint64_t)ch0 + nch > h->nch)
        return fail(PSK_SOFT_ERR_INVALID_ARG, "psk_soft_process: bad arguments");
    // plan on copies; commit only if every channel of the batch is accepted
    std::vector<psk::ChanCtl> next(h->ctl.begin() + ch0, h->ctl.begin() + ch0 + nch);
    std::vector<psk::ChanPlan> plans(nch);
    for (uint32_t i = 0; i < nch; i++) {
        if (next[i].props.samplesPerBaud > kSeqMaxS)
            return fail(PSK_SOFT_ERR_LIMIT, "samplesPerBaud > 1024");
        psk_soft_status st = psk::plan_call(next[i], h->lim, pkts[i], outs[i], plans[i]);
        if (st != PSK_SOFT_OK) {
            char buf[160];
            std::snprintf(buf, sizeof buf, "psk_soft_process: channel %u refused (status %d)", ch0 + i, (int)st);
            return fail(st, buf);
        }
        if (!h->dry && plans[i].mode != psk::PLAN_SKIP) {
            const psk::ChanPlan &p = plans[i];
            if ((p.n_in && !p.in) || ((uintptr_t)p.in & 7u) || ((uintptr_t)p.soft & 7u) || ((uintptr_t)p.bits & 3u) ||
                ((uintptr_t)p.phase & 3u) || ((uintptr_t)p.sidx & 3u))
                return fail(PSK_SOFT_ERR_INVALID_ARG,
                            "psk_soft_process: packet data must be 8-byte aligned, soft 8, bits 4, phase 4, sampleIndex 4");
        }
    }
    for (uint32_t i = 0; i < nch; i++) {
        h->ctl[ch0 + i] = next[i];
        h->last_mode[ch0 + i] = plans[i].mode;
    }
    if (h->dry)
        return PSK_SOFT_OK;

    // which kernels does this batch need?
    bool any = false, any_emit = false, any_seq = false, any_quiet = false;
    bool need_SH[17][5] = {};
    for (uint32_t i = 0; i < nch; i++) {
        const psk::ChanPlan &p = plans[i];
        if (p.mode == psk::PLAN_SKIP)
            continue;
        any = true;
        if (p.mode == psk::PLAN_FAST) {
            if (p.n_out) {
                any_emit = true;
                need_SH[p.S][p.A <= 128u ? 1 : p.A <= 256u ? 2 : 4] = true;
            } else {
                any_quiet = true;
            }
        } else {
            any_seq = true;
        }
    }
    if (!any)
        return PSK_SOFT_OK;

    PSK_HIP(hipSetDevice(h->device));
    hipStream_t stream = stream_v ? (hipStream_t)stream_v : h->stream;
    const int slot = h->slot;
    h->slot = (h->slot + 1) % kPlanSlots;
    if (h->ev_used[slot])
        PSK_HIP(hipEventSynchronize(h->ev[slot]));
    std::memcpy(h->h_plans[slot], plans.data(), sizeof(psk::ChanPlan) * nch);
    PSK_HIP(hipMemcpyAsync(h->d_plans[slot], h->h_plans[slot], sizeof(psk::ChanPlan) * nch, hipMemcpyHostToDevice,
                           stream));
    if (any_quiet)
        PSK_HIP(psk::launch_fast(0, 1, 0, h->d_plans[slot], ch0, nch, h->d_state, h->d_ring, h->lim.ring_cap, h->d_yv,
                                 h->lim.fit_cap, stream));
    // screened timing first; the exact-timing instantiation picks up the calls it refused, the
    // reference-order kernel (below) the calls both refused
    for (int exact = 0; exact <= 1; exact++)
        for (int S : kFastS)
            for (int H = 1; H <= 4; H++)
                if (need_SH[S][H])
                    PSK_HIP(psk::launch_fast(S, H, exact, h->d_plans[slot], ch0, nch, h->d_state, h->d_ring,
                                             h->lim.ring_cap, h->d_yv, h->lim.fit_cap, stream));
    if (any_seq || any_emit)  // any_emit: the exactness guard may hand calls over at run time
        PSK_HIP(psk::launch_seq(h->d_plans[slot], ch0, nch, h->d_state, h->d_ring, h->lim.ring_cap, h->d_yv,
                                h->lim.fit_cap, stream));
    PSK_HIP(hipEventRecord(h->ev[slot], stream));
    h->ev_used[slot] = true;
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
    const uint64_t in_cap = h->user.max_packet_complex;
    const uint64_t out_cap = (in_cap + 3) & ~1ull;  // even: every channel's rows stay 4-byte aligned
    if (!h->d_in) {
        const size_t n = h->nch;
        PSK_HIP(hipMalloc((void **)&h->d_in, sizeof(float) * 2 * in_cap * n));
        PSK_HIP(hipMalloc((void **)&h->d_soft, sizeof(float) * 2 * out_cap * n));
        PSK_HIP(hipMalloc((void **)&h->d_phase, sizeof(float) * out_cap * n));
        PSK_HIP(hipMalloc((void **)&h->d_bits, sizeof(int16_t) * 4 * out_cap * n));
        PSK_HIP(hipMalloc((void **)&h->d_sidx, sizeof(int16_t) * out_cap * n));
        h->stage_cap = out_cap;
    }
    std::vector<psk_soft_packet_t> dp(pkts, pkts + nch);
    std::vector<psk_soft_output_t> dout(outs, outs + nch);
    for (uint32_t i = 0; i < nch; i++) {
        const uint32_t ch = ch0 + i;
        if (pkts[i].present && pkts[i].n_floats / 2 > in_cap)
            return fail(PSK_SOFT_ERR_LIMIT, "psk_soft_process_host: packet longer than max_packet_complex");
        dp[i].data = h->d_in + (size_t)ch * 2 * in_cap;
        dout[i].soft = h->d_soft + (size_t)ch * 2 * out_cap;
        dout[i].phase = h->d_phase + (size_t)ch * out_cap;
        dout[i].bits = h->d_bits + (size_t)ch * 4 * out_cap;
        dout[i].sampleIndex = h->d_sidx + (size_t)ch * out_cap;
        dout[i].cap_symbols = out_cap;
        if (pkts[i].present && pkts[i].n_floats)
            PSK_HIP(hipMemcpyAsync((void *)dp[i].data, pkts[i].data, sizeof(float) * (pkts[i].n_floats & ~1ull),
                                   hipMemcpyHostToDevice, h->stream));
    }
    // the caller's capacity applies to the caller's buffers
    for (uint32_t i = 0; i < nch; i++) {
        psk::ChanCtl probe = h->ctl[ch0 + i];
        psk::ChanPlan pl;
        psk_soft_output_t o = outs[i];
        psk_soft_status st = psk::plan_call(probe, h->lim, pkts[i], o, pl);
        if (st == PSK_SOFT_OK && o.n_symbols > outs[i].cap_symbols)
            return fail(PSK_SOFT_ERR_CAPACITY, "psk_soft_process_host: output buffer too small");
    }
    psk_soft_status st = psk_soft_process_device(h, ch0, nch, dp.data(), dout.data(), h->stream);
    if (st != PSK_SOFT_OK)
        return st;
    for (uint32_t i = 0; i < nch; i++) {
        psk_soft_output_t &o = outs[i];
        const psk_soft_output_t &d = dout[i];
        o.ret = d.ret;
        o.n_symbols = d.n_symbols;
        o.n_bits = d.n_bits;
        o.n_sampleIndex = d.n_sampleIndex;
        o.sri_pushed = d.sri_pushed;
        o.sri_soft_xdelta = d.sri_soft_xdelta;
        o.sri_bits_xdelta = d.sri_bits_xdelta;
        o.n_warn = d.n_warn;
        if (d.n_symbols) {
            if (o.soft)
                PSK_HIP(hipMemcpyAsync(o.soft, d.soft, sizeof(float) * 2 * d.n_symbols, hipMemcpyDeviceToHost, h->stream));
            if (o.phase)
                PSK_HIP(hipMemcpyAsync(o.phase, d.phase, sizeof(float) * d.n_symbols, hipMemcpyDeviceToHost, h->stream));
            if (o.bits && d.n_bits)
                PSK_HIP(hipMemcpyAsync(o.bits, d.bits, sizeof(int16_t) * d.n_bits, hipMemcpyDeviceToHost, h->stream));
            if (o.sampleIndex && d.n_sampleIndex)
                PSK_HIP(hipMemcpyAsync(o.sampleIndex, d.sampleIndex, sizeof(int16_t) * d.n_sampleIndex,
                                       hipMemcpyDeviceToHost, h->stream));
        }
    }
    PSK_HIP(hipStreamSynchronize(h->stream));
    return PSK_SOFT_OK;
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
    return PSK_SOFT_OK;
}

psk_soft_status psk_soft_get_stats(psk_soft_handle_t *h, psk_soft_stats_t *stats)
{
    if (!h || !stats)
        return fail(PSK_SOFT_ERR_INVALID_ARG, "null argument");
    std::memset(stats, 0, sizeof *stats);
    if (h->dry)
        return PSK_SOFT_OK;
    psk_soft_status st = psk_soft_synchronize(h);
    if (st != PSK_SOFT_OK)
        return st;
    std::vector<psk::ChanState> s(h->nch);
    PSK_HIP(hipMemcpy(s.data(), h->d_state, sizeof(psk::ChanState) * h->nch, hipMemcpyDeviceToHost));
    for (uint32_t c = 0; c < h->nch; c++) {
        switch (h->last_mode[c]) {
        case psk::PLAN_FAST:
            if (s[c].guard == 2u) {
                stats->channels_sequential++;
                stats->channels_guard++;
            } else {
                stats->channels_fast++;
                if (s[c].guard == 3u)
                    stats->channels_exact_timing++;
                stats->unwrap_blocks += s[c].stat_blocks;
                stats->unwrap_extra_passes += s[c].stat_extra;
                stats->timing_exact_blocks += s[c].stat_exact;
            }
            break;
        case psk::PLAN_SEQ:
        case psk::PLAN_SEQ_S1: stats->channels_sequential++; break;
        default: break;
        }
    }
    return PSK_SOFT_OK;
}

psk_soft_status psk_soft_set_force_sequential(psk_soft_handle_t *h, int on)
{
    if (!h)
        return fail(PSK_SOFT_ERR_INVALID_ARG, "null handle");
    h->lim.force_seq = on != 0;
    return PSK_SOFT_OK;
}

uint64_t psk_soft_state_bytes(const psk_soft_handle_t *h)
{
    if (!h)
        return 0;
    return sizeof(psk::ChanCtl) + sizeof(psk::ChanState) + sizeof(float2) * (uint64_t)h->lim.ring_cap +
           sizeof(float) * (uint64_t)h->lim.fit_cap;
}

psk_soft_status psk_soft_export_state(psk_soft_handle_t *h, uint32_t ch, void *dst, uint64_t cap)
{
    if (!h || !dst || ch >= h->nch || cap < psk_soft_state_bytes(h))
        return fail(PSK_SOFT_ERR_INVALID_ARG, "psk_soft_export_state: bad arguments");
    uint8_t *p = (uint8_t *)dst;
    std::memset(p, 0, psk_soft_state_bytes(h));
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
    if (!h || !src || ch >= h->nch || bytes < psk_soft_state_bytes(h))
        return fail(PSK_SOFT_ERR_INVALID_ARG, "psk_soft_import_state: bad arguments");
    const uint8_t *p = (const uint8_t *)src;
    std::memcpy(&h->ctl[ch], p, sizeof(psk::ChanCtl));
    p += sizeof(psk::ChanCtl);
    if (h->dry)
        return PSK_SOFT_OK;
    psk_soft_status st = psk_soft_synchronize(h);
    if (st != PSK_SOFT_OK)
        return st;
    PSK_HIP(hipMemcpy(h->d_state + ch, p, sizeof(psk::ChanState), hipMemcpyHostToDevice));
    p += sizeof(psk::ChanState);
    float2 *ring = h->d_ring + ((size_t)ch * 2 + h->ctl[ch].ring_src) * h->lim.ring_cap;
    PSK_HIP(hipMemcpy(ring, p, sizeof(float2) * h->lim.ring_cap, hipMemcpyHostToDevice));
    p += sizeof(float2) * h->lim.ring_cap;
    PSK_HIP(hipMemcpy(h->d_yv + (size_t)ch * h->lim.fit_cap, p, sizeof(float) * h->lim.fit_cap, hipMemcpyHostToDevice));
    return PSK_SOFT_OK;
}

psk_soft_status psk_soft_peek(const psk_soft_handle_t *h, uint32_t ch, uint64_t *ring_len, uint64_t *index,
                              uint64_t *fit_len)
{
    if (!h || ch >= h->nch)
        return fail(PSK_SOFT_ERR_INVALID_ARG, "psk_soft_peek: bad channel");
    if (ring_len) *ring_len = h->ctl[ch].ring_len;
    if (index) *index = h->ctl[ch].index;
    if (fit_len) *fit_len = h->ctl[ch].lf_len;
    return PSK_SOFT_OK;
}

}  // extern "C"
