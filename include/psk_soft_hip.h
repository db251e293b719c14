/*
 * psk_soft_hip.h -- C ABI of libpsk_soft_hip.so, the MI355X (gfx950) implementation of
 * the hot path of REDHAWK rh.psk_soft.
 *
 * Drop-in boundary.  The library replaces, inside
 * psk_soft_i::serviceFunction(), everything from the reinterpretation of the
 * packet as complex samples to the end-of-call phase wrap
 *     reference cpp/psk_soft.cpp:365-603   (flag handling, resyncEnergy, LinearFit
 *                                           resets, the per-sample / per-symbol loop,
 *                                           the end-of-call wrap)
 *     reference cpp/psk_soft.cpp:35-185    (class LinearFit)
 *     reference cpp/psk_soft.cpp:619-651   (resyncEnergy, property listeners)
 * and leaves on the C++ host: getPacket / delete (:349-352, :616), the pushSRI and
 * pushPacket calls (:400-404, :605-615).  The host keeps calling it from the one
 * service thread REDHAWK gives the component.
 *
 * One "channel" = the complete state of one psk_soft_i instance (one stream).
 * Channels are independent; a handle owns a batch of them on ONE GPU.  All
 * functions return a psk_soft_status, never throw, never call back.  Caller owns
 * every buffer passed in for the duration of the call; the library owns the
 * per-channel demodulator state, which lives in HBM between calls.
 *
 * Plain pointers and sizes only -- no C++ or torch types cross this boundary.
 */
#ifndef PSK_SOFT_HIP_H
#define PSK_SOFT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PSK_SOFT_ABI_VERSION 2

typedef enum psk_soft_status {
    PSK_SOFT_OK = 0,
    PSK_SOFT_ERR_INVALID_ARG = 1,
    PSK_SOFT_ERR_NO_DEVICE = 2,     /* no usable gfx950 device / HIP runtime error at create */
    PSK_SOFT_ERR_HIP = 3,           /* a HIP call failed; see psk_soft_last_error()          */
    PSK_SOFT_ERR_LIMIT = 4,         /* a property exceeds the limits given at create          */
    PSK_SOFT_ERR_UNSUPPORTED = 5,   /* samplesPerBaud == 0 or phaseAvg == 0 (undefined in the
                                       reference: cpp/psk_soft.cpp:441,454 and :54-55,70)     */
    PSK_SOFT_ERR_CAPACITY = 6       /* an output buffer is too small                          */
} psk_soft_status;

/* serviceFunction() return codes, reference cpp/psk_soft.cpp:351,362,617 */
enum { PSK_SOFT_NOOP = 0, PSK_SOFT_NORMAL = 1 };

/* device == PSK_SOFT_DEVICE_NONE creates a control-plane-only handle: property,
 * flag, SRI and output-count logic run (they are host-side), no data is touched. */
#define PSK_SOFT_DEVICE_NONE (-1)

/* The six properties, reference psk_soft.prf.xml:23-60 / cpp/psk_soft_base.h:45-56.
 * Same names, types and defaults (cpp/psk_soft_base.cpp:96-148). */
typedef struct psk_soft_props {
    uint16_t samplesPerBaud;       /* ushort, default 10  */
    uint16_t constelationSize;     /* ushort, default 4   */
    uint32_t numAvg;               /* ulong,  default 100 */
    uint16_t phaseAvg;             /* ushort, default 50  */
    uint8_t differentialDecoding;  /* bool,   default 0   */
    uint8_t resetState;            /* bool,   default 0; self-clearing (cpp/psk_soft.cpp:365-372) */
} psk_soft_props_t;

/* Upper bounds that size the per-channel state in HBM. */
typedef struct psk_soft_limits {
    uint32_t max_window_samples;   /* >= samplesPerBaud*numAvg of every channel (the `samples` deque) */
    uint32_t max_phase_avg;        /* >= phaseAvg of every channel (LinearFit::yvals)                 */
    uint32_t max_packet_complex;   /* >= complex samples of one packet of one channel (host-buffer path) */
} psk_soft_limits_t;

/* One bulkio::InFloatPort::dataTransfer as serviceFunction() reads it
 * (reference cpp/psk_soft.cpp:349-359, 394, 428). */
typedef struct psk_soft_packet {
    const float *data;          /* dataBuffer: interleaved I,Q (device or host pointer, per entry point) */
    uint64_t n_floats;          /* dataBuffer.size()                                                      */
    double sri_xdelta;          /* SRI.xdelta                                                             */
    int32_t sri_mode;           /* SRI.mode; anything but 1 is dropped with a warning (:359-363)          */
    uint8_t sriChanged;
    uint8_t inputQueueFlushed;  /* forces resetState (:353-357)                                           */
    uint8_t present;            /* 0 = getPacket() returned NULL for this channel: NOOP (:350-352)        */
    uint8_t reserved;
} psk_soft_packet_t;

/* Where one channel's four output streams go, and what the call produced.
 * Layout advice for the device-pointer path: start every channel's row of every stream on a
 * 128-byte boundary (e.g. a row capacity that is a multiple of 64 symbols).  The kernels store
 * whole cache lines per wave; rows that straddle lines were measured 6 % slower end to end.
 * Pointer fields are inputs; the rest is filled in before the call returns
 * (output sizes depend only on packet sizes and properties, so they are exact
 * even on the asynchronous device-pointer path). */
typedef struct psk_soft_output {
    float *soft;                /* softDecision_dataFloat_out payload: re,im per symbol */
    int16_t *bits;              /* bits_dataShort_out: log2(M) shorts per symbol        */
    float *phase;               /* phase_dataFloat_out: one per symbol                  */
    int16_t *sampleIndex;       /* sampleIndex_dataShort_out: one per symbol            */
    uint64_t cap_symbols;       /* capacity of the buffers above, in symbols            */
    /* results */
    int32_t ret;                /* PSK_SOFT_NOOP / PSK_SOFT_NORMAL                      */
    uint64_t n_symbols;         /* symbols emitted: soft has 2*n, phase n                */
    uint64_t n_bits;            /* shorts written to bits (n * bitsPerBaud)             */
    uint64_t n_sampleIndex;     /* n, or 0 when samplesPerBaud == 1 (:459-469)          */
    int32_t sri_pushed;         /* 1: the host must pushSRI on soft/phase/bits (:393-405) */
    double sri_soft_xdelta;     /* xdelta for the soft and phase SRIs (:399)            */
    double sri_bits_xdelta;     /* xdelta for the bits SRI (:403)                       */
    int32_t n_warn;             /* LOG_WARN count (:355,361,566)                        */
} psk_soft_output_t;

typedef struct psk_soft_handle psk_soft_handle_t;

/* Runtime statistics of the last psk_soft_process_* call (read after psk_soft_synchronize).  A control-
 * plane-only handle (PSK_SOFT_DEVICE_NONE) fills in channels_fast / channels_sequential as PLANNED. */
typedef struct psk_soft_stats {
    uint64_t channels_fast;       /* channels handled by a wave-scan kernel                        */
    uint64_t channels_exact_timing; /* of those: calls whose timing screening refused (near-ties) and
                                       that the exact-timing wave-scan kernel redid                */
    uint64_t channels_sequential; /* channels handled by the reference-order kernel (planned)      */
    uint64_t channels_guard;      /* of those: sent there at run time by the exactness guard       */
    uint64_t unwrap_extra_passes; /* extra unwrap fixed-point passes summed over all 128-symbol blocks */
    uint64_t unwrap_blocks;       /* 128-symbol blocks processed by the wave-scan kernel           */
    uint64_t timing_exact_blocks; /* of those: blocks whose timing argmax needed the exact double pass
                                     (in the screened kernel, numAvg <= 128, or in the exact kernel) */
    uint64_t fit_chain_blocks;    /* of those: blocks whose LinearFit sums were redone in the reference's
                                     order of additions by the lane-after-lane chain (the wave-parallel
                                     candidates did not verify: sums crossing a binade or zero)          */
    uint64_t channels_tiled;      /* of channels_fast: calls carried by the time-tiled kernels (few channels,
                                     long packets: the call is cut along time, see PSK_SOFT_OPT_TIME_TILED) */
    uint64_t channels_parallel_fit; /* of channels_tiled: calls whose feedback unwrap and fit were guessed and
                                     verified in parallel along time instead of walked block by block   */
    uint64_t channels_parallel_fit_second_round; /* of channels_parallel_fit: on the second guess of the unwrap counts
                                     (the first, by consecutive raw phases, was wrong somewhere: a noisy stream)      */
    uint64_t parallel_fit_refusals; /* why tiled calls that tried did not: bit 0 a ySum that rounds, bit 1 an xySum
                                     certificate, bit 2 an unwrap count the guess got wrong (OR over the channels) */
} psk_soft_stats_t;

uint32_t psk_soft_abi_version(void);
const char *psk_soft_last_error(void);            /* thread-local text of the last failure */

/* lifetime -- replaces the psk_soft_i constructor (cpp/psk_soft.cpp:187-213): every
 * channel starts with the default properties, empty history and all three reset flags set. */
psk_soft_status psk_soft_create(int device, uint32_t n_channels, const psk_soft_limits_t *limits,
                                psk_soft_handle_t **out);
psk_soft_status psk_soft_destroy(psk_soft_handle_t *h);

/* configure() of channels [ch0, ch0+nch): stores the properties and runs the change
 * listeners the component registers (cpp/psk_soft.cpp:210-212, 638-651) for every
 * property whose value differs from the stored one.  Takes effect at the next process call
 * (the reference snapshots its properties at the top of serviceFunction, :374-378). */
psk_soft_status psk_soft_configure(psk_soft_handle_t *h, uint32_t ch0, uint32_t nch,
                                   const psk_soft_props_t *props /* [nch] */);
psk_soft_status psk_soft_query(const psk_soft_handle_t *h, uint32_t ch, psk_soft_props_t *props);
/* run one listener unconditionally: 0 samplesPerBaudChanged, 1 constelationSizeChanged, 2 phaseAvgChanged */
psk_soft_status psk_soft_fire_listener(psk_soft_handle_t *h, uint32_t ch, int which);

/* symbols one call can emit for a packet of n_complex samples: (n_complex + samplesPerBaud-1)/samplesPerBaud
 * bounds the reference's reserve() at cpp/psk_soft.cpp:434 */
uint64_t psk_soft_output_capacity(const psk_soft_handle_t *h, uint32_t ch, uint64_t n_complex);

/* One serviceFunction() body for channels [ch0, ch0+nch), one packet each.
 * _device: packet data and output buffers are DEVICE pointers; kernels are enqueued on
 *          `stream` (a hipStream_t, NULL = the handle's own stream) and the call returns
 *          without waiting; counts/SRI fields of `outs` are already final.
 * _host:   packet data and output buffers are HOST pointers; the library stages them
 *          through HBM and returns when the outputs are in place. */
psk_soft_status psk_soft_process_device(psk_soft_handle_t *h, uint32_t ch0, uint32_t nch,
                                        const psk_soft_packet_t *pkts /* [nch] */,
                                        psk_soft_output_t *outs /* [nch] */, void *stream);
psk_soft_status psk_soft_process_host(psk_soft_handle_t *h, uint32_t ch0, uint32_t nch,
                                      const psk_soft_packet_t *pkts, psk_soft_output_t *outs);
psk_soft_status psk_soft_synchronize(psk_soft_handle_t *h);

/* PSK_SOFT_OPT_DEFERRED_JOIN: make `stream` (a hipStream_t; NULL = the handle's own) wait for everything the calls made so
 * far have put on the handle's side streams -- the point in stream order behind which their results may be used.  A no-op
 * without pending deferred calls.  (There is no counterpart in the reference: its serviceFunction() is synchronous.) */
psk_soft_status psk_soft_join(psk_soft_handle_t *h, void *stream);
psk_soft_status psk_soft_get_stats(psk_soft_handle_t *h, psk_soft_stats_t *stats);
/* the same, one record per channel of [ch0, ch0+nch) (stats[nch]) */
psk_soft_status psk_soft_get_channel_stats(psk_soft_handle_t *h, uint32_t ch0, uint32_t nch, psk_soft_stats_t *stats);

/* Options of a handle (all default 0 = the reference's behaviour, quirks included).
 * PSK_SOFT_OPT_QPSK_SIGN_BITMAP: 1 = QPSK bits by the signs of the de-rotated symbol, as the
 *   constellation diagram at reference cpp/psk_soft.cpp:516-521 describes (A 00, B 01, C 10, D 11,
 *   least significant bit first), instead of the float->bool conversions of :523-526 that make
 *   every QPSK bit 0.  Opt-in; takes effect at the next process call. */
enum {
    PSK_SOFT_OPT_QPSK_SIGN_BITMAP = 1,
    /* 1 (default): a batch that mixes window classes (samplesPerBaud, numAvg <= 128 / 256 / 512 / 1024) launches its
     * classes side by side on streams of the handle, forked off and joined back into the caller's stream; 0: one
     * after the other on the caller's stream.  No effect on results. */
    PSK_SOFT_OPT_CONCURRENT_CLASSES = 2,
    /* Calls of few channels and many symbols are cut along time (tiles of a few hundred symbols spread over the
     * machine; the feedback unwrap and fit guessed in parallel and verified, else walked by one wave per channel):
     * 1 (default) = where it pays (a window class of the call with at most 64 channels and 2048 symbols or more out
     * per channel, or at most 512 channels and 24576 symbols; numAvg <= 128, samplesPerBaud 2 .. 16), 0 = never,
     * 2 = wherever the kernels exist (tests).  No effect on results.
     * The environment variable PSK_SOFT_TIME_TILED (0 / 1 / 2) sets the default of new handles. */
    PSK_SOFT_OPT_TIME_TILED = 3,
    /* The feedback unwrap and fit of a time-tiled call: 1 (default) = guessed in parallel along time and verified
     * position by position, a second round on corrected unwrap counts enqueued for a while after a call whose first
     * guess failed; 2 = the second round always enqueued; 0 = walked block by block, one wave per channel.  No effect on
     * results.  Environment: PSK_SOFT_PARALLEL_FIT. */
    PSK_SOFT_OPT_PARALLEL_FIT = 4,
    /* 1 = deferred join (default 0).  A batch that mixes window classes runs its classes on side streams of the handle.  By
     * default every call ends with the caller's stream waiting for them (results in stream order, like everything else).
     * With this option it does not: every class ends its calls on its own stream and the next call's launches of that class
     * queue behind them there, so a class with short launches runs ahead into the following calls while another is still
     * busy (configs[4] of the benchmark: 3.4 -> 2.8 ms per step).  The caller's stream -- or any stream -- sees the results of all
     * calls made so far after psk_soft_join(); psk_soft_synchronize() waits for them on the host.  Results are unchanged.
     * A call whose channels would not all stay on the stream they were on (a property or packet pattern that moves a channel
     * to another window class) joins everything first, by itself. */
    PSK_SOFT_OPT_DEFERRED_JOIN = 5
};
psk_soft_status psk_soft_set_option(psk_soft_handle_t *h, int option, int value);

/* Force every channel through the reference-order (sequential) kernel: 1 on, 0 off. */
psk_soft_status psk_soft_set_force_sequential(psk_soft_handle_t *h, int on);

/* Page-locked host memory that the GPU reads and writes directly (hipHostMalloc): packets and result
 * buffers allocated here can be handed to psk_soft_process_device as they are -- the kernels stream
 * them over PCIe with no staging copy (measured 55 GB/s in + 16 GB/s out, i.e. link rate, against
 * 32 + 9 GB/s for pageable buffers through psk_soft_process_host).  The replacement for the
 * std::vector storage of bulkio dataTransfer::dataBuffer (reference cpp/psk_soft.cpp:349, 428) in a
 * host that wants the link rate.  NULL on failure / without a GPU. */
void *psk_soft_host_alloc(size_t bytes);
void psk_soft_host_free(void *p);

/* Device (HBM) buffers on the handle's GPU for callers that keep packets and results resident and have no HIP
 * runtime of their own to allocate them with (hipMalloc / hipFree / hipMemcpy behind the handle's device; the
 * copies are synchronous).  A host that already owns device memory -- torch tensors, its own hipMalloc --
 * passes those pointers to psk_soft_process_device directly and never needs these. */
void *psk_soft_device_alloc(psk_soft_handle_t *h, size_t bytes);
void psk_soft_device_free(psk_soft_handle_t *h, void *p);
psk_soft_status psk_soft_device_upload(psk_soft_handle_t *h, void *dev_dst, const void *host_src, size_t bytes);
psk_soft_status psk_soft_device_download(psk_soft_handle_t *h, void *host_dst, const void *dev_src, size_t bytes);

/* Measurement support ("the empirical ceiling on the box -- a pure float4
 * read-reduce kernel over the same buffer"): reads `bytes` of device memory at `dev_ptr` (16-byte
 * aligned) `reps` times with 16-byte loads, nothing else, and returns the mean duration of one pass in
 * milliseconds, timed with HIP events on the handle's stream.  No counterpart in the reference. */
psk_soft_status psk_soft_probe_read_ms(psk_soft_handle_t *h, const void *dev_ptr, uint64_t bytes, int reps,
                                       double *ms_per_pass);

/* checkpoint / test support: opaque state blob of one channel */
uint64_t psk_soft_state_bytes(const psk_soft_handle_t *h);
psk_soft_status psk_soft_export_state(psk_soft_handle_t *h, uint32_t ch, void *dst, uint64_t cap);
psk_soft_status psk_soft_import_state(psk_soft_handle_t *h, uint32_t ch, const void *src, uint64_t bytes);
/* introspection of the mirrored control state (tests): samples.size(), index, yvals.size() */
psk_soft_status psk_soft_peek(const psk_soft_handle_t *h, uint32_t ch, uint64_t *ring_len,
                              uint64_t *index, uint64_t *fit_len);

#ifdef __cplusplus
}
#endif
#endif /* PSK_SOFT_HIP_H */
