/*
 * psk_soft_oracle.h -- CPU oracle for the psk_soft hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a sequential plain-C restatement of
 * psk_soft_i::serviceFunction() and class LinearFit of RedhawkSDR/psk_soft
 * (reference cpp/psk_soft.cpp:35-185, 346-651).  Only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() may load it; the shipped
 * product path (libpsk_soft_hip.so) never links, loads or calls it.
 *
 * PINNING STATUS (see oracle/README.md and DESIGN.md section "Oracle"):
 *   - the reference cannot be compiled in this image without writing stand-ins
 *     for the REDHAWK/omniORB/boost headers it includes (ossie/Component.h,
 *     bulkio/bulkio.h, boost/thread.hpp ...), so no oracle/_ref build exists;
 *   - the oracle is pinned by the reference's own six known-answer tests
 *     (tests/test_psk_soft.py:160-238, soft symbols within 1e-3 of the
 *     transmitted ones) and by the behaviours of the verbatim reference that
 *     SURVEY.md Appendix A recorded (output counts, all-zero QPSK bits,
 *     inf/NaN first differential output, pushSRI per call, gnu++98 pow value);
 *   - the bits / phase / sampleIndex streams have NO reference-side fixture
 *     at all ("parity unpinned" for those three streams beyond the recorded
 *     Appendix-A facts); they follow the reference line by line, and every
 *     arithmetic primitive is checked bit-for-bit against the same libstdc++ /
 *     libgcc / glibc routines the reference calls (oracle/prim_check.cpp).
 */
#ifndef PSK_SOFT_ORACLE_H
#define PSK_SOFT_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct psk_oracle psk_oracle_t;

/* serviceFunction() return codes (reference cpp/psk_soft.cpp:351,362,617). */
enum { PSK_ORACLE_NOOP = 0, PSK_ORACLE_NORMAL = 1 };

/* property ids, psk_soft.prf.xml:23-60 */
enum {
    PSK_ORACLE_PROP_samplesPerBaud = 0,
    PSK_ORACLE_PROP_numAvg = 1,
    PSK_ORACLE_PROP_constelationSize = 2,
    PSK_ORACLE_PROP_phaseAvg = 3,
    PSK_ORACLE_PROP_differentialDecoding = 4,
    PSK_ORACLE_PROP_resetState = 5
};

/* One bulkio dataTransfer as serviceFunction() sees it (cpp/psk_soft.cpp:349-359,394). */
typedef struct psk_oracle_packet {
    const float *data;      /* dataBuffer: interleaved I,Q                    */
    size_t n_floats;        /* dataBuffer.size()                              */
    double xdelta;          /* SRI.xdelta                                     */
    int mode;               /* SRI.mode (1 = complex)                         */
    int sriChanged;
    int inputQueueFlushed;
} psk_oracle_packet_t;

/* What one call pushed out (cpp/psk_soft.cpp:400-404, 605-615).  Pointers stay
 * valid until the next psk_oracle_service() on the same object. */
typedef struct psk_oracle_result {
    int ret;                /* NOOP / NORMAL                                  */
    const float *soft;      /* softDecision_dataFloat_out payload (re,im,...) */
    size_t n_soft_floats;
    const short *bits;      /* bits_dataShort_out                             */
    size_t n_bits;
    const float *phase;     /* phase_dataFloat_out                            */
    size_t n_phase;
    const short *index;     /* sampleIndex_dataShort_out                      */
    size_t n_index;
    int sri_pushed;         /* 1 if the three pushSRI calls ran this call     */
    double sri_soft_xdelta; /* xdelta pushed on soft + phase ports            */
    double sri_bits_xdelta; /* xdelta pushed on the bits port                 */
    int n_warn;             /* LOG_WARN count of this call                    */
} psk_oracle_result_t;

psk_oracle_t *psk_oracle_create(void);
void psk_oracle_destroy(psk_oracle_t *o);

/* configure() of one property.  Stores the value; when `fire` is non-zero the
 * registered change listener runs (cpp/psk_soft.cpp:210-212, 638-651).  The
 * sandbox-style wrapper passes fire = (new value != old value). */
void psk_oracle_set_property(psk_oracle_t *o, int prop, uint32_t value, int fire);
uint32_t psk_oracle_get_property(const psk_oracle_t *o, int prop);

int psk_oracle_service(psk_oracle_t *o, const psk_oracle_packet_t *pkt,
                       psk_oracle_result_t *res);

/* introspection used by tests */
size_t psk_oracle_ring_size(const psk_oracle_t *o);   /* samples.size()          */
size_t psk_oracle_index(const psk_oracle_t *o);       /* index                   */
float  psk_oracle_phase_estimate(const psk_oracle_t *o);
size_t psk_oracle_fit_history(const psk_oracle_t *o, float *dst, size_t cap);

/* arithmetic primitives, exported so prim_check.cpp can compare them with the
 * libstdc++ / libgcc / glibc routines the reference itself calls */
void  psk_oracle_prim_cmul(float a, float b, float c, float d, float *re, float *im);
void  psk_oracle_prim_cdiv(float a, float b, float c, float d, float *re, float *im);
void  psk_oracle_prim_cpow(float a, float b, unsigned n, float *re, float *im);
float psk_oracle_prim_norm(float a, float b);
void  psk_oracle_prim_polar1(float theta, float *re, float *im);
int   psk_oracle_prim_wrap_test(float phaseEstimate, float wrapValue);
float psk_oracle_prim_denominator(float xdelta, size_t pts);

#ifdef __cplusplus
}
#endif
#endif
