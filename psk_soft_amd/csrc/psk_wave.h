// psk_wave.h -- wave64 cross-lane primitives (DPP), the virtual sample stream of a call, and the
// LinearFit sum rebuild shared by the gfx950 kernels of the psk_soft hot path.
#ifndef PSK_WAVE_H
#define PSK_WAVE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "psk_device_math.h"
#include "psk_plan.h"

namespace psk {

constexpr int kWave = 64;
constexpr int kYRingMin = 512;      // LDS ring of unwrapped phases per wave (floats): the host picks a power of
constexpr int kYRingMax = 2048;     // two >= phaseAvg + 128 in this range (256 allowed where the energy ring is dynamic too)
constexpr int kSeqMaxS = 1024;      // reference-order kernel: symbolEnergy[] lives in LDS
constexpr int kSeqChunk = 512;      // reference-order kernel: packet staging chunk (complex samples)

// ---------------------------------------------------------------------------------
// cross-lane primitives (wave64, DPP)
// ---------------------------------------------------------------------------------
template <int CTRL, int ROW_MASK>
PSK_DEV int dpp_zero(int v)
{
    // lanes without a source lane, and rows masked off, receive 0
    return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xF, false);
}
template <int CTRL, int ROW_MASK>
PSK_DEV double dpp_zero_f64(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = dpp_zero<CTRL, ROW_MASK>(lo);
    hi = dpp_zero<CTRL, ROW_MASK>(hi);
    return __hiloint2double(hi, lo);
}
// row_shr:N with bound_ctrl:0 -- lanes whose source falls outside their row of 16 read 0, so no
// "old" value has to be materialised (saves two v_mov per 64-bit step)
template <int CTRL>
PSK_DEV int dpp_shr0(int v)
{
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}
template <int CTRL>
PSK_DEV double dpp_shr0_f64(double v)
{
    int lo = dpp_shr0<CTRL>(__double2loint(v)), hi = dpp_shr0<CTRL>(__double2hiint(v));
    return __hiloint2double(hi, lo);
}
// inclusive prefix sum over the 64 lanes: row_shr 1,2,4,8 inside each row of 16, then
// row_bcast:15 into rows 1,3 and row_bcast:31 into rows 2,3
PSK_DEV double wave_scan_f64(double v)
{
    v += dpp_shr0_f64<0x111>(v);
    v += dpp_shr0_f64<0x112>(v);
    v += dpp_shr0_f64<0x114>(v);
    v += dpp_shr0_f64<0x118>(v);
    v += dpp_zero_f64<0x142, 0xA>(v);
    v += dpp_zero_f64<0x143, 0xC>(v);
    return v;
}
// float versions, one instruction a step: the DPP source is folded into the add / max, and for
// the two cross-row steps the rows masked off simply keep their value (the compiler's own
// lowering needs a zeroed temporary and a separate add there).  Written as asm because that
// folding is not reachable from the builtins.  hipcc pads no hazard whose producer or consumer
// sits inside an asm statement, so every wait state is in the strings: a DPP instruction reads a
// register two wait states after its VALU write at the earliest, a v_readlane one; each statement
// opens with `s_nop 1` (the compiler may have scheduled the producer of the operand right in
// front) and closes with `s_nop 1` (the consumer right behind may be a DPP instruction or a
// v_readlane).  tools/isa_hazards.py scans the built ISA for violations of these rules.
#define PSK_DPP_STEP(op, ctl) op " %0, %0, %0 " ctl
#define PSK_DPP_ROW1 "row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
#define PSK_DPP_ROW2 "row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1"
#define PSK_DPP_ROW4 "row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1"
#define PSK_DPP_ROW8 "row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1"
#define PSK_DPP_BC15 "row_bcast:15 row_mask:0xa bank_mask:0xf"
#define PSK_DPP_BC31 "row_bcast:31 row_mask:0xc bank_mask:0xf"
}  // namespace psk
// N independent scans interleaved step by step, ONE asm statement per N (generated:
// tools/gen_wave_scan.py): wave_scan_f32_multi(float (&v)[N]), N = 1 .. 32
#include "psk_wave_scan_gen.h"
namespace psk {
PSK_DEV float wave_scan_f32(float v)
{
    float a[1] = {v};
    wave_scan_f32_multi(a);
    return a[0];
}
// max over the wave of non-negative values: lanes without a source read 0, so the running
// maximum of lane 63 is the wave maximum
PSK_DEV float wave_max_f32(float v)
{
    asm volatile("s_nop 1\n\t" PSK_DPP_STEP("v_max_f32_dpp", PSK_DPP_ROW1) "\n\ts_nop 1\n\t"
                 PSK_DPP_STEP("v_max_f32_dpp", PSK_DPP_ROW2) "\n\ts_nop 1\n\t"
                 PSK_DPP_STEP("v_max_f32_dpp", PSK_DPP_ROW4) "\n\ts_nop 1\n\t"
                 PSK_DPP_STEP("v_max_f32_dpp", PSK_DPP_ROW8) "\n\ts_nop 1\n\t"
                 PSK_DPP_STEP("v_max_f32_dpp", PSK_DPP_BC15) "\n\ts_nop 1\n\t"
                 PSK_DPP_STEP("v_max_f32_dpp", PSK_DPP_BC31) "\n\ts_nop 1"
                 : "+v"(v));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
// median of three signed integers (v_med3_i32)
PSK_DEV int med3_i32(int a, int b, int c)
{
    int r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
PSK_DEV int wave_scan_i32(int v)
{
    v += dpp_shr0<0x111>(v);
    v += dpp_shr0<0x112>(v);
    v += dpp_shr0<0x114>(v);
    v += dpp_shr0<0x118>(v);
    v += dpp_zero<0x142, 0xA>(v);
    v += dpp_zero<0x143, 0xC>(v);
    return v;
}
// value of lane-1 (wave_shr:1); lane 0 receives `carry`
PSK_DEV int wave_up1(int v, int carry) { return __builtin_amdgcn_update_dpp(carry, v, 0x138, 0xF, 0xF, false); }
PSK_DEV float wave_up1(float v, float carry)
{
    return __int_as_float(wave_up1(__float_as_int(v), __float_as_int(carry)));
}
PSK_DEV double wave_up1(double v, double carry)
{
    int lo = wave_up1(__double2loint(v), __double2loint(carry));
    int hi = wave_up1(__double2hiint(v), __double2hiint(carry));
    return __hiloint2double(hi, lo);
}
// ... lane 0 receives zero (bound_ctrl: no register has to be preloaded with the carry -- two moves less per double)
PSK_DEV int wave_up1_zero(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xF, 0xF, true); }
PSK_DEV float wave_up1_zero(float v) { return __int_as_float(wave_up1_zero(__float_as_int(v))); }
PSK_DEV double wave_up1_zero(double v)
{
    return __hiloint2double(wave_up1_zero(__double2hiint(v)), wave_up1_zero(__double2loint(v)));
}
PSK_DEV float read_lane(float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); }
PSK_DEV double read_lane(double v, int lane)
{
    int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
PSK_DEV double wave_sum_f64(double v) { return read_lane(wave_scan_f64(v), 63); }
PSK_DEV unsigned wave_max_u32(unsigned v)
{
#pragma unroll
    for (int o = 32; o; o >>= 1) {
        unsigned t = (unsigned)__shfl_xor((int)v, o);
        v = t > v ? t : v;
    }
    return v;
}
PSK_DEV unsigned wave_min_u32(unsigned v)
{
#pragma unroll
    for (int o = 32; o; o >>= 1) {
        unsigned t = (unsigned)__shfl_xor((int)v, o);
        v = t < v ? t : v;
    }
    return v;
}
// Orders this wave's LDS traffic: a later ds_read sees an earlier ds_write of another lane.  The
// LDS executes one wave's instructions in issue order, so all that is needed is that the COMPILER
// keeps the order (a wavefront-scope fence would also wait for every outstanding global load and
// store -- vmcnt(0) -- which exposes store latency inside the fit loop).
PSK_DEV void wave_lds_fence()
{
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

// ---------------------------------------------------------------------------------
// the virtual stream X = [ring of carried samples] ++ [packet]   (the `samples` deque)
// ---------------------------------------------------------------------------------
// Pointers that come out of a ChanPlan (packet, output rows) are generic to the compiler: it cannot
// trace them to a kernel argument, and emits FLAT loads and stores for them.  A flat access counts on
// vmcnt AND lgkmcnt, so a wait for an LDS result (energy ring, ds_bpermute, phase ring) also waits
// for the packet loads in flight.  They are global memory (device or page-locked host), and the
// wave-scan kernels say so (address space 1): -1 % at samplesPerBaud 8, -2 % at 6 and 10, -10 % at 12.
// (packet_global(S) = false keeps an instantiation on generic accesses, for A/B runs.)
#define PSK_GLOBAL __attribute__((address_space(1)))
typedef float f2g __attribute__((ext_vector_type(2)));
typedef float f4g __attribute__((ext_vector_type(4), aligned(8)));  // 16-byte access at 8-byte alignment
constexpr bool packet_global(int) { return true; }
template <bool G, class T>
struct MemPtr {
    typedef T *type;
};
template <class T>
struct MemPtr<true, T> {
    typedef PSK_GLOBAL T *type;
};
template <bool G, class T>
PSK_DEV typename MemPtr<G, T>::type mem_ptr(T *p)
{
    return (typename MemPtr<G, T>::type)p;
}
// pointer to a 16-byte vector at 8-byte alignment, generic or global (spelled out: a template
// parameter would drop the alignment attribute of the typedef)
template <bool G>
struct F4Ptr {
    typedef const f4g *type;
};
template <>
struct F4Ptr<true> {
    typedef const PSK_GLOBAL f4g *type;
};
// vector stores to the output rows at the alignment the ABI asks for (soft 8, phase / bits /
// sampleIndex 4 bytes), in the address space of the pointer
typedef float f4u __attribute__((ext_vector_type(4), aligned(8)));
typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
typedef short s2u __attribute__((ext_vector_type(2), aligned(4)));
typedef short s4u __attribute__((ext_vector_type(4), aligned(4)));
PSK_DEV void store_f4u(float *p, f4u v) { *reinterpret_cast<f4u *>(p) = v; }
// PSK_NT_STORE / PSK_NT_LOAD (experiments): the output rows / the packet with the non-temporal hint (written once, read once)
#ifndef PSK_NT_STORE
#define PSK_NT_STORE 0
#endif
#ifndef PSK_NT_LOAD
#define PSK_NT_LOAD 0
#endif
#if PSK_NT_STORE
#define PSK_ST(ptr, v) __builtin_nontemporal_store((v), (ptr))
#else
#define PSK_ST(ptr, v) (*(ptr) = (v))
#endif
PSK_DEV void store_f4u(PSK_GLOBAL float *p, f4u v) { PSK_ST((PSK_GLOBAL f4u *)p, v); }
PSK_DEV void store_f2u(float *p, f2u v) { *reinterpret_cast<f2u *>(p) = v; }
PSK_DEV void store_f2u(PSK_GLOBAL float *p, f2u v) { PSK_ST((PSK_GLOBAL f2u *)p, v); }
PSK_DEV void store_s2u(int16_t *p, s2u v) { *reinterpret_cast<s2u *>(p) = v; }
PSK_DEV void store_s2u(PSK_GLOBAL int16_t *p, s2u v) { PSK_ST((PSK_GLOBAL s2u *)p, v); }
PSK_DEV void store_s4u(int16_t *p, s4u v) { *reinterpret_cast<s4u *>(p) = v; }
PSK_DEV void store_s4u(PSK_GLOBAL int16_t *p, s4u v) { PSK_ST((PSK_GLOBAL s4u *)p, v); }
struct XView {
    const f2g *ring;
    const f2g *in;
    uint32_t L0;  // samples in the ring
};
PSK_DEV float2 x_at(const XView &X, uint64_t j)
{
    const f2g v = j < X.L0 ? X.ring[j] : X.in[j - X.L0];
    return make_float2(v.x, v.y);
}

template <int S>
PSK_DEV void load_symbol(const XView &X, uint64_t tau, bool valid, float2 (&x)[S])
{
#pragma unroll
    for (int k = 0; k < S; k++) x[k] = make_float2(0.0f, 0.0f);
    if (!valid)
        return;
    const uint64_t j0 = tau * (uint64_t)S;
    const f2g *p;
    if (j0 >= X.L0) {
        p = X.in + (j0 - X.L0);
    } else if (j0 + S <= X.L0) {
        p = X.ring + j0;
    } else {  // the one symbol that straddles ring and packet
#pragma unroll
        for (int k = 0; k < S; k++) x[k] = x_at(X, j0 + k);
        return;
    }
    if (S % 2 == 0) {
        // 16-byte loads; the address is only 8-byte aligned (gfx950 global loads allow that)
        const typename F4Ptr<packet_global(S)>::type q = (typename F4Ptr<packet_global(S)>::type)p;
#pragma unroll
        for (int k = 0; k < S / 2; k++) {
            const f4g t = q[k];
            x[2 * k] = make_float2(t.x, t.y);
            x[2 * k + 1] = make_float2(t.z, t.w);
        }
    } else {
        const auto q = mem_ptr<packet_global(S)>(p);
#pragma unroll
        for (int k = 0; k < S; k++) {
            const f2g t = q[k];
            x[k] = make_float2(t.x, t.y);
        }
    }
}

// ---------------------------------------------------------------------------------
// LinearFit pieces shared by both kernels
// ---------------------------------------------------------------------------------
// LinearFit::reset() tail (cpp/psk_soft.cpp:110-122) on `len` values y(j):
// ySum = sum y_j, xySum = sum fl32(fl32(j*xdelta)*y_j), both accumulated in double in the order j = 0, 1, ...
// The addends are float-valued.  Wave-parallel partial sums give the reference's bits whenever no
// addition rounds at all, which is certain while 24 + (exponent spread of the non-zero addends) +
// log2(len) + 1 <= 53; that is the case for any sane window, and is checked here.  Otherwise every lane
// runs the reference's loop (the values come out of LDS as broadcasts; a few microseconds, once per call).
PSK_DEV void addend_track(float v, unsigned &umax, unsigned &umin1)
{
    const unsigned b = __float_as_uint(v) & 0x7FFFFFFFu;  // a zero constrains neither bound
    umax = b > umax ? b : umax;
    umin1 = (b - 1u) < umin1 ? (b - 1u) : umin1;
}
PSK_DEV bool addends_order_free(unsigned umax, unsigned umin1, uint32_t len)
{
    umax = wave_max_u32(umax);
    umin1 = wave_min_u32(umin1);
    if (umin1 == 0xFFFFFFFFu)
        return true;  // all zero
    if (umax >= 0x7F800000u)
        return false;  // inf / NaN
    int emax = (int)(umax >> 23), emin = (int)((umin1 + 1u) >> 23);
    emax = emax < 1 ? 1 : emax;
    emin = emin < 1 ? 1 : emin;
    const int terms_log2 = 32 - __builtin_clz(len | 1u);
    return 24 + (emax - emin) + terms_log2 + 1 <= 53;
}
template <class YAt>
PSK_DEV void fit_rebuild_sums(YAt y_at, uint32_t len, float xdelta, double &ySum, double &xySum)
{
    const int lane = threadIdx.x & 63;
    double ys = 0.0, xys = 0.0;
    unsigned ymax = 0u, ymin1 = 0xFFFFFFFFu, tmax = 0u, tmin1 = 0xFFFFFFFFu;
    for (uint32_t j = lane; j < len; j += kWave) {
        float y = y_at(j);
        ys += (double)y;
        float jx = (float)j * xdelta;
        float jxy = jx * y;
        xys += (double)jxy;
        addend_track(y, ymax, ymin1);
        addend_track(jxy, tmax, tmin1);
    }
    ySum = wave_sum_f64(ys);
    xySum = wave_sum_f64(xys);
    const bool free_y = addends_order_free(ymax, ymin1, len), free_t = addends_order_free(tmax, tmin1, len);
    if (!(free_y && free_t)) {  // (wave-uniform)
        ys = 0.0, xys = 0.0;
#pragma unroll 1
        for (uint32_t j = 0; j < len; j++) {
            float y = y_at(j);
            ys += (double)y;
            float jx = (float)j * xdelta;
            float jxy = jx * y;
            xys += (double)jxy;
        }
        ySum = ys;
        xySum = xys;
    }
}

// ---------------------------------------------------------------------------------
// wave-scan kernel
// ---------------------------------------------------------------------------------
}  // namespace psk
#endif
