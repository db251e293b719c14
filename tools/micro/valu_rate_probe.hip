// valu_rate_probe.hip -- what one VALU instruction of each kind costs a SIMD of gfx950 at the wave-scan kernel's
// residency (four waves per SIMD, every CU busy): cycles per instruction per SIMD, from s_memtime around a loop of
// independent instructions of one kind (eight accumulators per wave, so that neither latency nor dependencies limit).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rate tools/micro/valu_rate_probe.hip && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

// eight independent instructions per statement
#define REP8_F32(op) \
    asm volatile(op " %0, %0, %8, %9\n\t" op " %1, %1, %8, %9\n\t" op " %2, %2, %8, %9\n\t" op " %3, %3, %8, %9\n\t" \
                 op " %4, %4, %8, %9\n\t" op " %5, %5, %8, %9\n\t" op " %6, %6, %8, %9\n\t" op " %7, %7, %8, %9" \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c))
#define REP8_2OP(op, T0, T1, T2, T3, T4, T5, T6, T7, B) \
    asm volatile(op " %0, %0, %8\n\t" op " %1, %1, %8\n\t" op " %2, %2, %8\n\t" op " %3, %3, %8\n\t" \
                 op " %4, %4, %8\n\t" op " %5, %5, %8\n\t" op " %6, %6, %8\n\t" op " %7, %7, %8" \
                 : "+v"(T0), "+v"(T1), "+v"(T2), "+v"(T3), "+v"(T4), "+v"(T5), "+v"(T6), "+v"(T7) : "v"(B))
#define REP8_1OP(op, T0, T1, T2, T3, T4, T5, T6, T7) \
    asm volatile(op " %0, %0\n\t" op " %1, %1\n\t" op " %2, %2\n\t" op " %3, %3\n\t" \
                 op " %4, %4\n\t" op " %5, %5\n\t" op " %6, %6\n\t" op " %7, %7" \
                 : "+v"(T0), "+v"(T1), "+v"(T2), "+v"(T3), "+v"(T4), "+v"(T5), "+v"(T6), "+v"(T7))

template <int KIND>
__global__ __launch_bounds__(64, 4) void k_rate(int iters, unsigned long long *cyc, float *sink)
{
    asm volatile("v_mov_b32 v120, 0" ::: "v120");  // 128 registers: four waves to a SIMD, as the kernel
    const int lane = threadIdx.x;
    float a0 = lane, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7, b = 1.0001f, c = 0.5f;
    double d0 = lane, d1 = d0 + 1, d2 = d0 + 2, d3 = d0 + 3, d4 = d0 + 4, d5 = d0 + 5, d6 = d0 + 6, d7 = d0 + 7, e = 1.0000001, f = 0.5;
    int i0 = lane, i1 = lane + 1, i2 = lane + 2, i3 = lane + 3, i4 = lane + 4, i5 = lane + 5, i6 = lane + 6, i7 = lane + 7;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (KIND == 0) REP8_F32("v_fma_f32");
            if (KIND == 1) REP8_2OP("v_add_f32", a0, a1, a2, a3, a4, a5, a6, a7, b);
            if (KIND == 2) REP8_2OP("v_mul_f32", a0, a1, a2, a3, a4, a5, a6, a7, b);
            if (KIND == 3)
                asm volatile("v_fma_f64 %0, %0, %8, %9\n\tv_fma_f64 %1, %1, %8, %9\n\tv_fma_f64 %2, %2, %8, %9\n\tv_fma_f64 %3, %3, %8, %9\n\t"
                             "v_fma_f64 %4, %4, %8, %9\n\tv_fma_f64 %5, %5, %8, %9\n\tv_fma_f64 %6, %6, %8, %9\n\tv_fma_f64 %7, %7, %8, %9"
                             : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(e), "v"(f));
            if (KIND == 4) REP8_2OP("v_add_f64", d0, d1, d2, d3, d4, d5, d6, d7, f);
            if (KIND == 5) REP8_2OP("v_mul_f64", d0, d1, d2, d3, d4, d5, d6, d7, e);
            if (KIND == 6) REP8_1OP("v_mov_b32", i0, i1, i2, i3, i4, i5, i6, i7);
            if (KIND == 7)
                asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n\tv_cndmask_b32 %1, %1, %8, vcc\n\tv_cndmask_b32 %2, %2, %8, vcc\n\tv_cndmask_b32 %3, %3, %8, vcc\n\t"
                             "v_cndmask_b32 %4, %4, %8, vcc\n\tv_cndmask_b32 %5, %5, %8, vcc\n\tv_cndmask_b32 %6, %6, %8, vcc\n\tv_cndmask_b32 %7, %7, %8, vcc"
                             : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(lane) : "vcc");
            if (KIND == 8)
                asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                             "v_mov_b32_dpp %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                             "v_mov_b32_dpp %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                             "v_mov_b32_dpp %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf"
                             : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7));
            if (KIND == 9)
                asm volatile("v_cvt_f64_f32 %0, %8\n\tv_cvt_f64_f32 %1, %9\n\tv_cvt_f64_f32 %2, %10\n\tv_cvt_f64_f32 %3, %11\n\t"
                             "v_cvt_f64_f32 %4, %12\n\tv_cvt_f64_f32 %5, %13\n\tv_cvt_f64_f32 %6, %14\n\tv_cvt_f64_f32 %7, %15"
                             : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7)
                             : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7));
            if (KIND == 10) REP8_1OP("v_rcp_f32", a0, a1, a2, a3, a4, a5, a6, a7);
            if (KIND == 11) REP8_1OP("v_rndne_f64", d0, d1, d2, d3, d4, d5, d6, d7);
            if (KIND == 12)
                asm volatile("v_readlane_b32 s20, %0, 3\n\tv_readlane_b32 s21, %1, 3\n\tv_readlane_b32 s22, %2, 3\n\tv_readlane_b32 s23, %3, 3\n\t"
                             "v_readlane_b32 s20, %4, 3\n\tv_readlane_b32 s21, %5, 3\n\tv_readlane_b32 s22, %6, 3\n\tv_readlane_b32 s23, %7, 3"
                             : : "v"(i0), "v"(i1), "v"(i2), "v"(i3), "v"(i4), "v"(i5), "v"(i6), "v"(i7) : "s20", "s21", "s22", "s23");
            if (KIND == 13) REP8_2OP("v_pk_mul_f32", d0, d1, d2, d3, d4, d5, d6, d7, e);
            if (KIND == 14) REP8_2OP("v_add_u32", i0, i1, i2, i3, i4, i5, i6, i7, lane);
            if (KIND == 15)
                asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cmp_lt_f32 vcc, %1, %2\n\tv_cmp_lt_f32 vcc, %2, %3\n\tv_cmp_lt_f32 vcc, %3, %4\n\t"
                             "v_cmp_lt_f32 vcc, %4, %5\n\tv_cmp_lt_f32 vcc, %5, %6\n\tv_cmp_lt_f32 vcc, %6, %7\n\tv_cmp_lt_f32 vcc, %7, %0"
                             : : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7) : "vcc");
            if (KIND == 16)
                asm volatile("s_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0");
            if (KIND == 18)  // v_cndmask with the condition in a scalar register pair
                asm volatile("v_cndmask_b32 %0, %0, %8, s[20:21]\n\tv_cndmask_b32 %1, %1, %8, s[20:21]\n\tv_cndmask_b32 %2, %2, %8, s[20:21]\n\tv_cndmask_b32 %3, %3, %8, s[20:21]\n\t"
                             "v_cndmask_b32 %4, %4, %8, s[20:21]\n\tv_cndmask_b32 %5, %5, %8, s[20:21]\n\tv_cndmask_b32 %6, %6, %8, s[20:21]\n\tv_cndmask_b32 %7, %7, %8, s[20:21]"
                             : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(lane) : "s20", "s21");
            if (KIND == 19)  // compare + select pairs, the way the compiler emits them
                asm volatile("v_cmp_lt_f32 vcc, %0, %8\n\tv_cndmask_b32 %0, %0, %8, vcc\n\tv_cmp_lt_f32 vcc, %1, %8\n\tv_cndmask_b32 %1, %1, %8, vcc\n\t"
                             "v_cmp_lt_f32 vcc, %2, %8\n\tv_cndmask_b32 %2, %2, %8, vcc\n\tv_cmp_lt_f32 vcc, %3, %8\n\tv_cndmask_b32 %3, %3, %8, vcc"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "vcc");
            if (KIND == 20) REP8_2OP("v_max_f32", a0, a1, a2, a3, a4, a5, a6, a7, b);
            if (KIND == 21) REP8_2OP("v_and_b32", i0, i1, i2, i3, i4, i5, i6, i7, lane);
            if (KIND == 22) REP8_2OP("v_lshlrev_b32", i0, i1, i2, i3, i4, i5, i6, i7, lane);
            if (KIND == 23) REP8_1OP("v_fract_f64", d0, d1, d2, d3, d4, d5, d6, d7);
            if (KIND == 24)
                asm volatile("v_cvt_f32_f64 %0, %8\n\tv_cvt_f32_f64 %1, %9\n\tv_cvt_f32_f64 %2, %10\n\tv_cvt_f32_f64 %3, %11\n\t"
                             "v_cvt_f32_f64 %4, %12\n\tv_cvt_f32_f64 %5, %13\n\tv_cvt_f32_f64 %6, %14\n\tv_cvt_f32_f64 %7, %15"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                             : "v"(d0), "v"(d1), "v"(d2), "v"(d3), "v"(d4), "v"(d5), "v"(d6), "v"(d7));
            if (KIND == 25) REP8_2OP("v_mul_lo_u32", i0, i1, i2, i3, i4, i5, i6, i7, lane);
            if (KIND == 26)
                asm volatile("v_med3_i32 %0, %0, %8, %9\n\tv_med3_i32 %1, %1, %8, %9\n\tv_med3_i32 %2, %2, %8, %9\n\tv_med3_i32 %3, %3, %8, %9\n\t"
                             "v_med3_i32 %4, %4, %8, %9\n\tv_med3_i32 %5, %5, %8, %9\n\tv_med3_i32 %6, %6, %8, %9\n\tv_med3_i32 %7, %7, %8, %9"
                             : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(lane), "v"(i0));
            if (KIND == 27)
                asm volatile("v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_add_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                             "v_add_f32_dpp %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_add_f32_dpp %3, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                             "v_add_f32_dpp %4, %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_add_f32_dpp %5, %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                             "v_add_f32_dpp %6, %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_add_f32_dpp %7, %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            if (KIND == 28)
                asm volatile("v_cmp_lt_f32 s[20:21], %0, %1\n\tv_cmp_lt_f32 s[22:23], %1, %2\n\tv_cmp_lt_f32 s[20:21], %2, %3\n\tv_cmp_lt_f32 s[22:23], %3, %4\n\t"
                             "v_cmp_lt_f32 s[20:21], %4, %5\n\tv_cmp_lt_f32 s[22:23], %5, %6\n\tv_cmp_lt_f32 s[20:21], %6, %7\n\tv_cmp_lt_f32 s[22:23], %7, %0"
                             : : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7) : "s20", "s21", "s22", "s23");
            if (KIND == 29)  // select without a condition register: v_cndmask with constant inline operands only
                asm volatile("v_cndmask_b32 %0, 0, %8, vcc\n\tv_cndmask_b32 %1, 0, %8, vcc\n\tv_cndmask_b32 %2, 0, %8, vcc\n\tv_cndmask_b32 %3, 0, %8, vcc\n\t"
                             "v_cndmask_b32 %4, 0, %8, vcc\n\tv_cndmask_b32 %5, 0, %8, vcc\n\tv_cndmask_b32 %6, 0, %8, vcc\n\tv_cndmask_b32 %7, 0, %8, vcc"
                             : "=v"(i0), "=v"(i1), "=v"(i2), "=v"(i3), "=v"(i4), "=v"(i5), "=v"(i6), "=v"(i7) : "v"(lane) : "vcc");
            if (KIND == 30) REP8_2OP("v_ldexp_f64", d0, d1, d2, d3, d4, d5, d6, d7, lane);
            if (KIND == 31)
                asm volatile("ds_bpermute_b32 %0, %8, %0\n\tds_bpermute_b32 %1, %8, %1\n\tds_bpermute_b32 %2, %8, %2\n\tds_bpermute_b32 %3, %8, %3\n\t"
                             "ds_bpermute_b32 %4, %8, %4\n\tds_bpermute_b32 %5, %8, %5\n\tds_bpermute_b32 %6, %8, %6\n\tds_bpermute_b32 %7, %8, %7\n\ts_waitcnt lgkmcnt(0)"
                             : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(lane));
            if (KIND == 32)  // one compare, four selects on its VCC
                asm volatile("v_cmp_lt_f32 vcc, %0, %8\n\tv_cndmask_b32 %0, %0, %8, vcc\n\tv_cndmask_b32 %1, %1, %8, vcc\n\tv_cndmask_b32 %2, %2, %8, vcc\n\tv_cndmask_b32 %3, %3, %8, vcc\n\t"
                             "v_cmp_lt_f32 vcc, %4, %8\n\tv_cndmask_b32 %4, %4, %8, vcc\n\tv_cndmask_b32 %5, %5, %8, vcc\n\tv_cndmask_b32 %6, %6, %8, vcc\n\tv_cndmask_b32 %7, %7, %8, vcc"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
            if (KIND == 33)  // VCC written by the scalar unit, then eight selects
                asm volatile("s_mov_b64 vcc, exec\n\tv_cndmask_b32 %0, %0, %8, vcc\n\tv_cndmask_b32 %1, %1, %8, vcc\n\tv_cndmask_b32 %2, %2, %8, vcc\n\tv_cndmask_b32 %3, %3, %8, vcc\n\t"
                             "v_cndmask_b32 %4, %4, %8, vcc\n\tv_cndmask_b32 %5, %5, %8, vcc\n\tv_cndmask_b32 %6, %6, %8, vcc\n\tv_cndmask_b32 %7, %7, %8, vcc"
                             : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(lane) : "vcc");
            if (KIND == 34)  // selects with distinct sources and destinations (no chain through the destination)
                asm volatile("v_cndmask_b32 %0, %8, %9, vcc\n\tv_cndmask_b32 %1, %8, %9, vcc\n\tv_cndmask_b32 %2, %8, %9, vcc\n\tv_cndmask_b32 %3, %8, %9, vcc\n\t"
                             "v_cndmask_b32 %4, %8, %9, vcc\n\tv_cndmask_b32 %5, %8, %9, vcc\n\tv_cndmask_b32 %6, %8, %9, vcc\n\tv_cndmask_b32 %7, %8, %9, vcc"
                             : "=v"(i0), "=v"(i1), "=v"(i2), "=v"(i3), "=v"(i4), "=v"(i5), "=v"(i6), "=v"(i7) : "v"(lane), "v"(i0) : "vcc");
            if (KIND == 35) REP8_2OP("v_xor_b32", i0, i1, i2, i3, i4, i5, i6, i7, lane);
            if (KIND == 36) REP8_2OP("v_sub_f32", a0, a1, a2, a3, a4, a5, a6, a7, b);
            if (KIND == 37)
                asm volatile("v_bfi_b32 %0, %8, %0, %9\n\tv_bfi_b32 %1, %8, %1, %9\n\tv_bfi_b32 %2, %8, %2, %9\n\tv_bfi_b32 %3, %8, %3, %9\n\t"
                             "v_bfi_b32 %4, %8, %4, %9\n\tv_bfi_b32 %5, %8, %5, %9\n\tv_bfi_b32 %6, %8, %6, %9\n\tv_bfi_b32 %7, %8, %7, %9"
                             : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(lane), "v"(i0));
            if (KIND == 38) REP8_2OP("v_min_f32", a0, a1, a2, a3, a4, a5, a6, a7, b);
            if (KIND == 39)
                asm volatile("v_mov_b64 %0, %0\n\tv_mov_b64 %1, %1\n\tv_mov_b64 %2, %2\n\tv_mov_b64 %3, %3\n\tv_mov_b64 %4, %4\n\tv_mov_b64 %5, %5\n\tv_mov_b64 %6, %6\n\tv_mov_b64 %7, %7"
                             : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7));
            if (KIND == 40)  // f32 fma with a scalar-register operand and with an inline constant
                asm volatile("v_fma_f32 %0, %0, s20, 0.5\n\tv_fma_f32 %1, %1, s20, 0.5\n\tv_fma_f32 %2, %2, s20, 0.5\n\tv_fma_f32 %3, %3, s20, 0.5\n\t"
                             "v_fma_f32 %4, %4, s20, 0.5\n\tv_fma_f32 %5, %5, s20, 0.5\n\tv_fma_f32 %6, %6, s20, 0.5\n\tv_fma_f32 %7, %7, s20, 0.5"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : : "s20");
            if (KIND == 41)  // f64 add with a 64-bit literal-free scalar pair operand
                asm volatile("v_add_f64 %0, %0, s[20:21]\n\tv_add_f64 %1, %1, s[20:21]\n\tv_add_f64 %2, %2, s[20:21]\n\tv_add_f64 %3, %3, s[20:21]\n\t"
                             "v_add_f64 %4, %4, s[20:21]\n\tv_add_f64 %5, %5, s[20:21]\n\tv_add_f64 %6, %6, s[20:21]\n\tv_add_f64 %7, %7, s[20:21]"
                             : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : : "s20", "s21");
            if (KIND == 42)  // f32 multiply with a 32-bit literal constant
                asm volatile("v_mul_f32 %0, 0x3f800347, %0\n\tv_mul_f32 %1, 0x3f800347, %1\n\tv_mul_f32 %2, 0x3f800347, %2\n\tv_mul_f32 %3, 0x3f800347, %3\n\t"
                             "v_mul_f32 %4, 0x3f800347, %4\n\tv_mul_f32 %5, 0x3f800347, %5\n\tv_mul_f32 %6, 0x3f800347, %6\n\tv_mul_f32 %7, 0x3f800347, %7"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            if (KIND == 43) REP8_2OP("v_add_f32", a0, a1, a2, a3, a4, a5, a6, a7, b);  // placeholder: identical to kind 1
            if (KIND == 44)  // VOP2 f32 with a scalar-register operand
                asm volatile("v_add_f32 %0, s20, %0\n\tv_add_f32 %1, s20, %1\n\tv_add_f32 %2, s20, %2\n\tv_add_f32 %3, s20, %3\n\t"
                             "v_add_f32 %4, s20, %4\n\tv_add_f32 %5, s20, %5\n\tv_add_f32 %6, s20, %6\n\tv_add_f32 %7, s20, %7"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : : "s20");
            if (KIND == 45)  // VOP3 f32 fma, vector registers and an inline constant
                asm volatile("v_fma_f32 %0, %0, %8, 0.5\n\tv_fma_f32 %1, %1, %8, 0.5\n\tv_fma_f32 %2, %2, %8, 0.5\n\tv_fma_f32 %3, %3, %8, 0.5\n\t"
                             "v_fma_f32 %4, %4, %8, 0.5\n\tv_fma_f32 %5, %5, %8, 0.5\n\tv_fma_f32 %6, %6, %8, 0.5\n\tv_fma_f32 %7, %7, %8, 0.5"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
            if (KIND == 46)  // VOP3 f32 add with the clamp modifier
                asm volatile("v_add_f32_e64 %0, %0, %8 clamp\n\tv_add_f32_e64 %1, %1, %8 clamp\n\tv_add_f32_e64 %2, %2, %8 clamp\n\tv_add_f32_e64 %3, %3, %8 clamp\n\t"
                             "v_add_f32_e64 %4, %4, %8 clamp\n\tv_add_f32_e64 %5, %5, %8 clamp\n\tv_add_f32_e64 %6, %6, %8 clamp\n\tv_add_f32_e64 %7, %7, %8 clamp"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
            if (KIND == 47)  // VOP3 f32 add with |.| and negation modifiers
                asm volatile("v_add_f32_e64 %0, |%0|, -%8\n\tv_add_f32_e64 %1, |%1|, -%8\n\tv_add_f32_e64 %2, |%2|, -%8\n\tv_add_f32_e64 %3, |%3|, -%8\n\t"
                             "v_add_f32_e64 %4, |%4|, -%8\n\tv_add_f32_e64 %5, |%5|, -%8\n\tv_add_f32_e64 %6, |%6|, -%8\n\tv_add_f32_e64 %7, |%7|, -%8"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
            if (KIND == 48)  // f64 fma with a scalar pair operand
                asm volatile("v_fma_f64 %0, %0, s[20:21], %8\n\tv_fma_f64 %1, %1, s[20:21], %8\n\tv_fma_f64 %2, %2, s[20:21], %8\n\tv_fma_f64 %3, %3, s[20:21], %8\n\t"
                             "v_fma_f64 %4, %4, s[20:21], %8\n\tv_fma_f64 %5, %5, s[20:21], %8\n\tv_fma_f64 %6, %6, s[20:21], %8\n\tv_fma_f64 %7, %7, s[20:21], %8"
                             : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(f) : "s20", "s21");
            if (KIND == 49)  // integer add with a scalar operand
                asm volatile("v_add_u32 %0, s20, %0\n\tv_add_u32 %1, s20, %1\n\tv_add_u32 %2, s20, %2\n\tv_add_u32 %3, s20, %3\n\t"
                             "v_add_u32 %4, s20, %4\n\tv_add_u32 %5, s20, %5\n\tv_add_u32 %6, s20, %6\n\tv_add_u32 %7, s20, %7"
                             : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : : "s20");
            if (KIND == 50)  // move from a scalar register
                asm volatile("v_mov_b32 %0, s20\n\tv_mov_b32 %1, s20\n\tv_mov_b32 %2, s20\n\tv_mov_b32 %3, s20\n\tv_mov_b32 %4, s20\n\tv_mov_b32 %5, s20\n\tv_mov_b32 %6, s20\n\tv_mov_b32 %7, s20"
                             : "=v"(i0), "=v"(i1), "=v"(i2), "=v"(i3), "=v"(i4), "=v"(i5), "=v"(i6), "=v"(i7) : : "s20");
            if (KIND == 51) REP8_2OP("v_ashrrev_i32", i0, i1, i2, i3, i4, i5, i6, i7, lane);
            if (KIND == 52) REP8_2OP("v_sub_u32", i0, i1, i2, i3, i4, i5, i6, i7, lane);
            if (KIND == 53)
                asm volatile("v_cvt_i32_f32 %0, %8\n\tv_cvt_i32_f32 %1, %9\n\tv_cvt_i32_f32 %2, %10\n\tv_cvt_i32_f32 %3, %11\n\t"
                             "v_cvt_i32_f32 %4, %12\n\tv_cvt_i32_f32 %5, %13\n\tv_cvt_i32_f32 %6, %14\n\tv_cvt_i32_f32 %7, %15"
                             : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7)
                             : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7));
            if (KIND == 54) REP8_1OP("v_rndne_f32", a0, a1, a2, a3, a4, a5, a6, a7);
            if (KIND == 17)  // the kernel's mix: two f32 to one f64
                asm volatile("v_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %4, %5\n\tv_fma_f64 %2, %2, %6, %7\n\t"
                             "v_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %4, %5\n\tv_fma_f64 %3, %3, %6, %7\n\tv_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %4, %5"
                             : "+v"(a0), "+v"(a1), "+v"(d0), "+v"(d1) : "v"(b), "v"(c), "v"(e), "v"(f));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0)
        cyc[blockIdx.x] = t1 - t0;
    const float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7) + (float)(i0 + i1 + i2 + i3 + i4 + i5 + i6 + i7);
    if (s == 1.2345678e-30f)
        *sink = s;
}

template <int KIND>
static void run(const char *name, int nw)
{
    const int iters = 2000;
    unsigned long long *d_cyc;
    float *sink;
    CK(hipMalloc(&d_cyc, 8 * (size_t)nw));
    CK(hipMalloc(&sink, 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_rate<KIND>, dim3(nw), dim3(64), 0, 0, 10, d_cyc, sink);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_rate<KIND>, dim3(nw), dim3(64), 0, 0, iters, d_cyc, sink);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> c(nw);
    CK(hipMemcpy(c.data(), d_cyc, 8 * (size_t)nw, hipMemcpyDeviceToHost));
    std::sort(c.begin(), c.end());
    const double n_inst = (double)iters * 64.0;  // per wave
    const int per_simd = nw / 1024 > 0 ? nw / 1024 : 1;
    std::printf("%-16s waves %4d: %.2f cycles per instruction per wave (median), %.2f per SIMD; kernel %.3f ms = %.2f GHz-cycles per instruction per SIMD at 2.4 GHz\n",
                name, nw, (double)c[nw / 2] / n_inst, (double)c[nw / 2] / n_inst / per_simd, ms, ms * 1e-3 * 2.4e9 / (n_inst * per_simd));
    CK(hipFree(d_cyc));
    CK(hipFree(sink));
}

int main()
{
    for (int nw : {4096}) {
        run<0>("v_fma_f32", nw);
        run<1>("v_add_f32", nw);
        run<2>("v_mul_f32", nw);
        run<3>("v_fma_f64", nw);
        run<4>("v_add_f64", nw);
        run<5>("v_mul_f64", nw);
        run<6>("v_mov_b32", nw);
        run<7>("v_cndmask_b32", nw);
        run<8>("v_mov_b32_dpp", nw);
        run<9>("v_cvt_f64_f32", nw);
        run<10>("v_rcp_f32", nw);
        run<11>("v_rndne_f64", nw);
        run<12>("v_readlane_b32", nw);
        run<13>("v_pk_mul_f32", nw);
        run<14>("v_add_u32", nw);
        run<15>("v_cmp_lt_f32", nw);
        run<16>("s_nop 0", nw);
        run<17>("mix 2 f32 : 1 f64", nw);
        run<18>("v_cndmask (sgpr cond)", nw);
        run<19>("v_cmp + v_cndmask (per pair)", nw);
        run<20>("v_max_f32", nw);
        run<21>("v_and_b32", nw);
        run<22>("v_lshlrev_b32", nw);
        run<23>("v_fract_f64", nw);
        run<24>("v_cvt_f32_f64", nw);
        run<25>("v_mul_lo_u32", nw);
        run<26>("v_med3_i32", nw);
        run<27>("v_add_f32_dpp", nw);
        run<28>("v_cmp_lt_f32 -> sgpr", nw);
        run<29>("v_cndmask 0, v, vcc", nw);
        run<30>("v_ldexp_f64", nw);
        run<31>("ds_bpermute_b32", nw);
        run<32>("v_cmp + 4 v_cndmask", nw);
        run<33>("s_mov vcc + 8 v_cndmask", nw);
        run<34>("v_cndmask d, a, b, vcc", nw);
        run<35>("v_xor_b32", nw);
        run<36>("v_sub_f32", nw);
        run<37>("v_bfi_b32", nw);
        run<38>("v_min_f32", nw);
        run<39>("v_mov_b64", nw);
        run<40>("v_fma_f32 v, s, const", nw);
        run<41>("v_add_f64 v, s[2]", nw);
        run<42>("v_mul_f32 literal", nw);
        run<44>("v_add_f32 v, s, v", nw);
        run<45>("v_fma_f32 v,v,v,0.5", nw);
        run<46>("v_add_f32 clamp", nw);
        run<47>("v_add_f32 |a|, -b", nw);
        run<48>("v_fma_f64 v,v,s[2],v", nw);
        run<49>("v_add_u32 v, s, v", nw);
        run<50>("v_mov_b32 v, s", nw);
        run<51>("v_ashrrev_i32", nw);
        run<52>("v_sub_u32", nw);
        run<53>("v_cvt_i32_f32", nw);
        run<54>("v_rndne_f32", nw);
    }
    return 0;
}
