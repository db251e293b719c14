// stream_floor.hip -- empirical memory ceiling for the psk_soft access geometry (diagnostic).
// 4096 single-wave workgroups, each streaming one 2 MiB channel row in blocks of 8 KiB and writing
// four output rows (16 / 8 / 4 / 8 bytes per lane per block), with different load patterns.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

typedef float f4u __attribute__((ext_vector_type(4), aligned(8)));

// MODE 0: lane-contiguous 128 B (8 x dwordx4 at lane stride 128 B)    [what psk_fast_kernel does]
// MODE 1: wave-coalesced (piece j at 1024*j + 16*lane)
// MODE 2: mode 0 loads, no stores
// MODE 3: mode 1 loads, no stores
// MODE 4: mode 0, stores only soft
// MODE 5: mode 0 loads, nontemporal stores
// MODE 6: nontemporal loads + nontemporal stores
// MODE 7: mode 0, the three narrow streams written every 4th block as 4x wider vectors
template <int MODE>
__global__ __launch_bounds__(64) void k_stream(const float* __restrict__ in, size_t row_floats, int n_blocks,
                                               float* __restrict__ soft, float* __restrict__ phase,
                                               short* __restrict__ sidx, short* __restrict__ bits, size_t cap)
{
    const int lane = threadIdx.x;
    const float* row = in + (size_t)blockIdx.x * row_floats;
    float* so = soft + (size_t)blockIdx.x * 2 * cap;
    float* ph = phase + (size_t)blockIdx.x * cap;
    short* sx = sidx + (size_t)blockIdx.x * cap;
    short* bi = bits + (size_t)blockIdx.x * 2 * cap;
    float acc = 0.f;
    for (int c = 0; c < n_blocks; c++) {
        const f4u* src = reinterpret_cast<const f4u*>(row + (size_t)c * 2048);
        f4u t[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (MODE == 6) t[j] = __builtin_nontemporal_load(&src[lane * 8 + j]);
            else t[j] = (MODE == 1 || MODE == 3) ? src[j * 64 + lane] : src[lane * 8 + j];
        }
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 8; j++) s += t[j].x * t[j].y + t[j].z * t[j].w;
        acc += s;
        if (MODE == 2 || MODE == 3) continue;
        const size_t i0 = (size_t)c * 128 + 2 * lane;
        typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
        typedef short s2u __attribute__((ext_vector_type(2), aligned(4)));
        typedef short s4u __attribute__((ext_vector_type(4), aligned(4)));
        f4u v = {s, acc, s, acc};
        if (MODE == 5 || MODE == 6) {
            typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
            typedef short s2u __attribute__((ext_vector_type(2), aligned(4)));
            typedef short s4u __attribute__((ext_vector_type(4), aligned(4)));
            __builtin_nontemporal_store(v, reinterpret_cast<f4u*>(so + 2 * i0));
            f2u p2 = {s, acc};
            __builtin_nontemporal_store(p2, reinterpret_cast<f2u*>(ph + i0));
            s2u x2 = {(short)(s > 0), (short)(acc > 0)};
            __builtin_nontemporal_store(x2, reinterpret_cast<s2u*>(sx + i0));
            s4u b4 = {(short)(s > 0), 0, (short)(acc > 0), 0};
            __builtin_nontemporal_store(b4, reinterpret_cast<s4u*>(bi + 2 * i0));
            continue;
        }
        *reinterpret_cast<f4u*>(so + 2 * i0) = v;
        if (MODE == 4) continue;
        if (MODE == 7) {
            // every 4th block: one lane writes 4 blocks' worth for 16 consecutive symbols... modelled as
            // lanes 0..15 writing 4x wider vectors (same bytes, quarter the instructions / wider segments)
            if ((c & 3) == 3) {
                const size_t base = (size_t)(c - 3) * 128;
                f4u pv = {s, acc, s, acc};
                f4u* pp = reinterpret_cast<f4u*>(ph + base);   // 512 floats = 128 f4u -> 2 per lane
                pp[lane] = pv; pp[lane + 64] = pv;
                typedef short s8u __attribute__((ext_vector_type(8), aligned(4)));
                s8u sv = {1, 0, 1, 0, 1, 0, 1, 0};
                reinterpret_cast<s8u*>(sx + base)[lane] = sv;             // 512 shorts = 64 x 8
                s8u* bb = reinterpret_cast<s8u*>(bi + 2 * base);           // 1024 shorts = 128 x 8
                bb[lane] = sv; bb[lane + 64] = sv;
            }
            continue;
        }
        if (MODE == 8) {
            // narrow streams every 2nd block: even lanes write the first block's 4 symbols, odd lanes the
            // second block's (what a pairwise lane exchange would produce): 16 / 16 / 8 bytes per lane
            if (c & 1) {
                const size_t base = (size_t)(c - 1) * 128 + (lane & 1) * 128 + (lane >> 1) * 4;
                f4u pv = {s, acc, s, acc};
                *reinterpret_cast<f4u*>(ph + base) = pv;
                typedef short s8u __attribute__((ext_vector_type(8), aligned(4)));
                typedef short s4u2 __attribute__((ext_vector_type(4), aligned(4)));
                s8u bv = {1, 0, 1, 0, 1, 0, 1, 0};
                *reinterpret_cast<s8u*>(bi + 2 * base) = bv;
                s4u2 xv = {1, 2, 3, 4};
                *reinterpret_cast<s4u2*>(sx + base) = xv;
            }
            continue;
        }
        f2u p2 = {s, acc};
        *reinterpret_cast<f2u*>(ph + i0) = p2;
        s2u x2 = {(short)(s > 0), (short)(acc > 0)};
        *reinterpret_cast<s2u*>(sx + i0) = x2;
        s4u b4 = {(short)(s > 0), 0, (short)(acc > 0), 0};
        *reinterpret_cast<s4u*>(bi + 2 * i0) = b4;
    }
    if (acc == 12345.678f) so[0] = acc;
}

template <int MODE>
float run(const float* in, size_t row_floats, int n_blocks, float* soft, float* phase, short* sidx, short* bits, size_t cap, int C)
{
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int i = 0; i < 2; i++) hipLaunchKernelGGL(k_stream<MODE>, dim3(C), dim3(64), 0, 0, in, row_floats, n_blocks, soft, phase, sidx, bits, cap);
    CHECK(hipEventRecord(a));
    const int reps = 10;
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL(k_stream<MODE>, dim3(C), dim3(64), 0, 0, in, row_floats, n_blocks, soft, phase, sidx, bits, cap);
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

// the same traffic with the NEXT block's loads issued before the current block is consumed (two blocks
// of a wave in flight): does the floor move with more bytes in flight per wave?
template <bool STORES>
__global__ __launch_bounds__(64) void k_stream_prefetch(const float* __restrict__ in, size_t row_floats, int n_blocks,
                                                        float* __restrict__ soft, float* __restrict__ phase,
                                                        short* __restrict__ sidx, short* __restrict__ bits, size_t cap)
{
    const int lane = threadIdx.x;
    const float* row = in + (size_t)blockIdx.x * row_floats;
    float* so = soft + (size_t)blockIdx.x * 2 * cap;
    float* ph = phase + (size_t)blockIdx.x * cap;
    short* sx = sidx + (size_t)blockIdx.x * cap;
    short* bi = bits + (size_t)blockIdx.x * 2 * cap;
    typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
    typedef short s2u __attribute__((ext_vector_type(2), aligned(4)));
    typedef short s4u __attribute__((ext_vector_type(4), aligned(4)));
    float acc = 0.f;
    f4u cur[8], nxt[8];
    {
        const f4u* src = reinterpret_cast<const f4u*>(row);
#pragma unroll
        for (int j = 0; j < 8; j++) cur[j] = src[lane * 8 + j];
    }
    for (int c = 0; c < n_blocks; c++) {
        if (c + 1 < n_blocks) {
            const f4u* src = reinterpret_cast<const f4u*>(row + (size_t)(c + 1) * 2048);
#pragma unroll
            for (int j = 0; j < 8; j++) nxt[j] = src[lane * 8 + j];
        }
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 8; j++) s += cur[j].x * cur[j].y + cur[j].z * cur[j].w;
        acc += s;
        if (STORES) {
            const size_t i0 = (size_t)c * 128 + 2 * lane;
            f4u v = {s, acc, s, acc};
            *reinterpret_cast<f4u*>(so + 2 * i0) = v;
            f2u p2 = {s, acc};
            *reinterpret_cast<f2u*>(ph + i0) = p2;
            s2u x2 = {(short)(s > 0), (short)(acc > 0)};
            *reinterpret_cast<s2u*>(sx + i0) = x2;
            s4u b4 = {(short)(s > 0), 0, (short)(acc > 0), 0};
            *reinterpret_cast<s4u*>(bi + 2 * i0) = b4;
        }
#pragma unroll
        for (int j = 0; j < 8; j++) cur[j] = nxt[j];
    }
    if (acc == 12345.678f) so[0] = acc;
}

template <bool STORES>
float run_prefetch(const float* in, size_t row_floats, int n_blocks, float* soft, float* phase, short* sidx, short* bits, size_t cap, int C)
{
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int i = 0; i < 2; i++) hipLaunchKernelGGL(k_stream_prefetch<STORES>, dim3(C), dim3(64), 0, 0, in, row_floats, n_blocks, soft, phase, sidx, bits, cap);
    CHECK(hipEventRecord(a));
    const int reps = 10;
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL(k_stream_prefetch<STORES>, dim3(C), dim3(64), 0, 0, in, row_floats, n_blocks, soft, phase, sidx, bits, cap);
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main(int argc, char **argv)
{
    // argv[1]: extra symbols of padding per output row (a multiple of 64 keeps the rows on 128-byte boundaries); argv[2] = 1: floor lines only
    const size_t pad = argc > 1 ? (size_t)atol(argv[1]) : 0; const bool quick = argc > 2 && atoi(argv[2]) != 0;
    const int C = 4096; const size_t N = 1 << 18; const size_t row_floats = 2 * N; const size_t cap = (N / 8 + 2 + 63) / 64 * 64 + pad;  /* output rows on 128-byte boundaries, as bench.py lays them out */ const int n_blocks = (int)(N / 8 / 128);
    float *in, *soft, *phase; short *sidx, *bits;
    const bool one_block = argc > 3 && atoi(argv[3]) != 0;  // argv[3] = 1: all five buffers carved out of ONE allocation
    if (one_block) {
        const size_t b_in = sizeof(float) * row_floats * C, b_soft = sizeof(float) * 2 * cap * C, b_ph = sizeof(float) * cap * C,
                     b_sx = sizeof(short) * cap * C, b_bi = sizeof(short) * 2 * cap * C;
        auto up = [](size_t v) { return (v + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1); };
        char *blk; CHECK(hipMalloc(&blk, up(b_in) + up(b_soft) + up(b_ph) + up(b_sx) + up(b_bi)));
        in = (float *)blk; soft = (float *)(blk + up(b_in)); phase = (float *)((char *)soft + up(b_soft));
        sidx = (short *)((char *)phase + up(b_ph)); bits = (short *)((char *)sidx + up(b_sx));
        CHECK(hipMemset(in, 0x3c, b_in));
    } else {
    CHECK(hipMalloc(&in, sizeof(float) * row_floats * C)); CHECK(hipMemset(in, 0x3c, sizeof(float) * row_floats * C));
    CHECK(hipMalloc(&soft, sizeof(float) * 2 * cap * C)); CHECK(hipMalloc(&phase, sizeof(float) * cap * C));
    CHECK(hipMalloc(&sidx, sizeof(short) * cap * C)); CHECK(hipMalloc(&bits, sizeof(short) * 2 * cap * C));
    }
    const double rd = 8.0 * N * C, wr = (double)C * (N / 8) * 18.0;
    float ms;
    ms = run<0>(in, row_floats, n_blocks, soft, phase, sidx, bits, cap, C); printf("mode0 lane-contiguous loads + 4 stores : %.3f ms  read %.2f TB/s total %.2f TB/s\n", ms, rd / ms / 1e9, (rd + wr) / ms / 1e9);
    if (quick) { printf("   in %p soft %p phase %p sidx %p bits %p cap %zu\n", (void*)in, (void*)soft, (void*)phase, (void*)sidx, (void*)bits, cap); return 0; }
    ms = run<1>(in, row_floats, n_blocks, soft, phase, sidx, bits, cap, C); printf("mode1 wave-coalesced loads + 4 stores  : %.3f ms  read %.2f TB/s total %.2f TB/s\n", ms, rd / ms / 1e9, (rd + wr) / ms / 1e9);
    ms = run<2>(in, row_floats, n_blocks, soft, phase, sidx, bits, cap, C); printf("mode2 lane-contiguous loads only        : %.3f ms  read %.2f TB/s\n", ms, rd / ms / 1e9);
    ms = run<3>(in, row_floats, n_blocks, soft, phase, sidx, bits, cap, C); printf("mode3 wave-coalesced loads only         : %.3f ms  read %.2f TB/s\n", ms, rd / ms / 1e9);
    ms = run<4>(in, row_floats, n_blocks, soft, phase, sidx, bits, cap, C); printf("mode4 lane-contiguous loads + soft only : %.3f ms  read %.2f TB/s\n", ms, rd / ms / 1e9);
    ms = run<5>(in, row_floats, n_blocks, soft, phase, sidx, bits, cap, C); printf("mode5 nontemporal stores                : %.3f ms  read %.2f TB/s total %.2f TB/s\n", ms, rd / ms / 1e9, (rd + wr) / ms / 1e9);
    ms = run<6>(in, row_floats, n_blocks, soft, phase, sidx, bits, cap, C); printf("mode6 nontemporal loads and stores       : %.3f ms  read %.2f TB/s total %.2f TB/s\n", ms, rd / ms / 1e9, (rd + wr) / ms / 1e9);
    ms = run<7>(in, row_floats, n_blocks, soft, phase, sidx, bits, cap, C); printf("mode7 narrow streams batched x4          : %.3f ms  read %.2f TB/s total %.2f TB/s\n", ms, rd / ms / 1e9, (rd + wr) / ms / 1e9);
    ms = run<8>(in, row_floats, n_blocks, soft, phase, sidx, bits, cap, C); printf("mode8 narrow streams batched x2 (pairs)   : %.3f ms  read %.2f TB/s total %.2f TB/s\n", ms, rd / ms / 1e9, (rd + wr) / ms / 1e9);
    ms = run_prefetch<true>(in, row_floats, n_blocks, soft, phase, sidx, bits, cap, C); printf("mode0 with the next block prefetched   : %.3f ms  read %.2f TB/s total %.2f TB/s\n", ms, rd / ms / 1e9, (rd + wr) / ms / 1e9);
    ms = run_prefetch<false>(in, row_floats, n_blocks, soft, phase, sidx, bits, cap, C); printf("mode2 with the next block prefetched   : %.3f ms  read %.2f TB/s\n", ms, rd / ms / 1e9);
    ms = run<0>(in, row_floats, n_blocks, soft, phase, sidx, bits, cap, C); printf("mode0 again                              : %.3f ms\n", ms);
    // does the power-of-two row stride of the input (2 MiB per channel) matter?  same kernels, padded rows
    CHECK(hipFree(in));
    for (size_t pad : {(size_t)64, (size_t)1024 + 64, (size_t)8192 + 192, (size_t)65536 + 1088}) {
        const size_t rf = row_floats + pad;
        CHECK(hipMalloc(&in, sizeof(float) * rf * C)); CHECK(hipMemset(in, 0x3c, sizeof(float) * rf * C));
        float m0 = run<0>(in, rf, n_blocks, soft, phase, sidx, bits, cap, C);
        float m2 = run<2>(in, rf, n_blocks, soft, phase, sidx, bits, cap, C);
        printf("row pad %6zu floats: mode0 %.3f ms  mode2 (loads only) %.3f ms\n", pad, m0, m2);
        CHECK(hipFree(in));
    }
    return 0;
}
