// prefetch_probe.hip -- does touching the next 8 KiB block one compute phase early help a wave that
// alternates "load 8 KiB -> long compute -> store" (the shape of psk_fast_kernel)?  (diagnostic)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float f4u __attribute__((ext_vector_type(4), aligned(8)));

// MODE 0: no prefetch   MODE 1: one dword per 128-B line of the next block   MODE 2: two dwords (both 64-B halves)
template <int MODE, int WORK>
__global__ __launch_bounds__(64, 4) void k_probe(const float* __restrict__ in, size_t row_floats, int n_blocks,
                                                 float* __restrict__ soft, size_t cap)
{
    const int lane = threadIdx.x;
    const float* row = in + (size_t)blockIdx.x * row_floats;
    float* so = soft + (size_t)blockIdx.x * 2 * cap;
    float acc = 0.f, pf = 0.f;
    for (int c = 0; c < n_blocks; c++) {
        const f4u* src = reinterpret_cast<const f4u*>(row + (size_t)c * 2048);
        acc += pf;  // consume the prefetched word (forces its wait here, before the real loads)
        f4u t[8];
#pragma unroll
        for (int j = 0; j < 8; j++) t[j] = src[lane * 8 + j];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 8; j++) s += t[j].x * t[j].y + t[j].z * t[j].w;
        if (MODE >= 1 && c + 1 < n_blocks) {
            const float* nx = row + (size_t)(c + 1) * 2048 + lane * 32;
            pf = nx[0];
            if (MODE == 2) pf += nx[16];
        }
        // a dependent ALU chain standing in for the per-block compute
        float v = s;
#pragma unroll 16
        for (int i = 0; i < WORK; i++) v = v * 1.0000001f + 0.5f;
        acc += v;
        const size_t i0 = (size_t)c * 128 + 2 * lane;
        f4u o = {s, acc, v, acc};
        *reinterpret_cast<f4u*>(so + 2 * i0) = o;
    }
    if (acc == 12345.678f) so[0] = acc;
}

template <int MODE, int WORK>
float run(const float* in, size_t row_floats, int n_blocks, float* soft, size_t cap, int C)
{
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int i = 0; i < 2; i++) hipLaunchKernelGGL((k_probe<MODE, WORK>), dim3(C), dim3(64), 0, 0, in, row_floats, n_blocks, soft, cap);
    CHECK(hipEventRecord(a));
    const int reps = 8;
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL((k_probe<MODE, WORK>), dim3(C), dim3(64), 0, 0, in, row_floats, n_blocks, soft, cap);
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main()
{
    const int C = 4096; const size_t N = 1 << 18; const size_t row_floats = 2 * N; const size_t cap = N / 8 + 2; const int n_blocks = (int)(N / 8 / 128);
    float *in, *soft;
    CHECK(hipMalloc(&in, sizeof(float) * row_floats * C)); CHECK(hipMemset(in, 0x3c, sizeof(float) * row_floats * C));
    CHECK(hipMalloc(&soft, sizeof(float) * 2 * cap * C));
#define ROW(W) printf("work %5d: none %.3f ms | 1 dword/line %.3f ms | 2 dwords/line %.3f ms\n", W, \
    run<0, W>(in, row_floats, n_blocks, soft, cap, C), run<1, W>(in, row_floats, n_blocks, soft, cap, C), run<2, W>(in, row_floats, n_blocks, soft, cap, C));
    ROW(0) ROW(256) ROW(512) ROW(1024) ROW(1536)
    return 0;
}
