// placement_probe.hip -- does the RELATIVE PLACEMENT of the input and output arrays in HBM change the
// streaming ceiling?  Same access pattern as stream_floor mode 0; one arena, the input placed at
// different offsets from its start, the four output arrays fixed behind it.  (diagnostic)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float f4u __attribute__((ext_vector_type(4), aligned(8)));
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
        for (int j = 0; j < 8; j++) t[j] = src[lane * 8 + j];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 8; j++) s += t[j].x * t[j].y + t[j].z * t[j].w;
        acc += s;
        const size_t i0 = (size_t)c * 128 + 2 * lane;
        typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
        typedef short s2u __attribute__((ext_vector_type(2), aligned(4)));
        typedef short s4u __attribute__((ext_vector_type(4), aligned(4)));
        f4u v = {s, acc, s, acc};
        *reinterpret_cast<f4u*>(so + 2 * i0) = v;
        f2u p2 = {s, acc};
        *reinterpret_cast<f2u*>(ph + i0) = p2;
        s2u x2 = {(short)(s > 0), (short)(acc > 0)};
        *reinterpret_cast<s2u*>(sx + i0) = x2;
        s4u b4 = {(short)(s > 0), 0, (short)(acc > 0), 0};
        *reinterpret_cast<s4u*>(bi + 2 * i0) = b4;
    }
    if (acc == 12345.678f) so[0] = acc;
}
int main()
{
    const int C = 4096; const size_t N = 1 << 18; const size_t row_floats = 2 * N; const int n_blocks = (int)(N / 8 / 128);
    const size_t in_bytes = sizeof(float) * row_floats * C, slack = 64u << 20;
    for (size_t cap : {(size_t)(N / 8), (size_t)(N / 8 + 2), (size_t)(N / 8 + 64), (size_t)(N / 8 + 1024)}) {
        char* arena; const size_t out_bytes = (8 + 4 + 2 + 4) * cap * C + 4096 * 8;
        CHECK(hipMalloc(&arena, in_bytes + slack + out_bytes)); CHECK(hipMemset(arena, 0x3c, in_bytes + slack));
        char* o = arena + in_bytes + slack;
        float* soft = (float*)o; o += 8 * cap * C; o = (char*)(((size_t)o + 4095) & ~(size_t)4095);
        float* phase = (float*)o; o += 4 * cap * C; o = (char*)(((size_t)o + 4095) & ~(size_t)4095);
        short* sidx = (short*)o; o += 2 * cap * C; o = (char*)(((size_t)o + 4095) & ~(size_t)4095);
        short* bits = (short*)o;
        printf("cap %zu symbols per output row, arena %p:", cap, (void*)arena);
        for (size_t off : {(size_t)0, (size_t)4096, (size_t)65536, (size_t)(1u << 20), (size_t)(3u << 20) + 8192, (size_t)(17u << 20)}) {
            const float* in = (const float*)(arena + off);
            hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
            for (int i = 0; i < 2; i++) hipLaunchKernelGGL(k_stream, dim3(C), dim3(64), 0, 0, in, row_floats, n_blocks, soft, phase, sidx, bits, cap);
            CHECK(hipEventRecord(a));
            for (int i = 0; i < 10; i++) hipLaunchKernelGGL(k_stream, dim3(C), dim3(64), 0, 0, in, row_floats, n_blocks, soft, phase, sidx, bits, cap);
            CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
            float ms; CHECK(hipEventElapsedTime(&ms, a, b));
            printf("  off %zuK: %.3f", off >> 10, ms / 10);
        }
        printf("\n");
        CHECK(hipFree(arena));
    }
    return 0;
}
