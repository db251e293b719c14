// Do two partly-filling launches on two streams share the machine?  (the mixed batch.)
// Kernel A: `na` one-wave workgroups that claim 256 vector registers each (two to a SIMD) and spin for `us` microseconds;
// kernel B: `nb` one-wave workgroups of 128 registers (four to a SIMD).  Timed: A alone, B alone, A then B on one stream,
// A and B on two non-blocking streams forked and joined by events (the arrangement of psk_capi.cpp's window classes).
// build + run (GPU box): hipcc --offload-arch=gfx950 -O3 -o /tmp/overlap_probe overlap_probe.hip && /tmp/overlap_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

__global__ __launch_bounds__(64) void spin_wide(unsigned long long ticks, unsigned *sink)
{
    asm volatile("v_mov_b32 v250, 0" ::: "v250");  // 256 registers: two waves to a SIMD
    const unsigned long long t0 = wall_clock64();
    unsigned acc = 0;
    while (wall_clock64() - t0 < ticks) acc += 1u;
    if (acc == 0xFFFFFFFFu) *sink = acc;
}
__global__ __launch_bounds__(64) void spin_narrow(unsigned long long ticks, unsigned *sink)
{
    asm volatile("v_mov_b32 v120, 0" ::: "v120");  // 128 registers: four waves to a SIMD
    const unsigned long long t0 = wall_clock64();
    unsigned acc = 0;
    while (wall_clock64() - t0 < ticks) acc += 1u;
    if (acc == 0xFFFFFFFFu) *sink = acc;
}

int main(int argc, char **argv)
{
    const int na = argc > 1 ? std::atoi(argv[1]) : 1360, nb = argc > 2 ? std::atoi(argv[2]) : 2736;
    const double us = argc > 3 ? std::atof(argv[3]) : 1000.0;
    const unsigned long long ticks = (unsigned long long)(us * 100.0);  // wall_clock64: 100 MHz
    unsigned *sink;
    CK(hipMalloc(&sink, 4));
    hipStream_t s0, s1;
    CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    hipEvent_t t0, t1, fork, join;
    CK(hipEventCreate(&t0));
    CK(hipEventCreate(&t1));
    CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
    auto timed = [&](const char *what, auto body) {
        float best = 1e30f;
        for (int r = 0; r < 5; r++) {
            CK(hipEventRecord(t0, s0));
            body();
            CK(hipEventRecord(t1, s0));
            CK(hipEventSynchronize(t1));
            float ms;
            CK(hipEventElapsedTime(&ms, t0, t1));
            best = ms < best ? ms : best;
        }
        std::printf("%-58s %.3f ms\n", what, best);
    };
    std::printf("A: %d waves x 256 VGPRs, B: %d waves x 128 VGPRs, each spinning %.0f us\n", na, nb, us);
    timed("A alone", [&] { spin_wide<<<na, 64, 0, s0>>>(ticks, sink); });
    timed("B alone", [&] { spin_narrow<<<nb, 64, 0, s0>>>(ticks, sink); });
    timed("A then B, one stream", [&] { spin_wide<<<na, 64, 0, s0>>>(ticks, sink); spin_narrow<<<nb, 64, 0, s0>>>(ticks, sink); });
    auto forked = [&](bool a_first) {
        CK(hipEventRecord(fork, s0));
        CK(hipStreamWaitEvent(s1, fork, 0));
        if (a_first) {
            spin_wide<<<na, 64, 0, s0>>>(ticks, sink);
            spin_narrow<<<nb, 64, 0, s1>>>(ticks, sink);
        } else {
            spin_narrow<<<nb, 64, 0, s0>>>(ticks, sink);
            spin_wide<<<na, 64, 0, s1>>>(ticks, sink);
        }
        CK(hipEventRecord(join, s1));
        CK(hipStreamWaitEvent(s0, join, 0));
    };
    timed("A and B on two streams (A first)", [&] { forked(true); });
    timed("A and B on two streams (B first)", [&] { forked(false); });
    // how full does one launch make the machine?  2048 wide waves are every slot there is
    timed("A with 2048 waves (every 256-register slot)", [&] { spin_wide<<<2048, 64, 0, s0>>>(ticks, sink); });
    timed("A with 2049 waves", [&] { spin_wide<<<2049, 64, 0, s0>>>(ticks, sink); });
    timed("B with 4096 waves (every 128-register slot)", [&] { spin_narrow<<<4096, 64, 0, s0>>>(ticks, sink); });
    timed("B with 4097 waves", [&] { spin_narrow<<<4097, 64, 0, s0>>>(ticks, sink); });
    return 0;
}
