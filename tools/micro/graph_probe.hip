// graph_probe.hip -- does a HIP graph shorten a chain of tiny dependent kernels on this box?
// Ten kernels of one wave each, back to back on one stream, against the same ten as a graph of kernel nodes:
// wall time per chain over many repetitions (what the time-tiled path's fifteen small launches would gain).
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/graph_probe tools/micro/graph_probe.hip && /tmp/graph_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void tiny(float *p, int k) { if (threadIdx.x == 0) p[k] += 1.0f; }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main()
{
    const int N = 10, reps = 2000;
    float *d;
    CK(hipMalloc(&d, 4096));
    CK(hipMemset(d, 0, 4096));
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    auto chain = [&]() { for (int k = 0; k < N; k++) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, st, d, k); };
    for (int i = 0; i < 50; i++) chain();
    CK(hipStreamSynchronize(st));
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < reps; i++) chain();
    CK(hipStreamSynchronize(st));
    double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
    printf("stream: %.1f us per chain of %d kernels (%.2f us each)\n", us, N, us / N);
    // the same chain as a graph
    hipGraph_t g;
    CK(hipGraphCreate(&g, 0));
    std::vector<hipGraphNode_t> nodes(N);
    std::vector<int> ks(N);
    for (int k = 0; k < N; k++) {
        ks[k] = k;
        void *args[2] = {&d, &ks[k]};
        hipKernelNodeParams kp = {};
        kp.func = (void *)tiny;
        kp.gridDim = dim3(1);
        kp.blockDim = dim3(64);
        kp.kernelParams = args;
        CK(hipGraphAddKernelNode(&nodes[k], g, k ? &nodes[k - 1] : nullptr, k ? 1 : 0, &kp));
    }
    hipGraphExec_t ge;
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int i = 0; i < 50; i++) CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < reps; i++) CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
    printf("graph : %.1f us per chain of %d kernels (%.2f us each)\n", us, N, us / N);
    return 0;
}
