// mfma_f64_probe.hip -- what v_mfma_f64_4x4x4f64 computes, lane by lane, and whether its accumulation is a chain of
// ROUNDED double additions in k order (then one instruction is four steps of a sequential sum: the lane-after-lane
// chain of psk_fast_loop.h could run on the matrix core).
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/mfma_probe tools/micro/mfma_f64_probe.hip && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <random>

__global__ void k_map(const double *a, const double *b, const double *c, double *d)
{
    const int l = threadIdx.x;
    d[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], c[l], 0, 0, 0);
}

int main()
{
    double ha[64], hb[64], hc[64], hd[64], *da, *db, *dc, *dd;
    hipMalloc(&da, 512); hipMalloc(&db, 512); hipMalloc(&dc, 512); hipMalloc(&dd, 512);
    auto run = [&]() {
        hipMemcpy(da, ha, 512, hipMemcpyHostToDevice); hipMemcpy(db, hb, 512, hipMemcpyHostToDevice);
        hipMemcpy(dc, hc, 512, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_map, dim3(1), dim3(64), 0, 0, da, db, dc, dd);
        hipMemcpy(hd, dd, 512, hipMemcpyDeviceToHost);
    };
    // 1. which B lanes feed which D lane: A = 1, B(l) = 2^l, C = 0
    for (int l = 0; l < 64; l++) { ha[l] = 1.0; hb[l] = ldexp(1.0, l % 52); hc[l] = 0.0; }
    // two passes so that lanes >= 52 are distinguishable
    for (int pass = 0; pass < 2; pass++) {
        for (int l = 0; l < 64; l++) hb[l] = (pass == 0) ? (l < 32 ? ldexp(1.0, l) : 0.0) : (l >= 32 ? ldexp(1.0, l - 32) : 0.0);
        run();
        printf("B lanes feeding each D lane (A = 1), pass %d (lanes %s):\n", pass, pass ? "32-63" : "0-31");
        for (int l = 0; l < 64; l++) {
            uint64_t m = (uint64_t)hd[l];
            printf(" D%02d<-", l);
            for (int s = 0; s < 32; s++) if (m >> s & 1) printf("%d,", s + 32 * pass);
            if (l % 4 == 3) printf("\n");
        }
    }
    // 2. which A lanes: B = 1, A(l) = 2^l
    for (int pass = 0; pass < 2; pass++) {
        for (int l = 0; l < 64; l++) { hb[l] = 1.0; ha[l] = (pass == 0) ? (l < 32 ? ldexp(1.0, l) : 0.0) : (l >= 32 ? ldexp(1.0, l - 32) : 0.0); hc[l] = 0; }
        run();
        printf("A lanes feeding each D lane (B = 1), pass %d:\n", pass);
        for (int l = 0; l < 64; l++) {
            uint64_t m = (uint64_t)hd[l];
            printf(" D%02d<-", l);
            for (int s = 0; s < 32; s++) if (m >> s & 1) printf("%d,", s + 32 * pass);
            if (l % 4 == 3) printf("\n");
        }
    }
    // 3. C lane: A = B = 0, C(l) = l
    for (int l = 0; l < 64; l++) { ha[l] = 0; hb[l] = 0; hc[l] = l; }
    run();
    printf("C lane of each D lane:");
    for (int l = 0; l < 64; l++) printf(" %d", (int)hd[l]);
    printf("\n");
    return 0;
}
