// mfma_f64_chain_probe.hip -- is v_mfma_f64_4x4x4f64 with A = 1 a chain of four ROUNDED double additions in k order,
//     D = fl(fl(fl(fl(C + B[k=0]) + B[1]) + B[2]) + B[3])   (k = lane / 16, see mfma_f64_probe.hip) ?
// and does `v_mov_b32_dpp row_newbcast:n` exist on gfx950?  Compares 10^6 random cases (operands of mixed sign and
// magnitude so that every step rounds) bit for bit with the four v_add_f64 of the same order, and with other orders.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <random>

__global__ void k_chain(const double *c, const double *b, double *d, double *dref, double *dalt, int n)
{
    const int l = threadIdx.x;
    for (int t = blockIdx.x; t < n; t += gridDim.x) {
        const double C = c[t];                       // uniform
        const double B = b[4 * t + (l >> 4)];        // row k = l / 16 holds term k
        const double D = __builtin_amdgcn_mfma_f64_4x4x4f64(1.0, B, C, 0, 0, 0);
        if (l == 0) {
            d[t] = D;
            const double b0 = b[4 * t], b1 = b[4 * t + 1], b2 = b[4 * t + 2], b3 = b[4 * t + 3];
            double r = C + b0;
            r = r + b1;
            r = r + b2;
            r = r + b3;
            dref[t] = r;
            dalt[t] = C + ((b0 + b1) + (b2 + b3));
        }
    }
}

__global__ void k_bcast(const int *src, int *dst)
{
    const int l = threadIdx.x;
    int v = src[l];
    dst[l] = __builtin_amdgcn_update_dpp(0, v, 0x150 + 5, 0xF, 0xF, false);  // row_newbcast:5
}

int main()
{
    const int n = 1 << 20;
    std::mt19937_64 rng(7);
    std::uniform_real_distribution<double> U(-1.0, 1.0);
    double *hc = new double[n], *hb = new double[4 * n], *hd = new double[n], *hr = new double[n], *ha = new double[n];
    for (int t = 0; t < n; t++) {
        const double s = ldexp(U(rng), (int)(rng() % 8));
        hc[t] = s;
        for (int k = 0; k < 4; k++) hb[4 * t + k] = ldexp(U(rng), (int)(rng() % 8) - 4) * (k & 1 ? 1 : -1);
    }
    double *dc, *db, *dd, *dr, *da;
    hipMalloc(&dc, 8 * n); hipMalloc(&db, 32 * n); hipMalloc(&dd, 8 * n); hipMalloc(&dr, 8 * n); hipMalloc(&da, 8 * n);
    hipMemcpy(dc, hc, 8 * n, hipMemcpyHostToDevice); hipMemcpy(db, hb, 32 * n, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_chain, dim3(1024), dim3(64), 0, 0, dc, db, dd, dr, da, n);
    hipMemcpy(hd, dd, 8 * n, hipMemcpyDeviceToHost); hipMemcpy(hr, dr, 8 * n, hipMemcpyDeviceToHost); hipMemcpy(ha, da, 8 * n, hipMemcpyDeviceToHost);
    long same_seq = 0, same_alt = 0, seq_ne_alt = 0;
    for (int t = 0; t < n; t++) {
        same_seq += memcmp(&hd[t], &hr[t], 8) == 0;
        same_alt += memcmp(&hd[t], &ha[t], 8) == 0;
        seq_ne_alt += memcmp(&hr[t], &ha[t], 8) != 0;
    }
    printf("cases %d: MFMA == sequential k-order chain of rounded adds: %ld; MFMA == pairwise order: %ld; (orders differ in %ld cases)\n",
           n, same_seq, same_alt, seq_ne_alt);
    int hs[64], ho[64], *ds, *dо;
    for (int l = 0; l < 64; l++) hs[l] = 100 + l;
    hipMalloc(&ds, 256); hipMalloc(&dо, 256);
    hipMemcpy(ds, hs, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_bcast, dim3(1), dim3(64), 0, 0, ds, dо);
    hipMemcpy(ho, dо, 256, hipMemcpyDeviceToHost);
    printf("row_newbcast:5 ->");
    for (int l = 0; l < 64; l += 8) printf(" [%d]=%d", l, ho[l]);
    printf("\n");
    return 0;
}
