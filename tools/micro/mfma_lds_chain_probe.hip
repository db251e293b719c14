// mfma_lds_chain_probe.hip -- the lane-after-lane recurrence of LinearFit's xySum (psk_fast_loop.h: fit_sums_chain)
// on the matrix core, with the operands staged through LDS instead of registers:
//     s_j = fl(fl(s_{j-1} - c_j) + t_j),   two symbols per lane, 128 per block (reference cpp/psk_soft.cpp:72, :78)
// v_mfma_f64_4x4x4f64 is four ROUNDED double additions in k order (mfma_f64_chain_probe.hip); D lane 0 takes its
// B operands from lanes 0, 16, 32, 48 and its A operands from the same lanes (mfma_f64_probe.hip).  Here: A = (-1, +1,
// -1, +1) in those lanes, B = (c0, t0, c1, t1) of step m read from LDS with one ds_read_b64 per step, D = C = the
// running sum, stored back to LDS (into the slot just consumed) by the same four lanes.  The loop has no VALU
// instruction at all.  Questions: (1) bit-identical to the v_add_f64 chain?  (2) does the MFMA ignore EXEC (the loop
// runs with only lanes 0/16/32/48 active)?  (3) cycles per 64-step chain, alone and next to VALU-bound waves, against
// the DPP chain of the kernel.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o /tmp/chain_probe tools/micro/mfma_lds_chain_probe.hip && /tmp/chain_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

__device__ __forceinline__ int up1_i(int v, int carry) { return __builtin_amdgcn_update_dpp(carry, v, 0x138, 0xF, 0xF, false); }
__device__ __forceinline__ double up1(double v, double carry)
{
    int lo = up1_i(__double2loint(v), __double2loint(carry));
    int hi = up1_i(__double2hiint(v), __double2hiint(carry));
    return __hiloint2double(hi, lo);
}

// the kernel's DPP chain (fit_sums_chain, xySum part)
__device__ __forceinline__ void chain_dpp(double s_in, double c0, double t0, double c1, double t1, double &x0, double &x1)
{
    double xs = s_in;
#pragma unroll 1
    for (int k = 0; k < 64; k += 4) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const double b = up1(xs, s_in);
            xs = ((b - c0) + t0 - c1) + t1;
        }
    }
    const double b = up1(xs, s_in);
    x0 = (b - c0) + t0;
    x1 = xs;
}

// operands through LDS, the additions on the matrix core.  lds: 2 KiB (64 steps x 4 doubles), 16-byte aligned.
template <bool MASKED>
__device__ __forceinline__ void chain_mfma(double s_in, double c0, double t0, double c1, double t1, double *lds, int lane,
                                           double &x0, double &x1)
{
    double2 *st = reinterpret_cast<double2 *>(lds) + 2 * lane;
    st[0] = make_double2(c0, t0);
    st[1] = make_double2(c1, t1);
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
    const int kq = lane >> 4;
    const double A = (kq & 1) ? 1.0 : -1.0;
    double *slot = lds + kq;  // + 4 * m
    double acc = s_in;
    if (!MASKED || (lane & 15) == 0) {
#pragma unroll 8
        for (int m = 0; m < 64; m++) {
            const double B = slot[4 * m];
            acc = __builtin_amdgcn_mfma_f64_4x4x4f64(A, B, acc, 0, 0, 0);
            if (MASKED || lane == 0)
                slot[4 * m] = acc;  // (lane 0: the chain; lanes 16, 32, 48 (MASKED): garbage into consumed operands)
        }
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    x1 = lds[4 * lane];
    const double b = lane ? lds[4 * (lane - 1)] : s_in;
    x0 = (b - c0) + t0;
}

// one lane walks the block: operands through LDS (one 16-byte read per symbol, issued ahead), two dependent additions per
// symbol, the sums back to LDS (into the slot just consumed); no cross-lane traffic at all.  lds: 2 KiB.
__device__ __forceinline__ void chain_serial(double s_in, double c0, double t0, double c1, double t1, double *lds, int lane,
                                             double &x0, double &x1)
{
    double2 *st = reinterpret_cast<double2 *>(lds) + 2 * lane;
    st[0] = make_double2(c0, t0);
    st[1] = make_double2(c1, t1);
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
    if (lane == 0) {
        double acc = s_in;
        const double2 *op = reinterpret_cast<const double2 *>(lds);
#pragma unroll 16
        for (int j = 0; j < 128; j++) {
            const double2 o = op[j];
            acc = (acc - o.x) + o.y;
            lds[2 * j] = acc;
        }
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    x0 = lds[4 * lane];
    x1 = lds[4 * lane + 2];
}

__device__ __forceinline__ uint64_t mix(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ double rnd_pm(uint64_t h, int emin, int espan)
{
    // +-[1,2) * 2^(emin + h % espan)
    const double m = 1.0 + (double)(h >> 12) * 0x1p-52;
    const int e = emin + (int)((h >> 3) % (uint64_t)espan);
    const double v = __hiloint2double((1023 + e) << 20, 0) * m;
    return (h & 1) ? -v : v;
}

// MODE 0: DPP chain, 1: MFMA + LDS (all lanes active, lane 0 stores), 2: MFMA + LDS with only lanes 0/16/32/48 active,
// 3: a VALU-bound filler (for the contention runs)
template <int MODE>
__global__ __launch_bounds__(64, 4) void k_chain(int reps, int filler_every, double *out, unsigned long long *cyc)
{
    __shared__ __attribute__((aligned(16))) double lds[256];
    const int lane = threadIdx.x;
    const int w = blockIdx.x;
    const bool filler = filler_every && w >= (int)gridDim.x / filler_every;  // (SIMD mates are w, w + grid/4, ...: one chain wave per SIMD)
    if (filler) {
        // VALU-issue-bound: eight independent chains of f32 fma and four of f64, no memory (what the chain's wave competes with)
        float a = (float)lane, a1 = a + 1, a2 = a + 2, a3 = a + 3, a4 = a + 4, a5 = a + 5, a6 = a + 6, a7 = a + 7, b = 1.0001f;
        double d = 1.0, d1 = 2.0, d2 = 3.0, d3 = 4.0;
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int r = 0; r < reps * 16; r++) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                a = __builtin_fmaf(a, b, 0.5f), a1 = __builtin_fmaf(a1, b, 0.5f), a2 = __builtin_fmaf(a2, b, 0.5f), a3 = __builtin_fmaf(a3, b, 0.5f);
                a4 = __builtin_fmaf(a4, b, 0.5f), a5 = __builtin_fmaf(a5, b, 0.5f), a6 = __builtin_fmaf(a6, b, 0.5f), a7 = __builtin_fmaf(a7, b, 0.5f);
                d = __builtin_fma(d, 1.0000001, 0.5), d1 = __builtin_fma(d1, 1.0000001, 0.5), d2 = __builtin_fma(d2, 1.0000001, 0.5), d3 = __builtin_fma(d3, 1.0000001, 0.5);
            }
        }
        a += a1 + a2 + a3 + a4 + a5 + a6 + a7;
        d += d1 + d2 + d3;
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        if (lane == 0)
            cyc[w] = t1 - t0;
        if (a == 1234.5f && d == 0.25)
            out[0] = d;
        return;
    }
    // operands: sums that hover around zero -- terms of mixed sign over ten binades, so that every step rounds
    const uint64_t seed = mix(0xC0FFEEull + (uint64_t)w * 64u + lane);
    double c0 = rnd_pm(mix(seed + 1), -12, 6), t0 = (double)(float)rnd_pm(mix(seed + 2), -10, 6);
    double c1 = rnd_pm(mix(seed + 3), -12, 6), t1 = (double)(float)rnd_pm(mix(seed + 4), -10, 6);
    double s = rnd_pm(mix(0xABCDull + w), -9, 4);
    double x0 = 0, x1 = 0;
    const unsigned long long t_0 = __builtin_amdgcn_s_memtime();
    const int creps = filler_every ? reps * 4 : reps;  // (next to fillers: for about as long as they run)
    for (int r = 0; r < creps; r++) {
        if (MODE == 6)
            break;
        if (MODE == 0)
            chain_dpp(s, c0, t0, c1, t1, x0, x1);
        else if (MODE == 1)
            chain_mfma<false>(s, c0, t0, c1, t1, lds, lane, x0, x1);
        else if (MODE == 2)
            chain_mfma<true>(s, c0, t0, c1, t1, lds, lane, x0, x1);
        else if (MODE == 4)
            chain_serial(s, c0, t0, c1, t1, lds, lane, x0, x1);
        else {  // MODE 5: 256 dependent additions and nothing else (the latency floor of any form of the chain)
            double a = s;
#pragma unroll 16
            for (int j = 0; j < 128; j++) a = (a - c0) + t0;
            x0 = a;
            x1 = a + c1;
        }
        // next rep: carry = the block's last sum (uniform), operands nudged so that nothing is loop-invariant
        s = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x1), 63), __builtin_amdgcn_readlane(__double2loint(x1), 63));
        c0 = -c0;
        t1 = -t1;
    }
    const unsigned long long t_1 = __builtin_amdgcn_s_memtime();
    out[(size_t)w * 128 + 2 * lane] = x0;
    out[(size_t)w * 128 + 2 * lane + 1] = x1;
    if (lane == 0)
        cyc[w] = t_1 - t_0;
}

template <int MODE>
static void run(int nw, int reps, int filler_every, std::vector<double> &out, double &cyc_chain, double &cyc_fill, float &ms)
{
    double *d_out;
    unsigned long long *d_cyc;
    CK(hipMalloc(&d_out, sizeof(double) * 128 * (size_t)nw));
    CK(hipMalloc(&d_cyc, 8 * (size_t)nw));
    CK(hipMemset(d_out, 0, sizeof(double) * 128 * (size_t)nw));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_chain<MODE>, dim3(nw), dim3(64), 0, 0, 4, filler_every, d_out, d_cyc);  // warm
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_chain<MODE>, dim3(nw), dim3(64), 0, 0, reps, filler_every, d_out, d_cyc);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    out.resize(128 * (size_t)nw);
    std::vector<unsigned long long> cyc(nw);
    CK(hipMemcpy(out.data(), d_out, sizeof(double) * out.size(), hipMemcpyDeviceToHost));
    CK(hipMemcpy(cyc.data(), d_cyc, 8 * (size_t)nw, hipMemcpyDeviceToHost));
    double a = 0, b = 0;
    int na = 0, nb = 0;
    for (int w = 0; w < nw; w++) {
        if (filler_every && w >= nw / filler_every)
            b += (double)cyc[w], nb++;
        else
            a += (double)cyc[w], na++;
    }
    cyc_chain = na ? a / na / reps : 0;
    cyc_fill = nb ? b / nb / reps : 0;
    CK(hipFree(d_out));
    CK(hipFree(d_cyc));
}

int main()
{
    const int reps = 64;
    for (int cfg = 0; cfg < 4; cfg++) {
        // waves: 256 = one per CU; 1024 = one per SIMD; 4096 = four per SIMD (the headline's residency);
        // last: 4096 with three of every four waves VALU-bound fillers
        const int nw = cfg == 0 ? 256 : cfg == 1 ? 1024 : 4096;
        const int fe = cfg == 3 ? 4 : 0;
        std::vector<double> o0, o1, o2, o4, o5;
        double c0, c1, c2, f0, f1, f2, c4, f4, c5, f5;
        float m0, m1, m2, m4, m5;
        run<0>(nw, reps, fe, o0, c0, f0, m0);
        run<1>(nw, reps, fe, o1, c1, f1, m1);
        run<2>(nw, reps, fe, o2, c2, f2, m2);
        run<4>(nw, reps, fe, o4, c4, f4, m4);
        run<5>(nw, reps, fe, o5, c5, f5, m5);
        if (fe) {
            std::vector<double> o6;
            double c6, f6;
            float m6;
            run<6>(nw, reps, fe, o6, c6, f6, m6);
            std::printf("fillers alone (the fourth wave of every SIMD absent): %.0f cycles per rep, kernel %.3f ms; next to 256 dependent additions: %.0f; next to the one-lane LDS chain: %.0f\n", f6, m6, f5, f4);
        }
        size_t bad4 = 0;
        for (int w = 0; w < nw; w++) {
            if (fe && w >= nw / fe)
                continue;
            for (int j = 0; j < 128; j++) bad4 += std::memcmp(&o0[(size_t)w * 128 + j], &o4[(size_t)w * 128 + j], 8) != 0;
        }
        std::printf("waves %4d%s: one lane through LDS: %.0f cycles per chain (kernel %.3f ms), %zu sums differ from the DPP chain's; 256 dependent additions alone: %.0f cycles\n",
                    nw, fe ? " (3 of 4 VALU fillers)" : "", c4, m4, bad4, c5);
        size_t bad1 = 0, bad2 = 0, n = 0;
        for (int w = 0; w < nw; w++) {
            if (fe && w >= nw / fe)
                continue;
            for (int j = 0; j < 128; j++, n++) {
                bad1 += std::memcmp(&o0[(size_t)w * 128 + j], &o1[(size_t)w * 128 + j], 8) != 0;
                bad2 += std::memcmp(&o0[(size_t)w * 128 + j], &o2[(size_t)w * 128 + j], 8) != 0;
            }
        }
        std::printf("waves %4d%s: cycles per 64-lane chain: DPP %.0f  MFMA+LDS %.0f  MFMA+LDS four lanes %.0f | kernel ms %.3f %.3f %.3f | "
                    "filler cycles per rep %.0f %.0f %.0f | sums differing from the DPP chain's: %zu / %zu of %zu\n",
                    nw, fe ? " (3 of 4 VALU fillers)" : "", c0, c1, c2, m0, m1, m2, f0, f1, f2, bad1, bad2, n);
    }
    return 0;
}
