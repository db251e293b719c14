// host_path_bench.cpp -- PCIe-inclusive rate of psk_soft_process_host through the C ABI, with
// packets and result buffers in ordinary (pageable) host memory as a BULKIO host has them.
// Never the bench.py `value`; reported in DESIGN.md section 5.
//   hipcc -O2 -std=c++17 -I include tools/host_path_bench.cpp -L psk_soft_amd -lpsk_soft_hip \
//       -Wl,-rpath,$PWD/psk_soft_amd -o /tmp/host_path_bench && /tmp/host_path_bench 4096 32768 8 [pinned]
// "pinned": packets and results live in hipHostMalloc memory and go to psk_soft_process_device as they
// are -- the kernels read and write host memory over PCIe (zero copy).
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "psk_soft_hip.h"

int main(int argc, char **argv)
{
    const uint32_t C = argc > 1 ? atoi(argv[1]) : 4096;
    const uint32_t N = argc > 2 ? atoi(argv[2]) : 32768;
    const int calls = argc > 3 ? atoi(argv[3]) : 8;
    const bool pinned = argc > 4 && argv[4][0] == 'p';
    psk_soft_limits_t lim = {16384, 512, N};
    psk_soft_handle_t *h = nullptr;
    if (psk_soft_create(0, C, &lim, &h) != PSK_SOFT_OK) {
        printf("create failed: %s\n", psk_soft_last_error());
        return 1;
    }
    std::vector<psk_soft_props_t> props(C);
    for (auto &p : props) p = psk_soft_props_t{8, 4, 100, 50, 0, 0};
    psk_soft_configure(h, 0, C, props.data());
    // QPSK-ish stimulus: shaped pulses, unit circle, a little noise (values do not matter for the rate)
    std::vector<std::vector<float>> iq(C, std::vector<float>(2 * (size_t)N));
    uint64_t st = 88172645463325252ull;
    for (uint32_t c = 0; c < C; c++)
        for (uint32_t i = 0; i < N; i++) {
            st ^= st << 13; st ^= st >> 7; st ^= st << 17;
            const int sym = (int)((i / 8 * 2654435761u + c) >> 7) & 3;
            const double a = 0.2 + 0.8 * std::sin(3.14159265 * ((i % 8) + 0.9) / 9.3), ph = 1.5707963 * sym + 0.3;
            iq[c][2 * i] = (float)(a * std::cos(ph) + 1e-2 * ((double)(st & 0xffff) / 65536.0 - 0.5));
            iq[c][2 * i + 1] = (float)(a * std::sin(ph) + 1e-2 * ((double)((st >> 16) & 0xffff) / 65536.0 - 0.5));
        }
    const uint64_t cap = (psk_soft_output_capacity(h, 0, N) + 63) / 64 * 64;
    std::vector<std::vector<float>> soft(C, std::vector<float>(2 * cap, 1.f)), phase(C, std::vector<float>(cap, 1.f));
    std::vector<std::vector<int16_t>> bits(C, std::vector<int16_t>(2 * cap, 1)), sidx(C, std::vector<int16_t>(cap, 1));
    float *p_in = nullptr, *p_soft = nullptr, *p_phase = nullptr;
    int16_t *p_bits = nullptr, *p_sidx = nullptr;
    if (pinned) {
        if (hipHostMalloc((void **)&p_in, sizeof(float) * 2 * (size_t)N * C) != hipSuccess ||
            hipHostMalloc((void **)&p_soft, sizeof(float) * 2 * cap * C) != hipSuccess ||
            hipHostMalloc((void **)&p_phase, sizeof(float) * cap * C) != hipSuccess ||
            hipHostMalloc((void **)&p_bits, sizeof(int16_t) * 2 * cap * C) != hipSuccess ||
            hipHostMalloc((void **)&p_sidx, sizeof(int16_t) * cap * C) != hipSuccess) {
            printf("hipHostMalloc failed\n");
            return 1;
        }
        for (uint32_t c = 0; c < C; c++) memcpy(p_in + 2 * (size_t)N * c, iq[c].data(), sizeof(float) * 2 * N);
    }
    std::vector<psk_soft_packet_t> pk(C);
    std::vector<psk_soft_output_t> out(C);
    std::vector<double> ms;
    for (int k = 0; k < calls; k++) {
        for (uint32_t c = 0; c < C; c++) {
            pk[c] = psk_soft_packet_t{pinned ? p_in + 2 * (size_t)N * c : iq[c].data(), 2ull * N, 0.01, 1, 0, 0, 1, 0};
            out[c] = psk_soft_output_t{};
            out[c].soft = pinned ? p_soft + 2 * cap * c : soft[c].data();
            out[c].bits = pinned ? p_bits + 2 * cap * c : bits[c].data();
            out[c].phase = pinned ? p_phase + cap * c : phase[c].data();
            out[c].sampleIndex = pinned ? p_sidx + cap * c : sidx[c].data();
            out[c].cap_symbols = cap;
        }
        auto t0 = std::chrono::steady_clock::now();
        psk_soft_status st = pinned ? psk_soft_process_device(h, 0, C, pk.data(), out.data(), nullptr)
                                    : psk_soft_process_host(h, 0, C, pk.data(), out.data());
        if (st == PSK_SOFT_OK && pinned)
            st = psk_soft_synchronize(h);
        if (st != PSK_SOFT_OK) {
            printf("process failed: %s\n", psk_soft_last_error());
            return 1;
        }
        ms.push_back(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    }
    double best = 1e30, sum = 0;
    for (int k = 2; k < calls; k++) { best = ms[k] < best ? ms[k] : best; sum += ms[k]; }
    const double avg = sum / (calls - 2);
    const double in_gb = (double)C * N * 8 / 1e9, out_gb = (double)C * out[0].n_symbols * 18 / 1e9;
    printf("%s channels=%u samples/packet=%u symbols/packet=%llu threads=%s stage_mb=%s\n",
           pinned ? "[zero copy from pinned memory]" : "[pageable memory, staged]", C, N,
           (unsigned long long)out[0].n_symbols, getenv("PSK_SOFT_HOST_THREADS") ? getenv("PSK_SOFT_HOST_THREADS") : "8",
           getenv("PSK_SOFT_STAGE_MB") ? getenv("PSK_SOFT_STAGE_MB") : "32");
    printf("  per call: avg %.1f ms, best %.1f ms -> %.2f Gsamples/s, %.1f GB/s in + %.1f GB/s out\n", avg, best,
           (double)C * N / avg / 1e6, in_gb / avg * 1e3, out_gb / avg * 1e3);
    psk_soft_destroy(h);
    return 0;
}
