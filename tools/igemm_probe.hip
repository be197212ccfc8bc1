// ablation probe: times igemm_fwd on the D l3 shape with parts of the main loop compiled out
#ifdef MMDGAN_TIMELINE
#include <hip/hip_runtime.h>
namespace mmdgan { __device__ unsigned long long g_timeline[4 * 8192]; }
#endif
#include "../mmd-gan_amd/csrc/conv_igemm.hip"
#include <algorithm>
#include <vector>
namespace mmdgan { void set_error(const char *, ...) {} bool outputs_prezeroed() { return false; } void *workspace(size_t) { return nullptr; } }
int main(int argc, char **argv) {
    using namespace mmdgan;
    int R = argc > 1 ? atoi(argv[1]) : 3;
    mmdgan_conv_geom g{128, 16, 16, 128, 128, R, 1};
    ConvDims d = conv_dims(g);
    size_t nx = (size_t)d.N * d.H * d.W * d.C, nw = (size_t)d.R * d.R * d.C * d.K, ny = (size_t)d.N * d.P * d.Q * d.K;
    float *x, *w, *y;
    (void)hipMalloc(&x, nx * 4); (void)hipMalloc(&w, nw * 4); (void)hipMalloc(&y, ny * 4);
    (void)hipMemset(x, 0, nx * 4); (void)hipMemset(w, 0, nw * 4);
    if (argc > 2) {      // random operands (data-dependent power: the clock under zeros is higher)
        std::vector<float> hx(nx), hw(nw);
        unsigned sd = 12345;
        for (auto &v : hx) { sd = sd * 1664525u + 1013904223u; v = (float)(sd >> 8) / 8388608.f - 1.f; }
        for (auto &v : hw) { sd = sd * 1664525u + 1013904223u; v = ((float)(sd >> 8) / 8388608.f - 1.f) * 0.05f; }
        (void)hipMemcpy(x, hx.data(), nx * 4, hipMemcpyHostToDevice); (void)hipMemcpy(w, hw.data(), nw * 4, hipMemcpyHostToDevice);
    }
    ConvEpilogue ep{nullptr, nullptr, nullptr, 0, kNoWrap, 0, false};
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) igemm_fwd(d, ep, x, w, y, 0);
    (void)hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) igemm_fwd(d, ep, x, w, y, 0);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%s R=%d stages=%d: %.1f us\n", VARIANT, R, R * R * 128 / 32, ms / 20 * 1e3);
#ifdef MMDGAN_TIMELINE
    {
        (void)hipDeviceSynchronize();
        igemm_fwd(d, ep, x, w, y, 0);
        (void)hipDeviceSynchronize();
        const int nb = (int)((long)d.N * d.P * d.Q / 64) * (d.K / 64);
        std::vector<unsigned long long> t(4 * nb);
        (void)hipMemcpyFromSymbol(t.data(), HIP_SYMBOL(g_timeline), sizeof(unsigned long long) * 4 * nb);
        unsigned long long t0 = ~0ull, t3 = 0;
        for (int b = 0; b < nb; ++b) { t0 = std::min(t0, t[4 * b]); t3 = std::max(t3, t[4 * b + 3]); }
        double s_start = 0, s_init = 0, s_loop = 0, s_epi = 0, last_start = 0, first_end = 1e30;
        for (int b = 0; b < nb; ++b) {
            s_start += (double)(t[4 * b] - t0); s_init += (double)(t[4 * b + 1] - t[4 * b]);
            s_loop += (double)(t[4 * b + 2] - t[4 * b + 1]); s_epi += (double)(t[4 * b + 3] - t[4 * b + 2]);
            last_start = std::max(last_start, (double)(t[4 * b] - t0)); first_end = std::min(first_end, (double)(t[4 * b + 3] - t0));
        }
        if (MMDGAN_TIMELINE & 16) {       // end time by XCD and by CU
            double xs[16] = {0}, xe[16] = {0}; int xn[16] = {0};
            for (int b = 0; b < nb; ++b) {
                const unsigned xcc = (unsigned)(t[4 * b + 1] >> 32) & 15, hw = (unsigned)t[4 * b + 1];
                xs[xcc] += (double)(t[4 * b] - t0); xe[xcc] += (double)(t[4 * b + 3] - t0); xn[xcc]++;
                if (b < 8) printf("   wg %d xcc %u hw_id %08x start %.2f end %.2f\n", b, xcc, hw, (double)(t[4 * b] - t0) * 0.01, (double)(t[4 * b + 3] - t0) * 0.01);
            }
            for (int x = 0; x < 16; ++x) if (xn[x]) printf("   xcc %d: %d wgs, mean start %.2f mean end %.2f\n", x, xn[x], xs[x] / xn[x] * 0.01, xe[x] / xn[x] * 0.01);
            std::vector<double> ends;
            for (int b = 0; b < nb; ++b) ends.push_back((double)(t[4 * b + 3] - t0) * 0.01);
            std::sort(ends.begin(), ends.end());
            printf("   end-time percentiles: 1%% %.1f 10%% %.1f 25%% %.1f 50%% %.1f 75%% %.1f 90%% %.1f 99%% %.1f\n", ends[nb / 100], ends[nb / 10], ends[nb / 4], ends[nb / 2], ends[3 * nb / 4], ends[9 * nb / 10], ends[99 * nb / 100]);
        }
        if (MMDGAN_TIMELINE & 32) {       // shader clock: cycles of clock64() per 100 MHz wall tick between entry and loop end
            double f = 0;
            for (int b = 0; b < nb; ++b) f += (double)(t[4 * b + 2] - t[4 * b + 1]) / (double)(t[4 * b + 3] - t[4 * b]);
            printf("   shader clock during the kernel: %.0f MHz (clock64 ticks / wall ticks, mean over workgroups)\n", f / nb * 100.0);
        }
        const double u = 0.01;   // 100 MHz ticks -> us
        printf("  %d workgroups: first start -> last end %.2f us; start skew avg %.2f max %.2f; init %.2f; loop %.2f; epilogue %.2f; first end at %.2f\n",
               nb, (double)(t3 - t0) * u, s_start / nb * u, last_start * u, s_init / nb * u, s_loop / nb * u, s_epi / nb * u, first_end * u);
    }
#endif
    return 0;
}
