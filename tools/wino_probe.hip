// ablation probe for the Winograd kernel: times wino_kernel<64> alone (weights pre-transformed) on a D-layer shape
#include "../mmd-gan_amd/csrc/conv_wino.hip"
#include <vector>
namespace mmdgan { void set_error(const char *, ...) {} bool outputs_prezeroed() { return false; }
static void *g_ws = nullptr; static size_t g_wsb = 0; void *workspace(size_t b) { return b <= g_wsb ? g_ws : nullptr; } }
int main(int argc, char **argv) {
    using namespace mmdgan;
    const int N = argc > 1 ? atoi(argv[1]) : 128, H = argc > 2 ? atoi(argv[2]) : 16, C = argc > 3 ? atoi(argv[3]) : 128, K = argc > 4 ? atoi(argv[4]) : 128;
    size_t nx = (size_t)N * H * H * C, nu = (size_t)16 * C * K, ny = (size_t)N * H * H * K;
    float *x, *U, *y, *w;
    (void)hipMalloc(&x, nx * 4); (void)hipMalloc(&U, nu * 4); (void)hipMalloc(&y, ny * 4); (void)hipMalloc(&w, (size_t)9 * C * K * 4);
    std::vector<float> h(nx > nu ? nx : nu);
    unsigned sd = 1;
    for (auto &v : h) { sd = sd * 1664525u + 1013904223u; v = ((float)(sd >> 8) / 8388608.f - 1.f) * 0.5f; }
    (void)hipMemcpy(x, h.data(), nx * 4, hipMemcpyHostToDevice); (void)hipMemcpy(U, h.data(), nu * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(w, h.data(), (size_t)9 * C * K * 4, hipMemcpyHostToDevice);
    ConvEpilogue ep{nullptr, nullptr, nullptr, 0, kNoWrap, 0, false};
    const long T = (long)N * (H / 2) * (H / 2);
#ifndef PBN
#define PBN 64
#endif
    const dim3 grid((unsigned)((T + 31) / 32), K / PBN);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((wino_kernel<PBN, false>), grid, dim3(256), (wino::Cfg<PBN>::LDS_BYTES), 0, N, H, H, C, K, ep, x, U, y, C / mmdgan::wino::BC);
    (void)hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((wino_kernel<PBN, false>), grid, dim3(256), (wino::Cfg<PBN>::LDS_BYTES), 0, N, H, H, C, K, ep, x, U, y, C / mmdgan::wino::BC);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const dim3 wg((K + 31) / 32, (C + 31) / 32);
    (void)hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(wino_weight_kernel<false>, wg, dim3(256), 0, 0, w, U, C, K);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms2; (void)hipEventElapsedTime(&ms2, e0, e1);
    printf("%s N=%d H=%d C=%d K=%d: %d wgs, %d stages: kernel %.1f us (%.1f TF effective), weight transform %.1f us\n", VARIANT, N, H, C, K,
           grid.x * grid.y, C / mmdgan::wino::BC, ms / 20 * 1e3, 2.0 * N * H * H * K * 9.0 * C / (ms / 20) / 1e9, ms2 / 20 * 1e3);
    return 0;
}
