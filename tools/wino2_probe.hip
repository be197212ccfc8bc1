// ablation probe for the F(2x2,2x2) kernel: D l2 geometry by default (conv 32x32x64 -> 16x16x128, batch 128)
#include "../mmd-gan_amd/csrc/conv_wino2.hip"
#include <vector>
namespace mmdgan { void set_error(const char *, ...) {} bool outputs_prezeroed() { return false; }
static void *g_ws = nullptr; static size_t g_wsb = 0; void *workspace(size_t b) { return b <= g_wsb ? g_ws : nullptr; }
bool plan_recording() { return false; } void plan_push(std::function<void()> &&) {}
hipError_t memset_async(void *p, int v, size_t b, hipStream_t s) { return hipMemsetAsync(p, v, b, s); } }
int main(int argc, char **argv) {
    using namespace mmdgan;
    const int N = argc > 1 ? atoi(argv[1]) : 128, H = argc > 2 ? atoi(argv[2]) : 32, C = argc > 3 ? atoi(argv[3]) : 64, K = argc > 4 ? atoi(argv[4]) : 128;
    const int dgrad = argc > 5 ? atoi(argv[5]) : 0;
    mmdgan_conv_geom g{N, H, H, C, K, 4, 2};
    const ConvDims d = conv_dims(g);
    size_t nx = (size_t)N * H * H * C, nu = (size_t)36 * C * K, ny = (size_t)N * d.P * d.Q * K;
    float *x, *U, *y;
    (void)hipMalloc(&x, nx * 4); (void)hipMalloc(&U, nu * 4); (void)hipMalloc(&y, ny * 4);
    std::vector<float> h(nx > ny ? nx : ny);
    unsigned sd = 1;
    for (auto &v : h) { sd = sd * 1664525u + 1013904223u; v = ((float)(sd >> 8) / 8388608.f - 1.f) * 0.5f; }
    (void)hipMemcpy(x, h.data(), nx * 4, hipMemcpyHostToDevice); (void)hipMemcpy(y, h.data(), ny * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(U, h.data(), (nu < h.size() ? nu : h.size()) * 4, hipMemcpyHostToDevice);
    ConvEpilogue ep{nullptr, nullptr, nullptr, 0, kNoWrap, 0, false};
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    auto run = [&]() { if (dgrad) wino2_dgrad(d, ep, y, nullptr, U, x, 0); else wino2_fwd(d, ep, x, nullptr, U, y, 0); };
    for (int i = 0; i < 3; ++i) run();
    (void)hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) run();
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%s %s N=%d H=%d C=%d K=%d: kernel %.1f us (%.1f TF effective)\n", VARIANT, dgrad ? "dgrad" : "fwd", N, H, C, K, ms / 20 * 1e3,
           2.0 * N * d.P * d.Q * K * 16.0 * C / (ms / 20) / 1e9);
    return 0;
}
