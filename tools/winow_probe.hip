// ablation probe for the Winograd weight-gradient kernel (D l3 geometry by default)
#include "../mmd-gan_amd/csrc/conv_wino.hip"
#include <vector>
namespace mmdgan { void set_error(const char *, ...) {} bool outputs_prezeroed() { return true; }
static void *g_ws = nullptr; static size_t g_wsb = 0; void *workspace(size_t b) { return b <= g_wsb ? g_ws : nullptr; } }
int main(int argc, char **argv) {
    using namespace mmdgan;
    const int N = argc > 1 ? atoi(argv[1]) : 128, H = argc > 2 ? atoi(argv[2]) : 16, C = argc > 3 ? atoi(argv[3]) : 128, K = argc > 4 ? atoi(argv[4]) : 128;
    mmdgan_conv_geom g{N, H, H, C, K, 3, 1};
    const ConvDims d = conv_dims(g);
    size_t nx = (size_t)N * H * H * C, ny = (size_t)N * H * H * K;
    float *x, *dy, *dw;
    (void)hipMalloc(&x, nx * 4); (void)hipMalloc(&dy, ny * 4); (void)hipMalloc(&dw, (size_t)9 * C * K * 4);
    std::vector<float> h(nx > ny ? nx : ny);
    unsigned sd = 1;
    for (auto &v : h) { sd = sd * 1664525u + 1013904223u; v = ((float)(sd >> 8) / 8388608.f - 1.f) * 0.5f; }
    (void)hipMemcpy(x, h.data(), nx * 4, hipMemcpyHostToDevice); (void)hipMemcpy(dy, h.data(), ny * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) wino_wgrad(d, x, dy, dw, 0);
    (void)hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) wino_wgrad(d, x, dy, dw, 0);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%s N=%d H=%d C=%d K=%d: kernel %.1f us (%.1f TF effective)\n", VARIANT, N, H, C, K, ms / 20 * 1e3, 2.0 * N * H * H * K * 9.0 * C / (ms / 20) / 1e9);
    return 0;
}
