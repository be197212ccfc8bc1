// ablation probe: times igemm_fwd on the D l3 shape with parts of the main loop compiled out
#include "../mmd-gan_amd/csrc/conv_igemm.hip"
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
    ConvEpilogue ep{nullptr, nullptr, nullptr, 0, kNoWrap, 0, false};
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) igemm_fwd(d, ep, x, w, y, 0);
    (void)hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) igemm_fwd(d, ep, x, w, y, 0);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%s R=%d stages=%d: %.1f us\n", VARIANT, R, R * R * 128 / 32, ms / 20 * 1e3);
    return 0;
}
