// What each part of wino2_wgrad_kernel costs: the shipped kernel source compiled with -DWG_ABLATE=<mask> (csrc/conv_wino2.hip:
// 1 patch loads, 2 input transform + V stores, 4 dY loads + stores, 8 operand reads from LDS, 16 the stage barrier, 32 the slab
// stores), timed alone.  Timing only: an ablated variant computes garbage.
//   args: [N H C K]   default: D l2's weight gradient at batch 128 (128 x 32 x 32 x 64 -> 128); MMDGAN_WGRAD_CUS sizes the grid (224)
#include "../mmd-gan_amd/csrc/conv_wino2.hip"
#include <vector>
namespace mmdgan { void set_error(const char *, ...) {} bool outputs_prezeroed() { return false; }
static void *g_ws = nullptr; static size_t g_wsb = 0;
void *workspace(size_t b) { return b <= g_wsb ? g_ws : nullptr; } void *workspace_acquire(size_t b, hipStream_t) { return b <= g_wsb ? g_ws : nullptr; }
void *wgrad_slabs_acquire(size_t b, hipStream_t, SlabReduceArgs *prev) { *prev = SlabReduceArgs{}; return b <= g_wsb ? g_ws : nullptr; }
static bool g_reduce = true;
int wgrad_slabs_release(const SlabReduceArgs &a, hipStream_t st) { if (g_reduce) slab_reduce_launch(a, st); return 0; }
bool plan_recording() { return false; } void plan_push(std::function<void()> &&) {} void plan_note_collective() {}
void plan_note_kernel(const void *, dim3, dim3, hipStream_t) {}
void addend_applied() {}
hipError_t memset_async(void *p, int v, size_t b, hipStream_t s) { return hipMemsetAsync(p, v, b, s); } }
extern "C" int mmdgan_colsum(const float *, long, int, float *, void *) { return 0; }
int main(int argc, char **argv) {
    using namespace mmdgan;
    const int N = argc > 1 ? atoi(argv[1]) : 128, H = argc > 2 ? atoi(argv[2]) : 32, C = argc > 3 ? atoi(argv[3]) : 64, K = argc > 4 ? atoi(argv[4]) : 128;
    mmdgan_conv_geom g{N, H, H, C, K, 4, 2};
    const ConvDims d = conv_dims(g);
    const size_t nx = (size_t)N * H * H * C, ny = (size_t)N * d.P * d.Q * K;
    float *x, *dy, *dw, *db;
    g_wsb = 128 << 20;
    (void)hipMalloc(&g_ws, g_wsb);
    (void)hipMalloc(&x, nx * 4); (void)hipMalloc(&dy, ny * 4); (void)hipMalloc(&dw, (size_t)16 * C * K * 4); (void)hipMalloc(&db, K * 4);
    std::vector<float> h(nx > ny ? nx : ny);
    unsigned sd = 1;
    for (auto &v : h) { sd = sd * 1664525u + 1013904223u; v = ((float)(sd >> 8) / 8388608.f - 1.f) * 0.5f; }
    (void)hipMemcpy(x, h.data(), nx * 4, hipMemcpyHostToDevice); (void)hipMemcpy(dy, h.data(), ny * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    bool dbd, dd;
    for (int mode = 0; mode < 2; ++mode) {
        g_reduce = mode == 0;
        for (int i = 0; i < 100; ++i) wino2_wgrad(d, x, dy, dw, db, &dbd, 0, nullptr, nullptr, &dd);
        float best = 1e9f, tot = 0;
        for (int r = 0; r < 5; ++r) {
            (void)hipEventRecord(e0);
            for (int i = 0; i < 50; ++i) wino2_wgrad(d, x, dy, dw, db, &dbd, 0, nullptr, nullptr, &dd);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            tot += ms; if (ms < best) best = ms;
        }
        printf("wgrad ablate %3d | N=%d H=%d C=%d K=%d cus=%d %s: %.2f us (best of 5 x 50: %.2f)  [%s]\n", WG_ABLATE, N, H, C, K, wgrad_cus(),
               mode == 0 ? "kernel + reduction" : "kernel only      ", tot / 250 * 1e3, best / 50 * 1e3, hipGetErrorString(hipGetLastError()));
    }
    return 0;
}
