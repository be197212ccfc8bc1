// the F(2x2,2x2) kernel alone with in-kernel wall-clock stamps (-DW2_DEBUG_TIMELINE, a temporary hook that is NOT in the
// shipped csrc): per workgroup and item, when the prologue, main loop, epilogue started / ended; printed for the two
// workgroups of a few CUs.  D l2 input-gradient geometry by default.
#define W2_DEBUG_TIMELINE 1
#include "../mmd-gan_amd/csrc/conv_wino2.hip"
#include <vector>
#include <map>
namespace mmdgan { void set_error(const char *, ...) {} bool outputs_prezeroed() { return false; }
static void *g_ws = nullptr; static size_t g_wsb = 0; void *workspace(size_t b) { return b <= g_wsb ? g_ws : nullptr; }
void *workspace_acquire(size_t b, hipStream_t) { return workspace(b); }
bool plan_recording() { return false; } void plan_push(std::function<void()> &&) {} void plan_note_collective() {}
hipError_t memset_async(void *p, int v, size_t b, hipStream_t s) { return hipMemsetAsync(p, v, b, s); } }
extern "C" int mmdgan_colsum(const float *, long, int, float *, void *) { return 0; }
int main(int argc, char **argv) {
    using namespace mmdgan;
    const int N = argc > 1 ? atoi(argv[1]) : 192, H = argc > 2 ? atoi(argv[2]) : 32, C = argc > 3 ? atoi(argv[3]) : 64, K = argc > 4 ? atoi(argv[4]) : 128;
    const int dgrad = argc > 5 ? atoi(argv[5]) : 1, stagger = argc > 6 ? atoi(argv[6]) : 0;
    mmdgan_conv_geom g{N, H, H, C, K, 4, 2};
    const ConvDims d = conv_dims(g);
    size_t nx = (size_t)N * H * H * C, nu = (size_t)36 * C * K, ny = (size_t)N * d.P * d.Q * K;
    float *x, *U, *y;
    unsigned long long *dbg;
    (void)hipMalloc(&x, nx * 4); (void)hipMalloc(&U, nu * 4); (void)hipMalloc(&y, ny * 4); (void)hipMalloc(&dbg, 512 * 64 * 8);
    std::vector<float> h(nx > ny ? nx : ny);
    unsigned sd = 1;
    for (auto &v : h) { sd = sd * 1664525u + 1013904223u; v = ((float)(sd >> 8) / 8388608.f - 1.f) * 0.5f; }
    (void)hipMemcpy(x, h.data(), nx * 4, hipMemcpyHostToDevice); (void)hipMemcpy(y, h.data(), ny * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(U, h.data(), (nu < h.size() ? nu : h.size()) * 4, hipMemcpyHostToDevice);
    float *dact; (void)hipMalloc(&dact, (size_t)N * H * H * C * 4); (void)hipMemcpy(dact, h.data(), (size_t)N * H * H * C * 4 < h.size() * 4 ? (size_t)N * H * H * C * 4 : h.size() * 4, hipMemcpyHostToDevice);
    ConvEpilogue ep{nullptr, nullptr, dgrad ? dact : nullptr, 2, dgrad ? (long)(N / 3 * 2) * H * H * C : kNoWrap, dgrad ? (long)(N / 3) * H * H * C : 0, false};
    wino2::Params P;
    P.N = d.N; P.TH = d.P / 2; P.TW = d.Q / 2;
    if (!dgrad) {
        P.IH = d.H; P.IW = d.W; P.Cr = d.C; P.nseg = 4;
        for (int s = 0; s < 4; ++s) { P.r0[s] = -1 + (s >> 1); P.c0[s] = -1 + (s & 1); P.o0r[s] = 0; P.o0c[s] = 0; }
        P.tstep = 4; P.pstep = 2; P.OH = d.P; P.OW = d.Q; P.Ko = d.K; P.otile = 2; P.ostep = 1;
    } else {
        P.IH = d.P; P.IW = d.Q; P.Cr = d.K; P.nseg = 1;
        for (int s = 0; s < 4; ++s) { P.r0[s] = (s >> 1) - 1; P.c0[s] = (s & 1) - 1; P.o0r[s] = s >> 1; P.o0c[s] = s & 1; }
        P.tstep = 2; P.pstep = 1; P.OH = d.H; P.OW = d.W; P.Ko = d.C; P.otile = 4; P.ostep = 2;
    }
    const long T = (long)d.N * P.TH * P.TW;
    P.ntb = (int)((T + 31) / 32); P.nkb = P.Ko / 64; P.nph = dgrad ? 4 : 1;
    const long nitems = (long)P.ntb * P.nkb * P.nph;
    P.stagger = stagger; P.dbg = dbg;
    const int grid = (int)(nitems < 512 ? nitems : 512);
    (void)hipFuncSetAttribute((const void *)wino2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)wino2::LDS_BYTES);
    const float *in = dgrad ? y : x; float *out = dgrad ? x : y;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 300; ++i) wino2_kernel<<<grid, 256, wino2::LDS_BYTES>>>(P, ep, in, U, out);
    (void)hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) wino2_kernel<<<grid, 256, wino2::LDS_BYTES>>>(P, ep, in, U, out);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("N=%d H=%d C=%d K=%d %s stagger %d: %ld items on %d workgroups, %.1f us per launch (with stamps)\n", N, H, C, K, dgrad ? "dgrad" : "fwd", stagger, nitems, grid, ms / 20 * 1e3);
    std::vector<unsigned long long> t(512 * 64);
    (void)hipMemcpy(t.data(), dbg, t.size() * 8, hipMemcpyDeviceToHost);
    unsigned long long t0 = ~0ull;
    for (int b = 0; b < grid; ++b) if (t[(b * 8) * 8] < t0) t0 = t[(b * 8) * 8];
    std::map<unsigned, std::vector<int>> cu;
    for (int b = 0; b < grid; ++b) { const unsigned hw = (unsigned)t[(b * 8 + 7) * 8], xcc = (unsigned)t[(b * 8 + 7) * 8 + 1] & 15; cu[(xcc << 16) | (((hw >> 13) & 7) << 8) | ((hw >> 8) & 15)].push_back(b); }
    int shown = 0;
    double sum[4] = {0, 0, 0, 0}; long cnt = 0;
    for (auto &kv : cu) {
        for (int b : kv.second) {
            const int count = (int)t[(b * 8 + 7) * 8 + 2];
            for (int it = 0; it < count && it < 7; ++it) {
                const unsigned long long *q = &t[(b * 8 + it) * 8];
                sum[0] += (q[1] - q[0]) * 0.01; sum[1] += (q[2] - q[1]) * 0.01; sum[2] += (q[3] - q[2]) * 0.01; ++cnt;
            }
        }
        if (shown++ >= 4) continue;
        printf("CU %06x\n", kv.first);
        for (int b : kv.second) {
            const int count = (int)t[(b * 8 + 7) * 8 + 2];
            printf("  wg %3d slot %u:", b, (unsigned)t[(b * 8 + 7) * 8] & 15);
            for (int it = 0; it < count && it < 7; ++it) {
                const unsigned long long *q = &t[(b * 8 + it) * 8];
                printf("  [P %.1f L %.1f-%.1f E -%.1f]", (q[0] - t0) * 0.01, (q[1] - t0) * 0.01, (q[2] - t0) * 0.01, (q[3] - t0) * 0.01);
            }
            printf("\n");
        }
    }
    { double e[5] = {0,0,0,0,0}; long n = 0;
      for (int b = 0; b < grid; ++b) { const int count = (int)t[(b * 8 + 7) * 8 + 2]; for (int it = 0; it < count && it < 7; ++it) { const unsigned long long *q = &t[(b * 8 + it) * 8];
        e[0] += (q[4] - q[2]) * 0.01; e[1] += (q[5] - q[4]) * 0.01; e[2] += (q[6] - q[5]) * 0.01; e[3] += (q[7] - q[6]) * 0.01; e[4] += (q[3] - q[7]) * 0.01; ++n; } }
      printf("epilogue (us): dact issue %.2f | block 0: write+barrier %.2f, transform+stores %.2f | block 1: barrier+write+barrier %.2f, transform+stores+barrier %.2f\n", e[0]/n, e[1]/n, e[2]/n, e[3]/n, e[4]/n); }
    printf("mean per item (us): prologue %.2f  main loop %.2f  epilogue %.2f   (%ld items)\n", sum[0] / cnt, sum[1] / cnt, sum[2] / cnt, cnt);
    return 0;
}
