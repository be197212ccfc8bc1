// What each part of wino2_kernel costs: the shipped kernel source compiled with -DW2_ABLATE=<mask> (csrc/conv_wino2.hip header:
// 1 patch loads, 2 B fragments, 4 A fragments, 8 input transform + V stores, 16 stage barrier, 32 epilogue operand loads,
// 64 epilogue stores, 128 the whole epilogue), timed alone on one geometry.  Timing only: an ablated variant computes garbage.
//   args: grid [N H C K dgrad]    default: 512 workgroups, the 3B-row input-gradient of D l2 (192 x 32 x 32 x 64 <- 128)
//   grid 256 = one workgroup per CU (what a workgroup does when it has the CU to itself)
// tools/wino2_ablate.sh builds the variants and runs them (on the GPU box, through gpurun).
#include "../mmd-gan_amd/csrc/conv_wino2.hip"
#include <vector>
namespace mmdgan { void set_error(const char *, ...) {} bool outputs_prezeroed() { return false; }
void *workspace(size_t) { return nullptr; } void *workspace_acquire(size_t, hipStream_t) { return nullptr; }
bool plan_recording() { return false; } void plan_push(std::function<void()> &&) {} void plan_note_collective() {}
void addend_applied() {}
void *wgrad_slabs_acquire(size_t, hipStream_t, SlabReduceArgs *) { return nullptr; }
int wgrad_slabs_release(const SlabReduceArgs &, hipStream_t) { return 0; }
bool wgrad_deferred() { return false; }
int wgrad_flush_pending() { return 0; }
double *bn_stats_request() { return nullptr; }
void bn_stats_applied() {}
int bn_slot_count(int) { return 1; }
hipError_t memset_async(void *p, int v, size_t b, hipStream_t s) { return hipMemsetAsync(p, v, b, s); } }
extern "C" int mmdgan_colsum(const float *, long, int, float *, void *) { return 0; }
int main(int argc, char **argv) {
    using namespace mmdgan;
    const int gridreq = argc > 1 ? atoi(argv[1]) : 512;
    const int N = argc > 2 ? atoi(argv[2]) : 192, H = argc > 3 ? atoi(argv[3]) : 32, C = argc > 4 ? atoi(argv[4]) : 64, K = argc > 5 ? atoi(argv[5]) : 128;
    const int dgrad = argc > 6 ? atoi(argv[6]) : 1;
    mmdgan_conv_geom g{N, H, H, C, K, 4, 2};
    const ConvDims d = conv_dims(g);
    const size_t nx = (size_t)N * H * H * C, nu = (size_t)36 * C * K, ny = (size_t)N * d.P * d.Q * K;
    float *x, *U, *y, *dact;
    (void)hipMalloc(&x, nx * 4); (void)hipMalloc(&U, nu * 4); (void)hipMalloc(&y, ny * 4); (void)hipMalloc(&dact, nx * 4);
    std::vector<float> h(nx > ny ? nx : ny);
    unsigned sd = 1;
    for (auto &v : h) { sd = sd * 1664525u + 1013904223u; v = ((float)(sd >> 8) / 8388608.f - 1.f) * 0.5f; }
    (void)hipMemcpy(x, h.data(), nx * 4, hipMemcpyHostToDevice); (void)hipMemcpy(y, h.data(), ny * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(U, h.data(), (nu < h.size() ? nu : h.size()) * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(dact, h.data(), nx * 4, hipMemcpyHostToDevice);
    ConvEpilogue ep{};
    ep.dact = dgrad ? dact : nullptr; ep.act = 2;                 // lrelu derivative from the layer's output, the 3B-row wrap
    ep.wrap_from = dgrad ? (long)(N / 3 * 2) * H * H * C : kNoWrap; ep.wrap_sub = dgrad ? (long)(N / 3) * H * H * C : 0;
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
    P.ksplit = 1; P.spp = P.nseg * (P.Cr / wino2::BC); P.contiguous = 0; P.slab_bytes = 0;
    const long nitems = (long)P.ntb * P.nkb * P.nph;
    const int grid = (int)(nitems < gridreq ? nitems : gridreq);
    (void)hipFuncSetAttribute((const void *)wino2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)wino2::LDS_BYTES);
    const float *in = dgrad ? y : x; float *out = dgrad ? x : y;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 200; ++i) wino2_kernel<<<grid, 256, wino2::LDS_BYTES>>>(P, ep, in, U, out);
    float best = 1e9f, tot = 0;
    for (int r = 0; r < 5; ++r) {
        (void)hipEventRecord(e0);
        for (int i = 0; i < 50; ++i) wino2_kernel<<<grid, 256, wino2::LDS_BYTES>>>(P, ep, in, U, out);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        tot += ms; if (ms < best) best = ms;
    }
    std::vector<float> o((size_t)(dgrad ? nx : ny));
    (void)hipMemcpy(o.data(), out, o.size() * 4, hipMemcpyDeviceToHost);
    double cs = 0, ca = 0;
    for (size_t i = 0; i < o.size(); ++i) { cs += o[i] * (double)(1 + (i % 7)); ca += fabs((double)o[i]); }
    printf("  output checksum %.9e  sum|.| %.9e\n", cs, ca);
    const double gflop = 2.0 * N * H * H * C * 16.0 * K / (dgrad ? 1 : 4) / 1e9 * (dgrad ? 1 : 4);
    printf("ablate %3d  grid %4d | N=%d H=%d C=%d K=%d %s: %ld items  %.2f us (best of 5 x 50: %.2f)  [%s]\n", W2_ABLATE, grid,
           N, H, C, K, dgrad ? "dgrad" : "fwd", nitems, tot / 250 * 1e3, best / 50 * 1e3, hipGetErrorString(hipGetLastError()));
    (void)gflop;
    return 0;
}
