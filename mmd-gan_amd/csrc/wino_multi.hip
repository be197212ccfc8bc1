// Winograd weight transforms of MANY kernels in one launch.  A training step transforms every eligible kernel of a
// network once per weight update (forward and input-gradient forms): 14 launches of 2-4 us at the head of a CIFAR step,
// or two of these (G's tensors, which its forward pass waits for, then D's).
#include "conv_internal.h"
#include "wino_weight.h"

namespace mmdgan {

constexpr int kWinoJobsMax = 24;
struct WinoJobTable {
    const float *w[kWinoJobsMax];
    float *u[kWinoJobsMax];
    int C[kWinoJobsMax], K[kWinoJobsMax];
    unsigned char kind[kWinoJobsMax];            // bit 0: input-gradient form, bit 1: 4x4 stride 2 (else 3x3), bit 2: 3x3 as F(4x4,3x3)
    unsigned first_block[kWinoJobsMax + 1];      // prefix sums of the jobs' workgroup counts
    int n;
};

__global__ __launch_bounds__(256) void wino_weight_multi_kernel(WinoJobTable t) {
    __shared__ float tile[9][32][33];
    int j = 0;
    while (j + 1 < t.n && blockIdx.x >= t.first_block[j + 1]) ++j;       // (uniform; n <= 24)
    unsigned b = blockIdx.x - t.first_block[j];
    const int C = t.C[j], K = t.K[j], kb = (K + 31) / 32, cb = (C + 31) / 32;
    const int bx = (int)(b % kb), by = (int)((b / kb) % cb), bz = (int)(b / ((unsigned)kb * cb));
    switch (t.kind[j]) {
    case 0: wino_weight_block<false>(tile, bx, by, t.w[j], t.u[j], C, K); break;
    case 1: wino_weight_block<true>(tile, bx, by, t.w[j], t.u[j], C, K); break;
    case 2: wino2_weight_block<false>(tile, bx, by, bz, t.w[j], t.u[j], C, K); break;
    case 3: wino2_weight_block<true>(tile, bx, by, bz, t.w[j], t.u[j], C, K); break;
    case 4: wino43_weight_block<false>(tile, bx, by, t.w[j], t.u[j], C, K); break;
    default: wino43_weight_block<true>(tile, bx, by, t.w[j], t.u[j], C, K); break;
    }
}

}  // namespace mmdgan

using namespace mmdgan;

extern "C" int mmdgan_wino_transform_multi(const mmdgan_wino_job *jobs, int n_jobs, void *stream) {
    MMDGAN_REQUIRE(jobs || n_jobs == 0, "wino_transform_multi: null job list");
    MMDGAN_REQUIRE(n_jobs >= 0, "wino_transform_multi: negative job count");
    for (int i0 = 0; i0 < n_jobs; i0 += kWinoJobsMax) {
        WinoJobTable t;
        t.n = n_jobs - i0 < kWinoJobsMax ? n_jobs - i0 : kWinoJobsMax;
        unsigned blocks = 0;
        for (int i = 0; i < t.n; ++i) {
            const mmdgan_wino_job &jb = jobs[i0 + i];
            MMDGAN_REQUIRE(jb.w && jb.u && jb.C >= 1 && jb.K >= 1, "wino_transform_multi: job %d: bad arguments", i0 + i);
            MMDGAN_REQUIRE((jb.R == 3 && jb.stride == 1) || (jb.R == 4 && jb.stride == 2),
                           "wino_transform_multi: job %d: 3x3 stride 1 or 4x4 stride 2 kernels only (got %dx%d stride %d)", i0 + i, jb.R,
                           jb.R, jb.stride);
            const bool f43 = jb.algo == MMDGAN_WINO_F43;
            MMDGAN_REQUIRE(jb.algo == MMDGAN_WINO_NONE || (jb.algo == MMDGAN_WINO_F23 && jb.R == 3) || (jb.algo == MMDGAN_WINO_F22S2 && jb.R == 4) ||
                           (f43 && jb.R == 3), "wino_transform_multi: job %d: algorithm %d does not fit a %dx%d kernel", i0 + i, jb.algo, jb.R, jb.R);
            if (f43)
                MMDGAN_REQUIRE((jb.dgrad ? jb.K : jb.C) % 8 == 0 && (jb.dgrad ? jb.C : jb.K) % 32 == 0,
                               "wino_transform_multi: job %d (F(4x4,3x3)): reduction-side channels must be a multiple of 8, output-side of 32", i0 + i);
            if (jb.R == 3)
                MMDGAN_REQUIRE((jb.dgrad ? jb.K : jb.C) % 8 == 0,
                               "wino_transform_multi: job %d (3x3): the reduction-side channel count must be a multiple of 8", i0 + i);
            else
                MMDGAN_REQUIRE(jb.C % 32 == 0 && jb.K % 32 == 0, "wino_transform_multi: job %d (4x4 stride 2): C and K must be multiples of 32",
                               i0 + i);
            t.w[i] = jb.w; t.u[i] = jb.u; t.C[i] = jb.C; t.K[i] = jb.K;
            t.kind[i] = (unsigned char)((jb.dgrad ? 1 : 0) | (jb.R == 4 ? 2 : 0) | (f43 ? 4 : 0));
            t.first_block[i] = blocks;
            blocks += (unsigned)((jb.K + 31) / 32) * (unsigned)((jb.C + 31) / 32) * (jb.R == 4 ? 4u : 1u);
        }
        t.first_block[t.n] = blocks;
        hipLaunchKernelGGL(wino_weight_multi_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, t);
        if (int rc = check_launch("wino_transform_multi")) return rc;
    }
    return MMDGAN_OK;
}
