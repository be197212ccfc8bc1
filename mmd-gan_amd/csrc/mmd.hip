// Fused pairwise squared distance + Gaussian kernel + repulsive / bounded MMD loss, forward and
// backward, for gfx950.  One launch; the three B x B matrices live only in registers.
//
// Replaces (reference, /root/reference/GeneralTools/math_func.py):
//   get_squared_dist  :799-840  (Gram form: d_i - 2 <a_i,b_j> + d_j, diag taken from the Gram
//                                matrix, clamped at 0 - reproduced literally so that the rmb clamp
//                                masks are bit-identical)
//   matrix_mean_wo_diagonal :1064, mmd_g :1312-1343, mmd_g_bounded :1380-1422,
//   GANLoss._repulsive_mmd_g_ / _bounded_ :2505-2550, and TF autodiff of all of it;
//   the same tile walk also serves the two non-repulsive Gaussian losses:
//   mixture_mmd_g :1435-1462 over sigma = [1, sqrt2, 2, sqrt8, 4] (GANLoss._mmd_g_ :2160-2173) and
//   mmd_g with bounds (GANLoss._mmd_g_bound_ :2175-2193).  The kernel is instantiated per loss so
//   the rep / rmb code on the training path is unchanged by them.
//   score_loss_kernel (bottom) is GANLoss._logistic_ :2128-2135 and _hinge_ :2137-2143.
//
// Mapping (wave = 64 lanes): one wave owns one row i; lane l owns column j = tile*64 + l.  A block
// (4 waves = 4 rows) stages a 64-row tile of s_gen and s_x in LDS (row stride d+1: conflict-free
// when every lane walks its own row), each wave evaluates the four distances
// (x_i,x_j) (x_i,y_j) (y_i,x_j) (y_i,y_j) with k-ordered fmaf chains, exponentiates, accumulates
// the kernel sums in double and the four gradient rows in 64 float accumulators per lane, then
// reduce-scatters them over the wave with 63 __shfl_xor steps (lane l ends up owning gradient
// element l).  This is O(B^2 d) VALU work on O(B d) bytes: not GEMM-shaped, no MFMA.
// Per-block partial sums go to the workspace; the last block to arrive (agent-scope release /
// acquire, placement independent) reduces them in block order and writes the 8 output scalars.
#include "common.h"
#include "tuning.h"

namespace mmdgan {

constexpr int kMmdRows = 4;     // rows (waves) per block
constexpr int kMmdTile = 64;    // columns per tile = one per lane
constexpr int kMmdMaxD = 256;   // LDS: 2 * 64 * (d+1) * 4 B <= 131 KB
constexpr int kMmdKC = 16;      // gradient k-chunk (64 accumulators = 4 vectors x 16)
constexpr int kNumSums = 6;     // kxx kxy kyy kxx_b kyy_b kxy_b

struct MmdArgs {
    const float *x, *y;   // s_gen, s_x
    int B, d, loss_type, dis_first;
    float w0, w1, lb, ub;
    double *partials;     // [gridDim.x][kNumSums]
    unsigned *counter;
    float *out, *grads, *dist;
    unsigned char *masks;
};

__device__ __forceinline__ float dotk(const float *a, const float *b, int d) {
    float acc = 0.f;
    for (int k = 0; k < d; ++k) acc = fmaf(a[k], b[k], acc);
    return acc;
}

// kernel value and -2 dK/dD of one squared distance: a single Gaussian (sigma 1), or the five-scale
// mixture whose 2 sigma^2 are the exact powers of two 2..32
template <bool MIX>
__device__ __forceinline__ void gauss(float D, float &K, float &G) {
    if (!MIX) {
        K = expf(-D / 2.0f);
        G = K;
    } else {
        K = 0.f; G = 0.f;
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            const float two_s2 = (float)(2 << q);
            const float e = expf(-D / two_s2);
            K += e;
            G += e * (2.0f / two_s2);
        }
    }
}

// DFIX: the score width as a compile-time constant (16 in every shipped architecture: my_test_cifar.py:37) - the dot products
// and the gradient update unroll, their LDS reads batch; 0: any d <= 256 at run time.
template <int LT, int DFIX>
__global__ __launch_bounds__(256) void mmd_kernel(MmdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int d = DFIX ? DFIX : a.d, ld = d + 1;
    float *tx = smem;                       // [64][ld]
    float *ty = tx + kMmdTile * ld;         // [64][ld]
    float *rx = ty + kMmdTile * ld;         // [4][d]   this block's rows of s_gen
    float *ry = rx + kMmdRows * d;          // [4][d]
    // 2*64*(d+1) + 8*d floats is even, so the double area is 8-byte aligned; the ticket lives in
    // the dynamic region too (a static __shared__ would shift the dynamic base off alignment)
    double *bsum = reinterpret_cast<double *>(ry + kMmdRows * d);
    unsigned &s_ticket = *reinterpret_cast<unsigned *>(bsum + kMmdRows * kNumSums);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int B = a.B;
    const int i = blockIdx.x * kMmdRows + wave;
    const bool row_ok = i < B;

    for (int t = tid; t < kMmdRows * d; t += 256) {
        int r = t / d, k = t - r * d, gi = blockIdx.x * kMmdRows + r;
        rx[t] = gi < B ? a.x[(size_t)gi * d + k] : 0.f;
        ry[t] = gi < B ? a.y[(size_t)gi * d + k] : 0.f;
    }
    __syncthreads();
    const float *xi = rx + wave * d, *yi = ry + wave * d;
    const float nxi = dotk(xi, xi, d), nyi = dotk(yi, yi, d);     // diag_part of the Gram matrices

    const float m = (float)B;
    const float inv = 1.0f / (m * (m - 1.0f));
    constexpr bool rmb = LT == MMDGAN_LOSS_RMB, mgb = LT == MMDGAN_LOSS_MGB, mix = LT == MMDGAN_LOSS_MMD_G;
    const bool yy_lower = a.w1 > 0.f;                             // math_func.py:1391-1394
    // coefficients of e(K_xx), e(K_xy), e(K_yy): loss_gen = (1,-2,1); loss_dis = (-1, w0, -w1)
    // (the host passes w = (2, 1) for mmd_g / mgb, whose loss_dis is -mmd)
    const float cg_xx = 1.f, cg_xy = -2.f, cg_yy = 1.f;
    const float cd_xx = -1.f, cd_xy = a.w0, cd_yy = -a.w1;

    double s_xx = 0, s_xy = 0, s_yy = 0, s_xxb = 0, s_yyb = 0, s_xyb = 0;
    const int ntiles = (B + kMmdTile - 1) / kMmdTile;
    const int nchunks = a.grads ? (d + kMmdKC - 1) / kMmdKC : 1;

    for (int kc = 0; kc < nchunks; ++kc) {
        float acc[64];
#pragma unroll
        for (int t = 0; t < 64; ++t) acc[t] = 0.f;
        const int k0 = kc * kMmdKC;

        for (int jt = 0; jt < ntiles; ++jt) {
            __syncthreads();
            for (int t = tid; t < kMmdTile * d; t += 256) {
                int r = t / d, k = t - r * d, gj = jt * kMmdTile + r;
                tx[r * ld + k] = gj < B ? a.x[(size_t)gj * d + k] : 0.f;
                ty[r * ld + k] = gj < B ? a.y[(size_t)gj * d + k] : 0.f;
            }
            __syncthreads();
            const int j = jt * kMmdTile + lane;
            const float *xj = tx + lane * ld, *yj = ty + lane * ld;
            const bool valid = row_ok && j < B;
            const bool off = valid && j != i;

            const float nxj = dotk(xj, xj, d), nyj = dotk(yj, yj, d);
            const float g_xx = dotk(xi, xj, d), g_xy = dotk(xi, yj, d);
            const float g_yx = dotk(xj, yi, d), g_yy = dotk(yi, yj, d);
            // math_func.py:805,833-834: max(d_i - 2*gram + d_j, 0)
            const float r_xx = (nxi - 2.0f * g_xx) + nxj, r_xy = (nxi - 2.0f * g_xy) + nyj;
            const float r_yx = (nxj - 2.0f * g_yx) + nyi, r_yy = (nyi - 2.0f * g_yy) + nyj;
            const float D_xx = fmaxf(r_xx, 0.f), D_xy = fmaxf(r_xy, 0.f);
            const float D_yx = fmaxf(r_yx, 0.f), D_yy = fmaxf(r_yy, 0.f);
            float K_xx, K_xy, K_yx, K_yy, G_xx, G_xy, G_yx, G_yy;
            gauss<mix>(D_xx, K_xx, G_xx); gauss<mix>(D_xy, K_xy, G_xy);
            gauss<mix>(D_yx, K_yx, G_yx); gauss<mix>(D_yy, K_yy, G_yy);

            if (kc == 0) {
                if (off) {
                    s_xx += (double)K_xx; s_xy += (double)K_xy; s_yy += (double)K_yy;
                    if (rmb) {
                        s_xxb += (double)expf(-fmaxf(D_xx, a.lb) / 2.0f);                        // :1386
                        s_yyb += (double)(yy_lower ? expf(-fmaxf(D_yy, a.lb) / 2.0f)
                                                   : expf(-fminf(D_yy, a.ub) / 2.0f));          // :1391-1394
                    }
                    if (mgb) {                                                                   // :1316-1322
                        s_xxb += (double)expf(-fmaxf(D_xx, a.lb) / 2.0f);
                        s_yyb += (double)expf(-fmaxf(D_yy, a.lb) / 2.0f);
                        s_xyb += (double)expf(-fminf(D_xy, a.ub) / 2.0f);
                    }
                }
                if (valid) {
                    const size_t n2 = (size_t)B * B, o = (size_t)i * B + j;
                    if (a.dist) { a.dist[o] = D_xx; a.dist[n2 + o] = D_xy; a.dist[2 * n2 + o] = D_yy; }
                    if (a.masks) {
                        a.masks[o] = D_xx < a.lb; a.masks[n2 + o] = D_xy > a.ub; a.masks[2 * n2 + o] = D_yy > a.ub;
                    }
                }
            }
            if (a.grads) {
                // dK/dD = -K/2; max(.,0) passes the gradient where the raw value is positive;
                // clamp-active entries of the bounded loss pass none
                const float p_xx = (off && r_xx > 0.f) ? 1.f : 0.f, p_xy = (off && r_xy > 0.f) ? 1.f : 0.f;
                const float p_yx = (off && r_yx > 0.f) ? 1.f : 0.f, p_yy = (off && r_yy > 0.f) ? 1.f : 0.f;
                float b_xx = p_xx, b_yy = p_yy, b_xy = p_xy, b_yx = p_yx;
                if (rmb) {
                    b_xx = (D_xx > a.lb) ? p_xx : 0.f;
                    b_yy = (yy_lower ? (D_yy > a.lb) : (D_yy < a.ub)) ? p_yy : 0.f;
                }
                if (mgb) {                       // tf.maximum / tf.minimum pass the gradient to the distance on a tie
                    b_xx = (D_xx >= a.lb) ? p_xx : 0.f; b_yy = (D_yy >= a.lb) ? p_yy : 0.f;
                    b_xy = (D_xy <= a.ub) ? p_xy : 0.f; b_yx = (D_yx <= a.ub) ? p_yx : 0.f;
                }
                const float Ag = -2.f * cg_xx * inv * G_xx * p_xx, Bg = -cg_xy * inv * G_xy * p_xy;
                const float Cg = -2.f * cg_yy * inv * G_yy * p_yy, Eg = -cg_xy * inv * G_yx * p_yx;
                const float Ad = -2.f * cd_xx * inv * G_xx * b_xx, Bd = -cd_xy * inv * G_xy * b_xy;
                const float Cd = -2.f * cd_yy * inv * G_yy * b_yy, Ed = -cd_xy * inv * G_yx * b_yx;
#pragma unroll
                for (int k = 0; k < kMmdKC; ++k) {
                    if (k0 + k < d) {
                        const float xik = xi[k0 + k], yik = yi[k0 + k], xjk = xj[k0 + k], yjk = yj[k0 + k];
                        const float dxx = xik - xjk, dxy = xik - yjk, dyy = yik - yjk, dyx = yik - xjk;
                        acc[k] += Ag * dxx + Bg * dxy;           // dLgen/dx_i
                        acc[16 + k] += Cg * dyy + Eg * dyx;      // dLgen/dy_i
                        acc[32 + k] += Ad * dxx + Bd * dxy;      // dLdis/dx_i
                        acc[48 + k] += Cd * dyy + Ed * dyx;      // dLdis/dy_i
                    }
                }
            }
        }
        if (a.grads) {
            // reduce-scatter over the wave: after the step with distance h, a lane whose bit h is
            // set keeps the upper half of the live accumulators; lane l ends owning element l.
#pragma unroll
            for (int h = 32; h > 0; h >>= 1) {
                const bool up = (lane & h) != 0;
#pragma unroll
                for (int t = 0; t < h; ++t) {
                    const float keep = up ? acc[t + h] : acc[t];
                    const float send = up ? acc[t] : acc[t + h];
                    acc[t] = keep + __shfl_xor(send, h, 64);
                }
            }
            const int vec = lane >> 4, k = k0 + (lane & 15);
            // MMDGAN_LOSS_FLAG_GRADS_DIS_FIRST: slots [dLdis/ds_x, dLdis/ds_gen, dLgen/ds_gen, dLgen/ds_x]
            const int slot = a.dis_first ? (0x0132 >> (4 * vec)) & 15 : vec;
            if (row_ok && k < d) a.grads[((size_t)slot * B + i) * d + k] = acc[0];
        }
    }

    // ---- kernel-sum reduction: wave -> block -> grid (last block finalises) -------------------
    double sums[kNumSums] = {s_xx, s_xy, s_yy, s_xxb, s_yyb, s_xyb};
#pragma unroll
    for (int q = 0; q < kNumSums; ++q) sums[q] = wave_sum(sums[q]);
    __syncthreads();
    if (lane == 0)
        for (int q = 0; q < kNumSums; ++q) bsum[wave * kNumSums + q] = sums[q];
    __syncthreads();
    if (tid == 0) {
        for (int q = 0; q < kNumSums; ++q) {
            double t = 0;
            for (int w = 0; w < kMmdRows; ++w) t += bsum[w * kNumSums + q];
            // 8-byte agent-scope atomics on BOTH sides of the hand-off (write-through stores here, L1-bypassing loads in the last
            // block): the 48 bytes of a block need no L2 write-back and no L1 invalidate - the release / acquire fence pair of
            // rounds 1-4 was ~3.4 us of a 14 us launch at B = 64 (MI355X_MICROARCH.md, "valid forms")
            __hip_atomic_store(a.partials + (size_t)blockIdx.x * kNumSums + q, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        s_ticket = __hip_atomic_fetch_add(a.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (s_ticket != gridDim.x - 1) return;
    if (wave == 0) {
        double tot[kNumSums];
#pragma unroll
        for (int q = 0; q < kNumSums; ++q) {
            double t = 0;
            for (int b = lane; b < (int)gridDim.x; b += 64)
                t += __hip_atomic_load(a.partials + (size_t)b * kNumSums + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            tot[q] = wave_sum(t);
        }
        if (lane == 0) {
            const double denom = (double)B * ((double)B - 1.0);
            const double e_xx = tot[0] / denom, e_xy = tot[1] / denom, e_yy = tot[2] / denom;
            const double e_xxb = (rmb || mgb) ? tot[3] / denom : e_xx, e_yyb = (rmb || mgb) ? tot[4] / denom : e_yy;
            // math_func.py:1341-1342,1421; in rmb e_kxy_b == e_kxy for every admissible weight pair (:1402)
            const double e_xyb = mgb ? tot[5] / denom : e_xy;
            a.out[0] = (float)(e_xx + e_yy - 2.0 * e_xy);
            a.out[1] = (float)((double)a.w0 * e_xyb - e_xxb - (double)a.w1 * e_yyb);
            a.out[2] = (float)e_xx; a.out[3] = (float)e_xy; a.out[4] = (float)e_yy;
            a.out[5] = (float)e_xxb; a.out[6] = (float)e_yyb; a.out[7] = (float)e_xyb;
            __hip_atomic_store(a.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ---- score-based losses: no pairwise term, one block --------------------------------------------
// hinge (math_func.py:2137-2143):    loss_dis = mean relu(1 + s_gen) + mean relu(1 - s_x); loss_gen = mean(-s_gen)
// logistic (math_func.py:2128-2135): loss_dis = mean(softplus(s_gen) + softplus(-s_x)); loss_gen = mean softplus(-s_gen)
__device__ __forceinline__ float softplus_f(float z) { return fmaxf(z, 0.f) + log1pf(expf(-fabsf(z))); }
__device__ __forceinline__ float sigmoid_f(float z) {
    const float e = expf(-fabsf(z));
    return z >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);
}

template <bool HINGE>
__global__ __launch_bounds__(256) void score_loss_kernel(const float *sg, const float *sx, int n, int dis_first,
                                                         float *out, float *grads) {
    __shared__ double red[4][3];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float invn = 1.0f / (float)n;
    // natural order [dLg/ds_gen, dLg/ds_x, dLd/ds_gen, dLd/ds_x]; dis_first as in mmd_kernel
    const int s0 = dis_first ? 2 : 0, s1 = dis_first ? 3 : 1, s2 = dis_first ? 1 : 2, s3 = dis_first ? 0 : 3;
    double a_dg = 0, a_dx = 0, a_g = 0;
    for (int t = tid; t < n; t += 256) {
        const float g = sg[t], x = sx[t];
        float gg, dg, dx;
        if (HINGE) {
            a_dg += (double)fmaxf(1.0f + g, 0.f); a_dx += (double)fmaxf(1.0f - x, 0.f); a_g += (double)(-g);
            gg = -invn; dg = (1.0f + g > 0.f) ? invn : 0.f; dx = (1.0f - x > 0.f) ? -invn : 0.f;
        } else {
            a_dg += (double)softplus_f(g); a_dx += (double)softplus_f(-x); a_g += (double)softplus_f(-g);
            gg = -sigmoid_f(-g) * invn; dg = sigmoid_f(g) * invn; dx = -sigmoid_f(-x) * invn;
        }
        if (grads) {
            grads[(size_t)s0 * n + t] = gg; grads[(size_t)s1 * n + t] = 0.f;
            grads[(size_t)s2 * n + t] = dg; grads[(size_t)s3 * n + t] = dx;
        }
    }
    a_dg = wave_sum(a_dg); a_dx = wave_sum(a_dx); a_g = wave_sum(a_g);
    if (lane == 0) { red[wave][0] = a_dg; red[wave][1] = a_dx; red[wave][2] = a_g; }
    __syncthreads();
    if (tid == 0) {
        double t0 = 0, t1 = 0, t2 = 0;
        for (int w = 0; w < 4; ++w) { t0 += red[w][0]; t1 += red[w][1]; t2 += red[w][2]; }
        const double dn = (double)n;
        out[0] = (float)(t2 / dn);
        out[1] = (float)(t0 / dn + t1 / dn);
        out[2] = (float)(t0 / dn); out[3] = (float)(t1 / dn);
        out[4] = out[5] = out[6] = out[7] = 0.f;
    }
}

static size_t mmd_lds_bytes(int d) {
    size_t fl = 2 * (size_t)kMmdTile * (d + 1) + 2 * (size_t)kMmdRows * d;       // always even
    return fl * sizeof(float) + (kMmdRows * kNumSums + 1) * sizeof(double);
}

}  // namespace mmdgan

using namespace mmdgan;

extern "C" size_t mmdgan_mmd_workspace_bytes(int B, int d) {
    (void)d;
    if (B < 1) return 0;
    size_t blocks = ((size_t)B + kMmdRows - 1) / kMmdRows;
    return 64 + blocks * kNumSums * sizeof(double);    // [counter | pad][partials]
}

// the pairwise launch itself (arguments validated by the callers)
static int launch_pairwise(const float *s_gen, const float *s_x, int B, int d, int loss_type, int dis_first, float w0, float w1,
                           float lower_bound, float upper_bound, float *out_scalars, float *grads, unsigned char *masks,
                           float *dist, void *workspace, hipStream_t st) {
    MmdArgs a;
    a.x = s_gen; a.y = s_x; a.B = B; a.d = d; a.loss_type = loss_type; a.dis_first = dis_first;
    a.w0 = w0; a.w1 = w1; a.lb = lower_bound; a.ub = upper_bound;
    a.counter = (unsigned *)workspace;
    a.partials = (double *)((char *)workspace + 64);
    a.out = out_scalars; a.grads = grads; a.masks = masks; a.dist = dist;
    if (zero_output(a.counter, 64, st) != hipSuccess) return check_launch("mmd_loss memset");
    const int blocks = (B + kMmdRows - 1) / kMmdRows;
    const size_t lds = mmd_lds_bytes(d);
    void (*kern)(MmdArgs);
    if (d == 16 && tuning().mmd_d16)
        kern = loss_type == MMDGAN_LOSS_REP ? mmd_kernel<MMDGAN_LOSS_REP, 16> : loss_type == MMDGAN_LOSS_RMB ? mmd_kernel<MMDGAN_LOSS_RMB, 16>
             : loss_type == MMDGAN_LOSS_MMD_G ? mmd_kernel<MMDGAN_LOSS_MMD_G, 16> : mmd_kernel<MMDGAN_LOSS_MGB, 16>;
    else
        kern = loss_type == MMDGAN_LOSS_REP ? mmd_kernel<MMDGAN_LOSS_REP, 0> : loss_type == MMDGAN_LOSS_RMB ? mmd_kernel<MMDGAN_LOSS_RMB, 0>
             : loss_type == MMDGAN_LOSS_MMD_G ? mmd_kernel<MMDGAN_LOSS_MMD_G, 0> : mmd_kernel<MMDGAN_LOSS_MGB, 0>;
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, st, a);
    return check_launch("mmd_loss");
}

extern "C" int mmdgan_mmd_loss(const float *s_gen, const float *s_x, int B, int d, int loss_type, float w0, float w1,
                               float lower_bound, float upper_bound, float *out_scalars, float *grads,
                               unsigned char *masks, float *dist, void *workspace, void *stream) {
    MMDGAN_REQUIRE(s_gen && s_x && out_scalars, "mmd_loss: null pointer");
    const int dis_first = (loss_type & MMDGAN_LOSS_FLAG_GRADS_DIS_FIRST) != 0;
    loss_type &= ~MMDGAN_LOSS_FLAG_GRADS_DIS_FIRST;
    MMDGAN_REQUIRE(loss_type >= MMDGAN_LOSS_REP && loss_type <= MMDGAN_LOSS_LOGISTIC, "mmd_loss: unknown loss %d", loss_type);
    hipStream_t st = (hipStream_t)stream;
    if (loss_type == MMDGAN_LOSS_HINGE || loss_type == MMDGAN_LOSS_LOGISTIC) {
        MMDGAN_REQUIRE(!masks && !dist, "mmd_loss: hinge / logistic have no pairwise distances");
        MMDGAN_REQUIRE(B >= 1 && d >= 1, "mmd_loss: empty scores");
        if (loss_type == MMDGAN_LOSS_HINGE)
            hipLaunchKernelGGL(score_loss_kernel<true>, dim3(1), dim3(256), 0, st, s_gen, s_x, B * d, dis_first, out_scalars, grads);
        else
            hipLaunchKernelGGL(score_loss_kernel<false>, dim3(1), dim3(256), 0, st, s_gen, s_x, B * d, dis_first, out_scalars, grads);
        return check_launch("score_loss");
    }
    MMDGAN_REQUIRE(workspace, "mmd_loss: null workspace");
    MMDGAN_REQUIRE(B >= 2, "mmd_loss: batch_size must be >= 2 (got %d)", B);
    MMDGAN_REQUIRE(d >= 1 && d <= kMmdMaxD, "mmd_loss: d must be in [1,%d] (got %d)", kMmdMaxD, d);
    if (loss_type == MMDGAN_LOSS_MMD_G || loss_type == MMDGAN_LOSS_MGB) { w0 = 2.0f; w1 = 1.0f; }   // loss_dis = -mmd
    MMDGAN_REQUIRE(w0 - w1 == 1.0f, "w[0]-w[1] must be 1");       // math_func.py:1340
    return launch_pairwise(s_gen, s_x, B, d, loss_type, dis_first, w0, w1, lower_bound, upper_bound, out_scalars, grads, masks,
                           dist, workspace, st);
}

// ------------------------------------------------------------------------------------------------
// 'mmd_g_mix' / 'fixed_g_mix' / 'sgm' (math_func.py:2195-2263): four launches on one stream
//   mix_prepare   coin, group row lists, the two mixed sets gathered          (get_mix_coin, slice_pairwise_distance)
//   mmd_kernel    loss_gen and its gradients on (s_gen, s_x)
//   mmd_kernel    -loss_mix and its gradients on the two mixed sets
//   mix_finish    gradients gathered back to s_gen / s_x rows, scalars, state UPDATE_OPS
// workspace (floats unless noted): [pairwise counter+partials][out2 8][out3 8][pos 2B int][XX 2B*d][G2 4*B*d][G3 4*B*d]
// ------------------------------------------------------------------------------------------------
namespace mmdgan {
constexpr int kMixMaxB = 8192;           // the coin's prefix counts live in LDS (33 KB)

struct MixLayout {
    size_t pair_bytes, out2, out3, misc, pos, xx, g2, g3, total;     // byte offsets
};
static MixLayout mix_layout(int B, int d) {
    MixLayout L;
    L.pair_bytes = (mmdgan_mmd_workspace_bytes(B, d) + 255) / 256 * 256;
    L.out2 = L.pair_bytes;
    L.out3 = L.out2 + 8 * sizeof(float);
    L.misc = L.out3 + 8 * sizeof(float);
    L.pos = L.misc + 8 * sizeof(float);
    L.xx = (L.pos + 2 * (size_t)B * sizeof(int) + 255) / 256 * 256;
    L.g2 = L.xx + 2 * (size_t)B * d * sizeof(float);
    L.g3 = L.g2 + 4 * (size_t)B * d * sizeof(float);
    L.total = L.g3 + 4 * (size_t)B * d * sizeof(float);
    return L;
}

// one block.  Row k of set X1 (XX rows 0..B-1) / X2 (rows B..2B-1) in tf.boolean_mask order:
//   X1 = [gen_i : coin_i] ++ [data_i : !coin_i],  X2 = [gen_i : !coin_i] ++ [data_i : coin_i]
// pos[i] = XX row of gen_i, pos[B + i] = XX row of data_i.
__global__ __launch_bounds__(256) void mix_prepare_kernel(const float *__restrict__ sg, const float *__restrict__ sx, int B, int d,
                                                          const float *__restrict__ uni, const float *__restrict__ state,
                                                          unsigned char *masks, int *pos, float *XX, float *misc) {
    extern __shared__ int sm[];
    int *coin = sm;                    // [B]
    int *tcount = sm + B;              // [256] chunk counts -> exclusive prefix
    __shared__ int total_true;
    const int tid = threadIdx.x;
    const float mix_prob = state[1];
    for (int i = tid; i < B; i += 256) coin[i] = uni[i] > mix_prob ? 1 : 0;       // tf.greater(uni, mix_prob), :2080
    __syncthreads();
    const int chunk = (B + 255) / 256;
    const int lo = min(tid * chunk, B), hi = min(lo + chunk, B);
    int c = 0;
    for (int i = lo; i < hi; ++i) c += coin[i];
    tcount[tid] = c;
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int t = 0; t < 256; ++t) { const int v = tcount[t]; tcount[t] = run; run += v; }
        total_true = run;
    }
    __syncthreads();
    const int T = total_true;
    int t_i = tcount[tid];
    for (int i = lo; i < hi; ++i) {
        const int ci = coin[i], f_i = i - t_i;
        pos[i] = ci ? t_i : B + f_i;                      // gen_i
        pos[B + i] = ci ? B + (B - T) + t_i : T + f_i;    // data_i
        if (masks) {
            masks[i] = (unsigned char)ci;                                                  // mix_indices
            masks[B + i] = (unsigned char)ci; masks[2 * B + i] = (unsigned char)(1 - ci);   // mix_group_1 = [idx, !idx]
            masks[3 * B + i] = (unsigned char)(1 - ci); masks[4 * B + i] = (unsigned char)ci;   // mix_group_2 = [!idx, idx]
        }
        t_i += ci;
    }
    __threadfence_block();
    __syncthreads();
    for (int t = tid; t < B * d; t += 256) {
        const int i = t / d, k = t - i * d;
        XX[(size_t)pos[i] * d + k] = sg[t];
        XX[(size_t)pos[B + i] * d + k] = sx[t];
    }
    if (tid == 0) misc[0] = (float)T;
}

__global__ __launch_bounds__(256) void mix_finish_kernel(int B, int d, int dis_first, const int *__restrict__ pos,
                                                         const float *__restrict__ G2, const float *__restrict__ G3,
                                                         const float *__restrict__ out2, const float *__restrict__ out3,
                                                         const float *__restrict__ misc, float threshold, float rho_avg, float rho_prob, float *state,
                                                         float *out, float *grads) {
    const size_t n = (size_t)B * d;
    // natural order [dLg/ds_gen, dLg/ds_x, dLd/ds_gen, dLd/ds_x]; dis_first as in mmd_kernel
    const int s0 = dis_first ? 2 : 0, s1 = dis_first ? 3 : 1, s2 = dis_first ? 1 : 2, s3 = dis_first ? 0 : 3;
    const float *Gd = G3 + 2 * n;          // the loss_dis slots of the mixed launch: d(-loss_mix)/d[X1 ; X2], 2B rows
    if (grads) {
        for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < n; t += (size_t)gridDim.x * 256) {
            const int i = (int)(t / d), k = (int)(t - (size_t)i * d);
            grads[s0 * n + t] = G2[t];
            grads[s1 * n + t] = G2[n + t];
            grads[s2 * n + t] = Gd[(size_t)pos[i] * d + k];
            grads[s3 * n + t] = Gd[(size_t)pos[B + i] * d + k];
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const float lg = out2[0], la = state[0], mp = state[1];
        out[0] = lg; out[1] = out3[1];
        out[2] = out2[2]; out[3] = out2[3]; out[4] = out2[4];
        out[5] = la; out[6] = mp; out[7] = misc[0];
        // UPDATE_OPS (math_func.py:2031-2033, 1999-2011): both right-hand sides read the pre-update values
        const float keep = (float)(1.0 - (double)rho_avg);
        state[0] = keep * la + rho_avg * lg;
        state[1] = fminf(fmaxf(mp + rho_prob * (la - threshold), 0.0f), 0.5f);
    }
}
}  // namespace mmdgan

extern "C" size_t mmdgan_mmd_mix_workspace_bytes(int B, int d) {
    if (B < 2 || d < 1) return 0;
    return mix_layout(B, d).total;
}

extern "C" int mmdgan_mmd_mix_loss(const float *s_gen, const float *s_x, int B, int d, int loss_type, const float *uni,
                                   float mix_threshold, float loss_average_update, float mix_prob_update, float *state,
                                   float *out_scalars, float *grads, unsigned char *masks, void *workspace, void *stream) {
    const int dis_first = (loss_type & MMDGAN_LOSS_FLAG_GRADS_DIS_FIRST) != 0;
    loss_type &= ~MMDGAN_LOSS_FLAG_GRADS_DIS_FIRST;
    MMDGAN_REQUIRE(loss_type == MMDGAN_LOSS_MMD_G_MIX || loss_type == MMDGAN_LOSS_SGM, "mmd_mix_loss: unknown loss %d", loss_type);
    MMDGAN_REQUIRE(B >= 2 && B <= kMixMaxB, "mmd_mix_loss: batch_size must be in [2,%d] (got %d)", kMixMaxB, B);
    MMDGAN_REQUIRE(d >= 1 && d <= kMmdMaxD, "mmd_mix_loss: d must be in [1,%d] (got %d)", kMmdMaxD, d);
    MMDGAN_REQUIRE(s_gen && s_x && uni && state && out_scalars && workspace, "mmd_mix_loss: null pointer");
    hipStream_t st = (hipStream_t)stream;
    const MixLayout L = mix_layout(B, d);
    char *ws = (char *)workspace;
    float *out2 = (float *)(ws + L.out2), *out3 = (float *)(ws + L.out3), *misc = (float *)(ws + L.misc);
    int *pos = (int *)(ws + L.pos);
    float *XX = (float *)(ws + L.xx), *G2 = (float *)(ws + L.g2), *G3 = (float *)(ws + L.g3);
    hipLaunchKernelGGL(mix_prepare_kernel, dim3(1), dim3(256), (size_t)(B + 256) * sizeof(int), st, s_gen, s_x, B, d, uni,
                       (const float *)state, masks, pos, XX, misc);
    if (int rc = check_launch("mmd_mix_loss prepare")) return rc;
    // the mixture of five Gaussians, or the single sigma-1 one; w = (2, 1) makes the loss_dis slots -MMD
    const int pair = loss_type == MMDGAN_LOSS_SGM ? MMDGAN_LOSS_REP : MMDGAN_LOSS_MMD_G;
    if (int rc = launch_pairwise(s_gen, s_x, B, d, pair, 0, 2.0f, 1.0f, 0.25f, 4.0f, out2, G2, nullptr, nullptr, ws, st)) return rc;
    if (int rc = launch_pairwise(XX, XX + (size_t)B * d, B, d, pair, 0, 2.0f, 1.0f, 0.25f, 4.0f, out3, G3, nullptr, nullptr, ws, st))
        return rc;
    const size_t n = (size_t)B * d;
    const int blocks = (int)((n + 255) / 256 < 256 ? (n + 255) / 256 : 256);
    hipLaunchKernelGGL(mix_finish_kernel, dim3(blocks), dim3(256), 0, st, B, d, dis_first, (const int *)pos, (const float *)G2,
                       (const float *)G3, (const float *)out2, (const float *)out3, (const float *)misc, mix_threshold,
                       loss_average_update,
                       mix_prob_update, state, out_scalars, grads);
    return check_launch("mmd_mix_loss finish");
}
