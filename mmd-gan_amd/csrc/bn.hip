// Batch normalisation over [rows, C] (NHWC flattened), training and inference, with the following
// activation fused.  Replaces tf.layers.batch_normalization(axis=1, training, fused=True) at
// layer_func.py:960-966 (TF defaults momentum .99, eps 1e-3) and its autodiff.
//
// HBM-bound.  Statistics are column sums accumulated in double (products of floats are exact in
// double, so E[x^2]-mean^2 carries no fp32 cancellation).  Two launches per direction: kernel A
// reduces a chunk of rows per workgroup and adds its 2*C sums to the totals in the workspace with
// fp64 atomics (256 workgroups x 2C adds: nothing); kernel B derives mean / 1/std (or dgamma / dbeta)
// for all channels from the totals into LDS at the top of every workgroup, applies, and workgroup 0
// writes the saved / moving statistics.  (A separate "finish" launch between the two cost 5 us + a
// launch gap per BN layer and direction on the generator's critical path.)  x is read twice, y
// written once.  The workspace (2*C doubles) must be zero on entry; the entries zero it themselves
// unless mmdgan_set_outputs_prezeroed(1) says the caller did.
#include "common.h"
#include <stdint.h>

namespace mmdgan {

constexpr int kBnMaxSplits = 256;

// partial[which*C + c] += this workgroup's sum; which 0: sum a, 1: sum b
template <int MODE>   // 0: a = x, b = x*x      1: a = dz, b = dz*xhat  (dz = dy*act'(y))
__global__ __launch_bounds__(256) void bn_partial_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                                         const float *__restrict__ dy, long rows, int C,
                                                         long rows_per_split, const float *mean, const float *invstd,
                                                         int act, double *partial) {
    __shared__ double red[2][4][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const long r0 = (long)blockIdx.y * rows_per_split;
    long r1 = r0 + rows_per_split;
    if (r1 > rows) r1 = rows;
    double sa = 0, sb = 0;
    if (c < C) {
        float mu = 0.f, is = 0.f;
        if (MODE == 1) { mu = mean[c]; is = invstd[c]; }
        for (long r = r0 + rl; r < r1; r += 4) {
            const long o = r * C + c;
            if (MODE == 0) {
                const double v = (double)x[o];
                sa += v; sb += v * v;
            } else {
                const float dz = dy[o] * act_bwd_from_out(y[o], act);
                const float xh = (x[o] - mu) * is;
                sa += (double)dz; sb += (double)dz * (double)xh;
            }
        }
    }
    red[0][rl][cl] = sa; red[1][rl][cl] = sb;
    __syncthreads();
    if (rl == 0 && c < C) {
        atomicAdd(partial + c, red[0][0][cl] + red[0][1][cl] + red[0][2][cl] + red[0][3][cl]);
        atomicAdd(partial + C + c, red[1][0][cl] + red[1][1][cl] + red[1][2][cl] + red[1][3][cl]);
    }
}

// float4 variant for C % 4 == 0: a block covers 64 channels as 16 float4 lanes x 16 row lanes, four
// independent rows in flight per thread (the scalar kernel above issues one 4-byte load per row and
// runs at ~1 TB/s; this one streams)
template <int MODE>
__global__ __launch_bounds__(256) void bn_partial_v4_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                                            const float *__restrict__ dy, long rows, int C,
                                                            long rows_per_split, const float *mean, const float *invstd,
                                                            int act, double *partial) {
    __shared__ double red[2][16][65];
    const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int c = blockIdx.x * 64 + cl * 4;
    const long r0 = (long)blockIdx.y * rows_per_split;
    long r1 = r0 + rows_per_split;
    if (r1 > rows) r1 = rows;
    double sa[4] = {0, 0, 0, 0}, sb[4] = {0, 0, 0, 0};
    if (c < C) {
        float mu[4] = {0, 0, 0, 0}, is[4] = {0, 0, 0, 0};
        if (MODE == 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { mu[j] = mean[c + j]; is[j] = invstd[c + j]; }
        }
        for (long r = r0 + rl; r < r1; r += 64) {
            float4 vx[4], vy[4], vd[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long rr = r + u * 16;
                const bool ok = rr < r1;
                const long o = (ok ? rr : r) * C + c;
                vx[u] = *reinterpret_cast<const float4 *>(x + o);
                if (MODE == 1) {
                    vy[u] = *reinterpret_cast<const float4 *>(y + o);
                    vd[u] = *reinterpret_cast<const float4 *>(dy + o);
                    if (!ok) vd[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                } else if (!ok) vx[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float xv[4] = {vx[u].x, vx[u].y, vx[u].z, vx[u].w};
                if (MODE == 0) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) { const double v = (double)xv[j]; sa[j] += v; sb[j] += v * v; }
                } else {
                    const float yv[4] = {vy[u].x, vy[u].y, vy[u].z, vy[u].w}, dv[4] = {vd[u].x, vd[u].y, vd[u].z, vd[u].w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float dz = dv[j] * act_bwd_from_out(yv[j], act);
                        const float xh = (xv[j] - mu[j]) * is[j];
                        sa[j] += (double)dz; sb[j] += (double)dz * (double)xh;
                    }
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { red[0][rl][cl * 4 + j] = sa[j]; red[1][rl][cl * 4 + j] = sb[j]; }
    __syncthreads();
    if (threadIdx.x < 128) {
        const int which = threadIdx.x >> 6, ch = threadIdx.x & 63;
        if (blockIdx.x * 64 + ch < C) {
            double t = 0;
#pragma unroll
            for (int k = 0; k < 16; ++k) t += red[which][k][ch];
            atomicAdd(partial + (size_t)which * C + blockIdx.x * 64 + ch, t);
        }
    }
}

// mean / 1/std of every channel from the totals into LDS (every workgroup; workgroup 0 also publishes the
// saved and moving statistics), then y = act((x - mean) * invstd * gamma + beta)
template <int VEC>
__global__ __launch_bounds__(256) void bn_train_apply_kernel(const float *__restrict__ x, long total, int C,
                                                             const float *__restrict__ gamma, const float *__restrict__ beta,
                                                             const double *__restrict__ totals, long rows, float eps,
                                                             float momentum, int unbiased, int act, float *__restrict__ y,
                                                             float *save_mean, float *save_invstd, const float *mm,
                                                             const float *mv, float *new_mm, float *new_mv) {
    extern __shared__ float stat[];                     // [mean C][invstd C]
    for (int c = threadIdx.x; c < C; c += 256) {
        const double n = (double)rows, mean = totals[c] / n;
        double var = totals[C + c] / n - mean * mean;   // biased batch variance
        if (var < 0) var = 0;
        const float is = (float)(1.0 / sqrt(var + (double)eps));
        stat[c] = (float)mean;
        stat[C + c] = is;
        if (blockIdx.x == 0) {
            save_mean[c] = (float)mean;
            save_invstd[c] = is;
            if (new_mm) {
                const double var_u = unbiased ? var * (n / (n > 1 ? n - 1.0 : 1.0)) : var;
                const float om = mm[c], ov = mv[c];     // reads precede writes (buffers may alias)
                new_mm[c] = om * momentum + (float)mean * (1.f - momentum);
                new_mv[c] = ov * momentum + (float)var_u * (1.f - momentum);
            }
        }
    }
    __syncthreads();
    const long stride = (long)gridDim.x * 256;
    if (VEC == 4) {
        for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < total / 4; q += stride) {
            const int c = (int)((q * 4) % C);
            const float4 v = reinterpret_cast<const float4 *>(x)[q];
            const float4 g = *reinterpret_cast<const float4 *>(gamma + c), b = *reinterpret_cast<const float4 *>(beta + c);
            const float in[4] = {v.x, v.y, v.z, v.w}, gv[4] = {g.x, g.y, g.z, g.w}, bv[4] = {b.x, b.y, b.z, b.w};
            float o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = act_fwd((in[j] - stat[c + j]) * stat[C + c + j] * gv[j] + bv[j], act);
            reinterpret_cast<float4 *>(y)[q] = make_float4(o[0], o[1], o[2], o[3]);
        }
    } else {
        for (long o = (long)blockIdx.x * 256 + threadIdx.x; o < total; o += stride) {
            const int c = o % C;
            y[o] = act_fwd((x[o] - stat[c]) * stat[C + c] * gamma[c] + beta[c], act);
        }
    }
}

__global__ __launch_bounds__(256) void bn_apply_kernel(const float *__restrict__ x, long total, int C,
                                                       const float *__restrict__ gamma, const float *__restrict__ beta,
                                                       const float *__restrict__ mean, const float *__restrict__ invstd,
                                                       float eps_for_var, int use_var, int act, float *__restrict__ y) {
    const long stride = (long)gridDim.x * 256;
    for (long o = (long)blockIdx.x * 256 + threadIdx.x; o < total; o += stride) {
        const int c = o % C;
        const float is = use_var ? rsqrtf(invstd[c] + eps_for_var) : invstd[c];   // infer: invstd holds the variance
        y[o] = act_fwd((x[o] - mean[c]) * is * gamma[c] + beta[c], act);
    }
}

__global__ __launch_bounds__(256) void bn_apply_v4_kernel(const float *__restrict__ x, long total4, int C,
                                                          const float *__restrict__ gamma, const float *__restrict__ beta,
                                                          const float *__restrict__ mean, const float *__restrict__ invstd,
                                                          float eps_for_var, int use_var, int act, float *__restrict__ y) {
    const long stride = (long)gridDim.x * 256;
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < total4; q += stride) {
        const int c = (int)((q * 4) % C);
        const float4 v = reinterpret_cast<const float4 *>(x)[q];
        const float4 g = *reinterpret_cast<const float4 *>(gamma + c), b = *reinterpret_cast<const float4 *>(beta + c);
        const float4 m = *reinterpret_cast<const float4 *>(mean + c), iv = *reinterpret_cast<const float4 *>(invstd + c);
        const float in[4] = {v.x, v.y, v.z, v.w}, gv[4] = {g.x, g.y, g.z, g.w}, bv[4] = {b.x, b.y, b.z, b.w};
        const float mv[4] = {m.x, m.y, m.z, m.w}, sv[4] = {iv.x, iv.y, iv.z, iv.w};
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float is = use_var ? rsqrtf(sv[j] + eps_for_var) : sv[j];
            o[j] = act_fwd((in[j] - mv[j]) * is * gv[j] + bv[j], act);
        }
        reinterpret_cast<float4 *>(y)[q] = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// dx = gamma*invstd*(dz - dbeta/n - xhat*dgamma/n)
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                                           const float *__restrict__ dy, long total, long rows, int C,
                                                           const float *__restrict__ gamma, const float *__restrict__ mean,
                                                           const float *__restrict__ invstd, const float *__restrict__ dgamma,
                                                           const float *__restrict__ dbeta, int act, float *__restrict__ dx) {
    const float invn = 1.0f / (float)rows;
    const long stride = (long)gridDim.x * 256;
    for (long o = (long)blockIdx.x * 256 + threadIdx.x; o < total; o += stride) {
        const int c = o % C;
        const float dz = dy[o] * act_bwd_from_out(y[o], act);
        const float xh = (x[o] - mean[c]) * invstd[c];
        dx[o] = gamma[c] * invstd[c] * (dz - dbeta[c] * invn - xh * dgamma[c] * invn);
    }
}

__global__ __launch_bounds__(256) void bn_bwd_apply_v4_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                                              const float *__restrict__ dy, long total4, long rows, int C,
                                                              const float *__restrict__ gamma, const float *__restrict__ mean,
                                                              const float *__restrict__ invstd, const float *__restrict__ dgamma,
                                                              const float *__restrict__ dbeta, int act, float *__restrict__ dx) {
    const float invn = 1.0f / (float)rows;
    const long stride = (long)gridDim.x * 256;
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < total4; q += stride) {
        const int c = (int)((q * 4) % C);
        const float4 a = reinterpret_cast<const float4 *>(x)[q], b = reinterpret_cast<const float4 *>(y)[q];
        const float4 d = reinterpret_cast<const float4 *>(dy)[q];
        const float4 g = *reinterpret_cast<const float4 *>(gamma + c), m = *reinterpret_cast<const float4 *>(mean + c);
        const float4 iv = *reinterpret_cast<const float4 *>(invstd + c);
        const float4 dg = *reinterpret_cast<const float4 *>(dgamma + c), db = *reinterpret_cast<const float4 *>(dbeta + c);
        const float xv[4] = {a.x, a.y, a.z, a.w}, yv[4] = {b.x, b.y, b.z, b.w}, dv[4] = {d.x, d.y, d.z, d.w};
        const float gv[4] = {g.x, g.y, g.z, g.w}, mv[4] = {m.x, m.y, m.z, m.w}, sv[4] = {iv.x, iv.y, iv.z, iv.w};
        const float dgv[4] = {dg.x, dg.y, dg.z, dg.w}, dbv[4] = {db.x, db.y, db.z, db.w};
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float dz = dv[j] * act_bwd_from_out(yv[j], act);
            const float xh = (xv[j] - mv[j]) * sv[j];
            o[j] = gv[j] * sv[j] * (dz - dbv[j] * invn - xh * dgv[j] * invn);
        }
        reinterpret_cast<float4 *>(dx)[q] = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// dbeta / dgamma of every channel from the totals into LDS (workgroup 0 publishes them), then
// dx = gamma*invstd*(dz - dbeta/n - xhat*dgamma/n)
template <int VEC>
__global__ __launch_bounds__(256) void bn_bwd_fused_apply_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                                                 const float *__restrict__ dy, long total, long rows, int C,
                                                                 const float *__restrict__ gamma, const float *__restrict__ mean,
                                                                 const float *__restrict__ invstd, const double *__restrict__ totals,
                                                                 float *dgamma, float *dbeta, int act, float *__restrict__ dx) {
    extern __shared__ float stat[];                     // [dbeta C][dgamma C]
    for (int c = threadIdx.x; c < C; c += 256) {
        const float db = (float)totals[c], dg = (float)totals[C + c];
        stat[c] = db;
        stat[C + c] = dg;
        if (blockIdx.x == 0) { dbeta[c] = db; dgamma[c] = dg; }
    }
    __syncthreads();
    const float invn = 1.0f / (float)rows;
    const long stride = (long)gridDim.x * 256;
    if (VEC == 4) {
        for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < total / 4; q += stride) {
            const int c = (int)((q * 4) % C);
            const float4 a = reinterpret_cast<const float4 *>(x)[q], b = reinterpret_cast<const float4 *>(y)[q];
            const float4 d = reinterpret_cast<const float4 *>(dy)[q];
            const float4 g = *reinterpret_cast<const float4 *>(gamma + c), m = *reinterpret_cast<const float4 *>(mean + c);
            const float4 iv = *reinterpret_cast<const float4 *>(invstd + c);
            const float xv[4] = {a.x, a.y, a.z, a.w}, yv[4] = {b.x, b.y, b.z, b.w}, dv[4] = {d.x, d.y, d.z, d.w};
            const float gv[4] = {g.x, g.y, g.z, g.w}, mv[4] = {m.x, m.y, m.z, m.w}, sv[4] = {iv.x, iv.y, iv.z, iv.w};
            float o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float dz = dv[j] * act_bwd_from_out(yv[j], act);
                const float xh = (xv[j] - mv[j]) * sv[j];
                o[j] = gv[j] * sv[j] * (dz - stat[c + j] * invn - xh * stat[C + c + j] * invn);
            }
            reinterpret_cast<float4 *>(dx)[q] = make_float4(o[0], o[1], o[2], o[3]);
        }
    } else {
        for (long o = (long)blockIdx.x * 256 + threadIdx.x; o < total; o += stride) {
            const int c = o % C;
            const float dz = dy[o] * act_bwd_from_out(y[o], act);
            const float xh = (x[o] - mean[c]) * invstd[c];
            dx[o] = gamma[c] * invstd[c] * (dz - stat[c] * invn - xh * stat[C + c] * invn);
        }
    }
}

static inline bool al16(const void *p) { return ((uintptr_t)p & 15) == 0; }

static int bn_splits(long rows, int C, long *rows_per_split) {
    const int cblocks = (C + 63) / 64;
    long splits = 1024 / cblocks;
    if (splits > kBnMaxSplits) splits = kBnMaxSplits;
    if (splits < 1) splits = 1;
    long rps = (rows + splits - 1) / splits;
    if (rps < 8) rps = 8;
    splits = (rows + rps - 1) / rps;
    *rows_per_split = rps;
    return (int)splits;
}

}  // namespace mmdgan

using namespace mmdgan;

extern "C" size_t mmdgan_bn_workspace_bytes(int C) { return C < 1 ? 0 : (size_t)2 * C * sizeof(double); }

extern "C" int mmdgan_bn_fwd_train(const float *x, long rows, int C, const float *gamma, const float *beta, float eps,
                                   float momentum, int unbiased_moving_var, int act, float *y, float *save_mean,
                                   float *save_invstd, const float *moving_mean, const float *moving_var,
                                   float *new_moving_mean, float *new_moving_var, void *workspace, void *stream) {
    MMDGAN_REQUIRE(x && gamma && beta && y && save_mean && save_invstd && workspace, "bn_fwd_train: null pointer");
    MMDGAN_REQUIRE(rows >= 1 && C >= 1, "bn_fwd_train: bad shape");
    MMDGAN_REQUIRE(!new_moving_mean || (moving_mean && moving_var && new_moving_var), "bn_fwd_train: moving stats");
    hipStream_t st = (hipStream_t)stream;
    long rps;
    const int splits = bn_splits(rows, C, &rps);
    double *part = (double *)workspace;
    if (zero_output(part, sizeof(double) * 2 * C, st) != hipSuccess) return check_launch("bn_fwd_train memset");
    const bool v4 = (C % 4) == 0 && al16(x) && al16(y) && al16(gamma) && al16(beta) && al16(save_mean) && al16(save_invstd);
    if (v4)
        hipLaunchKernelGGL(bn_partial_v4_kernel<0>, dim3((C + 63) / 64, splits), dim3(256), 0, st, x, nullptr, nullptr,
                           rows, C, rps, nullptr, nullptr, 0, part);
    else
        hipLaunchKernelGGL(bn_partial_kernel<0>, dim3((C + 63) / 64, splits), dim3(256), 0, st, x, nullptr, nullptr, rows,
                           C, rps, nullptr, nullptr, 0, part);
    const long total = rows * C;
    long blocks = ((v4 ? total / 4 : total) + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    const size_t lds = sizeof(float) * 2 * C;
    if (v4)
        hipLaunchKernelGGL(bn_train_apply_kernel<4>, dim3((unsigned)blocks), dim3(256), lds, st, x, total, C, gamma, beta, part, rows,
                           eps, momentum, unbiased_moving_var, act, y, save_mean, save_invstd, moving_mean, moving_var,
                           new_moving_mean, new_moving_var);
    else
        hipLaunchKernelGGL(bn_train_apply_kernel<1>, dim3((unsigned)blocks), dim3(256), lds, st, x, total, C, gamma, beta, part, rows,
                           eps, momentum, unbiased_moving_var, act, y, save_mean, save_invstd, moving_mean, moving_var,
                           new_moving_mean, new_moving_var);
    return check_launch("bn_fwd_train");
}

extern "C" int mmdgan_bn_fwd_infer(const float *x, long rows, int C, const float *gamma, const float *beta, float eps,
                                   int act, const float *moving_mean, const float *moving_var, float *y, void *stream) {
    MMDGAN_REQUIRE(x && gamma && beta && y && moving_mean && moving_var && rows >= 1 && C >= 1, "bn_fwd_infer: bad arguments");
    const long total = rows * C;
    long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (C % 4 == 0 && al16(x) && al16(y) && al16(gamma) && al16(beta) && al16(moving_mean) && al16(moving_var)) {
        blocks = (total / 4 + 255) / 256;
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(bn_apply_v4_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, total / 4, C,
                           gamma, beta, moving_mean, moving_var, eps, 1, act, y);
    } else
        hipLaunchKernelGGL(bn_apply_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, total, C, gamma,
                           beta, moving_mean, moving_var, eps, 1, act, y);
    return check_launch("bn_fwd_infer");
}

extern "C" int mmdgan_bn_bwd(const float *x, const float *y, const float *dy, long rows, int C, const float *gamma,
                             const float *save_mean, const float *save_invstd, int act, float *dx, float *dgamma,
                             float *dbeta, void *workspace, void *stream) {
    MMDGAN_REQUIRE(x && y && dy && gamma && save_mean && save_invstd && dx && dgamma && dbeta && workspace,
                   "bn_bwd: null pointer");
    MMDGAN_REQUIRE(rows >= 1 && C >= 1, "bn_bwd: bad shape");
    hipStream_t st = (hipStream_t)stream;
    long rps;
    const int splits = bn_splits(rows, C, &rps);
    double *part = (double *)workspace;
    if (zero_output(part, sizeof(double) * 2 * C, st) != hipSuccess) return check_launch("bn_bwd memset");
    const bool v4 = (C % 4) == 0 && al16(x) && al16(y) && al16(dy) && al16(dx) && al16(gamma) && al16(save_mean) &&
                    al16(save_invstd) && al16(dgamma) && al16(dbeta);
    if (v4)
        hipLaunchKernelGGL(bn_partial_v4_kernel<1>, dim3((C + 63) / 64, splits), dim3(256), 0, st, x, y, dy, rows, C, rps,
                           save_mean, save_invstd, act, part);
    else
        hipLaunchKernelGGL(bn_partial_kernel<1>, dim3((C + 63) / 64, splits), dim3(256), 0, st, x, y, dy, rows, C, rps,
                           save_mean, save_invstd, act, part);
    const long total = rows * C;
    long blocks = ((v4 ? total / 4 : total) + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    const size_t lds = sizeof(float) * 2 * C;
    if (v4)
        hipLaunchKernelGGL(bn_bwd_fused_apply_kernel<4>, dim3((unsigned)blocks), dim3(256), lds, st, x, y, dy, total, rows, C, gamma,
                           save_mean, save_invstd, part, dgamma, dbeta, act, dx);
    else
        hipLaunchKernelGGL(bn_bwd_fused_apply_kernel<1>, dim3((unsigned)blocks), dim3(256), lds, st, x, y, dy, total, rows, C, gamma,
                           save_mean, save_invstd, part, dgamma, dbeta, act, dx);
    return check_launch("bn_bwd");
}
