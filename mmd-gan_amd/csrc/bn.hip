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

#include "conv_internal.h"

namespace mmdgan {

// One workgroup per CU: kBnBlocks workgroups = (channel blocks of 64) x (row splits).
constexpr int kBnBlocks = 256;
// The apply kernels: at most kBnApplyBlocks workgroups (every one of them first reads all slots of the totals: 8 KB), each
// thread with kBnApplyInFlight float4 per tensor in flight.
constexpr int kBnApplyBlocks = 1024, kBnApplyInFlight = 4;
// The totals exist bn_slots(C) times over ([slot][2][C]): row split i adds into slot i % slots and the apply kernels sum the
// slots.  The fp64 atomics of all workgroups hit the same 2C addresses within the same microsecond and cost ~20 ns each
// per address, in series: 5 us of a 10.8 us kernel with 256 row splits on one set of totals (16.8 MB tensor, 64 channels).
// 8 slots per channel block of 64, fewer for wider tensors (their rows are split fewer ways): 32 adds per address always,
// and the 16 KB of totals every apply workgroup reads first does not grow with C.
__host__ __device__ inline int bn_slots(int C) {
    const int cblocks = (C + 63) / 64;
    return cblocks >= 8 ? 1 : 8 / cblocks;
}

// The affine map of the forward pass as ONE expression: the backward kernels that are not handed y decide the sign of
// relu / lrelu from it again, and must land on the bits bn_train_apply_kernel produced (same operations, same order,
// the fused multiply-add spelled out so that no compiler choice can differ between the two).
__device__ __forceinline__ float bn_affine(float x, float mean, float invstd, float gamma, float beta) {
    return __builtin_fmaf((x - mean) * invstd, gamma, beta);
}

// partial[which*C + c] += this workgroup's sum; which 0: sum a, 1: sum b
template <int MODE>   // 0: a = x, b = x*x      1: a = dz, b = dz*xhat  (dz = dy*act'(y))
__global__ __launch_bounds__(256) void bn_partial_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                                         const float *__restrict__ dy, long rows, int C,
                                                         long rows_per_split, const float *mean, const float *invstd,
                                                         const float *gamma, const float *beta, int act, double *partial) {
    __shared__ double red[2][4][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const long r0 = (long)blockIdx.y * rows_per_split;
    long r1 = r0 + rows_per_split;
    if (r1 > rows) r1 = rows;
    double sa = 0, sb = 0;
    if (c < C) {
        float mu = 0.f, is = 0.f, ga = 0.f, be = 0.f;
        if (MODE == 1) { mu = mean[c]; is = invstd[c]; if (!y) { ga = gamma[c]; be = beta[c]; } }
        for (long r = r0 + rl; r < r1; r += 4) {
            const long o = r * C + c;
            if (MODE == 0) {
                const double v = (double)x[o];
                sa += v; sb += v * v;
            } else {
                const float xo = x[o];
                const float dz = dy[o] * act_bwd_from_out(y ? y[o] : bn_affine(xo, mu, is, ga, be), act);
                const float xh = (xo - mu) * is;
                sa += (double)dz; sb += (double)dz * (double)xh;
            }
        }
    }
    red[0][rl][cl] = sa; red[1][rl][cl] = sb;
    __syncthreads();
    if (rl == 0 && c < C) {
        double *slot = partial + (size_t)(blockIdx.y % bn_slots(C)) * 2 * C;
        atomicAdd(slot + c, red[0][0][cl] + red[0][1][cl] + red[0][2][cl] + red[0][3][cl]);
        atomicAdd(slot + C + c, red[1][0][cl] + red[1][1][cl] + red[1][2][cl] + red[1][3][cl]);
    }
}

// float4 variant for C % 4 == 0: a block covers 64 channels as 16 float4 lanes x 16 row lanes, U independent rows in
// flight per thread.  (The scalar kernel above issues one 4-byte load per row and runs at ~1 TB/s.  With four rows in
// flight a CU held 16 KB of loads - one 256-thread workgroup per CU, 256 of them - and the kernel ran at the memory
// latency, 1.4-1.7 TB/s; U = 16 puts a whole 256-row split in flight at once.)
// MODE 1, y == nullptr: dz = dy * act'(t) with t = bn_affine(x) recomputed (relu / lrelu / linear: the sign is all the
// derivative needs) - a third less traffic than reading y back.
template <int MODE, int U>
__global__ __launch_bounds__(256) void bn_partial_v4_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                                            const float *__restrict__ dy, long rows, int C,
                                                            long rows_per_split, const float *mean, const float *invstd,
                                                            const float *gamma, const float *beta, int act, double *partial) {
    __shared__ double red[2][16][65];
    const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int c = blockIdx.x * 64 + cl * 4;
    const long r0 = (long)blockIdx.y * rows_per_split;
    long r1 = r0 + rows_per_split;
    if (r1 > rows) r1 = rows;
    double sa[4] = {0, 0, 0, 0}, sb[4] = {0, 0, 0, 0};
    if (c < C) {
        float mu[4] = {0, 0, 0, 0}, is[4] = {0, 0, 0, 0}, ga[4] = {0, 0, 0, 0}, be[4] = {0, 0, 0, 0};
        if (MODE == 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { mu[j] = mean[c + j]; is[j] = invstd[c + j]; }
            if (!y) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { ga[j] = gamma[c + j]; be[j] = beta[c + j]; }
            }
        }
        for (long r = r0 + rl; r < r1; r += 16 * U) {
            float4 vx[U], vy[U], vd[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const long rr = r + u * 16;
                const bool ok = rr < r1;
                const long o = (ok ? rr : r) * C + c;
                vx[u] = *reinterpret_cast<const float4 *>(x + o);
                if (MODE == 1) {
                    if (y) vy[u] = *reinterpret_cast<const float4 *>(y + o);
                    vd[u] = *reinterpret_cast<const float4 *>(dy + o);
                    if (!ok) vd[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                } else if (!ok) vx[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float xv[4] = {vx[u].x, vx[u].y, vx[u].z, vx[u].w};
                if (MODE == 0) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) { const double v = (double)xv[j]; sa[j] += v; sb[j] += v * v; }
                } else {
                    const float yv[4] = {vy[u].x, vy[u].y, vy[u].z, vy[u].w}, dv[4] = {vd[u].x, vd[u].y, vd[u].z, vd[u].w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float out = y ? yv[j] : bn_affine(xv[j], mu[j], is[j], ga[j], be[j]);
                        const float dz = dv[j] * act_bwd_from_out(out, act);
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
            atomicAdd(partial + (size_t)(blockIdx.y % bn_slots(C)) * 2 * C + (size_t)which * C + blockIdx.x * 64 + ch, t);
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
    // the first batch of x is requested BEFORE the totals are read: every workgroup reads the same few cache lines there
    // (8 KB of slots), which takes a microsecond or two per workgroup when a thousand of them ask at once
    const long stride = (long)gridDim.x * 256, n4 = VEC == 4 ? total / 4 : 0;
    long q0 = (long)blockIdx.x * 256 + threadIdx.x;
    float4 v[kBnApplyInFlight];
    if (VEC == 4) {
#pragma unroll
        for (int u = 0; u < kBnApplyInFlight; ++u)
            if (q0 + u * stride < n4) v[u] = reinterpret_cast<const float4 *>(x)[q0 + u * stride];
    }
    const int slots = bn_slots(C);
    for (int c = threadIdx.x; c < C; c += 256) {
        double t0 = 0, t1 = 0;
        double s0[8], s1[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {                   // unconditional loads (a slot that does not exist re-reads slot 0):
            const size_t o = (size_t)(k < slots ? k : 0) * 2 * C + c;      // all of them in flight together
            s0[k] = totals[o]; s1[k] = totals[o + C];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) { t0 += k < slots ? s0[k] : 0.0; t1 += k < slots ? s1[k] : 0.0; }
        const double n = (double)rows, mean = t0 / n;
        double var = t1 / n - mean * mean;              // biased batch variance
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
    if (VEC == 4) {
        while (q0 < n4) {
#pragma unroll
            for (int u = 0; u < kBnApplyInFlight; ++u) {
                const long q = q0 + u * stride;
                if (q >= n4) break;
                const int c = (int)((q * 4) % C);
                const float4 g = *reinterpret_cast<const float4 *>(gamma + c), b = *reinterpret_cast<const float4 *>(beta + c);
                const float in[4] = {v[u].x, v[u].y, v[u].z, v[u].w}, gv[4] = {g.x, g.y, g.z, g.w}, bv[4] = {b.x, b.y, b.z, b.w};
                float o[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = act_fwd(bn_affine(in[j], stat[c + j], stat[C + c + j], gv[j], bv[j]), act);
                reinterpret_cast<float4 *>(y)[q] = make_float4(o[0], o[1], o[2], o[3]);
            }
            q0 += kBnApplyInFlight * stride;
#pragma unroll
            for (int u = 0; u < kBnApplyInFlight; ++u)
                if (q0 + u * stride < n4) v[u] = reinterpret_cast<const float4 *>(x)[q0 + u * stride];
        }
    } else {
        for (long o = (long)blockIdx.x * 256 + threadIdx.x; o < total; o += stride) {
            const int c = o % C;
            y[o] = act_fwd(bn_affine(x[o], stat[c], stat[C + c], gamma[c], beta[c]), act);
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

// dbeta / dgamma of every channel from the totals into LDS (workgroup 0 publishes them), then
// dx = gamma*invstd*(dz - dbeta/n - xhat*dgamma/n)
// (y == nullptr: the activation's sign from the recomputed forward value, as in bn_partial_v4_kernel; the float4 form keeps
// kBnApplyInFlight independent float4 sets in flight per thread)
template <int VEC>
__global__ __launch_bounds__(256) void bn_bwd_fused_apply_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                                                 const float *__restrict__ dy, long total, long rows, int C,
                                                                 const float *__restrict__ gamma, const float *__restrict__ beta,
                                                                 const float *__restrict__ mean,
                                                                 const float *__restrict__ invstd, const double *__restrict__ totals,
                                                                 float *dgamma, float *dbeta, int act, float *__restrict__ dx) {
    extern __shared__ float stat[];                     // [dbeta C][dgamma C]
    const long stride = (long)gridDim.x * 256, n4 = VEC == 4 ? total / 4 : 0;      // (first batch before the totals: bn_train_apply_kernel)
    long q0 = (long)blockIdx.x * 256 + threadIdx.x;
    float4 a[kBnApplyInFlight], b[kBnApplyInFlight], d[kBnApplyInFlight];
    auto load_batch = [&]() {
#pragma unroll
        for (int u = 0; u < kBnApplyInFlight; ++u) {
            const long q = q0 + u * stride;
            if (q < n4) {
                a[u] = reinterpret_cast<const float4 *>(x)[q];
                if (y) b[u] = reinterpret_cast<const float4 *>(y)[q];
                d[u] = reinterpret_cast<const float4 *>(dy)[q];
            }
        }
    };
    if (VEC == 4) load_batch();
    const int slots = bn_slots(C);
    for (int c = threadIdx.x; c < C; c += 256) {
        double t0 = 0, t1 = 0;
        double s0[8], s1[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {                   // unconditional loads (a slot that does not exist re-reads slot 0):
            const size_t o = (size_t)(k < slots ? k : 0) * 2 * C + c;      // all of them in flight together
            s0[k] = totals[o]; s1[k] = totals[o + C];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) { t0 += k < slots ? s0[k] : 0.0; t1 += k < slots ? s1[k] : 0.0; }
        const float db = (float)t0, dg = (float)t1;
        stat[c] = db;
        stat[C + c] = dg;
        if (blockIdx.x == 0) { dbeta[c] = db; dgamma[c] = dg; }
    }
    __syncthreads();
    const float invn = 1.0f / (float)rows;
    if (VEC == 4) {
        while (q0 < n4) {
#pragma unroll
            for (int u = 0; u < kBnApplyInFlight; ++u) {
                const long q = q0 + u * stride;
                if (q >= n4) break;
                const int c = (int)((q * 4) % C);
                const float4 g = *reinterpret_cast<const float4 *>(gamma + c), m = *reinterpret_cast<const float4 *>(mean + c);
                const float4 iv = *reinterpret_cast<const float4 *>(invstd + c);
                float4 bt = make_float4(0.f, 0.f, 0.f, 0.f);
                if (!y) bt = *reinterpret_cast<const float4 *>(beta + c);
                const float xv[4] = {a[u].x, a[u].y, a[u].z, a[u].w}, yv[4] = {b[u].x, b[u].y, b[u].z, b[u].w};
                const float dv[4] = {d[u].x, d[u].y, d[u].z, d[u].w}, bv[4] = {bt.x, bt.y, bt.z, bt.w};
                const float gv[4] = {g.x, g.y, g.z, g.w}, mv[4] = {m.x, m.y, m.z, m.w}, sv[4] = {iv.x, iv.y, iv.z, iv.w};
                float o[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float out = y ? yv[j] : bn_affine(xv[j], mv[j], sv[j], gv[j], bv[j]);
                    const float dz = dv[j] * act_bwd_from_out(out, act);
                    const float xh = (xv[j] - mv[j]) * sv[j];
                    o[j] = gv[j] * sv[j] * (dz - stat[c + j] * invn - xh * stat[C + c + j] * invn);
                }
                reinterpret_cast<float4 *>(dx)[q] = make_float4(o[0], o[1], o[2], o[3]);
            }
            q0 += kBnApplyInFlight * stride;
            load_batch();
        }
    } else {
        for (long o = (long)blockIdx.x * 256 + threadIdx.x; o < total; o += stride) {
            const int c = o % C;
            const float xo = x[o];
            const float dz = dy[o] * act_bwd_from_out(y ? y[o] : bn_affine(xo, mean[c], invstd[c], gamma[c], beta[c]), act);
            const float xh = (xo - mean[c]) * invstd[c];
            dx[o] = gamma[c] * invstd[c] * (dz - stat[c] * invn - xh * stat[C + c] * invn);
        }
    }
}

static inline bool al16(const void *p) { return ((uintptr_t)p & 15) == 0; }

static int bn_splits(long rows, int C, long *rows_per_split) {
    const int cblocks = (C + 63) / 64;
    long splits = kBnBlocks / cblocks;
    if (splits < 1) splits = 1;
    long rps = (rows + splits - 1) / splits;
    if (rps < 8) rps = 8;
    splits = (rows + rps - 1) / rps;
    *rows_per_split = rps;
    return (int)splits;
}

// rows in flight per thread: 16 when a split is long enough to use them (16 row lanes x 16 = 256 rows), else 4
template <int MODE>
static void launch_partial_v4(dim3 grid, hipStream_t st, const float *x, const float *y, const float *dy, long rows, int C, long rps,
                              const float *mean, const float *invstd, const float *gamma, const float *beta, int act, double *part) {
    constexpr int UBIG = MODE == 0 ? 16 : 8;             // (MODE 1 holds two or three tensors per row)
    if (rps >= 16 * UBIG)
        hipLaunchKernelGGL((bn_partial_v4_kernel<MODE, UBIG>), grid, dim3(256), 0, st, x, y, dy, rows, C, rps, mean, invstd, gamma, beta, act, part);
    else
        hipLaunchKernelGGL((bn_partial_v4_kernel<MODE, 4>), grid, dim3(256), 0, st, x, y, dy, rows, C, rps, mean, invstd, gamma, beta, act, part);
}

int bn_slot_count(int C) { return bn_slots(C); }
int bn_stats_pass(const float *x, long rows, int C, double *totals, hipStream_t st) {
    long rps;
    const int splits = bn_splits(rows, C, &rps);
    if ((C % 4) == 0 && al16(x))
        launch_partial_v4<0>(dim3((C + 63) / 64, splits), st, x, nullptr, nullptr, rows, C, rps, nullptr, nullptr, nullptr, nullptr, 0, totals);
    else
        hipLaunchKernelGGL(bn_partial_kernel<0>, dim3((C + 63) / 64, splits), dim3(256), 0, st, x, nullptr, nullptr, rows,
                           C, rps, nullptr, nullptr, nullptr, nullptr, 0, totals);
    return check_launch("bn statistics");
}

}  // namespace mmdgan

using namespace mmdgan;

extern "C" size_t mmdgan_bn_workspace_bytes(int C) { return C < 1 ? 0 : (size_t)bn_slots(C) * 2 * C * sizeof(double); }

static int bn_fwd_train_impl(const float *x, long rows, int C, const float *gamma, const float *beta, float eps,
                             float momentum, int unbiased_moving_var, int act, float *y, float *save_mean,
                             float *save_invstd, const float *moving_mean, const float *moving_var,
                             float *new_moving_mean, float *new_moving_var, void *workspace, void *stream, bool have_totals) {
    MMDGAN_REQUIRE(x && gamma && beta && y && save_mean && save_invstd && workspace, "bn_fwd_train: null pointer");
    MMDGAN_REQUIRE(rows >= 1 && C >= 1, "bn_fwd_train: bad shape");
    MMDGAN_REQUIRE(!new_moving_mean || (moving_mean && moving_var && new_moving_var), "bn_fwd_train: moving stats");
    hipStream_t st = (hipStream_t)stream;
    long rps;
    const int splits = bn_splits(rows, C, &rps);
    double *part = (double *)workspace;
    const bool v4 = (C % 4) == 0 && al16(x) && al16(y) && al16(gamma) && al16(beta) && al16(save_mean) && al16(save_invstd);
    if (!have_totals) {
        if (zero_output(part, sizeof(double) * bn_slots(C) * 2 * C, st) != hipSuccess) return check_launch("bn_fwd_train memset");
        if (v4)
            launch_partial_v4<0>(dim3((C + 63) / 64, splits), st, x, nullptr, nullptr, rows, C, rps, nullptr, nullptr, nullptr, nullptr, 0, part);
        else
            hipLaunchKernelGGL(bn_partial_kernel<0>, dim3((C + 63) / 64, splits), dim3(256), 0, st, x, nullptr, nullptr, rows,
                               C, rps, nullptr, nullptr, nullptr, nullptr, 0, part);
    }
    const long total = rows * C;
    long blocks = ((v4 ? (total / 4 + kBnApplyInFlight - 1) / kBnApplyInFlight : total) + 255) / 256;
    if (blocks > (v4 ? kBnApplyBlocks : 4096)) blocks = v4 ? kBnApplyBlocks : 4096;
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
extern "C" int mmdgan_bn_fwd_train(const float *x, long rows, int C, const float *gamma, const float *beta, float eps,
                                   float momentum, int unbiased_moving_var, int act, float *y, float *save_mean,
                                   float *save_invstd, const float *moving_mean, const float *moving_var,
                                   float *new_moving_mean, float *new_moving_var, void *workspace, void *stream) {
    return bn_fwd_train_impl(x, rows, C, gamma, beta, eps, momentum, unbiased_moving_var, act, y, save_mean, save_invstd, moving_mean,
                             moving_var, new_moving_mean, new_moving_var, workspace, stream, false);
}
// ... with the totals of x already in `workspace` (mmdgan_conv2d_fwd_stats / _dgrad_stats produced x): normalise and update only
extern "C" int mmdgan_bn_fwd_apply(const float *x, long rows, int C, const float *gamma, const float *beta, float eps,
                                   float momentum, int unbiased_moving_var, int act, float *y, float *save_mean,
                                   float *save_invstd, const float *moving_mean, const float *moving_var,
                                   float *new_moving_mean, float *new_moving_var, void *workspace, void *stream) {
    return bn_fwd_train_impl(x, rows, C, gamma, beta, eps, momentum, unbiased_moving_var, act, y, save_mean, save_invstd, moving_mean,
                             moving_var, new_moving_mean, new_moving_var, workspace, stream, true);
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
                             const float *beta, const float *save_mean, const float *save_invstd, int act, float *dx,
                             float *dgamma, float *dbeta, void *workspace, void *stream) {
    MMDGAN_REQUIRE(x && dy && gamma && save_mean && save_invstd && dx && dgamma && dbeta && workspace, "bn_bwd: null pointer");
    MMDGAN_REQUIRE(rows >= 1 && C >= 1, "bn_bwd: bad shape");
    MMDGAN_REQUIRE(y || (beta && (act == MMDGAN_ACT_LINEAR || act == MMDGAN_ACT_RELU || act == MMDGAN_ACT_LRELU)),
                   "bn_bwd: without y the activation must be linear / relu / lrelu and beta must be given (act %d)", act);
    hipStream_t st = (hipStream_t)stream;
    long rps;
    const int splits = bn_splits(rows, C, &rps);
    double *part = (double *)workspace;
    if (zero_output(part, sizeof(double) * bn_slots(C) * 2 * C, st) != hipSuccess) return check_launch("bn_bwd memset");
    const bool v4 = (C % 4) == 0 && al16(x) && (!y || al16(y)) && al16(dy) && al16(dx) && al16(gamma) && (y || al16(beta)) &&
                    al16(save_mean) && al16(save_invstd) && al16(dgamma) && al16(dbeta);
    if (v4)
        launch_partial_v4<1>(dim3((C + 63) / 64, splits), st, x, y, dy, rows, C, rps, save_mean, save_invstd, gamma, beta, act, part);
    else
        hipLaunchKernelGGL(bn_partial_kernel<1>, dim3((C + 63) / 64, splits), dim3(256), 0, st, x, y, dy, rows, C, rps,
                           save_mean, save_invstd, gamma, beta, act, part);
    const long total = rows * C;
    long blocks = ((v4 ? (total / 4 + kBnApplyInFlight - 1) / kBnApplyInFlight : total) + 255) / 256;
    if (blocks > (v4 ? kBnApplyBlocks : 4096)) blocks = v4 ? kBnApplyBlocks : 4096;
    const size_t lds = sizeof(float) * 2 * C;
    if (v4)
        hipLaunchKernelGGL(bn_bwd_fused_apply_kernel<4>, dim3((unsigned)blocks), dim3(256), lds, st, x, y, dy, total, rows, C, gamma,
                           beta, save_mean, save_invstd, part, dgamma, dbeta, act, dx);
    else
        hipLaunchKernelGGL(bn_bwd_fused_apply_kernel<1>, dim3((unsigned)blocks), dim3(256), lds, st, x, y, dy, total, rows, C, gamma,
                           beta, save_mean, save_invstd, part, dgamma, dbeta, act, dx);
    return check_launch("bn_bwd");
}
