// Elementwise pieces of the reference's residual blocks (layer_func.py:1687-1842), NHWC fp32, gfx950.
//   ImageScaling 'avg'    :1155-1159  tf.nn.avg_pool, window = stride = f          -> resample_down (scale 1/f^2)
//   ImageScaling 'unpool' :1160-1163  four channel copies + depth_to_space = every pixel repeated f x f
//                                                                                   -> resample_up   (scale 1)
//   their gradients are each other with the other scale (sum over the window / spread over it),
//   the pre-activation of a block (:1785 _apply_activation_ on the block input, whose raw value also feeds the
//   shortcut, so it cannot ride on the producer's epilogue), and the branch sum (:1842).
// All of it is HBM-bound streaming: one thread per 16-byte channel quad, rows of C contiguous floats, so every
// load and store is a full coalesced line; nothing is staged through LDS because nothing is reused.
#include "common.h"

namespace mmdgan {

template <int VEC>
struct Pack;
template <>
struct Pack<4> {
    using T = float4;
    static __device__ __forceinline__ T zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
    static __device__ __forceinline__ void acc(T &a, const T &b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
    static __device__ __forceinline__ T scaled(const T &a, float s) { return make_float4(a.x * s, a.y * s, a.z * s, a.w * s); }
};
template <>
struct Pack<1> {
    using T = float;
    static __device__ __forceinline__ T zero() { return 0.f; }
    static __device__ __forceinline__ void acc(T &a, const T &b) { a += b; }
    static __device__ __forceinline__ T scaled(const T &a, float s) { return a * s; }
};

// y[n, p, q, :] = scale * sum_{i,j < f} x[n, p*f + i, q*f + j, :]      (x is [N, P*f, Q*f, C])
template <int VEC>
__global__ __launch_bounds__(256) void resample_down_kernel(const float *__restrict__ x, float *__restrict__ y, long total,
                                                            int P, int Q, int CV, int f, float scale, int accumulate) {
    using V = typename Pack<VEC>::T;
    const V *xv = reinterpret_cast<const V *>(x);
    V *yv = reinterpret_cast<V *>(y);
    const long stride = (long)gridDim.x * 256;
    const int W = Q * f;
    for (long o = (long)blockIdx.x * 256 + threadIdx.x; o < total; o += stride) {     // o = ((n*P + p)*Q + q)*CV + c
        const int c = (int)(o % CV);
        long t = o / CV;
        const int q = (int)(t % Q);
        t /= Q;
        const int p = (int)(t % P);
        const long n = t / P;
        V a = Pack<VEC>::zero();
        // the window is summed row by row, left to right: the order tf.nn.avg_pool's reference kernel uses
        for (int i = 0; i < f; ++i)
            for (int j = 0; j < f; ++j)
                Pack<VEC>::acc(a, xv[((n * (P * f) + p * f + i) * W + q * f + j) * CV + c]);
        a = Pack<VEC>::scaled(a, scale);
        if (accumulate) Pack<VEC>::acc(a, yv[o]);
        yv[o] = a;
    }
}

// y[n, h, w, :] = scale * x[n, h / f, w / f, :]                        (y is [N, P*f, Q*f, C])
template <int VEC>
__global__ __launch_bounds__(256) void resample_up_kernel(const float *__restrict__ x, float *__restrict__ y, long total,
                                                          int P, int Q, int CV, int f, float scale, int accumulate) {
    using V = typename Pack<VEC>::T;
    const V *xv = reinterpret_cast<const V *>(x);
    V *yv = reinterpret_cast<V *>(y);
    const long stride = (long)gridDim.x * 256;
    const int H = P * f, W = Q * f;
    for (long o = (long)blockIdx.x * 256 + threadIdx.x; o < total; o += stride) {     // o indexes y
        const int c = (int)(o % CV);
        long t = o / CV;
        const int w = (int)(t % W);
        t /= W;
        const int h = (int)(t % H);
        const long n = t / H;
        V a = Pack<VEC>::scaled(xv[((n * P + h / f) * Q + w / f) * CV + c], scale);
        if (accumulate) Pack<VEC>::acc(a, yv[o]);
        yv[o] = a;
    }
}

// periodic shuffling (layer_func.py:197-244: tf.depth_to_space / tf.space_to_depth, block-major channel order):
//   big[n, h*r + i, w*r + j, c] <-> small[n, h, w, (i*r + j)*C + c],  big [N, H*r, W*r, C], small [N, H, W, r*r*C].
// TO_BIG = depth_to_space (the up-sampling direction), else space_to_depth; each is the other's gradient.
// Runs of C contiguous floats move as they are: thread = one 16-byte (or 4-byte) element of the destination.
template <int VEC, bool TO_BIG>
__global__ __launch_bounds__(256) void shuffle_kernel(const float *__restrict__ src, float *__restrict__ dst, long total, int H,
                                                      int W, int CV, int r) {
    using V = typename Pack<VEC>::T;
    const V *sv = reinterpret_cast<const V *>(src);
    V *dv = reinterpret_cast<V *>(dst);
    const long stride = (long)gridDim.x * 256;
    for (long o = (long)blockIdx.x * 256 + threadIdx.x; o < total; o += stride) {     // o indexes the destination
        long t = o;
        if (TO_BIG) {                       // dst = big: ((n*H*r + y)*W*r + x)*CV + c
            const int c = (int)(t % CV); t /= CV;
            const int x = (int)(t % (W * r)); t /= (W * r);
            const int y = (int)(t % (H * r));
            const long n = t / (H * r);
            dv[o] = sv[((n * H + y / r) * W + x / r) * ((long)r * r * CV) + ((y % r) * r + x % r) * CV + c];
        } else {                            // dst = small: ((n*H + h)*W + w)*(r*r*CV) + (i*r + j)*CV + c
            const int c = (int)(t % CV); t /= CV;
            const int ij = (int)(t % (r * r)); t /= (r * r);
            const int w = (int)(t % W); t /= W;
            const int h = (int)(t % H);
            const long n = t / H;
            dv[o] = sv[((n * (H * r) + h * r + ij / r) * (W * r) + w * r + ij % r) * CV + c];
        }
    }
}

// ImageScaling 'bil' (layer_func.py:1128-1137): tf.image.resize_bilinear(align_corners=True).  Source coordinate of
// output row y: y * (H - 1) / (OH - 1) in fp32 as TF computes it; rows floor(.) and min(floor(.) + 1, H - 1) blended
// with the fractional part, columns likewise.  GRAD: the adjoint - every output gradient is scattered to its four
// sources with the same weights (fp32 atomics into a zeroed dx; not a hot path).
template <bool GRAD>
__global__ __launch_bounds__(256) void bilinear_kernel(const float *__restrict__ src, float *__restrict__ dst, long total, int H,
                                                       int W, int C, int OH, int OW, float sy, float sx) {
    const long stride = (long)gridDim.x * 256;
    for (long o = (long)blockIdx.x * 256 + threadIdx.x; o < total; o += stride) {     // o indexes the OH x OW side
        const int c = (int)(o % C);
        long t = o / C;
        const int x = (int)(t % OW);
        t /= OW;
        const int y = (int)(t % OH);
        const long n = t / OH;
        const float fy = (float)y * sy, fx = (float)x * sx;
        const int y0 = min((int)floorf(fy), H - 1), x0 = min((int)floorf(fx), W - 1);
        const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
        const float wy = fy - (float)y0, wx = fx - (float)x0;
        const long b = n * H;
        const long i00 = ((b + y0) * W + x0) * C + c, i01 = ((b + y0) * W + x1) * C + c;
        const long i10 = ((b + y1) * W + x0) * C + c, i11 = ((b + y1) * W + x1) * C + c;
        if (!GRAD) {
            const float top = src[i00] + (src[i01] - src[i00]) * wx;          // TF: top_left + (top_right - top_left) * x_lerp
            const float bot = src[i10] + (src[i11] - src[i10]) * wx;
            dst[o] = top + (bot - top) * wy;
        } else {
            const float g = src[o];
            atomicAdd(dst + i00, g * (1.f - wy) * (1.f - wx));
            atomicAdd(dst + i01, g * (1.f - wy) * wx);
            atomicAdd(dst + i10, g * wy * (1.f - wx));
            atomicAdd(dst + i11, g * wy * wx);
        }
    }
}

// ImageScaling 'bic' (layer_func.py:1138-1147): tf.image.resize_bicubic(align_corners=True) as TF 1.x computes it (legacy
// sampling, resize_bicubic_op.cc): source coordinate y * (H - 1) / (OH - 1) in fp32; its fractional part is rounded to a
// 1/1024 grid (lrintf(delta * 1024), the kernel's coefficient table) and the four taps floor-1 .. floor+2, clamped to the
// image, are weighted with the Keys cubic for A = -0.75 evaluated in double and rounded to fp32 (the table's entries):
//   w(x) = ((A+2)x - (A+3))x^2 + 1 for |x| <= 1,   ((Ax - 5A)x + 8A)x - 4A for 1 < |x| < 2.
// Rows are interpolated along x first, then along y (Interpolate1D order).  GRAD: the adjoint scatter (fp32 atomics).
__device__ __forceinline__ void bicubic_taps(int o, float scale, int limit, int idx[4], float wgt[4]) {
    const float loc = scale * (float)o;
    const int in = (int)loc;                                   // int64 cast of a non-negative float
    const float delta = loc - (float)in;
    const int off = (int)lrintf(delta * 1024.f);
    const double A = -0.75;
    auto near = [&](int i) { const double x = (double)((float)i / 1024.f); return (float)(((A + 2) * x - (A + 3)) * x * x + 1); };
    auto far = [&](int i) { const double x = (double)((float)i / 1024.f + 1.f); return (float)(((A * x - 5 * A) * x + 8 * A) * x - 4 * A); };
    wgt[0] = far(off); wgt[1] = near(off); wgt[2] = near(1024 - off); wgt[3] = far(1024 - off);
#pragma unroll
    for (int t = 0; t < 4; ++t) idx[t] = min(limit - 1, max(0, in - 1 + t));
}

template <bool GRAD>
__global__ __launch_bounds__(256) void bicubic_kernel(const float *__restrict__ src, float *__restrict__ dst, long total, int H,
                                                      int W, int C, int OH, int OW, float sy, float sx) {
    const long stride = (long)gridDim.x * 256;
    for (long o = (long)blockIdx.x * 256 + threadIdx.x; o < total; o += stride) {     // o indexes the OH x OW side
        const int c = (int)(o % C);
        long t = o / C;
        const int x = (int)(t % OW);
        t /= OW;
        const int y = (int)(t % OH);
        const long n = t / OH;
        int iy[4], ix[4];
        float wy[4], wx[4];
        bicubic_taps(y, sy, H, iy, wy);
        bicubic_taps(x, sx, W, ix, wx);
        if (!GRAD) {
            float rows[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const float *r = src + ((n * H + iy[a]) * W) * C + c;
                rows[a] = r[(long)ix[0] * C] * wx[0] + r[(long)ix[1] * C] * wx[1] + r[(long)ix[2] * C] * wx[2] + r[(long)ix[3] * C] * wx[3];
            }
            dst[o] = rows[0] * wy[0] + rows[1] * wy[1] + rows[2] * wy[2] + rows[3] * wy[3];
        } else {
            const float g = src[o];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) atomicAdd(dst + ((n * H + iy[a]) * W + ix[b]) * C + c, g * wy[a] * wx[b]);
        }
    }
}

// ImageScaling 'max' (layer_func.py:1149-1153): tf.nn.max_pool, window = stride = f.  Windows do not overlap, so the
// gradient needs no atomics: the thread of a window writes dy to its first maximum (row-major order, the element a
// strict '>' scan keeps - what TF's and the oracle's max-pool gradients pick) and zero to the rest.
template <bool GRAD>
__global__ __launch_bounds__(256) void maxpool_kernel(const float *__restrict__ x, const float *__restrict__ dy,
                                                      float *__restrict__ out, long total, int P, int Q, int C, int f) {
    const long stride = (long)gridDim.x * 256;
    const int W = Q * f;
    for (long o = (long)blockIdx.x * 256 + threadIdx.x; o < total; o += stride) {     // o = ((n*P + p)*Q + q)*C + c
        const int c = (int)(o % C);
        long t = o / C;
        const int q = (int)(t % Q);
        t /= Q;
        const int p = (int)(t % P);
        const long n = t / P;
        float best = -INFINITY;
        int arg = 0;
        for (int i = 0; i < f; ++i)
            for (int j = 0; j < f; ++j) {
                const float v = x[((n * (P * f) + p * f + i) * W + q * f + j) * C + c];
                if (v > best || (i == 0 && j == 0)) { best = v; arg = i * f + j; }
            }
        if (!GRAD) {
            out[o] = best;
        } else {
            const float g = dy[o];
            for (int i = 0; i < f; ++i)
                for (int j = 0; j < f; ++j)
                    out[((n * (P * f) + p * f + i) * W + q * f + j) * C + c] = (i * f + j == arg) ? g : 0.f;
        }
    }
}

// A residual block's scaling op and the 3x3 conv next to it are ONE linear map with a 4x4 stride-2 kernel:
//   MODE 0  avgpool/2 o conv3x3(W):  y[p] = 1/4 sum_{i<2} sum_a W[a] x[2p + i + a - 1]  = conv4x4/2 with  1/4 * (1_2 (*) W)
//   MODE 1  conv3x3(W) o unpool x2:  y[2p] = W[0] x[p-1] + (W[1]+W[2]) x[p],  y[2p+1] = (W[0]+W[1]) x[p] + W[2] x[p+1]
//           = the input-gradient form (transposed conv 4x4/2) with the flipped 1_2 (*) W, kernel stored [4,4,K,C]
// (1_2 (*) W = full convolution with [1,1], per dimension [W0, W0+W1, W1+W2, W2]).  4 taps per pixel instead of 9 and no
// up-sampled / un-pooled tensor in HBM.  GRAD: the adjoint map (gradient w.r.t. the 4x4 kernel -> gradient w.r.t. W).
template <int MODE, bool GRAD>
__global__ __launch_bounds__(256) void compose_kernel(const float *__restrict__ src, float *__restrict__ dst, int C, int K) {
    const long total = (long)C * K;
    const long stride = (long)gridDim.x * 256;
    for (long o = (long)blockIdx.x * 256 + threadIdx.x; o < total; o += stride) {
        // neighbouring threads walk the 4x4 kernel's innermost dimension (16 of the 25 accesses per element are on that side):
        // MODE 0 [4,4,C,K]: o = c*K + k;  MODE 1 [4,4,K,C]: o = k*C + c (the 9 accesses of W are strided there, and cached)
        const int c = MODE == 0 ? (int)(o / K) : (int)(o % C), k = MODE == 0 ? (int)(o - (long)c * K) : (int)(o / C);
        const float s = MODE == 0 ? 0.25f : 1.f;
        // F[r][a] = 1 where tap a of W contributes to tap r of the 4-wide kernel
        //   MODE 0: r = a + i (i = 0,1);  MODE 1: flipped: r = 3 - (a + i)
        auto idx4 = [&](int r, int t) -> long {
            return MODE == 0 ? ((long)(r * 4 + t) * C + c) * K + k : ((long)(r * 4 + t) * K + k) * C + c;
        };
        if (!GRAD) {
            float w[3][3];
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) w[a][b] = src[((long)(a * 3 + b) * C + c) * K + k];
            float out[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int t = 0; t < 4; ++t) out[r][t] = 0.f;
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int b = 0; b < 3; ++b)
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const int r = MODE == 0 ? a + i : 3 - (a + i), t = MODE == 0 ? b + j : 3 - (b + j);
                            out[r][t] += w[a][b];
                        }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int t = 0; t < 4; ++t) dst[idx4(r, t)] = out[r][t] * s;
        } else {
            float g[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int t = 0; t < 4; ++t) g[r][t] = src[idx4(r, t)];
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) {
                    float acc = 0.f;
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const int r = MODE == 0 ? a + i : 3 - (a + i), t = MODE == 0 ? b + j : 3 - (b + j);
                            acc += g[r][t];
                        }
                    dst[((long)(a * 3 + b) * C + c) * K + k] = acc * s;
                }
        }
    }
}

__global__ __launch_bounds__(256) void act_fwd_kernel(const float *__restrict__ x, float *__restrict__ y, long n, int act) {
    const long stride = (long)gridDim.x * 256;
    for (long o = (long)blockIdx.x * 256 + threadIdx.x; o < n; o += stride) y[o] = act_fwd(x[o], act);
}
// dx (+)= dy * act'(.), the derivative taken from the activation's OUTPUT y
__global__ __launch_bounds__(256) void act_bwd_kernel(const float *__restrict__ dy, const float *__restrict__ y,
                                                      float *__restrict__ dx, long n, int act, int accumulate) {
    const long stride = (long)gridDim.x * 256;
    for (long o = (long)blockIdx.x * 256 + threadIdx.x; o < n; o += stride) {
        const float g = dy[o] * act_bwd_from_out(y[o], act);
        dx[o] = accumulate ? dx[o] + g : g;
    }
}
__global__ __launch_bounds__(256) void axpby_kernel(const float *__restrict__ a, float alpha, const float *__restrict__ b,
                                                    float beta, float *__restrict__ out, long n) {
    const long stride = (long)gridDim.x * 256;
    for (long o = (long)blockIdx.x * 256 + threadIdx.x; o < n; o += stride) out[o] = alpha * a[o] + beta * b[o];
}

// ---- the 'padding' / 'dilation' keys of a conv layer as compositions around the 'SAME' stride-1 kernels -------------------
// strided slice: dst[n][p][q][c] = src[n][off + p*step][off + q*step][c]; adjoint = 1: src is the small tensor, dst the
// large one - every element of dst is written (the slice's positions get their value, the others zero): no zeroing, no atomics
__global__ __launch_bounds__(256) void strided_slice_kernel(const float *__restrict__ src, float *__restrict__ dst, int N, int H,
                                                            int W, int C, int off, int step, int P, int Q, int adjoint) {
    const long total = adjoint ? (long)N * H * W * C : (long)N * P * Q * C;
    const long stride = (long)gridDim.x * 256;
    for (long o = (long)blockIdx.x * 256 + threadIdx.x; o < total; o += stride) {
        const int c = (int)(o % C);
        long t = o / C;
        if (!adjoint) {
            const int q = (int)(t % Q);
            t /= Q;
            const int p = (int)(t % P), n = (int)(t / P);
            dst[o] = src[(((long)n * H + off + p * step) * W + off + q * step) * C + c];
        } else {
            const int w = (int)(t % W);
            t /= W;
            const int h = (int)(t % H), n = (int)(t / H);
            const int hp = h - off, wq = w - off;
            const bool hit = hp >= 0 && wq >= 0 && hp % step == 0 && wq % step == 0 && hp / step < P && wq / step < Q;
            dst[o] = hit ? src[(((long)n * P + hp / step) * Q + wq / step) * C + c] : 0.f;
        }
    }
}
// space <-> batch for a dilated conv (dilation d, stride 1): image n splits into d*d phase images of ceil(H/d) x ceil(W/d)
// (zero rows / columns where the size is no multiple of d - they act as the 'SAME' padding of the phase conv):
//   to_batch:  dst[(n*d + ph)*d + pw][hq][wq][c] = src[n][hq*d + ph][wq*d + pw][c]  (0 beyond the image)
//   from batch: dst[n][h][w][c] = src[(n*d + h%d)*d + w%d][h/d][w/d][c]             - each the other's adjoint
__global__ __launch_bounds__(256) void space_batch_kernel(const float *__restrict__ src, float *__restrict__ dst, int N, int H,
                                                          int W, int C, int d, int to_batch) {
    const int Hd = (H + d - 1) / d, Wd = (W + d - 1) / d;
    const long total = to_batch ? (long)N * d * d * Hd * Wd * C : (long)N * H * W * C;
    const long stride = (long)gridDim.x * 256;
    for (long o = (long)blockIdx.x * 256 + threadIdx.x; o < total; o += stride) {
        const int c = (int)(o % C);
        long t = o / C;
        if (to_batch) {
            const int wq = (int)(t % Wd);
            t /= Wd;
            const int hq = (int)(t % Hd);
            t /= Hd;
            const int pw = (int)(t % d);
            t /= d;
            const int ph = (int)(t % d), n = (int)(t / d);
            const int h = hq * d + ph, w = wq * d + pw;
            dst[o] = (h < H && w < W) ? src[(((long)n * H + h) * W + w) * C + c] : 0.f;
        } else {
            const int w = (int)(t % W);
            t /= W;
            const int h = (int)(t % H), n = (int)(t / H);
            dst[o] = src[((((long)n * d + h % d) * d + w % d) * Hd + h / d) * Wd * C + (long)(w / d) * C + c];
        }
    }
}

static inline int grid_of(long n) {
    long b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}
static inline bool al16(const void *p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace mmdgan

using namespace mmdgan;

extern "C" int mmdgan_resample_down(const float *x, float *y, int N, int P, int Q, int C, int factor, float scale,
                                    int accumulate, void *stream) {
    MMDGAN_REQUIRE(x && y && N >= 1 && P >= 1 && Q >= 1 && C >= 1 && factor >= 1, "resample_down: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    if (C % 4 == 0 && al16(x) && al16(y)) {
        const long total = (long)N * P * Q * (C / 4);
        hipLaunchKernelGGL(resample_down_kernel<4>, dim3(grid_of(total)), dim3(256), 0, st, x, y, total, P, Q, C / 4, factor, scale, accumulate);
    } else {
        const long total = (long)N * P * Q * C;
        hipLaunchKernelGGL(resample_down_kernel<1>, dim3(grid_of(total)), dim3(256), 0, st, x, y, total, P, Q, C, factor, scale, accumulate);
    }
    return check_launch("resample_down");
}

extern "C" int mmdgan_resample_up(const float *x, float *y, int N, int P, int Q, int C, int factor, float scale,
                                  int accumulate, void *stream) {
    MMDGAN_REQUIRE(x && y && N >= 1 && P >= 1 && Q >= 1 && C >= 1 && factor >= 1, "resample_up: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    if (C % 4 == 0 && al16(x) && al16(y)) {
        const long total = (long)N * P * factor * Q * factor * (C / 4);
        hipLaunchKernelGGL(resample_up_kernel<4>, dim3(grid_of(total)), dim3(256), 0, st, x, y, total, P, Q, C / 4, factor, scale, accumulate);
    } else {
        const long total = (long)N * P * factor * Q * factor * C;
        hipLaunchKernelGGL(resample_up_kernel<1>, dim3(grid_of(total)), dim3(256), 0, st, x, y, total, P, Q, C, factor, scale, accumulate);
    }
    return check_launch("resample_up");
}

extern "C" int mmdgan_strided_slice(const float *src, float *dst, int N, int H, int W, int C, int off, int step, int P, int Q,
                                    int adjoint, void *stream) {
    MMDGAN_REQUIRE(src && dst && N >= 1 && H >= 1 && W >= 1 && C >= 1 && off >= 0 && step >= 1 && P >= 1 && Q >= 1 &&
                   off + (P - 1) * step < H && off + (Q - 1) * step < W, "strided_slice: bad arguments");
    const long total = adjoint ? (long)N * H * W * C : (long)N * P * Q * C;
    hipLaunchKernelGGL(strided_slice_kernel, dim3(grid_of(total)), dim3(256), 0, (hipStream_t)stream, src, dst, N, H, W, C, off, step,
                       P, Q, adjoint);
    return check_launch("strided_slice");
}

extern "C" int mmdgan_space_batch(const float *src, float *dst, int N, int H, int W, int C, int dilation, int to_batch,
                                  void *stream) {
    MMDGAN_REQUIRE(src && dst && N >= 1 && H >= 1 && W >= 1 && C >= 1 && dilation >= 1, "space_batch: bad arguments");
    const int Hd = (H + dilation - 1) / dilation, Wd = (W + dilation - 1) / dilation;
    const long total = to_batch ? (long)N * dilation * dilation * Hd * Wd * C : (long)N * H * W * C;
    hipLaunchKernelGGL(space_batch_kernel, dim3(grid_of(total)), dim3(256), 0, (hipStream_t)stream, src, dst, N, H, W, C, dilation,
                       to_batch);
    return check_launch("space_batch");
}

extern "C" int mmdgan_act_fwd(const float *x, float *y, long n, int act, void *stream) {
    MMDGAN_REQUIRE(x && y && n >= 1 && act >= MMDGAN_ACT_LINEAR && act <= MMDGAN_ACT_TANH, "act_fwd: bad arguments");
    hipLaunchKernelGGL(act_fwd_kernel, dim3(grid_of(n)), dim3(256), 0, (hipStream_t)stream, x, y, n, act);
    return check_launch("act_fwd");
}

extern "C" int mmdgan_act_bwd(const float *dy, const float *y, float *dx, long n, int act, int accumulate, void *stream) {
    MMDGAN_REQUIRE(dy && y && dx && n >= 1 && act >= MMDGAN_ACT_LINEAR && act <= MMDGAN_ACT_TANH, "act_bwd: bad arguments");
    hipLaunchKernelGGL(act_bwd_kernel, dim3(grid_of(n)), dim3(256), 0, (hipStream_t)stream, dy, y, dx, n, act, accumulate);
    return check_launch("act_bwd");
}

extern "C" int mmdgan_axpby(const float *a, float alpha, const float *b, float beta, float *out, long n, void *stream) {
    MMDGAN_REQUIRE(a && b && out && n >= 1, "axpby: bad arguments");
    hipLaunchKernelGGL(axpby_kernel, dim3(grid_of(n)), dim3(256), 0, (hipStream_t)stream, a, alpha, b, beta, out, n);
    return check_launch("axpby");
}

extern "C" int mmdgan_periodic_shuffle(const float *src, float *dst, int N, int H, int W, int C, int factor, int to_big,
                                       void *stream) {
    MMDGAN_REQUIRE(src && dst && N >= 1 && H >= 1 && W >= 1 && C >= 1 && factor >= 1, "periodic_shuffle: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const bool v4 = C % 4 == 0 && al16(src) && al16(dst);
    const int cv = v4 ? C / 4 : C;
    const long total = (long)N * H * W * factor * factor * cv;
    const dim3 g(grid_of(total));
    if (v4) {
        if (to_big) hipLaunchKernelGGL((shuffle_kernel<4, true>), g, dim3(256), 0, st, src, dst, total, H, W, cv, factor);
        else hipLaunchKernelGGL((shuffle_kernel<4, false>), g, dim3(256), 0, st, src, dst, total, H, W, cv, factor);
    } else {
        if (to_big) hipLaunchKernelGGL((shuffle_kernel<1, true>), g, dim3(256), 0, st, src, dst, total, H, W, cv, factor);
        else hipLaunchKernelGGL((shuffle_kernel<1, false>), g, dim3(256), 0, st, src, dst, total, H, W, cv, factor);
    }
    return check_launch("periodic_shuffle");
}

extern "C" int mmdgan_bilinear_resize(const float *src, float *dst, int N, int H, int W, int C, int OH, int OW, int grad,
                                      void *stream) {
    MMDGAN_REQUIRE(src && dst && N >= 1 && H >= 1 && W >= 1 && C >= 1 && OH >= 1 && OW >= 1, "bilinear_resize: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const float sy = OH > 1 ? (float)(H - 1) / (float)(OH - 1) : 0.f, sx = OW > 1 ? (float)(W - 1) / (float)(OW - 1) : 0.f;
    const long total = (long)N * OH * OW * C;
    if (grad) {                              // src = dy [N,OH,OW,C], dst = dx [N,H,W,C]
        if (zero_output(dst, sizeof(float) * (size_t)N * H * W * C, st) != hipSuccess) return check_launch("bilinear_resize memset");
        hipLaunchKernelGGL(bilinear_kernel<true>, dim3(grid_of(total)), dim3(256), 0, st, src, dst, total, H, W, C, OH, OW, sy, sx);
    } else {
        hipLaunchKernelGGL(bilinear_kernel<false>, dim3(grid_of(total)), dim3(256), 0, st, src, dst, total, H, W, C, OH, OW, sy, sx);
    }
    return check_launch("bilinear_resize");
}

extern "C" int mmdgan_bicubic_resize(const float *src, float *dst, int N, int H, int W, int C, int OH, int OW, int grad,
                                     void *stream) {
    MMDGAN_REQUIRE(src && dst && N >= 1 && H >= 1 && W >= 1 && C >= 1 && OH >= 1 && OW >= 1, "bicubic_resize: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const float sy = OH > 1 ? (float)(H - 1) / (float)(OH - 1) : (float)H / (float)OH;      // CalculateResizeScale, align_corners
    const float sx = OW > 1 ? (float)(W - 1) / (float)(OW - 1) : (float)W / (float)OW;
    const long total = (long)N * OH * OW * C;
    if (grad) {                              // src = dy [N,OH,OW,C], dst = dx [N,H,W,C]
        if (zero_output(dst, sizeof(float) * (size_t)N * H * W * C, st) != hipSuccess) return check_launch("bicubic_resize memset");
        hipLaunchKernelGGL(bicubic_kernel<true>, dim3(grid_of(total)), dim3(256), 0, st, src, dst, total, H, W, C, OH, OW, sy, sx);
    } else {
        hipLaunchKernelGGL(bicubic_kernel<false>, dim3(grid_of(total)), dim3(256), 0, st, src, dst, total, H, W, C, OH, OW, sy, sx);
    }
    return check_launch("bicubic_resize");
}

extern "C" int mmdgan_max_pool(const float *x, const float *dy, float *out, int N, int P, int Q, int C, int factor,
                               void *stream) {
    MMDGAN_REQUIRE(x && out && N >= 1 && P >= 1 && Q >= 1 && C >= 1 && factor >= 1, "max_pool: bad arguments");
    const long total = (long)N * P * Q * C;
    if (dy) hipLaunchKernelGGL(maxpool_kernel<true>, dim3(grid_of(total)), dim3(256), 0, (hipStream_t)stream, x, dy, out, total, P, Q, C, factor);
    else hipLaunchKernelGGL(maxpool_kernel<false>, dim3(grid_of(total)), dim3(256), 0, (hipStream_t)stream, x, dy, out, total, P, Q, C, factor);
    return check_launch("max_pool");
}

extern "C" int mmdgan_compose_scaled_conv(const float *src, float *dst, int C, int K, int mode, int grad, void *stream) {
    MMDGAN_REQUIRE(src && dst && C >= 1 && K >= 1 && (mode == 0 || mode == 1), "compose_scaled_conv: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const dim3 g(grid_of((long)C * K));
    if (mode == 0) {
        if (grad) hipLaunchKernelGGL((compose_kernel<0, true>), g, dim3(256), 0, st, src, dst, C, K);
        else hipLaunchKernelGGL((compose_kernel<0, false>), g, dim3(256), 0, st, src, dst, C, K);
    } else {
        if (grad) hipLaunchKernelGGL((compose_kernel<1, true>), g, dim3(256), 0, st, src, dst, C, K);
        else hipLaunchKernelGGL((compose_kernel<1, false>), g, dim3(256), 0, st, src, dst, C, K);
    }
    return check_launch("compose_scaled_conv");
}
