// HBM-bound helpers: column sums, dot, spectral-norm norm/scale/fix-up steps, TF-Adam, layout seam.
// All are streaming kernels: float4 accesses where the layout allows, grid-stride loops capped at
// ~2048 blocks, double accumulation for reductions.
#include "common.h"
#include <stdint.h>

namespace mmdgan {

// ---------------------------------------------------------------------------------------------
// out[c] += sum over a row chunk of x[r,c]; out is zeroed by a memset node first.
// block = 64 columns x 4 row lanes.
__global__ __launch_bounds__(256) void colsum_kernel(const float *__restrict__ x, long rows, int cols,
                                                     long rows_per_block, float *out) {
    __shared__ double red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
    const long r0 = (long)blockIdx.y * rows_per_block;
    long r1 = r0 + rows_per_block;
    if (r1 > rows) r1 = rows;
    double acc = 0;
    if (c < cols)
        for (long r = r0 + rl; r < r1; r += 4) acc += (double)x[r * cols + c];
    red[rl][threadIdx.x & 63] = acc;
    __syncthreads();
    if (rl == 0 && c < cols) {
        double t = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        atomicAdd(out + c, (float)t);
    }
}

// cols % 4 == 0: float4 loads, 16 row lanes x 4 rows in flight per thread
__global__ __launch_bounds__(256) void colsum_v4_kernel(const float *__restrict__ x, long rows, int cols,
                                                        long rows_per_block, float *out) {
    __shared__ double red[16][65];
    const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int c = blockIdx.x * 64 + cl * 4;
    const long r0 = (long)blockIdx.y * rows_per_block;
    long r1 = r0 + rows_per_block;
    if (r1 > rows) r1 = rows;
    double acc[4] = {0, 0, 0, 0};
    if (c < cols)
        for (long r = r0 + rl; r < r1; r += 64) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long rr = r + u * 16;
                v[u] = *reinterpret_cast<const float4 *>(x + (rr < r1 ? rr : r) * cols + c);
                if (rr >= r1) v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                acc[0] += (double)v[u].x; acc[1] += (double)v[u].y; acc[2] += (double)v[u].z; acc[3] += (double)v[u].w;
            }
        }
#pragma unroll
    for (int j = 0; j < 4; ++j) red[rl][cl * 4 + j] = acc[j];
    __syncthreads();
    if (threadIdx.x < 64 && blockIdx.x * 64 + threadIdx.x < cols) {
        double t = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][threadIdx.x];
        atomicAdd(out + blockIdx.x * 64 + threadIdx.x, (float)t);
    }
}

__global__ __launch_bounds__(256) void dot_kernel(const float *__restrict__ a, const float *__restrict__ b, long n,
                                                  float *out) {
    __shared__ double red[4];
    double acc = 0;
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) acc += (double)a[i] * (double)b[i];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, (float)(red[0] + red[1] + red[2] + red[3]));
}

// ---------------------------------------------------------------------------------------------
// ||v||_2 and v/(||v||+1e-10) in ONE block (SN vectors are <= 64K elements on the DCGAN nets): deterministic, no
// inter-block traffic.  math_func.py:651,659.
__global__ __launch_bounds__(1024) void sn_norm_kernel(const float *__restrict__ v, long n, float *out_norm,
                                                       float *vn, float act_k, float *scale_out) {
    __shared__ double red[16];
    __shared__ float s_norm;
    double acc = 0;
    // 16-byte loads when the vector allows it (the residual nets' vectors reach 256K elements: 64 instead of 256
    // trips per thread); the products are exact in double either way, only the order of the sum changes
    const bool vec = (n & 3) == 0 && (((uintptr_t)v | (uintptr_t)vn) & 15) == 0;
    if (vec) {
        const float4 *v4 = reinterpret_cast<const float4 *>(v);
        for (long i = threadIdx.x; i < (n >> 2); i += 1024) {
            const float4 q = v4[i];
            acc += (double)q.x * (double)q.x + (double)q.y * (double)q.y + (double)q.z * (double)q.z + (double)q.w * (double)q.w;
        }
    } else {
        for (long i = threadIdx.x; i < n; i += 1024) acc += (double)v[i] * (double)v[i];
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0;
        for (int w = 0; w < 16; ++w) t += red[w];
        s_norm = (float)sqrt(t);
        if (out_norm) out_norm[0] = s_norm;
        if (scale_out) scale_out[0] = act_k / s_norm;          // layer_func.py:886-887
    }
    __syncthreads();
    if (vn) {
        const float inv = 1.0f / (s_norm + kEpsi);
        if (vec) {
            const float4 *v4 = reinterpret_cast<const float4 *>(v);
            float4 *o4 = reinterpret_cast<float4 *>(vn);
            for (long i = threadIdx.x; i < (n >> 2); i += 1024) {
                const float4 q = v4[i];
                o4[i] = make_float4(q.x * inv, q.y * inv, q.z * inv, q.w * inv);
            }
        } else {
            for (long i = threadIdx.x; i < n; i += 1024) vn[i] = v[i] * inv;
        }
    }
}

__global__ void sn_scale_kernel(const float *sigma, float act_k, float *out) { out[0] = act_k / sigma[0]; }

// dw = scale*G - (scale/sigma)*<G,W>*dsigma_dw   
__global__ __launch_bounds__(256) void sn_fixup_kernel(float *__restrict__ g, const float *__restrict__ ds,
                                                       const float *dot, const float *sigma, const float *scale,
                                                       long n) {
    const float sc = scale[0];
    const float c2 = sc / sigma[0] * dot[0];
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) g[i] = sc * g[i] - c2 * ds[i];
}

// ---------------------------------------------------------------------------------------------
// TF-Adam (graph_func.py:525-526 / tf.train.AdamOptimizer): epsilon OUTSIDE the bias correction.
// one thread: (optionally) advance the device step counter and derive lr_t in double, so the whole
// update is hipGraph-capturable (no host-computed value changes between replays)
__global__ void adam_prepare_kernel(int *step_counter, int step_host, float lr, float b1, float b2, float *lr_t_out) {
    int t = step_host;
    if (step_counter) { t = step_counter[0] + 1; step_counter[0] = t; }
    lr_t_out[0] = (float)((double)lr * sqrt(1.0 - pow((double)b2, (double)t)) / (1.0 - pow((double)b1, (double)t)));
}

__global__ __launch_bounds__(256) void adam_kernel(const void *const *ptrs, const long *sizes, const float *lr_t_ptr,
                                                   float b1, float b2, float eps, float gscale) {
    const float lr_t = lr_t_ptr[0];
    const int t = blockIdx.y;
    const long n = sizes[t];
    float *p = (float *)ptrs[4 * t];
    const float *g = (const float *)ptrs[4 * t + 1];
    float *m = (float *)ptrs[4 * t + 2], *v = (float *)ptrs[4 * t + 3];
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const float gi = g[i] * gscale;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] = p[i] - lr_t * mi / (sqrtf(vi) + eps);
    }
}

// The same update over ONE flat arena (p / g / m / v share offsets) cut into segments, with the spectral-norm fix-up of a
// segment's gradient folded in: a spectrally normalised kernel W enters the net as W * scale (scale = act_k / sigma(W)), so
//     dL/dW = scale * G - (scale / sigma) * <G, W> * dsigma/dW          (G = the gradient w.r.t. the scaled kernel)
// which is linear in G: the raw G stays in the gradient arena (and is what a data-parallel all-reduce sums, together with
// the scalar <G, W>), and Adam reads the effective gradient here - no pass over every kernel for the fix-up.
// Work is dealt in blocks of 1024 elements through a table (segment, first element), so a 3-element bias costs one block.
struct AdamSegment {       // == mmdgan_adam_segment (include/mmdgan_hip.h)
    long off, n;                         // elements, relative to the arena
    const float *dsigma, *dot, *sigma, *scale;      // NULL / unused for a plain segment
};
static_assert(sizeof(AdamSegment) == sizeof(mmdgan_adam_segment), "ABI");
__global__ __launch_bounds__(256) void adam_segments_kernel(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m,
                                                            float *__restrict__ v, const AdamSegment *__restrict__ segs,
                                                            const int2 *__restrict__ blocks, const float *lr_t_ptr, float b1,
                                                            float b2, float eps, float gscale, int apply_fixup) {
    const int2 bk = blocks[blockIdx.x];
    AdamSegment s = segs[bk.x];
    if (!apply_fixup) s.dsigma = nullptr;
    const float lr_t = lr_t_ptr[0];
    float a = gscale, c = 0.f;
    if (s.dsigma) {
        const float sc = s.scale[0];
        a = gscale * sc;
        c = gscale * (sc / s.sigma[0]) * s.dot[0];
    }
    const long i0 = (long)bk.y * 1024 + threadIdx.x * 4, base = s.off + i0;
    if (i0 + 4 <= s.n && (base & 3) == 0) {
        const float4 gv = *reinterpret_cast<const float4 *>(g + base);
        float4 mv = *reinterpret_cast<const float4 *>(m + base), vv = *reinterpret_cast<const float4 *>(v + base);
        float4 pv = *reinterpret_cast<const float4 *>(p + base);
        float4 dv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (s.dsigma) dv = *reinterpret_cast<const float4 *>(s.dsigma + i0);       // (16-byte aligned: checked by the host)
        const float ge[4] = {a * gv.x - c * dv.x, a * gv.y - c * dv.y, a * gv.z - c * dv.z, a * gv.w - c * dv.w};
        float *pm = reinterpret_cast<float *>(&mv), *pvv = reinterpret_cast<float *>(&vv), *pp = reinterpret_cast<float *>(&pv);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            pm[q] = b1 * pm[q] + (1.f - b1) * ge[q];
            pvv[q] = b2 * pvv[q] + (1.f - b2) * ge[q] * ge[q];
            pp[q] = pp[q] - lr_t * pm[q] / (sqrtf(pvv[q]) + eps);
        }
        *reinterpret_cast<float4 *>(m + base) = mv;
        *reinterpret_cast<float4 *>(v + base) = vv;
        *reinterpret_cast<float4 *>(p + base) = pv;
    } else {
        for (long i = i0; i < i0 + 4 && i < s.n; ++i) {
            const float gi = a * g[s.off + i] - (s.dsigma ? c * s.dsigma[i] : 0.f);
            const float mi = b1 * m[s.off + i] + (1.f - b1) * gi;
            const float vi = b2 * v[s.off + i] + (1.f - b2) * gi * gi;
            m[s.off + i] = mi;
            v[s.off + i] = vi;
            p[s.off + i] = p[s.off + i] - lr_t * mi / (sqrtf(vi) + eps);
        }
    }
}

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float *__restrict__ src, float *__restrict__ dst,
                                                           int N, int C, int HW) {
    const long total = (long)N * C * HW;
    const long stride = (long)gridDim.x * 256;
    for (long o = (long)blockIdx.x * 256 + threadIdx.x; o < total; o += stride) {   // o indexes dst (NHWC)
        const int c = o % C;
        const long t = o / C;
        const int hw = t % HW;
        const long n = t / HW;
        dst[o] = src[(n * C + c) * HW + hw];
    }
}
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float *__restrict__ src, float *__restrict__ dst,
                                                           int N, int C, int HW) {
    const long total = (long)N * C * HW;
    const long stride = (long)gridDim.x * 256;
    for (long o = (long)blockIdx.x * 256 + threadIdx.x; o < total; o += stride) {   // o indexes dst (NCHW)
        const int hw = o % HW;
        const long t = o / HW;
        const int c = t % C;
        const long n = t / C;
        dst[o] = src[(n * HW + hw) * C + c];
    }
}

// uint8 records -> fp32 NHWC in [-1, 1]: input_func.py:797-801 (decode_raw uint8, cast float32) and
// :839-842 (image / 127.5 - 1, reshape to (channels, height, width)).  One thread per pixel: byte reads are
// coalesced along each channel plane, the C floats a thread writes are contiguous with its neighbours'.
// IEEE division and subtraction, nothing to contract: bit-identical to the reference's arithmetic.
template <bool CHW>
__global__ __launch_bounds__(256) void u8_records_kernel(const unsigned char *__restrict__ src,
                                                         float *__restrict__ dst, long NP, int C, int HW) {
    const long stride = (long)gridDim.x * 256;
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < NP; t += stride) {        // t = n * HW + pixel
        const long n = t / HW;
        const int px = (int)(t - n * HW);
        for (int c = 0; c < C; ++c) {
            const unsigned char b = CHW ? src[(n * C + c) * HW + px] : src[t * C + c];
            dst[t * C + c] = __fsub_rn(__fdiv_rn((float)b, 127.5f), 1.0f);
        }
    }
}

// [C,H,W] records with H*W % 4 == 0: a thread takes four neighbouring pixels - one 4-byte load per channel plane,
// 4*C floats stored as C 16-byte vectors (the 12-byte-per-thread stores of the scalar kernel top out at ~50 % of HBM)
template <int C>
__global__ __launch_bounds__(256) void u8_records_chw4_kernel(const unsigned char *__restrict__ src,
                                                              float *__restrict__ dst, long NQ, int HW4) {
    const long stride = (long)gridDim.x * 256;
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < NQ; t += stride) {        // t = n * HW/4 + pixel quad
        const long n = t / HW4;
        const int q = (int)(t - n * HW4);
        float v[4 * C];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const uchar4 b = *reinterpret_cast<const uchar4 *>(src + ((n * C + c) * HW4 + q) * 4);
            v[0 * C + c] = __fsub_rn(__fdiv_rn((float)b.x, 127.5f), 1.0f);
            v[1 * C + c] = __fsub_rn(__fdiv_rn((float)b.y, 127.5f), 1.0f);
            v[2 * C + c] = __fsub_rn(__fdiv_rn((float)b.z, 127.5f), 1.0f);
            v[3 * C + c] = __fsub_rn(__fdiv_rn((float)b.w, 127.5f), 1.0f);
        }
        float4 *o = reinterpret_cast<float4 *>(dst + t * 4 * C);
#pragma unroll
        for (int k = 0; k < C; ++k) o[k] = make_float4(v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]);
    }
}

static inline int grid_for(long n, int per_block = 256, int cap = 2048) {
    long b = (n + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return (int)b;
}

}  // namespace mmdgan

using namespace mmdgan;

extern "C" int mmdgan_colsum(const float *x, long rows, int cols, float *out, void *stream) {
    MMDGAN_REQUIRE(x && out && rows >= 1 && cols >= 1, "colsum: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    if (zero_output(out, sizeof(float) * cols, st) != hipSuccess) return check_launch("colsum memset");
    const int cblocks = (cols + 63) / 64;
    long splits = 1024 / cblocks;
    if (splits < 1) splits = 1;
    long rpb = (rows + splits - 1) / splits;
    if (rpb < 16) rpb = 16;
    splits = (rows + rpb - 1) / rpb;
    if (cols % 4 == 0 && ((uintptr_t)x & 15) == 0)
        hipLaunchKernelGGL(colsum_v4_kernel, dim3(cblocks, (unsigned)splits), dim3(256), 0, st, x, rows, cols, rpb, out);
    else
        hipLaunchKernelGGL(colsum_kernel, dim3(cblocks, (unsigned)splits), dim3(256), 0, st, x, rows, cols, rpb, out);
    return check_launch("colsum");
}

extern "C" int mmdgan_dot(const float *a, const float *b, long n, float *out, void *stream) {
    MMDGAN_REQUIRE(a && b && out && n >= 1, "dot: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    if (zero_output(out, sizeof(float), st) != hipSuccess) return check_launch("dot memset");
    hipLaunchKernelGGL(dot_kernel, dim3(grid_for(n, 1024, 1024)), dim3(256), 0, st, a, b, n, out);
    return check_launch("dot");
}

extern "C" int mmdgan_sn_norm(const float *v, long n, float *out_norm, float *v_normalised, void *stream) {
    MMDGAN_REQUIRE(v && n >= 1 && (out_norm || v_normalised), "sn_norm: bad arguments");
    hipLaunchKernelGGL(sn_norm_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, v, n, out_norm, v_normalised, 0.f,
                       (float *)nullptr);
    return check_launch("sn_norm");
}

extern "C" int mmdgan_sn_norm_scale(const float *v, long n, float act_k, float *out_norm, float *scale_out,
                                    float *v_normalised, void *stream) {
    MMDGAN_REQUIRE(v && n >= 1 && (out_norm || v_normalised || scale_out), "sn_norm_scale: bad arguments");
    hipLaunchKernelGGL(sn_norm_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, v, n, out_norm, v_normalised, act_k,
                       scale_out);
    return check_launch("sn_norm_scale");
}

extern "C" int mmdgan_sn_scale(const float *sigma, float act_k, float *scale_out, void *stream) {
    MMDGAN_REQUIRE(sigma && scale_out, "sn_scale: null pointer");
    hipLaunchKernelGGL(sn_scale_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, sigma, act_k, scale_out);
    return check_launch("sn_scale");
}

extern "C" int mmdgan_sn_wgrad_fixup(float *g_inout, const float *dsigma_dw, const float *dot, const float *sigma,
                                     const float *scale, long n, void *stream) {
    MMDGAN_REQUIRE(g_inout && dsigma_dw && dot && sigma && scale && n >= 1, "sn_wgrad_fixup: bad arguments");
    hipLaunchKernelGGL(sn_fixup_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, g_inout, dsigma_dw, dot,
                       sigma, scale, n);
    return check_launch("sn_wgrad_fixup");
}

extern "C" int mmdgan_adam_multi(const void *const *ptrs, const long *sizes, int n_tensors, long max_size, float lr,
                                 float beta1, float beta2, float eps, int step, int *step_counter, float *lr_t_scratch,
                                 float grad_scale, void *stream) {
    MMDGAN_REQUIRE(ptrs && sizes && lr_t_scratch && n_tensors >= 1 && max_size >= 1, "adam_multi: bad arguments");
    MMDGAN_REQUIRE(step_counter || step >= 1, "adam_multi: step must be >= 1 when no device counter is given");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(adam_prepare_kernel, dim3(1), dim3(1), 0, st, step_counter, step, lr, beta1, beta2, lr_t_scratch);
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(max_size, 256, 1024), n_tensors), dim3(256), 0, st, ptrs, sizes,
                       lr_t_scratch, beta1, beta2, eps, grad_scale);
    return check_launch("adam_multi");
}

extern "C" int mmdgan_adam_segments(float *params, const float *grads, float *adam_m, float *adam_v,
                                    const mmdgan_adam_segment *segments_dev, int n_segments, const int *blocks_dev, long n_blocks,
                                    float lr, float beta1, float beta2, float eps, int step, int *step_counter,
                                    float *lr_t_scratch, float grad_scale, int apply_fixup, void *stream) {
    MMDGAN_REQUIRE(params && grads && adam_m && adam_v && segments_dev && blocks_dev && lr_t_scratch && n_segments >= 1 &&
                   n_blocks >= 1, "adam_segments: bad arguments");
    MMDGAN_REQUIRE(step_counter || step >= 1 || step == MMDGAN_ADAM_PREPARED,
                   "adam_segments: step must be >= 1 (or MMDGAN_ADAM_PREPARED) when no device counter is given");
    hipStream_t st = (hipStream_t)stream;
    if (step_counter || step != MMDGAN_ADAM_PREPARED)
        hipLaunchKernelGGL(adam_prepare_kernel, dim3(1), dim3(1), 0, st, step_counter, step, lr, beta1, beta2, lr_t_scratch);
    hipLaunchKernelGGL(adam_segments_kernel, dim3((unsigned)n_blocks), dim3(256), 0, st, params, grads, adam_m, adam_v,
                       reinterpret_cast<const AdamSegment *>(segments_dev), reinterpret_cast<const int2 *>(blocks_dev), lr_t_scratch,
                       beta1, beta2, eps, grad_scale, apply_fixup);
    return check_launch("adam_segments");
}

extern "C" int mmdgan_adam_prepare(float lr, float beta1, float beta2, int step, int *step_counter, float *lr_t_scratch,
                                   void *stream) {
    MMDGAN_REQUIRE(lr_t_scratch && (step_counter || step >= 1), "adam_prepare: bad arguments");
    hipLaunchKernelGGL(adam_prepare_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step_counter, step, lr, beta1, beta2, lr_t_scratch);
    return check_launch("adam_prepare");
}

namespace {
struct AdamPrepareTable { mmdgan_adam_prepare_job job[8]; };
__global__ void adam_prepare_multi_kernel(AdamPrepareTable t, int n) {
    const int i = threadIdx.x;
    if (i >= n) return;
    // (the by-value table, indexed where it lies: a dynamic index into the argument copy would go through scratch)
    const mmdgan_adam_prepare_job *jobs = (const mmdgan_adam_prepare_job *)__builtin_amdgcn_kernarg_segment_ptr();
    const mmdgan_adam_prepare_job j = jobs[i];
    int step = j.step;
    if (j.step_counter) { step = j.step_counter[0] + 1; j.step_counter[0] = step; }
    j.lr_t_scratch[0] = (float)((double)j.lr * sqrt(1.0 - pow((double)j.beta2, (double)step)) / (1.0 - pow((double)j.beta1, (double)step)));
}
}  // namespace

extern "C" int mmdgan_adam_prepare_multi(const mmdgan_adam_prepare_job *jobs, int n, void *stream) {
    MMDGAN_REQUIRE(jobs && n >= 1 && n <= 8, "adam_prepare_multi: 1..8 jobs");
    AdamPrepareTable t{};
    for (int i = 0; i < n; ++i) {
        MMDGAN_REQUIRE(jobs[i].lr_t_scratch && (jobs[i].step_counter || jobs[i].step >= 1), "adam_prepare_multi: bad job %d", i);
        t.job[i] = jobs[i];
    }
    hipLaunchKernelGGL(adam_prepare_multi_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, t, n);
    return check_launch("adam_prepare_multi");
}

extern "C" int mmdgan_nchw_to_nhwc(const float *src, float *dst, int N, int C, int H, int W, void *stream) {
    MMDGAN_REQUIRE(src && dst && N >= 1 && C >= 1 && H >= 1 && W >= 1, "nchw_to_nhwc: bad arguments");
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(grid_for((long)N * C * H * W)), dim3(256), 0, (hipStream_t)stream, src,
                       dst, N, C, H * W);
    return check_launch("nchw_to_nhwc");
}
extern "C" int mmdgan_u8_records_to_nhwc(const unsigned char *src, int src_is_chw, float *dst, int N, int C, int H,
                                         int W, void *stream) {
    MMDGAN_REQUIRE(src && dst && N >= 1 && C >= 1 && H >= 1 && W >= 1, "u8_records_to_nhwc: bad arguments");
    const long np = (long)N * H * W;
    const bool vec = src_is_chw && (H * W) % 4 == 0 && ((uintptr_t)src % 4 == 0) && ((uintptr_t)dst % 16 == 0);
    if (vec && (C == 1 || C == 3 || C == 4)) {
        const long nq = np / 4;
        const dim3 g(grid_for(nq, 256, 8192));
        if (C == 1) hipLaunchKernelGGL(u8_records_chw4_kernel<1>, g, dim3(256), 0, (hipStream_t)stream, src, dst, nq, H * W / 4);
        if (C == 3) hipLaunchKernelGGL(u8_records_chw4_kernel<3>, g, dim3(256), 0, (hipStream_t)stream, src, dst, nq, H * W / 4);
        if (C == 4) hipLaunchKernelGGL(u8_records_chw4_kernel<4>, g, dim3(256), 0, (hipStream_t)stream, src, dst, nq, H * W / 4);
        return check_launch("u8_records_to_nhwc");
    }
    if (src_is_chw)
        hipLaunchKernelGGL(u8_records_kernel<true>, dim3(grid_for(np)), dim3(256), 0, (hipStream_t)stream, src, dst, np, C, H * W);
    else
        hipLaunchKernelGGL(u8_records_kernel<false>, dim3(grid_for(np)), dim3(256), 0, (hipStream_t)stream, src, dst, np, C, H * W);
    return check_launch("u8_records_to_nhwc");
}
extern "C" int mmdgan_nhwc_to_nchw(const float *src, float *dst, int N, int C, int H, int W, void *stream) {
    MMDGAN_REQUIRE(src && dst && N >= 1 && C >= 1 && H >= 1 && W >= 1, "nhwc_to_nchw: bad arguments");
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(grid_for((long)N * C * H * W)), dim3(256), 0, (hipStream_t)stream, src,
                       dst, N, C, H * W);
    return check_launch("nhwc_to_nchw");
}
