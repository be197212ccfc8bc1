// Direct (non-GEMM) NHWC convolution kernels for the shapes the MFMA implicit-GEMM path does not
// take: the thin first/last layers of the model (D l1: C=3, G l5: K=3 - im2col depth 27 or an
// output width of 3 would waste >90 % of an MFMA tile and they are HBM-bound anyway) and channel
// counts that are not multiples of the GEMM tile (the width/8 parity nets).
// One thread per output element; consecutive lanes take consecutive channels so weight / output
// accesses coalesce and the activation reads broadcast.
#include "conv_internal.h"

namespace mmdgan {

__global__ __launch_bounds__(256) void direct_fwd_kernel(ConvDims d, ConvEpilogue ep, const float *__restrict__ x,
                                                         const float *__restrict__ w, float *__restrict__ y) {
    const long total = (long)d.N * d.P * d.Q * d.K;
    const long stride = (long)gridDim.x * 256;
    const float sc = ep.scale ? ep.scale[0] : 1.f;
    for (long o = (long)blockIdx.x * 256 + threadIdx.x; o < total; o += stride) {
        const int k = o % d.K;
        long t = o / d.K;
        const int q = t % d.Q; t /= d.Q;
        const int p = t % d.P;
        const int n = t / d.P;
        float acc = 0.f;
        for (int r = 0; r < d.R; ++r) {
            const int h = p * d.stride - d.pad + r;
            if (h < 0 || h >= d.H) continue;
            for (int s = 0; s < d.R; ++s) {
                const int ww = q * d.stride - d.pad + s;
                if (ww < 0 || ww >= d.W) continue;
                const float *xp = x + (((long)n * d.H + h) * d.W + ww) * d.C;
                const float *wp = w + ((long)(r * d.R + s) * d.C) * d.K + k;
                for (int c = 0; c < d.C; ++c) acc = fmaf(xp[c], wp[(long)c * d.K], acc);
            }
        }
        y[o] = ep.apply(acc * sc, k, o);
    }
}

// dx[n,h,w,c] = sum_{r,s,k} dy[n,p,q,k] * w[r,s,c,k] with p*stride - pad + r = h
__global__ __launch_bounds__(256) void direct_dgrad_kernel(ConvDims d, ConvEpilogue ep, const float *__restrict__ dy,
                                                           const float *__restrict__ w, float *__restrict__ dx) {
    const long total = (long)d.N * d.H * d.W * d.C;
    const long stride = (long)gridDim.x * 256;
    const float sc = ep.scale ? ep.scale[0] : 1.f;
    for (long o = (long)blockIdx.x * 256 + threadIdx.x; o < total; o += stride) {
        const int c = o % d.C;
        long t = o / d.C;
        const int ww = t % d.W; t /= d.W;
        const int h = t % d.H;
        const int n = t / d.H;
        float acc = 0.f;
        for (int r = 0; r < d.R; ++r) {
            const int hp = h + d.pad - r;
            if (hp < 0 || hp % d.stride) continue;
            const int p = hp / d.stride;
            if (p >= d.P) continue;
            for (int s = 0; s < d.R; ++s) {
                const int wq = ww + d.pad - s;
                if (wq < 0 || wq % d.stride) continue;
                const int q = wq / d.stride;
                if (q >= d.Q) continue;
                const float *gp = dy + (((long)n * d.P + p) * d.Q + q) * d.K;
                const float *wp = w + ((long)(r * d.R + s) * d.C + c) * d.K;
                for (int k = 0; k < d.K; ++k) acc = fmaf(gp[k], wp[k], acc);
            }
        }
        dx[o] = ep.apply(acc * sc, c, o);
    }
}

// dw[r,s,c,k] += sum over a chunk of output pixels; dw zeroed by a memset node first
__global__ __launch_bounds__(256) void direct_wgrad_kernel(ConvDims d, const float *__restrict__ x,
                                                           const float *__restrict__ dy, float *dw, long pix_per_block) {
    const long nout = (long)d.R * d.R * d.C * d.K;
    const long o = (long)blockIdx.x * 256 + threadIdx.x;
    if (o >= nout) return;
    const int k = o % d.K;
    long t = o / d.K;
    const int c = t % d.C; t /= d.C;
    const int s = t % d.R;
    const int r = t / d.R;
    const long npix = (long)d.N * d.P * d.Q;
    const long p0 = (long)blockIdx.y * pix_per_block;
    long p1 = p0 + pix_per_block;
    if (p1 > npix) p1 = npix;
    float acc = 0.f;
    for (long pix = p0; pix < p1; ++pix) {
        const int q = pix % d.Q;
        const long u = pix / d.Q;
        const int p = u % d.P;
        const int n = u / d.P;
        const int h = p * d.stride - d.pad + r, ww = q * d.stride - d.pad + s;
        if (h < 0 || h >= d.H || ww < 0 || ww >= d.W) continue;
        acc = fmaf(x[(((long)n * d.H + h) * d.W + ww) * d.C + c], dy[pix * d.K + k], acc);
    }
    atomicAdd(dw + o, acc);
}

int direct_fwd(const ConvDims &d, const ConvEpilogue &ep, const float *x, const float *w, float *y, hipStream_t st) {
    const long total = (long)d.N * d.P * d.Q * d.K;
    long blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(direct_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, st, d, ep, x, w, y);
    addend_applied();                                   // ConvEpilogue::apply
    return check_launch("conv2d_fwd(direct)");
}
int direct_dgrad(const ConvDims &d, const ConvEpilogue &ep, const float *dy, const float *w, float *dx, hipStream_t st) {
    const long total = (long)d.N * d.H * d.W * d.C;
    long blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(direct_dgrad_kernel, dim3((unsigned)blocks), dim3(256), 0, st, d, ep, dy, w, dx);
    addend_applied();
    return check_launch("conv2d_dgrad(direct)");
}
int direct_wgrad(const ConvDims &d, const float *x, const float *dy, float *dw, hipStream_t st) {
    const long nout = (long)d.R * d.R * d.C * d.K;
    if (zero_output(dw, sizeof(float) * nout, st) != hipSuccess) return check_launch("conv2d_wgrad memset");
    const long npix = (long)d.N * d.P * d.Q;
    const int oblocks = (int)((nout + 255) / 256);
    long splits = 2048 / oblocks;
    if (splits < 1) splits = 1;
    long ppb = (npix + splits - 1) / splits;
    if (ppb < 32) ppb = 32;
    splits = (npix + ppb - 1) / ppb;
    hipLaunchKernelGGL(direct_wgrad_kernel, dim3(oblocks, (unsigned)splits), dim3(256), 0, st, d, x, dy, dw, ppb);
    return check_launch("conv2d_wgrad(direct)");
}

}  // namespace mmdgan
