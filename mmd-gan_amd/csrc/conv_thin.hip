// Thin-layer convolution kernels: the first D layer (3 -> 64 channels) and the last G layer
// (64 -> 3), forward and both gradients.  Their im2col depth (27) or output width (3) would waste
// >90 % of an MFMA tile and together they are 0.2 % of the step's FLOPs, so they are HBM-bound
// VALU kernels built around three patterns:
//   thin_in   few input values per output pixel (<= 64 = taps x thin channels), wide output:
//             weights in LDS as [tap*i][n], each thread owns one pixel x 16 consecutive outputs
//             (4 x float4 stores).                       -> D l1 forward, G l5 input-gradient
//   thin_out  wide reduction (taps x C, C % 4 == 0), <= 4 outputs per pixel: one thread per
//             pixel, float4 activation loads, weights broadcast from LDS as [tap][j][i].
//                                                        -> G l5 forward, D l1 input-gradient
//   thin_wgrad  dw[tap][c][k] with one of C, K thin: a block walks a chunk of pixels, lane = wide
//             channel (coalesced), the 4 waves split the thin (tap, channel) entries, partial
//             sums are combined with fp32 atomics.       -> D l1 / G l5 weight-gradient
#include "conv_internal.h"

namespace mmdgan {

constexpr int kThinMaxRed = 64;     // thin_in: taps * thin channels
constexpr int kThinMaxOut = 4;      // thin_out: outputs per pixel

// input pixel feeding output pixel (oy, ox) through tap (r, t); false if outside / not on the grid
template <bool DGRAD>
__device__ __forceinline__ bool tap_source(const ConvDims &d, int oy, int ox, int r, int t, int &iy, int &ix) {
    if (!DGRAD) {
        iy = oy * d.stride - d.pad + r;
        ix = ox * d.stride - d.pad + t;
        return iy >= 0 && iy < d.H && ix >= 0 && ix < d.W;
    }
    const int hy = oy + d.pad - r, hx = ox + d.pad - t;
    if (hy < 0 || hx < 0 || hy % d.stride || hx % d.stride) return false;
    iy = hy / d.stride; ix = hx / d.stride;
    return iy < d.P && ix < d.Q;
}

// ---------------------------------------------------------------------------------------------
// thin_in: out[px][n] = sum_{tap, i < CI} in[src(px, tap)][i] * W(tap, i, n)
//   forward : in = x [N,H,W,CI=C],   out = y  [N,P,Q,NW=K], W(tap,i,n) = w[tap][i][n]
//   dgrad   : in = dy [N,P,Q,CI=K],  out = dx [N,H,W,NW=C], W(tap,i,n) = w[tap][n][i]
template <bool DGRAD>
__global__ __launch_bounds__(256) void thin_in_kernel(ConvDims d, ConvEpilogue ep, const float *__restrict__ in,
                                                      const float *__restrict__ w, float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float ws[];                // [tap*CI + i][NW]
    const int CI = DGRAD ? d.K : d.C, NW = DGRAD ? d.C : d.K;
    const int OH = DGRAD ? d.H : d.P, OW = DGRAD ? d.W : d.Q;
    const int IH = DGRAD ? d.P : d.H, IW = DGRAD ? d.Q : d.W;
    const int taps = d.R * d.R, red = taps * CI;
    for (int e = threadIdx.x; e < red * NW; e += 256) {
        const int n = e % NW, ti = e / NW, i = ti % CI, tap = ti / CI;
        ws[e] = DGRAD ? w[((long)tap * d.C + n) * d.K + i] : w[((long)tap * d.C + i) * d.K + n];
    }
    __syncthreads();
    const int groups = NW / 16;
    const long total = (long)d.N * OH * OW * groups;
    const float sc = ep.scale ? ep.scale[0] : 1.f;
    for (long gid = (long)blockIdx.x * 256 + threadIdx.x; gid < total; gid += (long)gridDim.x * 256) {
        const int g = gid % groups;
        long px = gid / groups;
        const int ox = px % OW;
        const long u = px / OW;
        const int oy = u % OH;
        const long n = u / OH;
        float acc[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = 0.f;
        for (int r = 0; r < d.R; ++r)
            for (int t = 0; t < d.R; ++t) {
                int iy, ix;
                if (!tap_source<DGRAD>(d, oy, ox, r, t, iy, ix)) continue;
                const float *ip = in + ((n * IH + iy) * IW + ix) * CI;
                const float *wp = ws + (long)((r * d.R + t) * CI) * NW + g * 16;
                for (int i = 0; i < CI; ++i) {
                    const float v = ip[i];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 wv = *reinterpret_cast<const float4 *>(wp + i * NW + q * 4);
                        acc[q * 4 + 0] = fmaf(v, wv.x, acc[q * 4 + 0]);
                        acc[q * 4 + 1] = fmaf(v, wv.y, acc[q * 4 + 1]);
                        acc[q * 4 + 2] = fmaf(v, wv.z, acc[q * 4 + 2]);
                        acc[q * 4 + 3] = fmaf(v, wv.w, acc[q * 4 + 3]);
                    }
                }
            }
        const long o = px * NW + g * 16;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 v;
            v.x = ep.apply(acc[q * 4 + 0] * sc, g * 16 + q * 4 + 0, o + q * 4 + 0);
            v.y = ep.apply(acc[q * 4 + 1] * sc, g * 16 + q * 4 + 1, o + q * 4 + 1);
            v.z = ep.apply(acc[q * 4 + 2] * sc, g * 16 + q * 4 + 2, o + q * 4 + 2);
            v.w = ep.apply(acc[q * 4 + 3] * sc, g * 16 + q * 4 + 3, o + q * 4 + 3);
            *reinterpret_cast<float4 *>(out + o + q * 4) = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// thin_out: out[px][j < NJ] = sum_{tap, i < CW} in[src(px, tap)][i] * W(tap, j, i),  CW % 4 == 0
//   forward : in = x,  out = y  [..,NJ=K], W(tap,j,i) = w[tap][i][j]
//   dgrad   : in = dy, out = dx [..,NJ=C], W(tap,j,i) = w[tap][j][i]
template <bool DGRAD>
__global__ __launch_bounds__(128) void thin_out_kernel(ConvDims d, ConvEpilogue ep, const float *__restrict__ in,
                                                       const float *__restrict__ w, float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float ws[];                // [tap][j][CW]
    const int CW = DGRAD ? d.K : d.C, NJ = DGRAD ? d.C : d.K;
    const int OH = DGRAD ? d.H : d.P, OW = DGRAD ? d.W : d.Q;
    const int IH = DGRAD ? d.P : d.H, IW = DGRAD ? d.Q : d.W;
    const int taps = d.R * d.R;
    for (int e = threadIdx.x; e < taps * NJ * CW; e += 128) {
        const int i = e % CW, tj = e / CW, j = tj % NJ, tap = tj / NJ;
        ws[e] = DGRAD ? w[((long)tap * d.C + j) * d.K + i] : w[((long)tap * d.C + i) * d.K + j];
    }
    __syncthreads();
    const long total = (long)d.N * OH * OW;
    const float sc = ep.scale ? ep.scale[0] : 1.f;
    for (long px = (long)blockIdx.x * 128 + threadIdx.x; px < total; px += (long)gridDim.x * 128) {
        const int ox = px % OW;
        const long u = px / OW;
        const int oy = u % OH;
        const long n = u / OH;
        float acc[kThinMaxOut] = {0.f, 0.f, 0.f, 0.f};
        for (int r = 0; r < d.R; ++r)
            for (int t = 0; t < d.R; ++t) {
                int iy, ix;
                if (!tap_source<DGRAD>(d, oy, ox, r, t, iy, ix)) continue;
                const float4 *ip = reinterpret_cast<const float4 *>(in + ((n * IH + iy) * IW + ix) * CW);
                const float4 *wp = reinterpret_cast<const float4 *>(ws + (long)((r * d.R + t) * NJ) * CW);
                for (int i4 = 0; i4 < CW / 4; ++i4) {
                    const float4 v = ip[i4];
#pragma unroll
                    for (int j = 0; j < kThinMaxOut; ++j)
                        if (j < NJ) {
                            const float4 wv = wp[j * (CW / 4) + i4];
                            acc[j] = fmaf(v.x, wv.x, fmaf(v.y, wv.y, fmaf(v.z, wv.z, fmaf(v.w, wv.w, acc[j]))));
                        }
                }
            }
#pragma unroll
        for (int j = 0; j < kThinMaxOut; ++j)
            if (j < NJ) out[px * NJ + j] = ep.apply(acc[j] * sc, j, px * NJ + j);
    }
}

// ---------------------------------------------------------------------------------------------
// thin_wgrad: dw[tap][c][k] += sum over a chunk of output rows.  lane = wide channel (64 per
// block.y), the block's 4 waves split the thin entries e = tap*T + thin_channel round-robin
// (<= 8 each).  A block walks whole output rows (n, p) with incremental addresses: per pixel one
// coalesced load of the wide operand is shared by the wave's entries and the thin operand is a
// wave-uniform (broadcast) load, so the inner loop is ~3 instructions per entry.
template <bool WIDE_K>
__global__ __launch_bounds__(256) void thin_wgrad_kernel(ConvDims d, const float *__restrict__ x,
                                                         const float *__restrict__ dy, float *dw, int rows_per_block,
                                                         float *partials) {
    // lane = (pixel group pg = lane/16, 4 consecutive wide channels w4 = lane%16): one float4 load
    // instruction covers 4 pixels x 64 wide channels - the kernel is bound by the NUMBER of vector
    // memory instructions, not by bytes (one-pixel-per-instruction measured 250 us on D l1).
    // WIDE_K  (thin C): slot s <-> entry e = wave + 4 s = (tap, c): 8 slots, x is a per-pixel scalar
    // !WIDE_K (thin K): slot g <-> tap = wave + 4 g (3 taps per wave); the K (<= 4) outputs of a tap
    //                   share one x load
    constexpr int NS = WIDE_K ? 8 : 3, NT = WIDE_K ? 1 : 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, pg = lane >> 4, w4 = lane & 15;
    const int T = WIDE_K ? d.C : d.K, WIDE = WIDE_K ? d.K : d.C;
    const int wide = blockIdx.y * 64 + w4 * 4;
    const int taps = d.R * d.R;
    const int entries = taps * T;
    const int nrows = d.N * d.P;
    const int row0 = blockIdx.x * rows_per_block;
    const int row1 = min(nrows, row0 + rows_per_block);
    const bool wide_ok = wide < WIDE;
    int er[NS], et[NS], eth[NS];
    bool live[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int e = wave + 4 * s;
        const int tap = WIDE_K ? e / T : e;
        live[s] = WIDE_K ? e < entries : tap < taps;
        eth[s] = WIDE_K ? e - tap * T : 0;
        er[s] = tap / d.R;
        et[s] = tap - er[s] * d.R;
    }
    float4 acc[NS][NT];
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int k = 0; k < NT; ++k) acc[s][k] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int row = row0; row < row1; ++row) {
        const int n = row / d.P, p = row - n * d.P;
        const float *dyrow = dy + (long)row * d.Q * d.K;
        int xbase[NS];          // offset of x[n, h, 0, 0] for each slot's tap row, or -1
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int h = p * d.stride - d.pad + er[s];
            xbase[s] = (live[s] && wide_ok && h >= 0 && h < d.H) ? ((n * d.H + h) * d.W) * d.C : -1;
        }
        for (int q = pg; q < d.Q; q += 4) {
            float4 wv = make_float4(0.f, 0.f, 0.f, 0.f);      // the wide operand of this pixel (WIDE_K: dy)
            float tv[NT];                                       // thin dy values (!WIDE_K)
            if (WIDE_K) { if (wide_ok) wv = *reinterpret_cast<const float4 *>(dyrow + q * d.K + wide); }
            else {
#pragma unroll
                for (int k = 0; k < NT; ++k) tv[k] = k < T ? dyrow[q * d.K + k] : 0.f;
            }
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int ww = q * d.stride - d.pad + et[s];
                if (xbase[s] < 0 || ww < 0 || ww >= d.W) continue;
                if (WIDE_K) {
                    const float xv = x[xbase[s] + ww * d.C + eth[s]];
                    acc[s][0].x = fmaf(xv, wv.x, acc[s][0].x); acc[s][0].y = fmaf(xv, wv.y, acc[s][0].y);
                    acc[s][0].z = fmaf(xv, wv.z, acc[s][0].z); acc[s][0].w = fmaf(xv, wv.w, acc[s][0].w);
                } else {
                    const float4 xv = *reinterpret_cast<const float4 *>(x + xbase[s] + ww * d.C + wide);
#pragma unroll
                    for (int k = 0; k < NT; ++k) {
                        acc[s][k].x = fmaf(xv.x, tv[k], acc[s][k].x); acc[s][k].y = fmaf(xv.y, tv[k], acc[s][k].y);
                        acc[s][k].z = fmaf(xv.z, tv[k], acc[s][k].z); acc[s][k].w = fmaf(xv.w, tv[k], acc[s][k].w);
                    }
                }
            }
        }
    }
    // combine the 4 pixel groups (lanes l, l^16, l^32, l^48), then one atomic per output from pg == 0
#pragma unroll
    for (int s = 0; s < NS; ++s) {
#pragma unroll
        for (int k = 0; k < NT; ++k) {
            float v[4] = {acc[s][k].x, acc[s][k].y, acc[s][k].z, acc[s][k].w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                v[c] += __shfl_xor(v[c], 16, 64);
                v[c] += __shfl_xor(v[c], 32, 64);
            }
            if (pg == 0 && live[s] && wide_ok && (WIDE_K || k < T)) {
                const int tap = er[s] * d.R + et[s];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const long o = WIDE_K ? ((long)tap * d.C + eth[s]) * d.K + wide + c : ((long)tap * d.C + wide + c) * d.K + k;
                    // every (wave, slot, lane) owns its outputs exclusively within a block: with a
                    // workspace the block's partial is a plain store and a second pass sums blocks
                    if (partials) partials[(long)blockIdx.x * ((long)taps * d.C * d.K) + o] = v[c];
                    else atomicAdd(dw + o, v[c]);
                }
            }
        }
    }
}

// dw[o] = sum_b partials[b][o]: 64 outputs per block (coalesced across lanes), the 4 waves split the
// partial blocks, 8 independent loads in flight per lane, LDS combine - no dependent load chain.
// wdot / dot (optional): dot[0] += <dw, wdot> over this block's 64 outputs (dot is zero on entry) - the scalar of the
// spectral-norm fix-up, formed where the finished dw is in registers (one atomic per block, 27 for a 3x3x3x64 kernel)
__global__ __launch_bounds__(256) void thin_wgrad_reduce_kernel(const float *__restrict__ partials, int nblocks, long nout,
                                                                float *__restrict__ dw, const float *__restrict__ wdot,
                                                                float *__restrict__ dot) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long o = (long)blockIdx.x * 64 + lane;
    float acc = 0.f;
    if (o < nout) {
        int b = wave;
        for (; b + 28 < nblocks; b += 32) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = partials[(long)(b + 4 * u) * nout + o];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u];
        }
        for (; b < nblocks; b += 4) acc += partials[(long)b * nout + o];
    }
    red[wave][lane] = acc;
    __syncthreads();
    if (wave != 0) return;
    double part = 0;
    if (o < nout) {
        const float v = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
        dw[o] = v;
        if (wdot) part = (double)v * (double)wdot[o];
    }
    if (wdot) {
        part = wave_sum(part);
        if (lane == 0) atomicAdd(dot, (float)part);
    }
}

// ---------------------------------------------------------------------------------------------
bool thin_fwd_in_ok(const ConvDims &d) { return d.R * d.R * d.C <= kThinMaxRed && d.K % 16 == 0 && d.K <= 256 && d.C < 16; }
bool thin_dgrad_in_ok(const ConvDims &d) { return d.R * d.R * d.K <= kThinMaxRed && d.C % 16 == 0 && d.C <= 256 && d.K < 16; }
bool thin_fwd_out_ok(const ConvDims &d) { return d.K <= kThinMaxOut && d.C % 4 == 0 && d.R * d.R * d.K * d.C * 4 <= 60 * 1024; }
bool thin_dgrad_out_ok(const ConvDims &d) { return d.C <= kThinMaxOut && d.K % 4 == 0 && d.R * d.R * d.K * d.C * 4 <= 60 * 1024; }
bool thin_wgrad_ok(const ConvDims &d) {
    return (d.R * d.R * d.C <= 32 && d.K >= 16 && d.K % 4 == 0) || (d.R * d.R <= 12 && d.K <= 4 && d.C >= 16 && d.C % 4 == 0);
}

static unsigned blocks_for(long work, int per_block, int cap) {
    long b = (work + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return (unsigned)b;
}

int thin_fwd(const ConvDims &d, const ConvEpilogue &ep, const float *x, const float *w, float *y, hipStream_t st) {
    if (thin_fwd_in_ok(d)) {
        const long work = (long)d.N * d.P * d.Q * (d.K / 16);
        hipLaunchKernelGGL(thin_in_kernel<false>, dim3(blocks_for(work, 256, 4096)), dim3(256),
                           sizeof(float) * d.R * d.R * d.C * d.K, st, d, ep, x, w, y);
    } else {
        const long work = (long)d.N * d.P * d.Q;
        const size_t lds = sizeof(float) * d.R * d.R * d.K * d.C;
        hipLaunchKernelGGL(thin_out_kernel<false>, dim3(blocks_for(work, 128, 4096)), dim3(128), lds, st, d, ep, x, w, y);
    }
    addend_applied();                                   // ConvEpilogue::apply in both kernels
    return check_launch("conv2d_fwd(thin)");
}
int thin_dgrad(const ConvDims &d, const ConvEpilogue &ep, const float *dy, const float *w, float *dx, hipStream_t st) {
    if (thin_dgrad_in_ok(d)) {
        const long work = (long)d.N * d.H * d.W * (d.C / 16);
        hipLaunchKernelGGL(thin_in_kernel<true>, dim3(blocks_for(work, 256, 4096)), dim3(256),
                           sizeof(float) * d.R * d.R * d.C * d.K, st, d, ep, dy, w, dx);
    } else {
        const long work = (long)d.N * d.H * d.W;
        const size_t lds = sizeof(float) * d.R * d.R * d.K * d.C;
        hipLaunchKernelGGL(thin_out_kernel<true>, dim3(blocks_for(work, 128, 4096)), dim3(128), lds, st, d, ep, dy, w, dx);
    }
    addend_applied();
    return check_launch("conv2d_dgrad(thin)");
}
int thin_wgrad(const ConvDims &d, const float *x, const float *dy, float *dw, hipStream_t st) {
    const long nout = (long)d.R * d.R * d.C * d.K;
    const bool wide_k = d.R * d.R * d.C <= 32 && d.K >= 16;
    const int wide = wide_k ? d.K : d.C;
    const int nrows = d.N * d.P;
    const int ychunks = (wide + 63) / 64;
    int rpb = (nrows * ychunks + 511) / 512;             // ~512 blocks: every block ends with one atomic per output,
                                                         // 2048 blocks made those 1728 addresses the bottleneck
    if (rpb < 1) rpb = 1;
    const int xblocks = (nrows + rpb - 1) / rpb;
    // measured: 512 blocks x 1728 contended fp32 atomics cost 125-265 us; partials + reduce ~15 us
    // (a batch-1 launch belongs to a power iteration, which shares the handle with concurrent chains: atomics, not the workspace)
    float *partials = d.N > 1 ? (float *)workspace_acquire(sizeof(float) * nout * xblocks, st) : nullptr;
    if (!partials && zero_output(dw, sizeof(float) * nout, st) != hipSuccess) return check_launch("conv2d_wgrad memset");
    if (wide_k) hipLaunchKernelGGL(thin_wgrad_kernel<true>, dim3(xblocks, ychunks), dim3(256), 0, st, d, x, dy, dw, rpb, partials);
    else hipLaunchKernelGGL(thin_wgrad_kernel<false>, dim3(xblocks, ychunks), dim3(256), 0, st, d, x, dy, dw, rpb, partials);
    if (partials)
        hipLaunchKernelGGL(thin_wgrad_reduce_kernel, dim3((unsigned)((nout + 63) / 64)), dim3(256), 0, st, partials, xblocks,
                           nout, dw, (const float *)nullptr, (float *)nullptr);
    return check_launch("conv2d_wgrad(thin)");
}

int thin_wgrad_reduce(const float *partials, int nblocks, long nout, float *dw, hipStream_t st, const float *wdot, float *dot) {
    hipLaunchKernelGGL(thin_wgrad_reduce_kernel, dim3((unsigned)((nout + 63) / 64)), dim3(256), 0, st, partials, nblocks, nout, dw,
                       wdot, dot);
    return check_launch("conv2d_wgrad(thin reduce)");
}

}  // namespace mmdgan
