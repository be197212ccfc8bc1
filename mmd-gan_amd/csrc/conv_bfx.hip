// Convolutions on the bf16 MFMA pipe at fp32 accuracy: every fp32 operand is split EXACTLY into three bf16 terms
// (x = hi + mid + lo, 8 mantissa bits each, by truncation: x - hi and (x - hi) - mid are exact in fp32) and a product
// a*b is issued as the six bf16 products whose weight is >= 2^-16 of it:
//     hi*hi + (hi*mid + mid*hi) + (hi*lo + lo*hi + mid*mid)            dropped: mid*lo, lo*mid, lo*lo  (<= 2^-23 of |a*b|)
// bf16 x bf16 is exact in the fp32 accumulator, so the result carries the rounding of an fp32 FMA chain (measured on
// random operands: 5e-9 of the output scale against 4e-7 for the fp32 MFMA chain).
// v_mfma_f32_32x32x16_bf16 retires 8x the reduction depth of v_mfma_f32_32x32x2_f32 in half the cycles: six of them
// cost 192 cycles where the fp32 pipe needs 512 for the same 32x32x16 block - 2.67x the rate, ~325 TFLOP/s fp32-equivalent
// (tools/mfma_bf16_peak.hip) against 157.  What the kernel must then keep up with is operand traffic, hence:
//   * one GENERIC gather-GEMM serves the forward convolution and (through tap tables) the input-gradient of stride-1
//     and stride-2 convolutions = the forward pass of transposed convolutions:
//         out[m, n] = sum_t sum_c in[pix(m, t), c] * Wt[t][n][c]
//   * a wave owns a 64x64 output tile (2x2 MFMA blocks): per 16-deep step it reads 12 KB of fragments from LDS for 24
//     MFMAs (768 cycles) = 16 B/clk - four SIMDs stay inside the CU's 128 B/clk LDS port;
//   * weights are split ONCE per weight update into Wp[3 planes][tap][n][c] bf16 (reduction index contiguous, the MFMA
//     B-fragment layout), so the B tile goes global -> LDS as plain 16-byte copies; activations are split on their way
//     from registers to LDS (5.5 VALU ops per element, ~6 % of the MFMA time);
//   * LDS rows are 80 bytes apart: the eight lanes of a b128 phase hit disjoint banks.
// Epilogue as everywhere: scale (SN, device scalar), bias, activation or activation derivative with the 3B-row wrap.
#include "conv_internal.h"
#include "bufload.h"

namespace mmdgan {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace bfx {
constexpr int BK = 32;                   // reduction channels per stage (two MFMA k-steps of 16)
constexpr int LDR = 40;                  // LDS row pitch in bf16 (32 + 8: 80 bytes)
constexpr int MAXT = 16;                 // taps per phase

struct Params {
    int N, IH, IW, Cin;                  // input  [N, IH, IW, Cin] fp32
    int OHt, OWt;                        // output pixels per image of one phase (the tile grid)
    int OH, OW, Cout;                    // output [N, OH, OW, Cout]
    int ostep, istep;                    // output pixel (oy*ostep + o0y, ox*ostep + o0x); input pixel (oy*istep + ty, ox*istep + tx)
    int nphase, ntaps, ttotal;           // phases (blockIdx.z), taps per phase, taps in Wp
    int o0y[4], o0x[4];
    signed char ty[4][MAXT], tx[4][MAXT], wt[4][MAXT];
};

template <int WM, int WN>
struct Cfg {
    static constexpr int BM = 64 * WM, BN = 64 * WN, NT = 64 * WM * WN;
    static constexpr int A_ELEMS = 3 * BM * LDR, B_ELEMS = 3 * BN * LDR;       // bf16 elements
    static constexpr int AV = BM * 8 / NT;                                     // float4 loads of A per thread and stage
    static constexpr int BV = 3 * BN * 4 / NT;                                 // 16-byte loads of B per thread and stage
    static constexpr size_t LDS_BYTES = 2 * (size_t)(A_ELEMS + B_ELEMS) + sizeof(long) * BM;
};
}  // namespace bfx

// Wp[p][t][n][c] (bf16), p = 0 hi, 1 mid, 2 lo.   FWD: n = k (output channel), c = input channel of w[t][c][k];
// DGRAD: n = c, reduction index = k (w's own order)
template <bool DGRAD>
__global__ __launch_bounds__(256) void bfx_weight_kernel(const float *__restrict__ w, unsigned short *__restrict__ Wp, int T, int C,
                                                         int K) {
    __shared__ float tile[32][33];
    const int t = blockIdx.z, c0 = blockIdx.y * 32, k0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, tq = threadIdx.x >> 5;
    const size_t plane = (size_t)T * C * K;
    auto put = [&](size_t idx, float x) {
        const unsigned u = __float_as_uint(x), h = u & 0xffff0000u;
        const float r = x - __uint_as_float(h);
        const unsigned m = __float_as_uint(r) & 0xffff0000u;
        const float r2 = r - __uint_as_float(m);
        Wp[idx] = (unsigned short)(h >> 16);
        Wp[plane + idx] = (unsigned short)(m >> 16);
        Wp[2 * plane + idx] = (unsigned short)(__float_as_uint(r2) >> 16);
    };
    if (DGRAD) {                       // straight: Wp[t][c][k]
        for (int cc = tq; cc < 32; cc += 8) {
            const int c = c0 + cc, k = k0 + tx;
            if (c < C && k < K) put(((size_t)t * C + c) * K + k, w[((size_t)t * C + c) * K + k]);
        }
    } else {                           // transposed: Wp[t][k][c]
        for (int cc = tq; cc < 32; cc += 8) {
            const int c = c0 + cc, k = k0 + tx;
            tile[cc][tx] = (c < C && k < K) ? w[((size_t)t * C + c) * K + k] : 0.f;
        }
        __syncthreads();
        for (int kk = tq; kk < 32; kk += 8) {
            const int k = k0 + kk, c = c0 + tx;
            if (c < C && k < K) put(((size_t)t * K + k) * C + c, tile[tx][kk]);
        }
    }
}

__device__ __forceinline__ unsigned pack_hi(unsigned x1, unsigned x0) {       // (x1 & 0xffff0000) | (x0 >> 16)
    return __builtin_amdgcn_perm(x1, x0, 0x07060302u);
}

template <int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN) void bfx_conv_kernel(bfx::Params P, ConvEpilogue ep, const float *__restrict__ in,
                                                               const unsigned short *__restrict__ Wp, float *__restrict__ out) {
    using namespace bfx;
    using C = Cfg<WM, WN>;
    constexpr int BM = C::BM, BN = C::BN, NT = C::NT, AV = C::AV, BV = C::BV;
    extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
    unsigned short *As = smem;                         // [3][BM][LDR]
    unsigned short *Bs = smem + C::A_ELEMS;            // [3][BN][LDR]
    long *obase = reinterpret_cast<long *>(reinterpret_cast<char *>(smem) + 2 * (size_t)(C::A_ELEMS + C::B_ELEMS));   // [BM]

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, kh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM, wn = wave / WM;
    const int phase = blockIdx.z;
    const long M = (long)P.N * P.OHt * P.OWt;
    const long m0 = (long)blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;

    // ---- per-row bookkeeping: output address of every tile row (LDS), input origin of this thread's A rows (registers)
    for (int r = tid; r < BM; r += NT) {
        const long m = m0 + r;
        long o = -1;
        if (m < M) {
            const int ox = (int)(m % P.OWt), oy = (int)((m / P.OWt) % P.OHt), img = (int)(m / ((long)P.OWt * P.OHt));
            o = (((long)img * P.OH + oy * P.ostep + P.o0y[phase]) * P.OW + ox * P.ostep + P.o0x[phase]) * P.Cout;
        }
        obase[r] = o;
    }
    const int aq = tid & 7;                            // which float4 of the 32-channel line
    int ay[AV], ax[AV], abase[AV];
#pragma unroll
    for (int j = 0; j < AV; ++j) {
        const int r = (tid >> 3) + j * (NT / 8);
        const long m = m0 + r;
        if (m < M) {
            const int ox = (int)(m % P.OWt), oy = (int)((m / P.OWt) % P.OHt), img = (int)(m / ((long)P.OWt * P.OHt));
            ay[j] = oy * P.istep; ax[j] = ox * P.istep;
            abase[j] = ((img * P.IH + ay[j]) * P.IW + ax[j]) * P.Cin + 4 * aq;
        } else {
            ay[j] = -(1 << 20); ax[j] = 0; abase[j] = 0;     // every tap out of range
        }
    }
    const __amdgpu_buffer_rsrc_t rin = make_rsrc(in, (long)P.N * P.IH * P.IW * P.Cin * 4);
    const size_t wplane = (size_t)P.ttotal * P.Cout * P.Cin;                 // elements per plane of Wp
    const int cchunks = P.Cin / BK;
    const int nstages = P.ntaps * cchunks;

    float4 ra[AV];
    uint4 rb[BV];
    auto gload = [&](int s) {
        const int t = s / cchunks, c0 = (s - t * cchunks) * BK;
        const int ty = P.ty[phase][t], tx = P.tx[phase][t], wt = P.wt[phase][t];
        const int toff = (ty * P.IW + tx) * P.Cin + c0;
#pragma unroll
        for (int j = 0; j < AV; ++j) {
            const int iy = ay[j] + ty, ix = ax[j] + tx;
            const bool ok = iy >= 0 && iy < P.IH && ix >= 0 && ix < P.IW;
            ra[j] = bufld4(rin, ok ? (unsigned)((abase[j] + toff) * 4) : kOOB);
        }
#pragma unroll
        for (int j = 0; j < BV; ++j) {
            const int e = tid + j * NT;                // piece index: plane-major, then row, then 16-byte quarter
            const int q = e & 3, row = (e >> 2) % BN, p = e / (4 * BN);
            const unsigned short *src = Wp + p * wplane + ((size_t)wt * P.Cout + n0 + row) * P.Cin + c0 + 8 * q;
            rb[j] = *reinterpret_cast<const uint4 *>(src);
        }
    };
    auto sstore = [&]() {
#pragma unroll
        for (int j = 0; j < AV; ++j) {
            const int r = (tid >> 3) + j * (NT / 8);
            const float xs[4] = {ra[j].x, ra[j].y, ra[j].z, ra[j].w};
            unsigned h[4], md[4], lo[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const unsigned u = __float_as_uint(xs[e]);
                h[e] = u;                                               // the pack takes the high half: truncation
                const float r1 = xs[e] - __uint_as_float(u & 0xffff0000u);
                md[e] = __float_as_uint(r1);
                lo[e] = __float_as_uint(r1 - __uint_as_float(md[e] & 0xffff0000u));
            }
            unsigned short *dst = As + r * LDR + 4 * aq;
            *reinterpret_cast<uint2 *>(dst) = make_uint2(pack_hi(h[1], h[0]), pack_hi(h[3], h[2]));
            *reinterpret_cast<uint2 *>(dst + BM * LDR) = make_uint2(pack_hi(md[1], md[0]), pack_hi(md[3], md[2]));
            *reinterpret_cast<uint2 *>(dst + 2 * BM * LDR) = make_uint2(pack_hi(lo[1], lo[0]), pack_hi(lo[3], lo[2]));
        }
#pragma unroll
        for (int j = 0; j < BV; ++j) {
            const int e = tid + j * NT;
            const int q = e & 3, row = (e >> 2) % BN, p = e / (4 * BN);
            *reinterpret_cast<uint4 *>(Bs + (p * BN + row) * LDR + 8 * q) = rb[j];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    gload(0);
    for (int s = 0; s < nstages; ++s) {
        __syncthreads();                               // the previous stage's fragment reads are done
        sstore();
        __syncthreads();
        if (s + 1 < nstages) gload(s + 1);             // in flight underneath the MFMAs below
        const unsigned short *ar = As + (wm * 64 + l31) * LDR + 8 * kh;
        const unsigned short *br = Bs + (wn * 64 + l31) * LDR + 8 * kh;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 fa[2][3], fb[2][3];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    fa[mi][p] = *reinterpret_cast<const bf16x8 *>(ar + (p * BM + mi * 32) * LDR + 16 * ks);
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    fb[ni][p] = *reinterpret_cast<const bf16x8 *>(br + (p * BN + ni * 32) * LDR + 16 * ks);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    f32x16 a = acc[mi][ni];            // small terms first
                    a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mi][1], fb[ni][1], a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mi][0], fb[ni][2], a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mi][2], fb[ni][0], a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mi][0], fb[ni][1], a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mi][1], fb[ni][0], a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mi][0], fb[ni][0], a, 0, 0, 0);
                    acc[mi][ni] = a;
                }
        }
    }

    // ---- epilogue: register r of lane l holds row (r&3) + 8*(r>>2) + 4*kh, column l31 of its 32x32 block
    const float sc = ep.scale ? ep.scale[0] : 1.f;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int ch = n0 + wn * 64 + ni * 32 + l31;
        const float bv = ep.bias ? ep.bias[ch] : 0.f;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                const long ob = obase[row];
                if (ob < 0) continue;
                const long o = ob + ch;
                float v = acc[mi][ni][r] * sc + bv;
                v = ep.dact ? v * act_bwd_from_out(ep.dact[ep.dact_index(o)], ep.act) : act_fwd(v, ep.act);
                out[o] = v;
            }
    }
}

// ------------------------------------------------------------------------------------------------
static int bfx_mode() {                  // MMDGAN_BFX=0 keeps every convolution on the fp32-MFMA kernels; 1 (default) uses the
    static int v = -1;                   // split-bf16 kernel where it measured faster; 2: wherever the shape is eligible
    if (v < 0) { const char *e = getenv("MMDGAN_BFX"); v = e ? atoi(e) : 1; }
    return v;
}

static bool bfx_shape_ok(const ConvDims &d, bool dgrad) {
    if (bfx_mode() == 0) return false;
    const int cin = dgrad ? d.K : d.C, cout = dgrad ? d.C : d.K;
    if (cin % bfx::BK || cout % 64 || d.R * d.R > bfx::MAXT) return false;
    if (d.stride == 1) return (d.R & 1) && d.pad == d.R / 2 && d.P == d.H && d.Q == d.W;
    if (d.stride == 2) return d.R == 4 && d.pad == 1 && d.H % 2 == 0 && d.W % 2 == 0;
    return false;
}
bool bfx_eligible(const ConvDims &d, bool dgrad) { return bfx_shape_ok(d, dgrad); }
size_t bfx_weight_bytes(const ConvDims &d) { return 3 * sizeof(unsigned short) * (size_t)d.R * d.R * d.C * d.K; }

int bfx_transform(const ConvDims &d, const float *w, bool dgrad, void *Wp, hipStream_t st) {
    const dim3 grid((d.K + 31) / 32, (d.C + 31) / 32, d.R * d.R);
    if (dgrad) hipLaunchKernelGGL(bfx_weight_kernel<true>, grid, dim3(256), 0, st, w, (unsigned short *)Wp, d.R * d.R, d.C, d.K);
    else hipLaunchKernelGGL(bfx_weight_kernel<false>, grid, dim3(256), 0, st, w, (unsigned short *)Wp, d.R * d.R, d.C, d.K);
    return check_launch("bfx_transform");
}

template <int WM, int WN>
static void bfx_launch_cfg(const bfx::Params &P, const ConvEpilogue &ep, const float *in, const void *Wp, float *out, hipStream_t st) {
    using C = bfx::Cfg<WM, WN>;
    const long M = (long)P.N * P.OHt * P.OWt;
    const dim3 grid((unsigned)((M + C::BM - 1) / C::BM), P.Cout / C::BN, P.nphase);
    static bool cap = false;
    if (!cap) {
        (void)hipFuncSetAttribute((const void *)bfx_conv_kernel<WM, WN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES);
        cap = true;
    }
    hipLaunchKernelGGL((bfx_conv_kernel<WM, WN>), grid, dim3(C::NT), C::LDS_BYTES, st, P, ep, in, (const unsigned short *)Wp, out);
}

static int bfx_launch(const ConvDims &d, const ConvEpilogue &ep, const float *in, const float *w, const void *Wp, float *out,
                      bool dgrad, hipStream_t st) {
    if (!Wp) {
        void *ws = workspace(bfx_weight_bytes(d));
        if (int rc = bfx_transform(d, w, dgrad, ws, st)) return rc;
        Wp = ws;
    }
    bfx::Params P;
    memset(&P, 0, sizeof(P));
    P.N = d.N; P.ttotal = d.R * d.R;
    if (!dgrad) {
        P.IH = d.H; P.IW = d.W; P.Cin = d.C; P.OHt = d.P; P.OWt = d.Q; P.OH = d.P; P.OW = d.Q; P.Cout = d.K;
        P.ostep = 1; P.istep = d.stride; P.nphase = 1; P.ntaps = d.R * d.R;
        for (int r = 0; r < d.R; ++r)
            for (int c = 0; c < d.R; ++c) {
                const int t = r * d.R + c;
                P.ty[0][t] = (signed char)(r - d.pad); P.tx[0][t] = (signed char)(c - d.pad); P.wt[0][t] = (signed char)t;
            }
    } else if (d.stride == 1) {
        P.IH = d.P; P.IW = d.Q; P.Cin = d.K; P.OHt = d.H; P.OWt = d.W; P.OH = d.H; P.OW = d.W; P.Cout = d.C;
        P.ostep = 1; P.istep = 1; P.nphase = 1; P.ntaps = d.R * d.R;
        for (int r = 0; r < d.R; ++r)
            for (int c = 0; c < d.R; ++c) {
                const int t = r * d.R + c;
                P.ty[0][t] = (signed char)(d.pad - r); P.tx[0][t] = (signed char)(d.pad - c); P.wt[0][t] = (signed char)t;
            }
    } else {                             // 4x4 stride 2 pad 1: phase (a, b) = output parity, 2x2 taps each
        P.IH = d.P; P.IW = d.Q; P.Cin = d.K; P.OHt = d.H / 2; P.OWt = d.W / 2; P.OH = d.H; P.OW = d.W; P.Cout = d.C;
        P.ostep = 2; P.istep = 1; P.nphase = 4; P.ntaps = 4;
        for (int a = 0; a < 2; ++a)
            for (int b = 0; b < 2; ++b) {
                const int ph = a * 2 + b;
                P.o0y[ph] = a; P.o0x[ph] = b;
                int t = 0;
                for (int r = (a + 1) & 1; r < 4; r += 2)
                    for (int c = (b + 1) & 1; c < 4; c += 2, ++t) {
                        P.ty[ph][t] = (signed char)((a + 1 - r) / 2); P.tx[ph][t] = (signed char)((b + 1 - c) / 2);
                        P.wt[ph][t] = (signed char)(r * 4 + c);
                    }
            }
    }
    const long M = (long)P.N * P.OHt * P.OWt;
    // the largest tile that still gives every CU two workgroups; 64x64 (one wave) for the small layers
    const long t128 = ((M + 127) / 128) * (P.Cout / 64) * P.nphase;
    static int force = -2;
    if (force == -2) { const char *e = getenv("MMDGAN_BFX_TILE"); force = e ? atoi(e) : -1; }
    int cfg = t128 >= 512 ? 1 : 0;
    if (P.Cout % 128 == 0 && ((M + 127) / 128) * (P.Cout / 128) * P.nphase >= 768) cfg = 2;
    if (force >= 0) cfg = force;
    if (cfg == 2 && P.Cout % 128) cfg = 1;
    if (cfg == 2) bfx_launch_cfg<2, 2>(P, ep, in, Wp, out, st);
    else if (cfg == 1) bfx_launch_cfg<2, 1>(P, ep, in, Wp, out, st);
    else bfx_launch_cfg<1, 1>(P, ep, in, Wp, out, st);
    return check_launch(dgrad ? "conv2d_dgrad(bf16x6)" : "conv2d_fwd(bf16x6)");
}

bool bfx_fwd_ok(const ConvDims &d) { return bfx_shape_ok(d, false) && workspace(bfx_weight_bytes(d)) != nullptr; }
bool bfx_dgrad_ok(const ConvDims &d) { return bfx_shape_ok(d, true) && workspace(bfx_weight_bytes(d)) != nullptr; }
int bfx_fwd(const ConvDims &d, const ConvEpilogue &ep, const float *x, const float *w, const void *Wp, float *y, hipStream_t st) {
    return bfx_launch(d, ep, x, w, Wp, y, false, st);
}
int bfx_dgrad(const ConvDims &d, const ConvEpilogue &ep, const float *dy, const float *w, const void *Wp, float *dx, hipStream_t st) {
    return bfx_launch(d, ep, dy, w, Wp, dx, true, st);
}

}  // namespace mmdgan
