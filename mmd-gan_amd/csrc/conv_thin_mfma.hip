// MFMA versions of the thin first/last layers (D l1: 3 -> 64 channels, G l5: 64 -> 3; 3x3, stride 1)
// and their gradients.  As plain implicit GEMMs these layers waste an MFMA tile (im2col depth 27, or
// 3 output columns), so the first implementation (conv_thin.hip) used the vector ALU - and ran at a
// quarter of the HBM rate because every FMA costs an issue slot.  Regrouped, each of them is a GEMM
// whose small dimension is J = taps x thin channels = 27 (padded to 32), which fills a
// v_mfma_f32_32x32x2_f32 tile to 84 %:
//
//   n2w  "narrow to wide"  out[px][ch] = sum_j patch[px][j] * Wm[j][ch]        J = reduction
//        (D l1 forward, G l5 input-gradient)   M = 32 wide channels, N = 32 pixels, K = J;
//        Wm lives in registers, the patch operand is gathered straight from global/L1 (the thin
//        tensor is tiny and every element is reused 9x), the epilogue writes float4 channel groups.
//   w2n  "wide to narrow"  T[px'][j] = sum_c in[px'][c] * Wm[c][j];  out[px][n] = sum_tap T[px+tap][tap,n]
//        (G l5 forward, D l1 input-gradient)   M = J, N = 32 pixels, K = wide channels; a workgroup
//        owns a band of image rows (+ halo), T goes through LDS, then a 9-term gather-sum.
//   wgrad  dW[j][ch] = sum_px patch[px][j] * wide[px][ch]                      M = J, N = 32 wide channels,
//        K = pixels; a wave walks image rows two pixels per MFMA, partial sums per workgroup go to the
//        library workspace and are combined in fixed order (deterministic).
//
// All three are then HBM-bound on the wide tensor (33.5 MB for D l1 at batch 128).
// Round 4, measured and NOT kept (tools/bench_conv.py, CIFAR batch 64, us): n2w with the operands swapped - MFMA rows = pixels,
// so a lane owns one channel and every accumulator register leaves as two whole 128-byte lines, no LDS transpose, no address
// arithmetic - D l1 forward 20.7 -> 22.6, G l5 input-gradient 15.0 -> 20.8: 32 dword stores per tile instead of 8 float4
// stores cost more in the vector-memory instruction path than the LDS round trip and its ~200 VALU instructions saved;
// w2n with eight waves per band (one tile each) instead of four waves of two: G l5 forward 15.6 -> 18.3, D l1
// input-gradient 19.0 -> 21.9; more n2w workgroups (768 / 1024 instead of 512): unchanged.  The whole step did not move in
// any of the three (1.891-1.896 ms).
// Layout of v_mfma_f32_32x32x2_f32 (wave64): A lane l -> row l%32, k = l/32; B lane l -> col l%32,
// k = l/32; D register r of lane l -> row (r&3) + 8*(r>>2) + 4*(l/32), col l%32.
#include "conv_internal.h"
#include "bufload.h"

// Ablation hooks (tools/thin_ablate.sh: what each part of the n2w kernel costs): a compile-time bit mask, 0 in the library build.
// 1: the wide tensor's stores, 2: the patch gathers, 4: all MFMAs but the first of a tile (a VALU FMA keeps the operands live).
// Round 5, D l1 forward at batch 128 (33.5 MB written), graph replay: 20.7 us shipped; 15.2 without the stores, 19.8 without the
// gathers, 17.5 without the MFMAs, 11.1 without all three - launch, weight prologue, the LDS transposes and the epilogue's
// arithmetic of two tiles per wave.  The stores' 5.5 us are 6 TB/s: the kernel is bound by its fixed latencies at this size
// (131 072 pixels), not by the wide tensor's bandwidth (profiles/r05_thin_ablation.txt).
#ifndef THIN_ABLATE
#define THIN_ABLATE 0
#endif

namespace mmdgan {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kJP = 16;                  // MFMA k-pairs / rows for J <= 32

// offset of tap index along one axis: forward-type gathers read in[p + r - pad], gradient-type
// gathers read in[p + pad - r]
template <bool FLIP>
__device__ __forceinline__ int tap_off(int r, int pad) { return FLIP ? pad - r : r - pad; }

// per patch column j = (tap, thin channel): byte offset of the gathered element relative to the
// output pixel, the tap's bit in the validity mask and where its weights start - filled on the host
// (32 x 2 runtime divisions per wave cost more than the wave's 32 MFMAs)
struct PatchTab {
    int delta[32];
    unsigned tbit[32];
    int woff[32];
};

template <bool FLIP>
static PatchTab make_patch_tab(const ConvDims &d, int Cn) {
    PatchTab t;
    const int J = d.R * d.R * Cn;
    for (int j = 0; j < 32; ++j) {
        const bool used = j < J;
        const int jj = used ? j : 0;
        const int tap = jj / Cn, cn = jj - tap * Cn, r = tap / d.R, c = tap - r * d.R;
        const int dh = FLIP ? d.pad - r : r - d.pad, dw = FLIP ? d.pad - c : c - d.pad;
        t.delta[j] = ((dh * d.W + dw) * Cn + cn) * 4;
        t.tbit[j] = used ? 1u << tap : 0u;
        t.woff[j] = used ? (FLIP ? tap * d.C * d.K + cn : jj * d.K) : -1;
    }
    return t;
}

// ------------------------------------------------------------------------------------------------
// n2w.  FLIP = false: forward (in = x [N,H,W,Cn=C], Wm[j][ch] = w[j*K + ch]);
//       FLIP = true : input-gradient (in = dy [N,H,W,Cn=K], Wm[(tap,k)][c] = w[(tap*C + c)*K + k])
template <bool FLIP, int WB>
__global__ __launch_bounds__(256) void thinm_n2w_kernel(ConvDims d, ConvEpilogue ep, const float *__restrict__ in,
                                                        const float *__restrict__ w, float *__restrict__ out,
                                                        int ntiles, PatchTab tab) {
    constexpr int Wd = WB * 32;
    const int Cn = FLIP ? d.K : d.C;
    const int R = d.R, IH = d.H, IW = d.W;
    const long M = (long)d.N * IH * IW;
    const int lane = threadIdx.x & 63, l31 = lane & 31, kh = lane >> 5, wave = threadIdx.x >> 6;
    float a[WB][kJP];
    int delta[kJP];
    unsigned tbit[kJP];
#pragma unroll
    for (int jp = 0; jp < kJP; ++jp) {
        delta[jp] = kh ? tab.delta[2 * jp + 1] : tab.delta[2 * jp];
        tbit[jp] = kh ? tab.tbit[2 * jp + 1] : tab.tbit[2 * jp];
        const int wo = kh ? tab.woff[2 * jp + 1] : tab.woff[2 * jp];
#pragma unroll
        for (int wb = 0; wb < WB; ++wb) {
            const int ch = wb * 32 + l31;
            a[wb][jp] = wo >= 0 ? w[wo + (FLIP ? ch * d.K : ch)] : 0.f;
        }
    }
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(in, M * Cn * 4);
    const float sc = ep.scale ? ep.scale[0] : 1.f;
    // the patch gather of tile t+1 is in flight while tile t is multiplied and stored: a wave walks several tiles
    // (launch_n2w sizes the grid for ~4), so the weight prologue and the gather latency are paid once, not per tile
    auto gather = [&](int tile, float (&bq)[kJP]) {
        const long m = (long)tile * 32 + l31;
        const bool ok = tile < ntiles && m < M;
        const int mm = ok ? (int)m : 0;
        const int wq = mm % IW, hq = (mm / IW) % IH;
        unsigned cols = 0, mask = 0;
        for (int t = 0; t < R; ++t) {
            const int x = wq + tap_off<FLIP>(t, d.pad);
            cols |= (unsigned)(x >= 0 && x < IW) << t;
        }
        for (int r = 0; r < R; ++r) {
            const int y = hq + tap_off<FLIP>(r, d.pad);
            mask |= (y >= 0 && y < IH) ? cols << (r * R) : 0u;
        }
        if (!ok) mask = 0;
        const unsigned pixbase = (unsigned)(mm * Cn * 4);
#pragma unroll
        for (int jp = 0; jp < kJP; ++jp)
            bq[jp] = (THIN_ABLATE & 2) ? (float)(mask & tbit[jp]) : bufld1(rs, (mask & tbit[jp]) ? pixbase + (unsigned)delta[jp] : kOOB);
    };
    const int tstep = gridDim.x * 4;
    float bnext[kJP];
    gather(blockIdx.x * 4 + wave, bnext);
    for (int tile = blockIdx.x * 4 + wave; tile < ntiles; tile += tstep) {
        const long m = (long)tile * 32 + l31;
        const bool ok = m < M;
        float b[kJP];
#pragma unroll
        for (int jp = 0; jp < kJP; ++jp) b[jp] = bnext[jp];
        gather(tile + tstep, bnext);
        f32x16 acc[WB];
#pragma unroll
        for (int wb = 0; wb < WB; ++wb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[wb][r] = 0.f;
#pragma unroll
        for (int jp = 0; jp < kJP; ++jp)
#pragma unroll
            for (int wb = 0; wb < WB; ++wb)
                if (!(THIN_ABLATE & 4) || jp == 0) acc[wb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[wb][jp], b[jp], acc[wb], 0, 0, 0);
                else acc[wb][jp] += a[wb][jp] * b[jp];
        if constexpr (WB <= 2) {
            // The wide tensor is what this kernel moves (33.5 MB for D l1 at batch 128): its stores decide the time.
            // A lane holds one PIXEL's channel quads, so direct stores put 16 bytes at a 256-byte stride per lane - every
            // 128-byte line is written by four different instructions (18 % of the HBM rate, profiles/r01_conv_layers.txt).
            // Transposed through a wave-private LDS slab [32 pixels][Wd + 4] the same data leaves as whole pixel rows:
            // lane L stores the float4 of channel quad L % (Wd/4) of pixel L / (Wd/4) - 1 KB contiguous per instruction.
            constexpr int LDP = Wd + 4;                    // +4 floats: the 16 pixel rows of a b128 phase cover all 64 banks
            __shared__ __attribute__((aligned(16))) float slab[4][32 * LDP];
            float *sl = slab[wave];
#pragma unroll
            for (int wb = 0; wb < WB; ++wb)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<float4 *>(sl + l31 * LDP + wb * 32 + 8 * g + 4 * kh) =
                        make_float4(acc[wb][4 * g], acc[wb][4 * g + 1], acc[wb][4 * g + 2], acc[wb][4 * g + 3]);
            __builtin_amdgcn_wave_barrier();               // same wave, in-order LDS queue: a scheduling fence is enough
            asm volatile("" ::: "memory");
            constexpr int QP = Wd / 4, PPI = 64 / QP;      // lanes per pixel row, pixels per store instruction
            const int cq = lane % QP, pp = lane / QP, ch = 4 * cq;
            float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ep.bias) bv = *reinterpret_cast<const float4 *>(ep.bias + ch);
            const long m0 = (long)tile * 32;
#pragma unroll
            for (int it = 0; it < 32 / PPI; ++it) {
                const int px = it * PPI + pp;
                const long mo = m0 + px;
                float4 v = *reinterpret_cast<const float4 *>(sl + px * LDP + ch);
                if (mo < M) {
                    const long o = mo * Wd + ch;
                    v.x = v.x * sc + bv.x; v.y = v.y * sc + bv.y; v.z = v.z * sc + bv.z; v.w = v.w * sc + bv.w;
                    if (ep.dact) {
                        const float4 y = *reinterpret_cast<const float4 *>(ep.dact + ep.dact_index(o));
                        v.x *= act_bwd_from_out(y.x, ep.act); v.y *= act_bwd_from_out(y.y, ep.act);
                        v.z *= act_bwd_from_out(y.z, ep.act); v.w *= act_bwd_from_out(y.w, ep.act);
                    } else {
                        v.x = act_fwd(v.x, ep.act); v.y = act_fwd(v.y, ep.act);
                        v.z = act_fwd(v.z, ep.act); v.w = act_fwd(v.w, ep.act);
                    }
                    if (!(THIN_ABLATE & 1) || v.x == 123.456f) *reinterpret_cast<float4 *>(out + o) = ep.add4(v, o);
                }
            }
            __builtin_amdgcn_wave_barrier();
            asm volatile("" ::: "memory");
            continue;
        }
        if (!ok) continue;
        // (128 wide channels: the slab would not fit the default LDS cap) lane = one pixel; registers 4g..4g+3 are 4
        // consecutive channels -> float4 stores
#pragma unroll
        for (int wb = 0; wb < WB; ++wb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int ch = wb * 32 + 8 * g + 4 * kh;
                const long o = m * Wd + ch;
                float4 v = make_float4(acc[wb][4 * g] * sc, acc[wb][4 * g + 1] * sc, acc[wb][4 * g + 2] * sc, acc[wb][4 * g + 3] * sc);
                if (ep.bias) {
                    const float4 bv = *reinterpret_cast<const float4 *>(ep.bias + ch);
                    v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
                }
                if (ep.dact) {
                    const float4 y = *reinterpret_cast<const float4 *>(ep.dact + ep.dact_index(o));
                    v.x *= act_bwd_from_out(y.x, ep.act); v.y *= act_bwd_from_out(y.y, ep.act);
                    v.z *= act_bwd_from_out(y.z, ep.act); v.w *= act_bwd_from_out(y.w, ep.act);
                } else {
                    v.x = act_fwd(v.x, ep.act); v.y = act_fwd(v.y, ep.act);
                    v.z = act_fwd(v.z, ep.act); v.w = act_fwd(v.w, ep.act);
                }
                *reinterpret_cast<float4 *>(out + o) = ep.add4(v, o);
            }
    }
}

// ------------------------------------------------------------------------------------------------
// w2n.  FLIP = false: forward (in = x [N,H,W,CW=C], out = y [N,H,W,Cn=K], Wm[c][(tap,k)] = w[(tap*C + c)*K + k]);
//       FLIP = true : input-gradient (in = dy [N,H,W,CW=K], out = dx [N,H,W,Cn=C], Wm[k][(tap,c)] = w[(tap*C + c)*K + k])
// One workgroup = one band of RB output rows of one image; T covers the band plus R-1 halo rows.
template <bool FLIP, int CW>
__global__ __launch_bounds__(256) void thinm_w2n_kernel(ConvDims d, ConvEpilogue ep, const float *__restrict__ in,
                                                        const float *__restrict__ w, float *__restrict__ out, int RB,
                                                        int bands, int ldp) {
    extern __shared__ __attribute__((aligned(16))) float T[];       // [J][ldp]
    constexpr int CP = CW / 2;
    const int Cn = FLIP ? d.C : d.K;
    const int R = d.R, J = R * R * Cn, IH = d.H, IW = d.W;
    const int n = blockIdx.x / bands, band = blockIdx.x - n * bands;
    const int h0 = band * RB;
    const int rows_out = min(RB, IH - h0);
    const int lo = FLIP ? d.pad - (R - 1) : -d.pad;                 // smallest row offset of a tap
    const int trows = rows_out + R - 1, tpix = trows * IW, tr0 = h0 + lo;
    const int lane = threadIdx.x & 63, l31 = lane & 31, kh = lane >> 5, wave = threadIdx.x >> 6;
    float a[CP];
    {
        const bool used = l31 < J;
        const int j = used ? l31 : 0;
        const int tap = j / Cn, cn = j - tap * Cn;
#pragma unroll
        for (int cp = 0; cp < CP; ++cp) {
            const int c = kh * CP + cp;                              // this lane's k index <-> wide channel
            const float v = FLIP ? w[((long)tap * d.C + cn) * d.K + c] : w[((long)tap * d.C + c) * d.K + cn];
            a[cp] = used ? v : 0.f;
        }
    }
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(in, (long)d.N * IH * IW * CW * 4);
    const int ntile = (tpix + 31) / 32;
    for (int tile = wave; tile < ntile; tile += 4) {
        const int pl = tile * 32 + l31;
        const int row = tr0 + pl / IW, col = pl % IW;
        const bool ok = pl < tpix && row >= 0 && row < IH;
        const unsigned base = (unsigned)((((n * IH + row) * IW + col) * CW + kh * CP) * 4);
        float b[CP];
#pragma unroll
        for (int q = 0; q < CP / 4; ++q) {
            const float4 v = bufld4(rs, ok ? base + q * 16 : kOOB);
            b[4 * q] = v.x; b[4 * q + 1] = v.y; b[4 * q + 2] = v.z; b[4 * q + 3] = v.w;
        }
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int cp = 0; cp < CP; ++cp) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cp], b[cp], acc, 0, 0, 0);
        if (pl < tpix) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (j < J) T[j * ldp + pl] = acc[r];
            }
        }
    }
    __syncthreads();
    const float sc = ep.scale ? ep.scale[0] : 1.f;
    const int npo = rows_out * IW;
    for (int idx = threadIdx.x; idx < npo * Cn; idx += 256) {
        const int cn = idx / npo, px = idx - cn * npo;
        const int ro = px / IW, wo = px - ro * IW;
        float v = 0.f;
        for (int r = 0; r < R; ++r) {
            const int trow = ro + tap_off<FLIP>(r, d.pad) - lo;
            for (int t = 0; t < R; ++t) {
                const int c = wo + tap_off<FLIP>(t, d.pad);
                if (c >= 0 && c < IW) v += T[((r * R + t) * Cn + cn) * ldp + trow * IW + c];
            }
        }
        const long o = (((long)n * IH + h0 + ro) * IW + wo) * Cn + cn;
        out[o] = ep.apply(v * sc, cn, o);
    }
}

// ------------------------------------------------------------------------------------------------
// wgrad.  FLIP = false: thin input channels (D l1): patch from x [N,H,W,Cn=C], wide = dy [N,H,W,K],
//                       dW index j*K + ch;
//         FLIP = true : thin output channels (G l5): patch from dy [N,H,W,Cn=K] (taken at px - tap),
//                       wide = x [N,H,W,C], dW index (tap*C + ch)*K + k.
template <bool FLIP, int WB>
__global__ __launch_bounds__(256) void thinm_wgrad_kernel(ConvDims d, const float *__restrict__ narrow,
                                                          const float *__restrict__ wide, float *__restrict__ partials,
                                                          int rows_per_wave) {
    __shared__ float red[4][WB * 16 * 64];
    constexpr int Wd = WB * 32, U = 8;
    const int Cn = FLIP ? d.K : d.C;
    const int R = d.R, J = R * R * Cn, IH = d.H, IW = d.W;
    const int nrows = d.N * IH;
    const int lane = threadIdx.x & 63, l31 = lane & 31, kh = lane >> 5, wave = threadIdx.x >> 6;
    // this lane's patch column j = l31
    const bool used = l31 < J;
    const int jj = used ? l31 : 0;
    const int tap = jj / Cn, cn = jj - tap * Cn, tr = tap / R, tt = tap - tr * R;
    const int dh = tap_off<FLIP>(tr, d.pad), dw = tap_off<FLIP>(tt, d.pad);
    const __amdgpu_buffer_rsrc_t rn = make_rsrc(narrow, (long)nrows * IW * Cn * 4);
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(wide, (long)nrows * IW * Wd * 4);
    f32x16 acc[WB];
#pragma unroll
    for (int wb = 0; wb < WB; ++wb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[wb][r] = 0.f;
    const int row0 = (blockIdx.x * 4 + wave) * rows_per_wave;
    for (int row = row0; row < min(nrows, row0 + rows_per_wave); ++row) {
        const int h = row % IH;
        const bool rowok = used && h + dh >= 0 && h + dh < IH;
        // narrow: (row + dh, w + dw, cn), w = 2i + kh;  wide: (row, w, wb*32 + l31)
        const int nbase = (((row + dh) * IW + dw + kh) * Cn + cn) * 4;
        const unsigned wbase = (unsigned)(((row * IW + kh) * Wd + l31) * 4);
        for (int i0 = 0; i0 < IW / 2; i0 += U) {
            float pa[U], pb[WB][U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = i0 + u, c = 2 * i + kh + dw;
                const bool ok = rowok && i < IW / 2 && c >= 0 && c < IW;
                pa[u] = bufld1(rn, ok ? (unsigned)(nbase + i * 2 * Cn * 4) : kOOB);
#pragma unroll
                for (int wb = 0; wb < WB; ++wb)
                    pb[wb][u] = bufld1(rw, i < IW / 2 ? wbase + (unsigned)((i * 2 * Wd + wb * 32) * 4) : kOOB);
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int wb = 0; wb < WB; ++wb) acc[wb] = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[u], pb[wb][u], acc[wb], 0, 0, 0);
        }
    }
    // sum the 4 waves (lane-major parking: conflict-free), then one partial per workgroup
#pragma unroll
    for (int wb = 0; wb < WB; ++wb)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wave][(wb * 16 + r) * 64 + lane] = acc[wb][r];
    __syncthreads();
    const long nout = (long)R * R * d.C * d.K;
    float *dst = partials + (long)blockIdx.x * nout;
    for (int e = threadIdx.x; e < WB * 16 * 64; e += 256) {
        const int ln = e & 63, r = (e >> 6) & 15, wb = e >> 10;
        const int j = (r & 3) + 8 * (r >> 2) + 4 * (ln >> 5), ch = wb * 32 + (ln & 31);
        if (j >= J) continue;
        const float v = red[0][e] + red[1][e] + red[2][e] + red[3][e];
        const int jt = j / Cn, jc = j - jt * Cn;
        dst[FLIP ? ((long)jt * d.C + ch) * d.K + jc : (long)j * d.K + ch] = v;
    }
}

// ------------------------------------------------------------------------------------------------
static bool common_ok(const ConvDims &d) { return d.stride == 1 && (d.R & 1) && d.R <= 5 && d.pad == d.R / 2 && d.P == d.H && d.Q == d.W; }
static bool wide_ok(int c) { return c == 32 || c == 64 || c == 128; }

bool thinm_fwd_n2w_ok(const ConvDims &d) { return common_ok(d) && d.R * d.R * d.C <= 32 && wide_ok(d.K); }
bool thinm_dgrad_n2w_ok(const ConvDims &d) { return common_ok(d) && d.R * d.R * d.K <= 32 && wide_ok(d.C); }
static bool band_fits(const ConvDims &d, int narrow) { return sizeof(float) * (size_t)d.R * d.R * narrow * d.R * d.W <= 60 * 1024; }
bool thinm_fwd_w2n_ok(const ConvDims &d) { return common_ok(d) && d.R * d.R * d.K <= 32 && wide_ok(d.C) && band_fits(d, d.K); }
bool thinm_dgrad_w2n_ok(const ConvDims &d) { return common_ok(d) && d.R * d.R * d.C <= 32 && wide_ok(d.K) && band_fits(d, d.C); }
bool thinm_wgrad_ok(const ConvDims &d) {
    return common_ok(d) && d.W % 2 == 0 &&
           ((d.R * d.R * d.C <= 32 && (d.K == 32 || d.K == 64)) || (d.R * d.R * d.K <= 32 && (d.C == 32 || d.C == 64)));
}

template <bool FLIP>
static void launch_n2w(const ConvDims &d, const ConvEpilogue &ep, const float *in, const float *w, float *out, int wide,
                       hipStream_t st) {
    const long M = (long)d.N * d.H * d.W;
    const int ntiles = (int)((M + 31) / 32);
    int blocks = (ntiles + 3) / 4;
    if (blocks > 512) blocks = 512;      // two workgroups per CU, each wave walks its tiles (768 / 1024: measured the same)
    const PatchTab tab = make_patch_tab<FLIP>(d, FLIP ? d.K : d.C);
    if (wide == 32) hipLaunchKernelGGL((thinm_n2w_kernel<FLIP, 1>), dim3(blocks), dim3(256), 0, st, d, ep, in, w, out, ntiles, tab);
    else if (wide == 64) hipLaunchKernelGGL((thinm_n2w_kernel<FLIP, 2>), dim3(blocks), dim3(256), 0, st, d, ep, in, w, out, ntiles, tab);
    else hipLaunchKernelGGL((thinm_n2w_kernel<FLIP, 4>), dim3(blocks), dim3(256), 0, st, d, ep, in, w, out, ntiles, tab);
}

template <bool FLIP>
static void launch_w2n(const ConvDims &d, const ConvEpilogue &ep, const float *in, const float *w, float *out, int wide,
                       int narrow, hipStream_t st) {
    const int target = d.W <= 32 ? 256 : 512;                       // pixels of T per workgroup
    int RB = target / d.W - (d.R - 1);
    if (RB < 1) RB = 1;
    if (RB > d.H) RB = d.H;
    while (RB > 1 && sizeof(float) * (size_t)d.R * d.R * narrow * (RB + d.R - 1) * d.W > 60 * 1024) --RB;   // default LDS cap
    const int bands = (d.H + RB - 1) / RB;
    const int ldp = (RB + d.R - 1) * d.W;
    const size_t lds = sizeof(float) * (size_t)d.R * d.R * narrow * ldp;
    const dim3 grid(d.N * bands);
    if (wide == 32) {
        hipLaunchKernelGGL((thinm_w2n_kernel<FLIP, 32>), grid, dim3(256), lds, st, d, ep, in, w, out, RB, bands, ldp);
    } else if (wide == 64) {
        hipLaunchKernelGGL((thinm_w2n_kernel<FLIP, 64>), grid, dim3(256), lds, st, d, ep, in, w, out, RB, bands, ldp);
    } else {
        hipLaunchKernelGGL((thinm_w2n_kernel<FLIP, 128>), grid, dim3(256), lds, st, d, ep, in, w, out, RB, bands, ldp);
    }
}

int thinm_fwd(const ConvDims &d, const ConvEpilogue &ep, const float *x, const float *w, float *y, hipStream_t st) {
    if (thinm_fwd_n2w_ok(d)) launch_n2w<false>(d, ep, x, w, y, d.K, st);
    else launch_w2n<false>(d, ep, x, w, y, d.C, d.K, st);
    addend_applied();                                   // n2w: add4 at its stores; w2n: ConvEpilogue::apply
    return check_launch("conv2d_fwd(thin mfma)");
}

int thinm_dgrad(const ConvDims &d, const ConvEpilogue &ep, const float *dy, const float *w, float *dx, hipStream_t st) {
    if (thinm_dgrad_n2w_ok(d)) launch_n2w<true>(d, ep, dy, w, dx, d.C, st);
    else launch_w2n<true>(d, ep, dy, w, dx, d.K, d.C, st);
    addend_applied();
    return check_launch("conv2d_dgrad(thin mfma)");
}

// needs the library workspace for the per-workgroup partial sums; returns 1 if it is not available
// (the caller then falls back to the VALU kernel)
int thinm_wgrad(const ConvDims &d, const float *x, const float *dy, float *dw, hipStream_t st, const float *wdot, float *dot) {
    const long nout = (long)d.R * d.R * d.C * d.K;
    const int nrows = d.N * d.H;
    int blocks = (nrows + 3) / 4;
    if (blocks > 256) blocks = 256;
    const int rpw = (nrows + blocks * 4 - 1) / (blocks * 4);
    blocks = (nrows + rpw * 4 - 1) / (rpw * 4);
    if (d.N == 1) return 1;                       // batch-1 (power iteration, concurrent chains): never the shared workspace
    float *partials = (float *)workspace_acquire(sizeof(float) * nout * blocks, st);
    if (!partials) return 1;
    const bool thin_in = d.R * d.R * d.C <= 32 && (d.K == 32 || d.K == 64);
    if (thin_in) {
        if (d.K == 32) hipLaunchKernelGGL((thinm_wgrad_kernel<false, 1>), dim3(blocks), dim3(256), 0, st, d, x, dy, partials, rpw);
        else hipLaunchKernelGGL((thinm_wgrad_kernel<false, 2>), dim3(blocks), dim3(256), 0, st, d, x, dy, partials, rpw);
    } else {
        if (d.C == 32) hipLaunchKernelGGL((thinm_wgrad_kernel<true, 1>), dim3(blocks), dim3(256), 0, st, d, dy, x, partials, rpw);
        else hipLaunchKernelGGL((thinm_wgrad_kernel<true, 2>), dim3(blocks), dim3(256), 0, st, d, dy, x, partials, rpw);
    }
    return thin_wgrad_reduce(partials, blocks, nout, dw, st, wdot, dot);
}

}  // namespace mmdgan
