// Winograd F(2x2, 3x3) convolution for the 3x3 / stride-1 layers of the discriminator (D l3, l5, l7:
// 43 % of the step's FLOPs), forward and input-gradient.  Why: the fp32 MFMA pipe is the bound and,
// under it, the clock (1.8-1.9 GHz on real data: power) - the direct implicit GEMM already runs at
// 83-89 % of what the pipe delivers at that clock, so the remaining lever is fewer MFMAs per output:
// 16 multiplies per 2x2 output tile instead of 36 (2.25x).
//
//   y = A^T [ sum_c (G g G^T) (.) (B^T d B) ] A      g: 3x3 filter, d: 4x4 input patch, y: 2x2 outputs
//
// As GEMMs: for each of the 16 "frequencies" f, M_f[tile][k] = sum_c V_f[tile][c] * U_f[c][k].
//   * U = G g G^T is computed once per launch by wino_weight_kernel into the library workspace
//     ([16][C][K]; for the input-gradient the taps are flipped and the channel roles swapped).
//   * one workgroup = 32 tiles (MFMA rows) x BN output channels, ALL 16 frequencies: wave w owns
//     f = 4w..4w+3 (i = w, j = 0..3), i.e. 4 x BN/32 accumulators of v_mfma_f32_32x32x2_f32.
//   * the input transform is fused into the load path: a thread owns row r of one tile for four
//     channels (4 float4 loads per stage; padding = buffer-load range check baked into the offsets),
//     does the row pass in registers, fetches the one other row it needs with a quad-permute DPP
//     move and writes its 16 values of V to LDS.  A stage is 8 input channels: 32 MFMAs per wave.
//   * the output transform is the epilogue: the j direction inside each wave's registers, the i
//     direction across the 4 waves through LDS, then bias / SN scale / activation (or activation
//     derivative with the 3B-row wrap) and coalesced stores - same ConvEpilogue as the direct kernels.
// Numerics: all transform constants are 0, +-1, +-1/2 (exact); the result differs from the direct
// kernel by fp32 rounding of a different summation order (measured <= 2e-6 relative on the parity cases).
#include "conv_internal.h"
#include "bufload.h"
#include "wino_weight.h"

namespace mmdgan {

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace wino {
constexpr int BC = 8;                  // reduction channels per stage = 4 MFMA k-pairs: 32 MFMAs per wave between barriers
constexpr int ROW = 32 * 4 + 4;        // V: floats per (frequency, k half) row = [32 tiles][4 k-pairs] + 4
constexpr int FSV = 2 * ROW;           // V: floats per frequency
constexpr int IPAD = 8;                // ... + 8 floats per frequency ROW i = f >> 2: the four patch rows of a producer wave's
                                       // store (f = 4 pr + j) land 8 banks apart, every offset stays 16-byte aligned
__host__ __device__ constexpr int voff(int f) { return f * FSV + (f >> 2) * IPAD; }
static_assert((voff(4) - voff(0)) % 32 == 8 && voff(1) % 4 == 0, "V frequency stride");
constexpr int V_FLOATS = voff(16);     // one stage of transformed activations
template <int BN>
struct Cfg {
    static constexpr int NCB = BN / 32;
    static constexpr int ZS_FLOATS = 4 * 2 * 32 * 32;        // epilogue exchange buffer
    static constexpr int SMEM_FLOATS = 2 * V_FLOATS > ZS_FLOATS ? 2 * V_FLOATS : ZS_FLOATS;
    static constexpr size_t LDS_BYTES = sizeof(float) * SMEM_FLOATS + sizeof(long) * 32;   // + output offset of each tile
};
}  // namespace wino

template <bool FLIP>
__global__ __launch_bounds__(256) void wino_weight_kernel(const float *__restrict__ w, float *__restrict__ U, int C, int K) {
    __shared__ float tile[8][32][33];                    // (layout of U: wino_weight.h)
    wino_weight_block<FLIP>(tile, blockIdx.x, blockIdx.y, w, U, C, K);
}

// x [N,H,W,Cr] (*) U [16][Cr][Ko] -> out [N,H,W,Ko], 'SAME' padding, stride 1
// Only V goes through LDS: wave w is the sole consumer of U[4w..4w+3], so its B fragments (one dword per
// lane, 128 contiguous bytes per half-wave) are loaded from L2 straight into the MFMA operand registers,
// each refilled for the next stage right after the MFMA that consumed it (measured on the D l3 shape:
// staging U through LDS cost 10 of 60 us).
// SPLIT: blockIdx.z takes a slice of the channel reduction and writes its A^T M A (the transform is linear) into slab
// blockIdx.z of `out` (= the library workspace then, slab_elems apart); slab_epilogue() sums the slabs and applies
// scale / bias / activation.  For launches whose tile count alone cannot fill the chip (D l7 forward at batch 128:
// 16 x 8 workgroups on 256 CUs).  No zeroing, no atomics: the same bits every run.
#ifndef WINO_WAVES
#define WINO_WAVES 2
#endif
template <int BN, bool SPLIT>
__global__ __launch_bounds__(256, WINO_WAVES) void wino_kernel(int N, int H, int W, int Cr, int Ko, ConvEpilogue ep,
                                                      const float *__restrict__ x, const float *__restrict__ U,
                                                      float *__restrict__ out, int stages_per_split, long slab_elems) {
    using Cf = wino::Cfg<BN>;
    constexpr int BC = wino::BC, ROW = wino::ROW, FSV = wino::FSV, NCB = Cf::NCB, VF = wino::V_FLOATS;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, kh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int TH = H >> 1, TW = W >> 1;
    const long T = (long)N * TH * TW;
    const int t0 = blockIdx.x * 32, n0 = blockIdx.y * BN;
    // ---- producer: thread = (tile pt, channel quad pp, patch row pr); pr in the low lane bits (DPP quad)
    const int pt = tid >> 3, pp = (tid >> 2) & 1, pr = tid & 3;
    unsigned xoff[4];
    {
        const long id = (long)t0 + pt;
        const bool ok = id < T;
        const long ii = ok ? id : 0;
        const int tx = ii % TW, ty = (ii / TW) % TH, n = ii / ((long)TW * TH);
        const int y = 2 * ty - 1 + pr;
        const bool rowok = ok && y >= 0 && y < H;
        if ((tid & 7) == 0)      // element offset of output pixel (2ty, 2tx), channel 0, for the epilogue (-1: no such tile)
            reinterpret_cast<long *>(smem + Cf::SMEM_FLOATS)[pt] = ok ? (((long)n * H + 2 * ty) * W + 2 * tx) * Ko : -1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int xx = 2 * tx - 1 + j;
            xoff[j] = (rowok && xx >= 0 && xx < W) ? (unsigned)(((((long)n * H + y) * W + xx) * Cr + 4 * pp) * 4) : kOOB;
        }
    }
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(x, (long)N * H * W * Cr * 4);
    const __amdgpu_buffer_rsrc_t ru = make_rsrc(U, (long)16 * Cr * Ko * 4);
    const float sa = pr == 3 ? -1.f : 1.f, sb = (pr & 1) ? 1.f : -1.f;      // V[i=pr] = sa * X[pr] + sb * X[other]
    // V[f][k half][tile][4 k-pairs]: channel 4 pp + e of the stage -> k half e & 1, k-pair 2 pp + (e >> 1); f = 4 pr + j
    const int vdst = wino::voff(4 * pr) + pt * 4 + pp * 2;
    // B fragments of frequency 4 wave + fl, column block cb: U[f][stage][kh][n0 + cb*32 + l31][4 k-pairs]: one 16-byte load
    const unsigned ubase = (unsigned)(((long)kh * Ko + n0 + l31) * 16);
    const unsigned ustage = (unsigned)(2 * Ko * 16), ufreq = (unsigned)((long)Cr * Ko * 4);
    const int s_begin = SPLIT ? blockIdx.z * stages_per_split : 0;
    const int nstages = SPLIT ? min(Cr / BC, s_begin + stages_per_split) : Cr / BC;      // = end of this block's stage range

    f32x16 acc[4][NCB];
#pragma unroll
    for (int fl = 0; fl < 4; ++fl)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[fl][cb][r] = 0.f;

    float4 rin[4];
    float4 fb[4][NCB], fa[2];
    float X[4][4];
    // row pass X = d B along the 4 columns of this thread's patch row, 4 channels
#define WINO_ROWPASS                                                                                     \
    X[0][0] = rin[0].x - rin[2].x; X[0][1] = rin[0].y - rin[2].y; X[0][2] = rin[0].z - rin[2].z; X[0][3] = rin[0].w - rin[2].w; \
    X[1][0] = rin[1].x + rin[2].x; X[1][1] = rin[1].y + rin[2].y; X[1][2] = rin[1].z + rin[2].z; X[1][3] = rin[1].w + rin[2].w; \
    X[2][0] = rin[2].x - rin[1].x; X[2][1] = rin[2].y - rin[1].y; X[2][2] = rin[2].z - rin[1].z; X[2][3] = rin[2].w - rin[1].w; \
    X[3][0] = rin[1].x - rin[3].x; X[3][1] = rin[1].y - rin[3].y; X[3][2] = rin[1].z - rin[3].z; X[3][3] = rin[1].w - rin[3].w;
    // column pass V = B^T X: row i = pr needs its own row and row {2,2,1,1}[pr] of the same quad; frequency 4 pr + J gets the
    // channel pairs (0,2) -> k half 0 and (1,3) -> k half 1 as two 8-byte stores
#define WINO_VSTORE(BUF, J)                                                                              \
    {                                                                                                    \
        float v_[4];                                                                                     \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                  \
            const int other = __builtin_amdgcn_update_dpp(0, __float_as_int(X[J][e]), 0x5A, 0xF, 0xF, false); /* quad_perm [2,2,1,1] */ \
            v_[e] = fmaf(sb, __int_as_float(other), sa * X[J][e]);                                       \
        }                                                                                                \
        *reinterpret_cast<float2 *>((BUF) + vdst + (J) * FSV) = make_float2(v_[0], v_[2]);               \
        *reinterpret_cast<float2 *>((BUF) + vdst + (J) * FSV + ROW) = make_float2(v_[1], v_[3]);         \
    }
#define WINO_XLOAD(S)                                                                                    \
    {                                                                                                    \
        const unsigned sx = (unsigned)((S) * BC * 4);          /* padded taps: kOOB + sx stays out of range */ \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) rin[j] = bufld4(rx, xoff[j] + sx);                 \
    }
    // the B fragments (all four k-pairs) of frequency 4 wave + FL for stage S
#define WINO_BLOAD(FL, S)                                                                                \
    _Pragma("unroll") for (int cb = 0; cb < NCB; ++cb)                                                   \
        fb[FL][cb] = bufld4s(ru, ubase, (unsigned)(4 * wave + (FL)) * ufreq + (unsigned)(S) * ustage + cb * 512);
#define WINO_ALOAD(SL, BUF, FL) fa[SL] = *reinterpret_cast<const float4 *>((BUF) + abase + (FL) * FSV);

    // prologue: tile 0 -> LDS, B fragments of stage 0
    WINO_XLOAD(s_begin)
#pragma unroll
    for (int fl = 0; fl < 4; ++fl) WINO_BLOAD(fl, s_begin)
    WINO_ROWPASS
    WINO_VSTORE(smem, 0) WINO_VSTORE(smem, 1) WINO_VSTORE(smem, 2) WINO_VSTORE(smem, 3)
    __syncthreads();
    // A stage = 8 channels = 4 MFMA k-pairs.  The wave walks its four frequencies one after the other: the A fragments of a
    // frequency are ONE ds_read_b128 (fetched while the previous frequency is multiplied), its B fragments ONE 16-byte
    // load per column block, refilled for the next stage right after the last MFMA that reads them (3/4 of a stage ahead
    // of their use) - a quarter of the LDS / vector-memory instructions of the one-dword-per-MFMA form.  Every accumulator
    // still sees its k-pairs in ascending order (bitwise the same sums).  The global loads of tile s+1 sit behind the
    // first MFMA group, its transform + LDS stores behind the last two (sched_barrier keeps hipcc from re-clumping them): one
    // stage ahead, nothing in registers across the barrier (two stages ahead until round 4: CIFAR step 1.855 -> 1.843 ms).
    const int abase = wino::voff(4 * wave) + kh * ROW + l31 * 4;
    for (int s = s_begin; s < nstages; ++s) {
        const float *cur = smem + ((s - s_begin) & 1) * VF;
        float *nxt = smem + ((s - s_begin + 1) & 1) * VF;
        const int sn = s + 1 < nstages ? s + 1 : s;           // the last refill re-reads the last stage (unused)
        WINO_ALOAD(0, cur, 0)
#pragma unroll
        for (int fl = 0; fl < 4; ++fl) {
            if (fl + 1 < 4) WINO_ALOAD((fl + 1) & 1, cur, fl + 1)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float a = q == 0 ? fa[fl & 1].x : q == 1 ? fa[fl & 1].y : q == 2 ? fa[fl & 1].z : fa[fl & 1].w;
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) {
                    const float b = q == 0 ? fb[fl][cb].x : q == 1 ? fb[fl][cb].y : q == 2 ? fb[fl][cb].z : fb[fl][cb].w;
                    acc[fl][cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[fl][cb], 0, 0, 0);
                }
                if (q == 3) WINO_BLOAD(fl, sn)
                // tile s+1: activations, then row pass and one frequency column per half-group
                if (fl == 0 && q == 0) WINO_XLOAD(s + 1)
                else if (fl == 2 && q == 0) { WINO_ROWPASS WINO_VSTORE(nxt, 0) }
                else if (fl == 2 && q == 2) WINO_VSTORE(nxt, 1)
                else if (fl == 3 && q == 0) WINO_VSTORE(nxt, 2)
                else if (fl == 3 && q == 2) WINO_VSTORE(nxt, 3)
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
    }
#undef WINO_ROWPASS
#undef WINO_VSTORE
#undef WINO_XLOAD
#undef WINO_BLOAD
#undef WINO_ALOAD

    // ---- output transform + epilogue
    const float sc = ep.scale ? ep.scale[0] : 1.f;
    float *Zs = smem;                                   // [wave i][b][tile][k 32]
    const long *obase = reinterpret_cast<const long *>(smem + Cf::SMEM_FLOATS);
    const int kq = tid & 7, tb = tid >> 3;              // 4 channels kq*4.., item tb = (tile, b) and tb + 32
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int tile = (r & 3) + 8 * (r >> 2) + 4 * kh;
            const float m0 = acc[0][cb][r], m1 = acc[1][cb][r], m2 = acc[2][cb][r], m3 = acc[3][cb][r];
            Zs[((wave * 2 + 0) * 32 + tile) * 32 + l31] = m0 + m1 + m2;
            Zs[((wave * 2 + 1) * 32 + tile) * 32 + l31] = m1 - m2 - m3;
        }
        __syncthreads();
        const int ch = n0 + cb * 32 + kq * 4;
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ep.bias) bv = *reinterpret_cast<const float4 *>(ep.bias + ch);
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int item = tb + 32 * it, tile = item >> 1, b = item & 1;
            const long ob = obase[tile];
            if (ob >= 0) {
                const float4 z0 = *reinterpret_cast<const float4 *>(Zs + ((0 * 2 + b) * 32 + tile) * 32 + kq * 4);
                const float4 z1 = *reinterpret_cast<const float4 *>(Zs + ((1 * 2 + b) * 32 + tile) * 32 + kq * 4);
                const float4 z2 = *reinterpret_cast<const float4 *>(Zs + ((2 * 2 + b) * 32 + tile) * 32 + kq * 4);
                const float4 z3 = *reinterpret_cast<const float4 *>(Zs + ((3 * 2 + b) * 32 + tile) * 32 + kq * 4);
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    float4 v;
                    if (a == 0) v = make_float4(z0.x + z1.x + z2.x, z0.y + z1.y + z2.y, z0.z + z1.z + z2.z, z0.w + z1.w + z2.w);
                    else v = make_float4(z1.x - z2.x - z3.x, z1.y - z2.y - z3.y, z1.z - z2.z - z3.z, z1.w - z2.w - z3.w);
                    const long o = ob + ((long)a * W + b) * Ko + ch;
                    if (SPLIT) {                          // this part's plain sums into its slab; the epilogue follows the slab sum
                        *reinterpret_cast<float4 *>(out + (long)blockIdx.z * slab_elems + o) = v;
                        continue;
                    }
                    v.x = v.x * sc + bv.x; v.y = v.y * sc + bv.y; v.z = v.z * sc + bv.z; v.w = v.w * sc + bv.w;
                    if (ep.dact) {
                        const float4 yv = *reinterpret_cast<const float4 *>(ep.dact + ep.dact_index(o));
                        v.x *= act_bwd_from_out(yv.x, ep.act); v.y *= act_bwd_from_out(yv.y, ep.act);
                        v.z *= act_bwd_from_out(yv.z, ep.act); v.w *= act_bwd_from_out(yv.w, ep.act);
                    } else {
                        v.x = act_fwd(v.x, ep.act); v.y = act_fwd(v.y, ep.act);
                        v.z = act_fwd(v.z, ep.act); v.w = act_fwd(v.w, ep.act);
                    }
                    *reinterpret_cast<float4 *>(out + o) = ep.add4(v, o);
                }
            }
        }
        __syncthreads();
    }
}

// y = act(y + bias[ch])  or  y *= act'(dact[...])  in place: the non-linear part of the epilogue after a split launch
__global__ __launch_bounds__(256) void epilogue_pass_kernel(float *__restrict__ y, long total4, int Ko, ConvEpilogue ep) {
    const long stride = (long)gridDim.x * 256;
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < total4; q += stride) {
        const long o = q * 4;
        const int ch = (int)(o % Ko);
        float4 v = reinterpret_cast<float4 *>(y)[q];
        if (ep.bias) {
            const float4 bv = *reinterpret_cast<const float4 *>(ep.bias + ch);
            v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
        }
        if (ep.dact) {
            const float4 yv = *reinterpret_cast<const float4 *>(ep.dact + ep.dact_index(o));
            v.x *= act_bwd_from_out(yv.x, ep.act); v.y *= act_bwd_from_out(yv.y, ep.act);
            v.z *= act_bwd_from_out(yv.z, ep.act); v.w *= act_bwd_from_out(yv.w, ep.act);
        } else {
            v.x = act_fwd(v.x, ep.act); v.y = act_fwd(v.y, ep.act);
            v.z = act_fwd(v.z, ep.act); v.w = act_fwd(v.w, ep.act);
        }
        reinterpret_cast<float4 *>(y)[q] = v;
    }
}

int epilogue_pass(float *y, long total, int Ko, const ConvEpilogue &ep, hipStream_t st) {
    if (!ep.bias && !ep.dact && ep.act == MMDGAN_ACT_LINEAR) return MMDGAN_OK;
    long blocks = (total / 4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(epilogue_pass_kernel, dim3((unsigned)blocks), dim3(256), 0, st, y, total / 4, Ko, ep);
    return check_launch("conv epilogue pass");
}

// ------------------------------------------------------------------------------------------------
static bool wino_enabled() { return tuning().wino != 0; }      // MMDGAN_WINO=0: the 3x3 layers on the implicit-GEMM kernels

// below ~512 tiles the grid no longer fills the chip and the direct kernel wins (measured: D l7 forward at batch
// 128 = 512 tiles: 68 us with 32-channel column blocks vs 92 direct; 768 tiles: 98 vs 160 us).  Splitting the
// channel reduction (wino_kernel<.., true>) does not rescue the small launches either - memset + atomics + the
// activation pass cost what the extra workgroups gain (D l7 forward 98 vs 92 direct, 3B dgrad 139 vs 101
// unsplit) - so it is only used below 96 workgroups.  MMDGAN_WINO_MIN_TILES overrides (the parity tests use small
// problems, which is also what exercises the split path).
// Round 3: with weights transformed by the caller a small launch splits its channel reduction over workspace slabs
// (wino_launch) and fills the chip whatever its tile count, so the line for THAT form is 128 tiles: the 512 -> 512 block of the
// ResNet-SN discriminator at 4x4 (384 tiles at 3B rows) 118 -> ~45 us per input-gradient, the ResNet step 4.92 -> 4.81 ms.
static long wino_min_tiles(bool caller_transformed = false) {
    const long v = tuning().wino_min_tiles;
    return v >= 0 ? v : (caller_transformed ? 128 : 512);
}

static bool wino_shape_ok(const ConvDims &d, int cr, int ko, bool caller_transformed = false) {
    return wino_enabled() && d.R == 3 && d.stride == 1 && d.pad == 1 && d.H % 2 == 0 && d.W % 2 == 0 && cr % wino::BC == 0 &&
           cr >= 32 && ko % 64 == 0 && (long)d.N * (d.H / 2) * (d.W / 2) >= wino_min_tiles(caller_transformed);
}
// would the library run this geometry through Winograd, given transformed weights?
bool wino_eligible(const ConvDims &d, bool dgrad) { return dgrad ? wino_shape_ok(d, d.K, d.C, true) : wino_shape_ok(d, d.C, d.K, true); }
// (without transformed weights the call transforms into the workspace: not for batch-1 launches, which run on concurrent chains)
bool wino_fwd_ok(const ConvDims &d) { return d.N > 1 && wino_shape_ok(d, d.C, d.K) && workspace(sizeof(float) * 16 * (size_t)d.C * d.K) != nullptr; }
bool wino_dgrad_ok(const ConvDims &d) { return d.N > 1 && wino_shape_ok(d, d.K, d.C) && workspace(sizeof(float) * 16 * (size_t)d.C * d.K) != nullptr; }

int wino_transform(const ConvDims &d, const float *w, bool flip, float *U, hipStream_t st) {
    if ((flip ? d.K : d.C) % 8) {
        set_error("wino_transform (3x3): the reduction-side channel count (%d) must be a multiple of 8", flip ? d.K : d.C);
        return MMDGAN_E_ARG;
    }
    const dim3 wg((d.K + 31) / 32, (d.C + 31) / 32);
    if (flip) hipLaunchKernelGGL(wino_weight_kernel<true>, wg, dim3(256), 0, st, w, U, d.C, d.K);
    else hipLaunchKernelGGL(wino_weight_kernel<false>, wg, dim3(256), 0, st, w, U, d.C, d.K);
    return check_launch("wino_transform");
}

// U == nullptr: transform w into the library workspace first
static int wino_launch(const ConvDims &d, const ConvEpilogue &ep, const float *in, const float *w, const float *U, float *out,
                       bool flip, hipStream_t st) {
    const int cr = flip ? d.K : d.C, ko = flip ? d.C : d.K;
    const bool own_u = U != nullptr;
    if (!U) {
        float *ws = (float *)workspace_acquire(sizeof(float) * 16 * (size_t)cr * ko, st);
        if (!ws) { set_error("conv2d (winograd): no workspace for the transformed weights"); return MMDGAN_E_ARG; }
        if (int rc = wino_transform(d, w, flip, ws, st)) return rc;
        U = ws;
    }
    const long T = (long)d.N * (d.H / 2) * (d.W / 2);
    const long wgs = ((T + 31) / 32) * (ko / 64);
    const int nstages = cr / wino::BC;
    // too few tiles for two workgroups per CU, weights transformed by the caller (the workspace is free then): split the
    // channel reduction over workspace slabs, as the 4x4 stride-2 kernel does (conv_wino2.hip: wino2_ksplit).  Each part
    // keeps >= 8 stages (64 channels, 256 MFMAs per wave).  Measured (CIFAR batch 64): D l7 forward (128 workgroups) 69.7 ->
    // 52.4 us, its 3B-row input-gradient (192) 87.8 -> 75.7; ms per CIFAR / STL step with the split applied to grids below
    // 0 / 129 / 193 / 257 / 385 workgroups: 2.045 / 2.014 / 1.999 / 1.999 / 1.974 and 3.909 / 3.915 / 3.902 / 3.912 / 3.887
    // (here the 384-workgroup launch, D l5's 3B-row input-gradient, gains too - the 4x4 kernel's did not).
    const long below = tuning().wino_ksplit_below;
    int split = 1;
    if (own_u && d.N > 1 && wgs < below)
        while (split < 8 && wgs * split < 512 && nstages % (2 * split) == 0 && nstages / (2 * split) >= 8) split *= 2;
    const long total = (long)d.N * d.H * d.W * ko;
    float *slabs = nullptr;
    if (split > 1) slabs = (float *)workspace_acquire(sizeof(float) * (size_t)split * total, st);
    if (slabs) {
        const dim3 grid((unsigned)((T + 31) / 32), ko / 64, split);
        ConvEpilogue plain{};
        plain.wrap_from = kNoWrap;
        hipLaunchKernelGGL((wino_kernel<64, true>), grid, dim3(256), (wino::Cfg<64>::LDS_BYTES), st, d.N, d.H, d.W, cr, ko, plain, in,
                           U, slabs, nstages / split, total);
        if (int rc = check_launch(flip ? "conv2d_dgrad(winograd split)" : "conv2d_fwd(winograd split)")) return rc;
        return slab_epilogue(slabs, split, total, ko, ep, out, st);
    }
    // 32-channel column blocks (half the accumulators: 4 waves per SIMD instead of 2, twice the workgroups, but the
    // input transform is redone per column block) pay off at the grid sizes where 64 leaves CUs idle - measured:
    // 128 workgroups (D l7 forward) 68 vs 94 us, 384 (D l5 3B dgrad) 83 vs 91; 192 / 256 / 512+ are better at 64
    if (wgs <= 128 || (wgs > 256 && wgs < 512)) {
        const dim3 grid((unsigned)((T + 31) / 32), ko / 32);
        hipLaunchKernelGGL((wino_kernel<32, false>), grid, dim3(256), (wino::Cfg<32>::LDS_BYTES), st, d.N, d.H, d.W, cr, ko, ep, in,
                           U, out, nstages, 0L);
    } else {
        const dim3 grid((unsigned)((T + 31) / 32), ko / 64);
        hipLaunchKernelGGL((wino_kernel<64, false>), grid, dim3(256), (wino::Cfg<64>::LDS_BYTES), st, d.N, d.H, d.W, cr, ko, ep, in,
                           U, out, nstages, 0L);
    }
    addend_applied();                                   // in the kernel's stores (the split form above: in slab_epilogue)
    return check_launch(flip ? "conv2d_dgrad(winograd)" : "conv2d_fwd(winograd)");
}

int wino_fwd(const ConvDims &d, const ConvEpilogue &ep, const float *x, const float *w, const float *U, float *y, hipStream_t st) {
    return wino_launch(d, ep, x, w, U, y, false, st);
}
int wino_dgrad(const ConvDims &d, const ConvEpilogue &ep, const float *dy, const float *w, const float *U, float *dx, hipStream_t st) {
    return wino_launch(d, ep, dy, w, U, dx, true, st);
}



// ================================================================================================
// Weight gradient of the same 3x3 / stride-1 layers in the Winograd domain:
//   dW = G^T [ sum_tiles (B^T d B) (.) (A dY A^T) ] G
// i.e. per frequency f a GEMM dU_f[c][k] = sum_tiles V_f[tile][c] * dM_f[tile][k] whose reduction runs over the
// 2x2 output tiles - 16 multiplies per tile and (c,k) instead of 36.
//   * workgroup = 32 input channels x 64 output channels x all 16 frequencies (wave w: i = w, j = 0..3), a slice
//     of the tile range (blockIdx.z); a stage is 8 tiles = 4 MFMA k-pairs = 32 MFMAs per wave.
//   * V = B^T d B goes through LDS exactly as in the forward kernel (thread = tile, channel quad, patch row).
//   * dM = A dY A^T never touches LDS: wave w needs row i = w only, and a lane (tile 2kp+kh, channel k) gets its
//     four j-values from the four dY pixels of its tile with 6 adds - B fragments are built in registers from
//     coalesced dword loads.
//   * epilogue: G^T dU G (linear, so it is applied to the partial sums): j direction in registers, i direction
//     across the waves through LDS, then 9 fp32 atomics per (c,k) into the zeroed dW - the same atomic traffic as
//     the split direct kernel.
struct TileWalk {             // running (tx, ty, image) of a tile index advanced by a fixed step
    int tx, ty, n;
};
struct TileStep {
    int dtx, dty, dn;
};
__device__ __forceinline__ TileStep make_step(int d, int TW, int TH) { return {d % TW, (d / TW) % TH, d / (TW * TH)}; }
__device__ __forceinline__ TileWalk make_walk(long id, int TW, int TH) {
    return {(int)(id % TW), (int)((id / TW) % TH), (int)(id / ((long)TW * TH))};
}
__device__ __forceinline__ void advance(TileWalk &t, const TileStep &s, int TW, int TH) {
    t.tx += s.dtx;
    const bool c1 = t.tx >= TW;
    t.tx -= c1 ? TW : 0;
    t.ty += s.dty + (c1 ? 1 : 0);
    const bool c2 = t.ty >= TH;
    t.ty -= c2 ? TH : 0;
    t.n += s.dn + (c2 ? 1 : 0);
}

namespace winow {
constexpr int BT = 8;                         // tiles per stage
constexpr int LDC = 33;                       // V: floats per tile row (32 channels + 1)
constexpr int FS = BT * LDC + 2;              // V: floats per frequency
static_assert((4 * FS) % 32 == 8, "V frequency stride");
constexpr int V_FLOATS = 16 * FS;
constexpr int TS_FLOATS = 4 * 3 * 32 * 32;    // epilogue exchange buffer, one column block at a time
constexpr int SMEM_FLOATS = 2 * V_FLOATS > TS_FLOATS ? 2 * V_FLOATS : TS_FLOATS;
constexpr size_t LDS_BYTES = sizeof(float) * SMEM_FLOATS + sizeof(unsigned) * 4 * BT;   // + ring of dY tile offsets
}  // namespace winow

__global__ __launch_bounds__(256, 2) void wino_wgrad_kernel(int N, int H, int W, int C, int K, const float *__restrict__ x,
                                                            const float *__restrict__ dy, float *__restrict__ dw,
                                                            int stages_per_split) {
    using namespace winow;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, kh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int TH = H >> 1, TW = W >> 1;
    const long T = (long)N * TH * TW;
    const int nst_all = (int)((T + BT - 1) / BT);
    const int c0 = blockIdx.x * 32, n0 = blockIdx.y * 64;
    const int s0 = blockIdx.z * stages_per_split, s1 = min(nst_all, s0 + stages_per_split);
    if (s0 >= s1) return;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(x, (long)N * H * W * C * 4);
    const __amdgpu_buffer_rsrc_t rdy = make_rsrc(dy, (long)N * H * W * K * 4);
    // ---- producer of V: thread = (tile of the stage pt, channel quad pp, patch row pr)
    const int pt = tid >> 5, pp = (tid >> 2) & 7, pr = tid & 3;
    TileWalk pw = make_walk((long)s0 * BT + pt, TW, TH);
    const TileStep step8 = make_step(BT, TW, TH);
    const float sa = pr == 3 ? -1.f : 1.f, sb = (pr & 1) ? 1.f : -1.f;
    const int vdst = (pr * 4) * FS + pt * LDC + 4 * pp;
    float4 rin[4];
    unsigned *dyoff = reinterpret_cast<unsigned *>(smem + SMEM_FLOATS);      // [stage & 3][tile of the stage]
    int pstage = s0;                          // stage the producer is loading
    auto xload = [&]() {                      // the 4 pixels of patch row pr of the producer's current tile, then advance
        if ((tid & 31) == 0)                  // byte offset of dY pixel (2ty, 2tx), channel 0, of this tile (consumers add the rest)
            dyoff[(pstage & 3) * BT + pt] = pw.n < N ? (unsigned)((((pw.n * H + 2 * pw.ty) * W + 2 * pw.tx) * K) * 4) : kOOB;
        ++pstage;
        const int y = 2 * pw.ty - 1 + pr;
        const bool rowok = pw.n < N && y >= 0 && y < H;
        const unsigned base = (unsigned)((((pw.n * H + y) * W + 2 * pw.tx - 1) * C + c0 + 4 * pp) * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int xx = 2 * pw.tx - 1 + j;
            rin[j] = bufld4(rx, (rowok && xx >= 0 && xx < W) ? base + (unsigned)(j * C * 4) : kOOB);
        }
        advance(pw, step8, TW, TH);
    };
    auto vstore = [&](float *buf) {
        float X[4][4];
        X[0][0] = rin[0].x - rin[2].x; X[0][1] = rin[0].y - rin[2].y; X[0][2] = rin[0].z - rin[2].z; X[0][3] = rin[0].w - rin[2].w;
        X[1][0] = rin[1].x + rin[2].x; X[1][1] = rin[1].y + rin[2].y; X[1][2] = rin[1].z + rin[2].z; X[1][3] = rin[1].w + rin[2].w;
        X[2][0] = rin[2].x - rin[1].x; X[2][1] = rin[2].y - rin[1].y; X[2][2] = rin[2].z - rin[1].z; X[2][3] = rin[2].w - rin[1].w;
        X[3][0] = rin[1].x - rin[3].x; X[3][1] = rin[1].y - rin[3].y; X[3][2] = rin[1].z - rin[3].z; X[3][3] = rin[1].w - rin[3].w;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int other = __builtin_amdgcn_update_dpp(0, __float_as_int(X[j][e]), 0x5A, 0xF, 0xF, false);   // quad_perm [2,2,1,1]
                buf[vdst + j * FS + e] = fmaf(sb, __int_as_float(other), sa * X[j][e]);
            }
    };
    // ---- consumer: this lane's tile for k-pair kp of a stage is s*8 + 2*kp + kh, its channel n0 + cb*32 + l31
    const unsigned kcol = (unsigned)((n0 + l31) * 4);
    const unsigned dyrow = (unsigned)(W * K * 4), dypix = (unsigned)(K * 4);
    // A dY A^T, row i = wave: with e[b] = wa*dY[0][b] + wb*dY[1][b]  ->  dM[i][.] = (e0, e0 + e1, e0 - e1, -e1)
    const float wa = wave == 3 ? 0.f : 1.f, wb = wave == 0 ? 0.f : (wave == 1 ? 1.f : -1.f);

    f32x16 acc[4][2];
#pragma unroll
    for (int fl = 0; fl < 4; ++fl)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[fl][cb][r] = 0.f;

    float dyv[2][2][4];                        // [slot][cb][pixel a*2+b] of the k-pair being prefetched
    auto dyload = [&](int slot, int stage, int kp) {
        const unsigned off = dyoff[(stage & 3) * BT + 2 * kp + kh] + kcol;     // kOOB + kcol stays out of range
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                dyv[slot][cb][p] = bufld1s(rdy, off, (unsigned)(cb * 128) + (p >> 1) * dyrow + (p & 1) * dypix);
            }
    };

    xload();
    vstore(smem);
    xload();
    __syncthreads();
    dyload(0, s0, 0);
    const int abase = (4 * wave) * FS + kh * LDC + l31;
    for (int s = s0; s < s1; ++s) {
        const float *cur = smem + ((s - s0) & 1) * V_FLOATS;
        float *nxt = smem + ((s - s0 + 1) & 1) * V_FLOATS;
#pragma unroll
        for (int kp = 0; kp < 4; ++kp) {
            float fa[4];
#pragma unroll
            for (int fl = 0; fl < 4; ++fl) fa[fl] = cur[abase + fl * FS + 2 * kp * LDC];
            if (kp < 3) dyload((kp + 1) & 1, s, kp + 1);      // next k-pair; the last one prefetches the next stage's first
            else dyload(0, s + 1, 0);
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                const float *d = dyv[kp & 1][cb];
                const float e0 = fmaf(wb, d[2], wa * d[0]), e1 = fmaf(wb, d[3], wa * d[1]);
                const float bq[4] = {e0, e0 + e1, e0 - e1, -e1};
#pragma unroll
                for (int fl = 0; fl < 4; ++fl)
                    acc[fl][cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[fl], bq[fl], acc[fl][cb], 0, 0, 0);
            }
            if (kp == 0) vstore(nxt);          // tile s+1 -> LDS
            else if (kp == 2) xload();         // tile s+2 -> registers
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }

    // ---- G^T dU G on the partial sums, then atomics.  Ts[i][t][c][k 32], one column block at a time
    float *Ts = smem;
    const int kk = tid & 31, cr = tid >> 5;              // lanes along k: one atomic instruction = two 128-byte runs
    const long CK = (long)C * K;
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int cl = (r & 3) + 8 * (r >> 2) + 4 * kh;
            const float a0 = acc[0][cb][r], a1 = acc[1][cb][r], a2 = acc[2][cb][r], a3 = acc[3][cb][r];
            const float h = 0.5f * (a1 + a2);
            Ts[((wave * 3 + 0) * 32 + cl) * 32 + l31] = a0 + h;
            Ts[((wave * 3 + 1) * 32 + cl) * 32 + l31] = 0.5f * (a1 - a2);
            Ts[((wave * 3 + 2) * 32 + cl) * 32 + l31] = h + a3;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int cc = cr + 8 * q;
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const float z0 = Ts[((0 * 3 + t) * 32 + cc) * 32 + kk], z1 = Ts[((1 * 3 + t) * 32 + cc) * 32 + kk];
                const float z2 = Ts[((2 * 3 + t) * 32 + cc) * 32 + kk], z3 = Ts[((3 * 3 + t) * 32 + cc) * 32 + kk];
                float *dst = dw + ((long)t * CK) + (long)(c0 + cc) * K + n0 + cb * 32 + kk;      // dw[r][t][c][k], r = 0
                const float h = 0.5f * (z1 + z2);
                atomicAdd(dst, z0 + h);
                atomicAdd(dst + 3 * CK, 0.5f * (z1 - z2));
                atomicAdd(dst + 6 * CK, h + z3);
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// Second design of the same product (what the stride-2 kernel in conv_wino2.hip taught): workgroup = 32 input channels x
// 128 output channels, eight waves = (column quarter, frequency rows {0,1} | {2,3}), each with eight 32x32 accumulators;
//   * a stage is 16 tiles = 8 k-pairs = 64 MFMAs per wave; V as above (all 512 threads: tile, channel quad, patch row) but
//     stored [f][k-pair][tile parity][32 c] with one ds_write_b128 per frequency (the A fragment of a k-pair is 64 consecutive
//     words);
//   * the 2x2 dY pixels of the stage's tiles go through LDS (coalesced float4 loads by all threads, 4 each) instead of eight
//     dword gathers per lane and k-pair; dM rows are built in registers from four LDS words: 12 LDS words per 8 MFMAs;
//   * the two frequency-row halves swap half of their j-transformed partial sums through LDS, finish G^T dU G in registers
//     and store the workgroup's 9 x 32 x 128 partial result into its slab of the library workspace [split][9][C][K]; the
//     slab reduction of conv_wino2.hip sums them (no zeroing, no atomics, deterministic) together with the bias gradient,
//     which rides along in the workgroups of channel block 0.
namespace winos {
constexpr int BT = 16, BC = 32, BK = 128, NT = 512;
constexpr int FS = BT * BC + 4;               // V: floats per frequency (+4: the four patch-row lanes of a quad hit different banks)
constexpr int V_FLOATS = 16 * FS;
constexpr int DYT = 4 * BK + 32;              // dY stage tile: floats per tile, [4 pixels][128 channels] (+32: half-waves on disjoint banks)
constexpr int DY_FLOATS = BT * DYT;
constexpr int EX_FLOATS = 8 * 48 * 64;        // epilogue exchange: [wave][48 values][lane]
constexpr int MAIN_FLOATS = 2 * V_FLOATS + 2 * DY_FLOATS;
constexpr int SMEM_FLOATS = MAIN_FLOATS > EX_FLOATS ? MAIN_FLOATS : EX_FLOATS;
constexpr size_t LDS_BYTES = sizeof(float) * SMEM_FLOATS + sizeof(unsigned) * 4 * BT;
constexpr int DYV = BT * 4 * (BK / 4) / NT;   // float4 items of the dY stage tile per thread (4)
static_assert(LDS_BYTES <= 160 * 1024, "LDS");
}  // namespace winos

template <bool DBIAS>
__global__ __launch_bounds__(winos::NT) void wino_wgrad_slab_kernel(int N, int H, int W, int C, int K, const float *__restrict__ x,
                                                                    const float *__restrict__ dy, float *__restrict__ part,
                                                                    float *__restrict__ dbpart, int stages_per_split, SlabReduceArgs prev) {
    using namespace winos;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, kh = lane >> 5;
    // the previous weight-gradient launch of this stream left its slabs un-summed (mmdgan_wgrad_defer): this workgroup's share first
    slab_reduce_share(prev, blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), gridDim.x * gridDim.y * gridDim.z, tid, NT,
                      reinterpret_cast<double *>(smem));
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wk = wv & 3, fh = wv >> 2;                       // 32-column quarter of the 128, frequency rows 2fh, 2fh + 1
    const int TH = H >> 1, TW = W >> 1;
    const long T = (long)N * TH * TW;
    const int nst_all = (int)((T + BT - 1) / BT);
    // The workgroups of one pixel range (same blockIdx.z, different channel / column blocks) read the same x and dY
    // stages: walk the grid so that they sit on ONE XCD, next to each other in its dispatch order (hardware deals
    // linear workgroup ids round-robin over the 8 XCDs), and the second reader is served by that XCD's L2.
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    {
        const int per = gridDim.x * gridDim.y, n = per * gridDim.z;
        const int id = bx + gridDim.x * (by + gridDim.y * bz), xcd = id & 7;
        const int v = xcd * (n >> 3) + min(xcd, n & 7) + (id >> 3);     // XCD j holds ids j, j + 8, ...: n/8 of them, one more if j < n%8
        bx = v % gridDim.x;
        by = (v / gridDim.x) % gridDim.y;
        bz = v / per;
    }
    const int c0 = bx * BC, n0 = by * BK;
    const int s0 = bz * stages_per_split, s1 = min(nst_all, s0 + stages_per_split);
    if (s0 >= s1) return;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(x, (long)N * H * W * C * 4);
    const __amdgpu_buffer_rsrc_t rdy = make_rsrc(dy, (long)N * H * W * K * 4);
    float *Vs = smem, *DYs = smem + 2 * V_FLOATS;
    unsigned *dyoff = reinterpret_cast<unsigned *>(smem + SMEM_FLOATS);      // [stage & 3][tile of the stage]
    // ---- producer of V: thread = (tile of the stage pt, channel quad pp, patch row pr)
    const int pt = tid >> 5, pp = (tid >> 2) & 7, pr = tid & 3;
    TileWalk pw = make_walk((long)s0 * BT + pt, TW, TH);
    const TileStep stepT = make_step(BT, TW, TH);
    const float sa = pr == 3 ? -1.f : 1.f, sb = (pr & 1) ? 1.f : -1.f;
    const int vdst = (pr * 4) * FS + (pt >> 1) * 64 + (pt & 1) * 32 + 4 * pp;
    float4 rin[4];
    int pstage = s0;                          // stage the producer is loading
    auto xload = [&]() {                      // the 4 pixels of patch row pr of the producer's current tile, then advance
        const bool ok = pw.n < N && pstage < s1;             // beyond the batch (ragged last stage) or the split: no traffic
        if ((tid & 31) == 0)                  // byte offset of dY pixel (2ty, 2tx), channel 0, of this tile (the dY producers add the rest)
            dyoff[(pstage & 3) * BT + pt] = ok ? (unsigned)((((pw.n * H + 2 * pw.ty) * W + 2 * pw.tx) * K) * 4) : kOOB;
        ++pstage;
        const int y = 2 * pw.ty - 1 + pr;
        const bool rowok = ok && y >= 0 && y < H;
        const unsigned base = (unsigned)((((pw.n * H + y) * W + 2 * pw.tx - 1) * C + c0 + 4 * pp) * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int xx = 2 * pw.tx - 1 + j;
            rin[j] = bufld4(rx, (rowok && xx >= 0 && xx < W) ? base + (unsigned)(j * C * 4) : kOOB);
        }
        advance(pw, stepT, TW, TH);
    };
    auto vstore = [&](float *buf) {
        float X[4][4];
        X[0][0] = rin[0].x - rin[2].x; X[0][1] = rin[0].y - rin[2].y; X[0][2] = rin[0].z - rin[2].z; X[0][3] = rin[0].w - rin[2].w;
        X[1][0] = rin[1].x + rin[2].x; X[1][1] = rin[1].y + rin[2].y; X[1][2] = rin[1].z + rin[2].z; X[1][3] = rin[1].w + rin[2].w;
        X[2][0] = rin[2].x - rin[1].x; X[2][1] = rin[2].y - rin[1].y; X[2][2] = rin[2].z - rin[1].z; X[2][3] = rin[2].w - rin[1].w;
        X[3][0] = rin[1].x - rin[3].x; X[3][1] = rin[1].y - rin[3].y; X[3][2] = rin[1].z - rin[3].z; X[3][3] = rin[1].w - rin[3].w;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int other = __builtin_amdgcn_update_dpp(0, __float_as_int(X[j][e]), 0x5A, 0xF, 0xF, false);   // quad_perm [2,2,1,1]
                o[e] = fmaf(sb, __int_as_float(other), sa * X[j][e]);
            }
            *reinterpret_cast<float4 *>(buf + vdst + j * FS) = make_float4(o[0], o[1], o[2], o[3]);
        }
    };
    // ---- producer of the dY stage tile: item = (tile, pixel of its 2x2, channel quad of the 128), all threads
    const unsigned dyrow = (unsigned)(W * K * 4), dypix = (unsigned)(K * 4);
    const bool dosum = DBIAS && bx == 0;
    float4 dbs = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 rdyv[DYV];
    auto dyload = [&](int stage) {            // needs dyoff[stage], written at least one barrier ago
#pragma unroll
        for (int it = 0; it < DYV; ++it) {
            const int e = tid + it * NT;
            const int tile = e >> 7, px = (e >> 5) & 3, kq = e & 31;
            const unsigned off = dyoff[(stage & 3) * BT + tile];
            rdyv[it] = bufld4(rdy, off == kOOB ? kOOB : off + (px >> 1) * dyrow + (px & 1) * dypix + (unsigned)((n0 + 4 * kq) * 4));
        }
    };
    auto dystore = [&](float *buf) {          // (tiles beyond the split were not loaded: zeros)
#pragma unroll
        for (int it = 0; it < DYV; ++it) {
            const int e = tid + it * NT;
            const int tile = e >> 7, px = (e >> 5) & 3, kq = e & 31;
            *reinterpret_cast<float4 *>(buf + tile * DYT + px * BK + 4 * kq) = rdyv[it];
            if (dosum) { dbs.x += rdyv[it].x; dbs.y += rdyv[it].y; dbs.z += rdyv[it].z; dbs.w += rdyv[it].w; }
        }
    };
    // A dY A^T, A = [1 0; 1 1; 1 -1; 0 -1]: this wave's rows i = 2fh, 2fh + 1 of e = A dY:  (1,0),(1,1) | (1,-1),(0,-1)
    const float a0 = 1.f, a1 = fh ? -1.f : 0.f, b0 = fh ? 0.f : 1.f, b1 = fh ? -1.f : 1.f;

    f32x16 acc[8];                             // [row of the wave 2][j 4]
#pragma unroll
    for (int f = 0; f < 8; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;

    xload();
    vstore(Vs);
    xload();
    __syncthreads();                          // dyoff[s0], dyoff[s0 + 1] visible
    dyload(s0);
    dystore(DYs);
    __syncthreads();
    const int abase = (8 * fh) * FS + lane;                  // + f * FS + kp * 64
    const int dbase = kh * DYT + wk * 32 + l31;              // + 2 kp * DYT + pixel * BK
    float fa[8], dv[4];
    auto opload = [&](const float *cur, const float *dcur, int kp) {     // operands of k-pair kp: 8 + 4 LDS words
#pragma unroll
        for (int f = 0; f < 8; ++f) fa[f] = cur[abase + f * FS + kp * 64];
#pragma unroll
        for (int q = 0; q < 4; ++q) dv[q] = dcur[dbase + 2 * kp * DYT + q * BK];
    };
    for (int s = s0; s < s1; ++s) {
        const int b = (s - s0) & 1;
        const float *cur = Vs + b * V_FLOATS, *dcur = DYs + b * DY_FLOATS;
        float *nxt = Vs + (b ^ 1) * V_FLOATS, *dnxt = DYs + (b ^ 1) * DY_FLOATS;
        opload(cur, dcur, 0);
#pragma unroll
        for (int kp = 0; kp < BT / 2; ++kp) {
            const float ea0 = fmaf(a1, dv[2], a0 * dv[0]), ea1 = fmaf(a1, dv[3], a0 * dv[1]);     // row 2fh
            const float eb0 = fmaf(b1, dv[2], b0 * dv[0]), eb1 = fmaf(b1, dv[3], b0 * dv[1]);     // row 2fh + 1
            const float bm[8] = {ea0, ea0 + ea1, ea0 - ea1, -ea1, eb0, eb0 + eb1, eb0 - eb1, -eb1};
            float a[8];
#pragma unroll
            for (int f = 0; f < 8; ++f) a[f] = fa[f];
            if (kp + 1 < BT / 2) opload(cur, dcur, kp + 1);           // next k-pair's operands fly under this one's MFMAs
#pragma unroll
            for (int f = 0; f < 8; ++f) acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[f], bm[f], acc[f], 0, 0, 0);
            if (kp == 0) { vstore(nxt); dyload(s + 1); }      // tile s+1: V -> LDS, its dY -> registers
            else if (kp == 4) xload();                        // tile s+2 -> registers (and its dY offsets)
            else if (kp == 6) dystore(dnxt);
        }
        __syncthreads();
    }

    if (dosum) {                               // (workgroup-uniform) 16 threads per channel quad -> one partial row of the split
        *reinterpret_cast<float4 *>(smem + (tid >> 5) * BK + 4 * (tid & 31)) = dbs;
        __syncthreads();
        if (tid < BK) {
            float t = 0.f;
#pragma unroll
            for (int q = 0; q < NT / 32; ++q) t += smem[q * BK + tid];
            dbpart[(long)bz * K + n0 + tid] = t;
        }
        __syncthreads();
    }
    // ---- G^T dU G, G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1].  j direction in registers: rows (2fh, 2fh+1) x taps t = 0..2;
    // then the wave keeps accumulator registers [8 fh, 8 fh + 8) and gets the partner wave's rows for them through LDS.
    float *ex = smem;
    {
        float *mine = ex + (wv * 48) * 64 + lane;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int r = q + 8 * (1 - fh);              // the half the partner keeps
#pragma unroll
            for (int row = 0; row < 2; ++row) {
                const float u0 = acc[4 * row][r], u1 = acc[4 * row + 1][r], u2 = acc[4 * row + 2][r], u3 = acc[4 * row + 3][r];
                const float h = 0.5f * (u1 + u2);
                mine[((row * 3 + 0) * 8 + q) * 64] = u0 + h;
                mine[((row * 3 + 1) * 8 + q) * 64] = 0.5f * (u1 - u2);
                mine[((row * 3 + 2) * 8 + q) * 64] = h + u3;
            }
        }
    }
    __syncthreads();
    const long CK = (long)C * K;
    float *dst0 = part + (long)bz * 9 * CK + (long)(c0 + 4 * kh) * K + n0 + wk * 32 + l31;
    const float *theirs = ex + ((wv ^ 4) * 48) * 64 + lane;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int r = q + 8 * fh;
        float *dr = dst0 + (long)((r & 3) + 8 * (r >> 2)) * K;
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            float own[2];
#pragma unroll
            for (int row = 0; row < 2; ++row) {
                const float u0 = acc[4 * row][r], u1 = acc[4 * row + 1][r], u2 = acc[4 * row + 2][r], u3 = acc[4 * row + 3][r];
                const float h = 0.5f * (u1 + u2);
                own[row] = t == 0 ? u0 + h : (t == 1 ? 0.5f * (u1 - u2) : h + u3);
            }
            const float o0 = theirs[((0 * 3 + t) * 8 + q) * 64], o1 = theirs[((1 * 3 + t) * 8 + q) * 64];
            // rows i = 0..3 of the j-transformed sums
            const float T0 = fh ? o0 : own[0], T1 = fh ? o1 : own[1], T2 = fh ? own[0] : o0, T3 = fh ? own[1] : o1;
            const float hh = 0.5f * (T1 + T2);
            dr[(long)(0 * 3 + t) * CK] = T0 + hh;
            dr[(long)(1 * 3 + t) * CK] = 0.5f * (T1 - T2);
            dr[(long)(2 * 3 + t) * CK] = hh + T3;
        }
    }
}

bool wino_wgrad_slab_ok(const ConvDims &d) {
    return tuning().wino_wgrad_slab && d.C % winos::BC == 0 && d.K % winos::BK == 0;
}

bool wino_wgrad_ok(const ConvDims &d) {
    return tuning().wino_wgrad && d.N > 1 && wino_enabled() && d.R == 3 && d.stride == 1 && d.pad == 1 && d.H % 2 == 0 && d.W % 2 == 0 && d.C % 32 == 0 &&
           d.K % 64 == 0 && (long)d.N * (d.H / 2) * (d.W / 2) >= (wino_min_tiles() < 256 ? wino_min_tiles() : 256);   // 79 vs 89 us at 512 tiles (D l7)
}

// dbias (optional): the column sums of dy; *dbias_done says whether they were produced here (slab path)
int wino_wgrad(const ConvDims &d, const float *x, const float *dy, float *dw, float *dbias, bool *dbias_done, hipStream_t st,
               const float *wdot, float *dot, bool *dot_done) {
    const long T = (long)d.N * (d.H / 2) * (d.W / 2);
    if (dbias_done) *dbias_done = false;
    if (dot_done) *dot_done = false;
    if (wino_wgrad_slab_ok(d)) {
        const int nst = (int)((T + winos::BT - 1) / winos::BT);
        const long base = (long)(d.C / winos::BC) * (d.K / winos::BK);
        int split = (int)(wgrad_cus() / base);                                  // one 8-wave workgroup per CU, one round
        if (split > nst / 2) split = nst / 2;                           // >= 2 stages (128 MFMAs per wave) per workgroup
        if (split < 1) split = 1;
        const int sps = (nst + split - 1) / split;
        split = (nst + sps - 1) / sps;                                  // every slab gets written
        const size_t n = 9 * (size_t)d.C * d.K;
        SlabReduceArgs prev{};
        if (float *part = (float *)wgrad_slabs_acquire(sizeof(float) * (n + d.K) * split, st, &prev)) {
            static bool cap_raised = false;
            if (!cap_raised) {
                (void)hipFuncSetAttribute((const void *)wino_wgrad_slab_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)winos::LDS_BYTES);
                (void)hipFuncSetAttribute((const void *)wino_wgrad_slab_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)winos::LDS_BYTES);
                cap_raised = true;
            }
            float *dbpart = part + n * split;
            const dim3 grid(d.C / winos::BC, d.K / winos::BK, split);
            if (dbias)
                hipLaunchKernelGGL(wino_wgrad_slab_kernel<true>, grid, dim3(winos::NT), winos::LDS_BYTES, st, d.N, d.H, d.W, d.C, d.K, x, dy, part,
                                   dbpart, sps, prev);
            else
                hipLaunchKernelGGL(wino_wgrad_slab_kernel<false>, grid, dim3(winos::NT), winos::LDS_BYTES, st, d.N, d.H, d.W, d.C, d.K, x, dy, part,
                                   dbpart, sps, prev);
            if (int rc = slab_reduce(part, split, n, dw, dbpart, dbias ? d.K : 0, dbias, st, wdot, dot)) return rc;
            if (dbias_done) *dbias_done = dbias != nullptr;
            if (dot_done) *dot_done = wdot != nullptr;
            return check_launch("conv2d_wgrad(winograd)");
        }
    }
    const int nst = (int)((T + winow::BT - 1) / winow::BT);
    const long base = (long)(d.C / 32) * (d.K / 64);
    int split = (int)((512 + base - 1) / base);                    // ~512 workgroups
    if (split > nst / 8) split = nst / 8 > 0 ? nst / 8 : 1;         // >= 8 stages (256 MFMAs per wave) per workgroup
    int sps = (nst + split - 1) / split;
    split = (nst + sps - 1) / sps;
    if (zero_output(dw, sizeof(float) * 9 * (size_t)d.C * d.K, st) != hipSuccess) return check_launch("conv2d_wgrad memset");
    hipLaunchKernelGGL(wino_wgrad_kernel, dim3(d.C / 32, d.K / 64, split), dim3(256), winow::LDS_BYTES, st, d.N, d.H, d.W, d.C,
                       d.K, x, dy, dw, sps);
    return check_launch("conv2d_wgrad(winograd)");
}

}  // namespace mmdgan
