// Winograd F(4x4, 3x3) WEIGHT GRADIENT of the 3x3 / stride-1 layers (layer_func.py:912-916, op 'c'): the third kernel of
// the family in conv_wino43.hip (same points 0, 1, -1, 1/2, -2, infinity, same matrices):
//
//   dW = G^T [ sum_tiles (B^T d B) (.) (A dY A^T) ] G       d: 6x6 input patch, dY: the tile's 4x4 output gradients
//
// i.e. per frequency f = 6 i + j a GEMM  dU_f[c][k] = sum_tiles V_f[tile][c] * M_f[tile][k]  whose reduction runs over the
// 4x4 output tiles: 36 multiplies per tile and (c, k) instead of 144 direct, or 64 as F(2x2,3x3) tiles (conv_wino.hip).
//   * one workgroup = 32 input channels x 32 output channels x ALL 36 frequencies x a slice of the tile range, EIGHT waves
//     in three roles (one consumer and one producer on every SIMD - the matrix pipe and the VALU are separate pipes):
//       waves 0-3 (consumers): 9 frequencies each = 9 accumulators of v_mfma_f32_32x32x2_f32; per window of 8 tiles (4 MFMA
//                 k-pairs) 36 ds_read_b64 (both operands come from LDS - neither is a weight) and 36 MFMAs - nothing else;
//       waves 4-5 (V producers): lane = (input channel, k half, k-pair pair) owns TWO tiles of the window: their 6x6 patches
//                 as 72 dword loads two windows ahead (zero padding = the buffer range check, as in the forward kernel),
//                 B^T d B on both tiles at once (packed f32x2), 36 conflict-free ds_write_b64;
//       waves 6-7 (M producers): the same for the output gradients: 2 x 16 dword loads, A dY A^T, 36 ds_write_b64; the
//                 workgroups of channel block 0 also add up what they load: the bias gradient rides along.
//     One barrier per window, both operands double-buffered (2 x 72 KB).
//   * epilogue: G^T dU G needs all 36 frequencies of a (c, k) and they sit in four waves: the accumulators go through LDS
//     ([36][32][32] = 144 KB over the operand buffers), every thread then owns two (c, k), reads their 36 values, applies the
//     transform in registers and stores 9 values into the workgroup's slab of the library workspace [split][9][C][K]; the slab
//     reduction (slab_reduce.h: stand-alone pass, or the prologue of the stream's next weight-gradient launch) sums the
//     slabs in a fixed order together with the bias rows - no zeroing, no atomics, bit-reproducible, and the 4x larger tile
//     leaves a quarter of the slabs F(2x2,3x3) needs for one round of workgroups.
#include "conv_internal.h"
#include "bufload.h"

namespace mmdgan {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace w43w {
constexpr int NT = 512;
constexpr int BT = 8;                           // tiles per window = 4 MFMA k-pairs
constexpr int BC = 32, BK = 32;                 // channels of x / of dy per workgroup
constexpr int FS = 2 * 2 * 32 * 2;              // floats per frequency of one operand: [k half][k-pair pair][channel 32][2 k-pairs]
constexpr int OP_FLOATS = 36 * FS;              // V (or M) of one window
constexpr int WIN_FLOATS = 2 * OP_FLOATS;       // V then M: 73,728 bytes
constexpr int Z_FLOATS = 36 * 32 * 32;          // the epilogue's exchange buffer [f][c][k]
static_assert(Z_FLOATS == 2 * WIN_FLOATS, "the exchange buffer is the double buffer");
constexpr size_t LDS_BYTES = sizeof(float) * (Z_FLOATS + 4 * BK);      // + the four partial bias rows
constexpr unsigned kPad = 0x40000000u;          // an out-of-image row / column / tile: keeps the offset out of range
constexpr long kMaxBytes = 0x40000000L;         // ... for tensors below 1 GiB
}  // namespace w43w

// compile-time ablation masks (tools/wino43w_ablate.sh); 0 in the library build
//   1: no patch / dY loads   2: no transforms, no operand stores   4: no MFMAs   8: no operand reads (LDS)   16: no epilogue
#ifndef W43W_ABLATE
#define W43W_ABLATE 0
#endif

__device__ __forceinline__ f32x2 w43w_fma(float c, f32x2 a, f32x2 b) { return __builtin_elementwise_fma(f32x2{c, c}, a, b); }
// B^T d for one six-point line (conv_wino43.hip: W43_BT6)
#define W43W_BT6(d0, d1, d2, d3, d4, d5, t0, t1, t2, t3, t4, t5)                       \
    {                                                                                  \
        const f32x2 e_ = (d3) - (d1), g_ = (d4) - (d2);                                \
        t0 = w43w_fma(-2.f, (d2), w43w_fma(1.5f, e_, (d0) + (d4)));                    \
        t1 = w43w_fma(2.5f, (d3), w43w_fma(0.5f, (d2), (d4) - (d1)));                  \
        t2 = w43w_fma(0.5f, (d3), w43w_fma(-2.5f, (d2), (d4) + (d1)));                 \
        t3 = w43w_fma(2.f, e_, g_);                                                    \
        t4 = w43w_fma(-0.5f, e_, g_);                                                  \
        t5 = w43w_fma(-2.f, (d3), w43w_fma(1.5f, g_, (d5) + (d1)));                    \
    }
// A e for one four-point line: A = (A^T)^T = [1 0 0 0; 1 1 1 1; 1 -1 1 -1; 1 1/2 1/4 1/8; 1 -2 4 -8; 0 0 0 1]
#define W43W_A4(e0, e1, e2, e3, m0, m1, m2, m3, m4, m5)                                \
    {                                                                                  \
        const f32x2 s_ = (e0) + (e2), u_ = (e1) + (e3);                                \
        m0 = (e0);                                                                     \
        m1 = s_ + u_;                                                                  \
        m2 = s_ - u_;                                                                  \
        m3 = w43w_fma(0.125f, (e3), w43w_fma(0.25f, (e2), w43w_fma(0.5f, (e1), (e0)))); \
        m4 = w43w_fma(-8.f, (e3), w43w_fma(4.f, (e2), w43w_fma(-2.f, (e1), (e0))));    \
        m5 = (e3);                                                                     \
    }
// G^T m for one six-point line: G^T = [1 1/3 -1/3 -16/15 1/15 0; 0 1/3 1/3 -8/15 -2/15 0; 0 1/3 -1/3 -4/15 4/15 1]
#define W43W_GT6(m0, m1, m2, m3, m4, m5, g0, g1, g2)                                   \
    {                                                                                  \
        const float p_ = (1.f / 3.f) * ((m1) - (m2)), q_ = (1.f / 3.f) * ((m1) + (m2)); \
        g0 = fmaf(1.f / 15.f, (m4), fmaf(-16.f / 15.f, (m3), (m0) + p_));              \
        g1 = fmaf(-2.f / 15.f, (m4), fmaf(-8.f / 15.f, (m3), q_));                     \
        g2 = fmaf(4.f / 15.f, (m4), fmaf(-4.f / 15.f, (m3), p_ + (m5)));               \
    }

struct W43wWalk {             // running (tx, ty, image) of a tile index advanced by the window's 8 tiles
    int tx, ty, n;
};

// x [N,H,W,C], dy [N,H,W,K] -> part [split][9][C][K] (+ dbpart [split][K]); H and W multiples of 4, C and K of 32
template <bool DBIAS>
__global__ __launch_bounds__(w43w::NT) void wino43_wgrad_kernel(int N, int H, int W, int C, int K, const float *__restrict__ x,
                                                                const float *__restrict__ dy, float *__restrict__ part,
                                                                float *__restrict__ dbpart, int nblk_c, int nblk, int windows_per_split,
                                                                SlabReduceArgs prev) {
    using namespace w43w;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, kh = lane >> 5;
    // the previous weight-gradient launch of this stream left its slabs un-summed (mmdgan_wgrad_defer): this workgroup's share first
    slab_reduce_share(prev, blockIdx.x, gridDim.x, tid, NT, reinterpret_cast<double *>(smem));
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int TH = H >> 2, TW = W >> 2;
    const int T = N * TH * TW;
    // The workgroups of one tile range (same x and dy, different channel blocks) sit next to each other on ONE XCD (hardware
    // deals linear workgroup ids round-robin over the 8 XCDs): the other readers of a patch are served by that XCD's L2.
    int blk, bz;
    {
        const int n = gridDim.x, id = blockIdx.x, xcd = id & 7;
        const int v = xcd * (n >> 3) + min(xcd, n & 7) + (id >> 3);
        blk = v % nblk;
        bz = v / nblk;
    }
    const int c0 = (blk % nblk_c) * BC, k0 = (blk / nblk_c) * BK;
    const int w_begin = bz * windows_per_split;
    const int nw_all = (T + BT - 1) / BT;
    const int S = min(nw_all, w_begin + windows_per_split) - w_begin;          // windows of this workgroup (>= 1: the launcher's split)
    const int t_end = min(T, (w_begin + S) * BT);
    float *dbsum = smem + Z_FLOATS;                                             // [kh 2 x pair 2][k 32]

    if (wave >= 4) {
        // ------------------------------------------------------------ producers: lane = (channel l31, k half kh, k-pair pair pp)
        // owns the window's tiles 4 pp + kh (k-pair 2 pp) and 4 pp + kh + 2 (k-pair 2 pp + 1): .x / .y of every f32x2 below
        const bool is_m = wave >= 6;
        const int pp = (wave - 4) & 1;
        const int tl = 4 * pp + kh;
        const int vd = (is_m ? OP_FLOATS : 0) + kh * 128 + pp * 64 + l31 * 2;
        W43wWalk wa, wb;
        {
            const int ia = w_begin * BT + tl, ib = ia + 2;
            wa = {ia % TW, (ia / TW) % TH, ia / (TW * TH)};
            wb = {ib % TW, (ib / TW) % TH, ib / (TW * TH)};
        }
        const int s_tx = BT % TW, s_ty = (BT / TW) % TH, s_n = BT / (TW * TH);
        auto advance = [&](W43wWalk &t) {
            t.tx += s_tx;
            const bool c1 = t.tx >= TW;
            t.tx -= c1 ? TW : 0;
            t.ty += s_ty + (c1 ? 1 : 0);
            const bool c2 = t.ty >= TH;
            t.ty -= c2 ? TH : 0;
            t.n += s_n + (c2 ? 1 : 0);
        };
        int ida = w_begin * BT + tl;                    // tile index of .x of the next window to be requested (.y: + 2)
        if (!is_m) {
            // ---------------------------------------------------------------- V = B^T d B of the patches
            const __amdgpu_buffer_rsrc_t rx = make_rsrc(x, (long)N * H * W * C * 4);
            const unsigned chb = (unsigned)((c0 + l31) * 4);
            f32x2 da[36], db[36];
            auto xload = [&](f32x2(&d)[36]) __attribute__((always_inline)) {   // the next window's two patches, then advance
                if (W43W_ABLATE & 1) {
#pragma unroll
                    for (int e = 0; e < 36; ++e) {
                        float o = (float)(lane + e);
                        asm volatile("" : "+v"(o));
                        d[e] = f32x2{o, o};
                    }
                    return;
                }
                unsigned rowa[6], cola[6], rowb[6], colb[6];
                const bool oka = ida < t_end, okb = ida + 2 < t_end;
#pragma unroll
                for (int r = 0; r < 6; ++r) {
                    const int ya = 4 * wa.ty - 1 + r, xa = 4 * wa.tx - 1 + r, yb = 4 * wb.ty - 1 + r, xb = 4 * wb.tx - 1 + r;
                    rowa[r] = (oka && ya >= 0 && ya < H) ? (unsigned)((wa.n * H + ya) * W) * (unsigned)(C * 4) + chb : kPad;
                    cola[r] = (xa >= 0 && xa < W) ? (unsigned)(xa * C * 4) : kPad;
                    rowb[r] = (okb && yb >= 0 && yb < H) ? (unsigned)((wb.n * H + yb) * W) * (unsigned)(C * 4) + chb : kPad;
                    colb[r] = (xb >= 0 && xb < W) ? (unsigned)(xb * C * 4) : kPad;
                }
#pragma unroll
                for (int r = 0; r < 6; ++r)
#pragma unroll
                    for (int j = 0; j < 6; ++j) {
                        d[r * 6 + j].x = bufld1(rx, rowa[r] + cola[j]);
                        d[r * 6 + j].y = bufld1(rx, rowb[r] + colb[j]);
                    }
                advance(wa);
                advance(wb);
                ida += BT;
            };
            auto transform_dump = [&](f32x2(&d)[36], float *buf) __attribute__((always_inline)) {
                if (W43W_ABLATE & 2) return;
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < 6; ++r) {
                    f32x2 t0, t1, t2, t3, t4, t5;
                    W43W_BT6(d[r * 6 + 0], d[r * 6 + 1], d[r * 6 + 2], d[r * 6 + 3], d[r * 6 + 4], d[r * 6 + 5], t0, t1, t2, t3, t4, t5)
                    d[r * 6 + 0] = t0; d[r * 6 + 1] = t1; d[r * 6 + 2] = t2; d[r * 6 + 3] = t3; d[r * 6 + 4] = t4; d[r * 6 + 5] = t5;
                }
#pragma unroll
                for (int jj = 0; jj < 6; ++jj) {
                    f32x2 v0, v1, v2, v3, v4, v5;
                    W43W_BT6(d[0 + jj], d[6 + jj], d[12 + jj], d[18 + jj], d[24 + jj], d[30 + jj], v0, v1, v2, v3, v4, v5)
                    *reinterpret_cast<f32x2 *>(buf + (0 + jj) * FS + vd) = v0;
                    *reinterpret_cast<f32x2 *>(buf + (6 + jj) * FS + vd) = v1;
                    *reinterpret_cast<f32x2 *>(buf + (12 + jj) * FS + vd) = v2;
                    *reinterpret_cast<f32x2 *>(buf + (18 + jj) * FS + vd) = v3;
                    *reinterpret_cast<f32x2 *>(buf + (24 + jj) * FS + vd) = v4;
                    *reinterpret_cast<f32x2 *>(buf + (30 + jj) * FS + vd) = v5;
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            // window j: the consumers multiply window j; its successor's operands are written during it, and the patches of the
            // one after that are requested into the registers that were transformed a window ago (beyond the slice's last tile:
            // out-of-range offsets - the loads return zeros without traffic, their window is never multiplied)
            xload(da);
            xload(db);
            transform_dump(da, smem);
            __syncthreads();
            for (int j = 0;;) {
                xload(da);
                transform_dump(db, smem + ((j + 1) & 1) * WIN_FLOATS);
                __syncthreads();
                if (++j >= S) break;
                xload(db);
                transform_dump(da, smem + ((j + 1) & 1) * WIN_FLOATS);
                __syncthreads();
                if (++j >= S) break;
            }
        } else {
            // ---------------------------------------------------------------- M = A dY A^T of the tiles' output gradients
            const __amdgpu_buffer_rsrc_t rdy = make_rsrc(dy, (long)N * H * W * K * 4);
            const unsigned chb = (unsigned)((k0 + l31) * 4);
            const unsigned rowbytes = (unsigned)(W * K * 4), pixbytes = (unsigned)(K * 4);
            const bool dosum = DBIAS && c0 == 0;
            f32x2 dbs = {0.f, 0.f};
            f32x2 ea[16], eb[16];
            auto yload = [&](f32x2(&d)[16]) __attribute__((always_inline)) {
                if (W43W_ABLATE & 1) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        float o = (float)(lane + e);
                        asm volatile("" : "+v"(o));
                        d[e] = f32x2{o, o};
                    }
                    return;
                }
                const unsigned oa = ida < t_end ? (unsigned)((wa.n * H + 4 * wa.ty) * W + 4 * wa.tx) * pixbytes + chb : kPad;
                const unsigned ob = ida + 2 < t_end ? (unsigned)((wb.n * H + 4 * wb.ty) * W + 4 * wb.tx) * pixbytes + chb : kPad;
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) {           // (the pixel's part of the offset is wave-uniform: the scalar offset)
                        d[a * 4 + b].x = bufld1s(rdy, oa, (unsigned)a * rowbytes + (unsigned)b * pixbytes);
                        d[a * 4 + b].y = bufld1s(rdy, ob, (unsigned)a * rowbytes + (unsigned)b * pixbytes);
                    }
                advance(wa);
                advance(wb);
                ida += BT;
            };
            auto transform_dump = [&](f32x2(&d)[16], float *buf) __attribute__((always_inline)) {
                if (W43W_ABLATE & 2) return;
                __builtin_amdgcn_sched_barrier(0);
                if (dosum) {
                    f32x2 s = (d[0] + d[1]) + (d[2] + d[3]);
#pragma unroll
                    for (int a = 1; a < 4; ++a) s += (d[a * 4] + d[a * 4 + 1]) + (d[a * 4 + 2] + d[a * 4 + 3]);
                    dbs += s;
                }
                f32x2 t[4][6];                              // t[a][j] = sum_b dY[a][b] A[j][b]
#pragma unroll
                for (int a = 0; a < 4; ++a)
                    W43W_A4(d[a * 4 + 0], d[a * 4 + 1], d[a * 4 + 2], d[a * 4 + 3], t[a][0], t[a][1], t[a][2], t[a][3], t[a][4], t[a][5])
#pragma unroll
                for (int jj = 0; jj < 6; ++jj) {            // M[i][j] = sum_a A[i][a] t[a][j]
                    f32x2 v0, v1, v2, v3, v4, v5;
                    W43W_A4(t[0][jj], t[1][jj], t[2][jj], t[3][jj], v0, v1, v2, v3, v4, v5)
                    *reinterpret_cast<f32x2 *>(buf + (0 + jj) * FS + vd) = v0;
                    *reinterpret_cast<f32x2 *>(buf + (6 + jj) * FS + vd) = v1;
                    *reinterpret_cast<f32x2 *>(buf + (12 + jj) * FS + vd) = v2;
                    *reinterpret_cast<f32x2 *>(buf + (18 + jj) * FS + vd) = v3;
                    *reinterpret_cast<f32x2 *>(buf + (24 + jj) * FS + vd) = v4;
                    *reinterpret_cast<f32x2 *>(buf + (30 + jj) * FS + vd) = v5;
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            yload(ea);
            yload(eb);
            transform_dump(ea, smem);
            __syncthreads();
            for (int j = 0;;) {
                yload(ea);
                transform_dump(eb, smem + ((j + 1) & 1) * WIN_FLOATS);
                __syncthreads();
                if (++j >= S) break;
                yload(eb);
                transform_dump(ea, smem + ((j + 1) & 1) * WIN_FLOATS);
                __syncthreads();
                if (++j >= S) break;
            }
            // (the transforms beyond window S - 1 saw zeros: tiles >= t_end are out of range)
            if (dosum) dbsum[((wave - 6) * 2 + kh) * BK + l31] = dbs.x + dbs.y;
        }
    } else {
        // ---------------------------------------------------------------- consumers: 9 frequencies x 32 c x 32 k
        f32x16 acc[9];
#pragma unroll
        for (int fl = 0; fl < 9; ++fl)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[fl][r] = 0.f;
        const int f0 = 9 * wave;
        const int obase = f0 * FS + kh * 128 + l31 * 2;     // + fl * FS + pair * 64 (+ OP_FLOATS: M)
        __syncthreads();
        for (int j = 0; j < S; ++j) {
            const float *cur = smem + (j & 1) * WIN_FLOATS + obase;
            // frequencies in pairs (the last one alone): consecutive MFMAs go to different accumulators, the operands of the
            // next pair are fetched while this one is multiplied
            float2 va[2][2], vb[2][2];                      // [frequency of the pair][k-pair pair]
            auto opload = [&](int fl, int slot) __attribute__((always_inline)) {
                if (W43W_ABLATE & 8) {
                    va[slot][0] = va[slot][1] = vb[slot][0] = vb[slot][1] = make_float2(1.f, 1.f);
                    return;
                }
                va[slot][0] = *reinterpret_cast<const float2 *>(cur + fl * FS);
                va[slot][1] = *reinterpret_cast<const float2 *>(cur + fl * FS + 64);
                vb[slot][0] = *reinterpret_cast<const float2 *>(cur + fl * FS + OP_FLOATS);
                vb[slot][1] = *reinterpret_cast<const float2 *>(cur + fl * FS + OP_FLOATS + 64);
            };
            opload(0, 0);
            opload(1, 1);
#pragma unroll
            for (int fp = 0; fp < 9; fp += 2) {
                const float2 a00 = va[0][0], a01 = va[0][1], b00 = vb[0][0], b01 = vb[0][1];
                const float2 a10 = va[1][0], a11 = va[1][1], b10 = vb[1][0], b11 = vb[1][1];
                if (fp + 2 < 9) opload(fp + 2, 0);
                if (fp + 3 < 9) opload(fp + 3, 1);
                if (!(W43W_ABLATE & 4)) {
#define W43W_MM(ACC, A_, B_) ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(A_, B_, ACC, 0, 0, 0);
                    if (fp + 1 < 9) {
                        W43W_MM(acc[fp], a00.x, b00.x) W43W_MM(acc[fp + 1], a10.x, b10.x) W43W_MM(acc[fp], a00.y, b00.y)
                        W43W_MM(acc[fp + 1], a10.y, b10.y) W43W_MM(acc[fp], a01.x, b01.x) W43W_MM(acc[fp + 1], a11.x, b11.x)
                        W43W_MM(acc[fp], a01.y, b01.y) W43W_MM(acc[fp + 1], a11.y, b11.y)
                    } else {
                        W43W_MM(acc[fp], a00.x, b00.x) W43W_MM(acc[fp], a00.y, b00.y) W43W_MM(acc[fp], a01.x, b01.x)
                        W43W_MM(acc[fp], a01.y, b01.y)
                    }
#undef W43W_MM
                } else {
                    acc[fp][0] += a00.x * b00.x + a00.y * b00.y + a01.x * b01.x + a01.y * b01.y + a10.x * b10.x + a11.y * b11.y;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();
        }
        // the accumulators into the exchange buffer [f][c][k] (over the operand buffers: every wave is past the last window's
        // barrier, the producers' last store was a window earlier)
        if (!(W43W_ABLATE & 16)) {
#pragma unroll
            for (int fl = 0; fl < 9; ++fl)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * kh;            // input channel held by accumulator register r
                    smem[((f0 + fl) * 32 + row) * 32 + l31] = acc[fl][r];
                }
        }
    }

    // ---------------------------------------------------------------- G^T dU G + slab stores (all eight waves)
    __syncthreads();
    if (DBIAS && c0 == 0 && tid < BK)
        dbpart[(long)bz * K + k0 + tid] = (dbsum[tid] + dbsum[BK + tid]) + (dbsum[2 * BK + tid] + dbsum[3 * BK + tid]);
    if (W43W_ABLATE & 16) return;
    const int ok = tid & 31, oc = tid >> 5;             // this thread's output channel; its input channels are oc and oc + 16
    const long CK = (long)C * K;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int cc = oc + 16 * it;
        float g[3][6];                                  // g[r][j] = sum_i G[i][r] dU[i][j]
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            float m[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) m[i] = smem[((i * 6 + j) * 32 + cc) * 32 + ok];
            W43W_GT6(m[0], m[1], m[2], m[3], m[4], m[5], g[0][j], g[1][j], g[2][j])
        }
        float *dst = part + (long)bz * 9 * CK + (long)(c0 + cc) * K + k0 + ok;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            float o0, o1, o2;                           // dw[r][t] = sum_j g[r][j] G[j][t]
            W43W_GT6(g[r][0], g[r][1], g[r][2], g[r][3], g[r][4], g[r][5], o0, o1, o2)
            dst[(long)(r * 3 + 0) * CK] = o0;
            dst[(long)(r * 3 + 1) * CK] = o1;
            dst[(long)(r * 3 + 2) * CK] = o2;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// MMDGAN_WINO43_WGRAD=0: never; 1 (default): from wino43_wgrad_min_tiles 4x4 tiles on; 2: every eligible shape (parity tests)
bool wino43_wgrad_ok(const ConvDims &d) {
    const int mode = tuning().wino43_wgrad;
    if (mode == 0 || d.R != 3 || d.stride != 1 || d.pad != 1 || d.N < 2 || d.H % 4 || d.W % 4 || d.C % w43w::BC || d.K % w43w::BK) return false;
    if ((long)d.N * d.H * d.W * d.C * 4 >= w43w::kMaxBytes || (long)d.N * d.H * d.W * d.K * 4 >= w43w::kMaxBytes) return false;
    const long T = (long)d.N * (d.H / 4) * (d.W / 4);
    return mode == 2 ? T >= 8 : T >= tuning().wino43_wgrad_min_tiles;
}

// returns 0 (done; *dbias_done / *dot_done say what rode along), an error code, or 1: no workspace for the slabs (the caller
// takes another kernel)
int wino43_wgrad(const ConvDims &d, const float *x, const float *dy, float *dw, float *dbias, bool *dbias_done, hipStream_t st,
                 const float *wdot, float *dot, bool *dot_done) {
    using namespace w43w;
    if (dbias_done) *dbias_done = false;
    if (dot_done) *dot_done = false;
    const long T = (long)d.N * (d.H / 4) * (d.W / 4);
    const int nw = (int)((T + BT - 1) / BT);
    const int nblk_c = d.C / BC, nblk = nblk_c * (d.K / BK);
    int split = wgrad_cus() / nblk;                                 // one 8-wave workgroup per CU (144 KB of LDS), one round
    if (split > nw / 4) split = nw / 4;                             // >= 4 windows (144 MFMAs per wave) per workgroup
    if (split < 1) split = 1;
    const int wps = (nw + split - 1) / split;
    split = (nw + wps - 1) / wps;                                   // every slab gets written
    const size_t n = 9 * (size_t)d.C * d.K;
    SlabReduceArgs prev{};
    float *part = (float *)wgrad_slabs_acquire(sizeof(float) * (n + d.K) * split, st, &prev);
    if (!part) return 1;
    static bool cap_raised = false;
    if (!cap_raised) {
        (void)hipFuncSetAttribute((const void *)wino43_wgrad_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES);
        (void)hipFuncSetAttribute((const void *)wino43_wgrad_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES);
        cap_raised = true;
    }
    float *dbpart = part + n * split;
    const dim3 grid((unsigned)(nblk * split));
    if (dbias)
        hipLaunchKernelGGL(wino43_wgrad_kernel<true>, grid, dim3(NT), LDS_BYTES, st, d.N, d.H, d.W, d.C, d.K, x, dy, part, dbpart, nblk_c,
                           nblk, wps, prev);
    else
        hipLaunchKernelGGL(wino43_wgrad_kernel<false>, grid, dim3(NT), LDS_BYTES, st, d.N, d.H, d.W, d.C, d.K, x, dy, part, dbpart, nblk_c,
                           nblk, wps, prev);
    if (int rc = slab_reduce(part, split, n, dw, dbpart, dbias ? d.K : 0, dbias, st, wdot, dot)) return rc;
    if (dbias_done) *dbias_done = dbias != nullptr;
    if (dot_done) *dot_done = wdot != nullptr;
    return check_launch("conv2d_wgrad(winograd 4x4)");
}

}  // namespace mmdgan
