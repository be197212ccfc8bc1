// Winograd F(4x4, 3x3) convolution for the 3x3 / stride-1 layers (layer_func.py:912-916, op 'c'), forward and
// input-gradient: 36 multiplies per 4x4 output tile instead of 144 direct (F(2x2,3x3), conv_wino.hip: 64).
//
//   y = A^T [ sum_c (G g G^T) (.) (B^T d B) ] A      g: 3x3 filter, d: 6x6 input patch, y: 4x4 outputs
//
// with the interpolation points 0, 1, -1, 1/2, -2, infinity (tools/wino43_gate.py: 3.3-5.0e-6 of the output scale
// element-wise on D's three layers, forward / input-gradient / weight-gradient - Lavin's 0, +-1, +-2 measured 8e-6 - 1.3e-5).
// As GEMMs: for each of the 36 frequencies f = 6 i + j,  M_f[tile][k] = sum_c V_f[tile][c] * U_f[c][k].
//   * U = G g G^T comes from wino43_transform ([36][Cr / 8][2][Ko][4], the B-operand layout of conv_wino.hip).
//   * one workgroup = 32 tiles (MFMA rows) x 32 output channels x ALL 36 frequencies, EIGHT waves in two roles:
//       waves 0-3 (consumers): 9 frequencies each = 9 accumulators of v_mfma_f32_32x32x2_f32; per 8-channel stage 9
//                 ds_read_b128 (A fragments), 9 16-byte loads from L2 (B fragments, refilled right after their use) and
//                 36 MFMAs - nothing else;
//       waves 4-7 (producers): lane = (tile, channel pair of the window's 16): the 36 pixels of its 6x6 patch as 8-byte loads
//                 one window ahead (zero padding = the buffer range check: an out-of-image row or column adds 2^30 to the
//                 offset), B^T d B in registers on both channels at once (12 six-point passes of 16 packed operations), 36
//                 conflict-free ds_write_b64.  (First cut: one dword per lane and pixel, 144 vector-memory instructions per
//                 8 channels = 2450 cycles of the CU's address path against 2304 of MFMAs.  16-byte loads with a wave owning a
//                 whole stage needed 144 + 144 registers, spilled, and ran slower than that.)
//     One barrier per window of 16 channels (two stages of U), V double-buffered (2 x 72 KB).  A consumer wave and a producer wave share each SIMD: the matrix
//     pipe and the VALU are separate issue ports, so the transform runs beside the MFMAs instead of between them (the
//     F(2x2,3x3) kernel interleaves both roles in every wave by hand).
//   * the output transform needs all 36 frequencies of an element and they sit in four waves: after the last stage the
//     accumulators go through LDS ([36][32 tiles][32 channels] = 144 KB, the workgroup has the CU to itself); every thread
//     then owns two (tile, channel), reads their 36 values, does A^T M A in registers, transposes the 4 x 4 outputs with the
//     three other channels of its lane quad (two DPP butterfly rounds) and handles ONE output row of 4 pixels x 4 channels:
//     16-byte loads (activation derivative, addend) and stores, the activation switch once per row - not per element.
//   * SPLIT: blockIdx's third coordinate takes a slice of the channel reduction and writes its plain A^T M A sums into a
//     workspace slab; slab_epilogue() (conv_wino2.hip) sums the slabs and applies the epilogue.  For launches whose tile
//     count cannot fill the chip (D l5 / l7 at batch 64: 128 / 64 workgroups).
#include "conv_internal.h"
#include "bufload.h"
#include "wino_weight.h"

namespace mmdgan {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace w43 {
constexpr int NT = 512;                     // 4 consumer + 4 producer waves
[[maybe_unused]] constexpr int BC = 8;      // reduction channels per stage = 4 MFMA k-pairs (the granularity of U)
constexpr int BW = 16;                      // ... per WINDOW (one barrier, one V buffer) = two stages: a patch pixel is then 64
                                            // contiguous bytes - the CU's address path takes ~2 cycles per contiguous run whatever its
                                            // length, and at 32-byte runs (8 channels) that alone was 2450 cycles per 2304 of MFMAs
constexpr int FSV = 512;                    // V: floats per frequency = [stage of the window 2][k half 2][tile 32, swizzled][k-pair 4]
constexpr int V_FLOATS = 36 * FSV;          // one window of transformed activations (73,728 bytes)
constexpr int Z_FLOATS = 36 * 32 * 32;      // the epilogue's exchange buffer [36][32 tiles][32 channels] (over the V buffers)
constexpr int SMEM_FLOATS = Z_FLOATS;
static_assert(Z_FLOATS >= 2 * V_FLOATS, "exchange buffer covers the V double buffer");
constexpr size_t LDS_BYTES = sizeof(float) * SMEM_FLOATS + sizeof(long) * 32;   // + output offset of each tile
constexpr unsigned kPad = 0x40000000u;      // an out-of-image row or column: one or two of these keep the offset out of range
constexpr long kMaxBytes = 0x40000000L;     // ... for tensors below 1 GiB
}  // namespace w43

// B^T d for one six-point line: B^T = [1 -3/2 -2 3/2 1 0; 0 -1 1/2 5/2 1 0; 0 1 -5/2 1/2 1 0; 0 -2 -1 2 1 0; 0 1/2 -1 -1/2 1 0;
// 0 1 -3/2 -2 3/2 1]
#define W43_BT6(d0, d1, d2, d3, d4, d5, t0, t1, t2, t3, t4, t5)                        \
    {                                                                                  \
        const auto e_ = (d3) - (d1), g_ = (d4) - (d2);                                 \
        t0 = W43_FMA(-2.f, (d2), W43_FMA(1.5f, e_, (d0) + (d4)));                      \
        t1 = W43_FMA(2.5f, (d3), W43_FMA(0.5f, (d2), (d4) - (d1)));                    \
        t2 = W43_FMA(0.5f, (d3), W43_FMA(-2.5f, (d2), (d4) + (d1)));                   \
        t3 = W43_FMA(2.f, e_, g_);                                                     \
        t4 = W43_FMA(-0.5f, e_, g_);                                                   \
        t5 = W43_FMA(-2.f, (d3), W43_FMA(1.5f, g_, (d5) + (d1)));                      \
    }
__device__ __forceinline__ f32x2 w43_fma(float c, f32x2 a, f32x2 b) { return __builtin_elementwise_fma(f32x2{c, c}, a, b); }
__device__ __forceinline__ float w43_fma(float c, float a, float b) { return fmaf(c, a, b); }
#define W43_FMA(c, a, b) w43_fma((c), (a), (b))
// A^T m for one six-point line: A^T = [1 1 1 1 1 0; 0 1 -1 1/2 -2 0; 0 1 1 1/4 4 0; 0 1 -1 1/8 -8 1]
#define W43_AT6(m0, m1, m2, m3, m4, m5, y0, y1, y2, y3)                                \
    {                                                                                  \
        const float s_ = (m1) + (m2), d_ = (m1) - (m2);                                \
        y0 = (m0) + s_ + ((m3) + (m4));                                                \
        y1 = fmaf(-2.f, (m4), fmaf(0.5f, (m3), d_));                                   \
        y2 = fmaf(4.f, (m4), fmaf(0.25f, (m3), s_));                                   \
        y3 = fmaf(-8.f, (m4), fmaf(0.125f, (m3), d_)) + (m5);                          \
    }

template <bool FLIP>
__global__ __launch_bounds__(256) void wino43_weight_kernel(const float *__restrict__ w, float *__restrict__ U, int C, int K) {
    __shared__ float tile[9][32][33];
    wino43_weight_block<FLIP>(tile, blockIdx.x, blockIdx.y, w, U, C, K);
}

// compile-time ablation masks (tools/wino43_ablate.sh); 0 in the library build
#ifndef W43_ABLATE
#define W43_ABLATE 0
#endif
// 1: no patch loads (producers transform zeros)   2: no transform / V stores   4: no MFMAs   8: no B-fragment loads
// 16: no epilogue exchange / output transform (stores zeros)   32: no stores at all
// 256: no V stores (the transform stays)
// 64: every patch load reads the tensor's first pixel (L1 hits: what the address path alone costs)   128: every tile reads the same 6x6 pixels

#ifndef W43_PACE
#define W43_PACE 0
#endif
#if W43_PACE == 0
#define W43_PACE_NOPS
#elif W43_PACE == 1
#define W43_PACE_NOPS __builtin_amdgcn_sched_barrier(0); asm volatile("s_nop 15"); __builtin_amdgcn_sched_barrier(0);
#elif W43_PACE == 2
#define W43_PACE_NOPS __builtin_amdgcn_sched_barrier(0); asm volatile("s_nop 15\n\ts_nop 15"); __builtin_amdgcn_sched_barrier(0);
#elif W43_PACE == 3
#define W43_PACE_NOPS __builtin_amdgcn_sched_barrier(0); asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15"); __builtin_amdgcn_sched_barrier(0);
#else
#define W43_PACE_NOPS __builtin_amdgcn_sched_barrier(0); asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 7"); __builtin_amdgcn_sched_barrier(0);
#endif
#ifdef W43_TRACE      // measurement builds only (tools/wino43_trace.py): shader-clock stamps of one workgroup's windows
__device__ long g_w43_trace[8 * 256];
#define W43_STAMP(ROW, J) \
    if (blockIdx.x == W43_TRACE && lane == 0 && (J) < 256) g_w43_trace[(ROW) * 256 + (J)] = (long)__builtin_readcyclecounter();
#else
#define W43_STAMP(ROW, J)
#endif

// x [N,H,W,Cr] (*) U [36][Cr][Ko] -> out [N,H,W,Ko], 'SAME' padding, stride 1, H and W multiples of 4
template <bool SPLIT>
__global__ __launch_bounds__(w43::NT) void wino43_kernel(int N, int H, int W, int Cr, int Ko, ConvEpilogue ep,
                                                         const float *__restrict__ x, const float *__restrict__ U,
                                                         float *__restrict__ out, int ntb, int ncb, int stages_per_split,
                                                         long slab_elems) {
    using namespace w43;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, kh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int TH = H >> 2, TW = W >> 2;
    const long T = (long)N * TH * TW;
    // The workgroups of one tile block (same patches, different output-channel blocks) sit next to each other on ONE XCD
    // (hardware deals linear workgroup ids round-robin over the 8 XCDs): the second reader of a patch is served by that L2.
    int cb, tb, bz;
    {
        const int n = gridDim.x, id = blockIdx.x, xcd = id & 7;
        const int v = xcd * (n >> 3) + min(xcd, n & 7) + (id >> 3);
        cb = v % ncb;
        tb = (v / ncb) % ntb;
        bz = v / (ncb * ntb);
    }
    const int t0 = tb * 32, n0 = cb * 32;
    const int s_begin = SPLIT ? bz * stages_per_split : 0;                  // in WINDOWS of 16 channels
    const int s_end = SPLIT ? min(Cr / BW, s_begin + stages_per_split) : Cr / BW;
    long *obase = reinterpret_cast<long *>(smem + SMEM_FLOATS);
    if (wave == 0) { W43_STAMP(7, 0) }

    if (wave >= 4) {
#ifdef W43_PRODUCER_PRIO
        __builtin_amdgcn_s_setprio(W43_PRODUCER_PRIO);
#endif
        // ---------------------------------------------------------------- producers: patches -> B^T d B -> V
        // lane of the four waves = (tile, channel PAIR of the window's 16): everything below works on f32x2 (v_pk_fma_f32 /
        // v_pk_add_f32, two channels per instruction).  With the stage's channel c on k half (c >> 1) & 1, k-pair (c & 1) +
        // 2 (c >> 2) - the order wino43_weight_block lays U out in - a pair is two consecutive k-pairs of one k half: one
        // 8-byte store of V per frequency.
        const int pl = (wave - 4) * 64 + lane, pt = pl >> 3, cp = pl & 7, sub = cp >> 2, pkh = cp & 1, pkp = cp & 2;
        unsigned rowb[6], colb[6];
        {
            const long id = (long)t0 + pt;
            const bool ok = id < T;
            const long ii = ok ? id : 0;
            const int tx = ii % TW, ty = (ii / TW) % TH, n = ii / ((long)TW * TH);
            if (cp == 0) obase[pt] = ok ? (((long)n * H + 4 * ty) * W + 4 * tx) * Ko : -1;   // output pixel (4ty, 4tx), channel 0
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                const int y = 4 * ty - 1 + r, xx = 4 * tx - 1 + r;
                rowb[r] = (ok && y >= 0 && y < H) ? (unsigned)(((((long)n * H + y) * W) * Cr + 2 * cp) * 4) : kPad;
                colb[r] = (xx >= 0 && xx < W) ? (unsigned)((long)xx * Cr * 4) : kPad;
                if (W43_ABLATE & 64) { rowb[r] = (unsigned)(2 * cp * 4); colb[r] = 0; }     // every load from the tensor's first pixel
            }
        }
        unsigned poff[36];
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int jj = 0; jj < 6; ++jj) poff[r * 6 + jj] = rowb[r] + colb[jj];
        const __amdgpu_buffer_rsrc_t rx = make_rsrc(x, (long)N * H * W * Cr * 4);
        // V[f][stage of the window][k half][tile ^ 4 kh ^ 2 stage][k-pair]: the swizzle spreads a quarter-wave of the stores
        // (2 tiles x 8 pairs) over all 32 banks and keeps the consumers' 16-byte A-fragment reads conflict-free
        const int vd = sub * 256 + pkh * 128 + ((pt ^ (pkh << 2) ^ (sub << 1)) << 2) + pkp;
        f32x2 da[36], db[36];                           // raw patches: pixel (r, j) x 2 channels, two windows in flight
        const int S = s_end - s_begin;
        auto xload = [&](f32x2(&d)[36], int j) __attribute__((always_inline)) {   // the patches of window j (beyond the last:
            if (W43_ABLATE & 1) {                       // re-reads it, unused - no branch, no zero fill: either makes the wave wait
#pragma unroll                                          // for every load in flight)
                for (int e = 0; e < 36; ++e) {
                    float o = (float)(lane + e);        // (opaque: the transform below is not folded away)
                    asm volatile("" : "+v"(o));
                    d[e] = f32x2{o, o};
                }
                return;
            }
            // 36 vector-memory instructions and NOTHING else: a vector-memory instruction does not use the SIMD's VALU / MFMA issue
            // port (everything that does - MFMA passes included - runs strictly one after the other on a SIMD, whichever wave
            // it comes from), so a request proceeds beside the consumer's MFMAs; the 36 offsets live in registers for that
            const unsigned so = (unsigned)((s_begin + min(j, S - 1)) * BW * 4);
#pragma unroll
            for (int e = 0; e < 36; ++e) d[e] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rx, poff[e], so, 0));
        };
        // V <- B^T d B: along the columns of each patch row (in place), then along the rows of each frequency column - the six
        // frequencies of a column go to LDS as soon as they exist, so nothing but d itself stays alive
        auto transform_dump = [&](f32x2(&d)[36], float *buf) __attribute__((always_inline)) {
            if (W43_ABLATE & 2) return;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                f32x2 t0, t1, t2, t3, t4, t5;
                W43_BT6(d[r * 6 + 0], d[r * 6 + 1], d[r * 6 + 2], d[r * 6 + 3], d[r * 6 + 4], d[r * 6 + 5], t0, t1, t2, t3, t4, t5)
                d[r * 6 + 0] = t0; d[r * 6 + 1] = t1; d[r * 6 + 2] = t2; d[r * 6 + 3] = t3; d[r * 6 + 4] = t4; d[r * 6 + 5] = t5;
            }
#pragma unroll
            for (int jj = 0; jj < 6; ++jj) {
                f32x2 v0, v1, v2, v3, v4, v5;
                W43_BT6(d[0 + jj], d[6 + jj], d[12 + jj], d[18 + jj], d[24 + jj], d[30 + jj], v0, v1, v2, v3, v4, v5)
                if (W43_ABLATE & 256) {                 // results kept alive, no LDS stores
                    asm volatile("" ::"v"(v0), "v"(v1), "v"(v2), "v"(v3), "v"(v4), "v"(v5));
                } else {
                    *reinterpret_cast<f32x2 *>(buf + (0 + jj) * FSV + vd) = v0;
                    *reinterpret_cast<f32x2 *>(buf + (6 + jj) * FSV + vd) = v1;
                    *reinterpret_cast<f32x2 *>(buf + (12 + jj) * FSV + vd) = v2;
                    *reinterpret_cast<f32x2 *>(buf + (18 + jj) * FSV + vd) = v3;
                    *reinterpret_cast<f32x2 *>(buf + (24 + jj) * FSV + vd) = v4;
                    *reinterpret_cast<f32x2 *>(buf + (30 + jj) * FSV + vd) = v5;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        // window j = the consumers multiply window j (of this block's S); V of window j + 1 is written during it.  Per window:
        // request the patches two windows on into the registers that were transformed a window ago, then transform the ones
        // requested a window ago and store V.  (One register set, requested a window ahead: the chain request -> ~4700 cycles until the 36 loads of all
        // four waves have landed -> transform -> next request made the window 7000 cycles against 4608 of MFMAs -
        // tools/wino43_trace.py.  And no register spill anywhere in this loop: a scratch reload is a vector-memory load behind
        // the 36 just requested, and waiting for it waits for all of them.)
        xload(da, 0);
        xload(db, 1);
        transform_dump(da, smem);
        __syncthreads();
        for (int j = 0;;) {                             // window j: request window j + 2 into the registers transformed a window ago,
            if (wave == 4) { W43_STAMP(2, j) }          // transform + store window j + 1 (requested a window ago)
            xload(da, j + 2);
            if (wave == 4) { W43_STAMP(4, j) }
            transform_dump(db, smem + ((j + 1) & 1) * V_FLOATS);     // (past the last window: a copy of it, into the buffer nobody reads)
            if (wave == 4) { W43_STAMP(5, j) }
            __syncthreads();
            if (++j >= S) break;
            if (wave == 4) { W43_STAMP(2, j) }
            xload(db, j + 2);
            if (wave == 4) { W43_STAMP(4, j) }
            transform_dump(da, smem + ((j + 1) & 1) * V_FLOATS);
            if (wave == 4) { W43_STAMP(5, j) }
            __syncthreads();
            if (++j >= S) break;
        }
    } else {
        // ---------------------------------------------------------------- consumers: 9 frequencies x 32 tiles x 32 channels
        f32x16 acc[9];                                  // (declared HERE: visible to both roles, its zeroing is hoisted above the
#pragma unroll                                          //  role branch and the producers lose 144 registers to it)
        for (int fl = 0; fl < 9; ++fl)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[fl][r] = 0.f;
        const __amdgpu_buffer_rsrc_t ru = make_rsrc(U, (long)36 * Cr * Ko * 4);
        // B fragments of frequency f, stage s: U[f][s][kh][n0 + l31][4 k-pairs] - one 16-byte load
        const unsigned ubase = (unsigned)(((long)kh * Ko + n0 + l31) * 16);
        const unsigned ustage = (unsigned)(2 * Ko * 16), ufreq = (unsigned)((long)Cr * Ko * 4);
        const int f0 = 9 * wave;
        // A fragments: V[f][stage of the window][kh][tile (swizzled)][4 k-pairs], one ds_read_b128
        const int abase0 = kh * 128 + ((l31 ^ (kh << 2)) << 2), abase1 = 256 + kh * 128 + ((l31 ^ (kh << 2) ^ 2) << 2);
        float4 fb[9], fa[2];                            // (B fragments a whole window ahead instead of a stage: measured, no gain)
#pragma unroll
        for (int fl = 0; fl < 9; ++fl)
            fb[fl] = (W43_ABLATE & 8) ? make_float4(1.f, 1.f, 1.f, 1.f)
                                      : bufld4s(ru, ubase, (unsigned)(f0 + fl) * ufreq + (unsigned)(2 * s_begin) * ustage);
        __syncthreads();
        if (wave == 0) { W43_STAMP(7, 1) }
        const int st_end = 2 * s_end;                   // in stages of 8 channels
        for (int s = s_begin; s < s_end; ++s) {
            if (wave == 0) { W43_STAMP(0, s - s_begin) }
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                const float *cur = smem + ((s - s_begin) & 1) * V_FLOATS + f0 * FSV + (sub ? abase1 : abase0);
                const int sn = 2 * s + sub + 1 < st_end ? 2 * s + sub + 1 : 2 * s + sub;     // the last refill re-reads the last stage
                // frequencies in pairs (the last one alone): consecutive MFMAs go to different accumulators, the A fragments of
                // the next pair are fetched while this one is multiplied, each B fragment is refilled for the next stage right
                // after its use
                fa[0] = *reinterpret_cast<const float4 *>(cur);
                fa[1] = *reinterpret_cast<const float4 *>(cur + FSV);
#pragma unroll
                for (int fp = 0; fp < 9; fp += 2) {
                    const float4 a0 = fa[0], a1 = fa[1];
                    if (fp + 2 < 9) fa[0] = *reinterpret_cast<const float4 *>(cur + (fp + 2) * FSV);
                    if (fp + 3 < 9) fa[1] = *reinterpret_cast<const float4 *>(cur + (fp + 3) * FSV);
                    const float4 b0 = fb[fp], b1 = fb[fp + 1 < 9 ? fp + 1 : fp];
                    if (!(W43_ABLATE & 4)) {
                        // W43_PACE: after every MFMA the wave idles on s_nop until the matrix pipe is about to take the next
                        // one, instead of standing at that MFMA - the SIMD's issue port then goes to the producer wave
#define W43_MM(ACC, A_, B_)                                                        \
    ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(A_, B_, ACC, 0, 0, 0);             \
    W43_PACE_NOPS
                        if (fp + 1 < 9) {
                            W43_MM(acc[fp], a0.x, b0.x) W43_MM(acc[fp + 1], a1.x, b1.x) W43_MM(acc[fp], a0.y, b0.y)
                            W43_MM(acc[fp + 1], a1.y, b1.y) W43_MM(acc[fp], a0.z, b0.z) W43_MM(acc[fp + 1], a1.z, b1.z)
                            W43_MM(acc[fp], a0.w, b0.w) W43_MM(acc[fp + 1], a1.w, b1.w)
                        } else {
                            W43_MM(acc[fp], a0.x, b0.x) W43_MM(acc[fp], a0.y, b0.y) W43_MM(acc[fp], a0.z, b0.z) W43_MM(acc[fp], a0.w, b0.w)
                        }
#undef W43_MM
                    } else {
                        acc[fp][0] += a0.x * b0.x + a0.y * b0.y + a0.z * b0.z + a0.w * b0.w + a1.x * b1.x;
                    }
                    if (!(W43_ABLATE & 8)) {
                        fb[fp] = bufld4s(ru, ubase, (unsigned)(f0 + fp) * ufreq + (unsigned)sn * ustage);
                        if (fp + 1 < 9) fb[fp + 1] = bufld4s(ru, ubase, (unsigned)(f0 + fp + 1) * ufreq + (unsigned)sn * ustage);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (wave == 0) { W43_STAMP(1, s - s_begin) }
            __syncthreads();
        }
        if (wave == 0) { W43_STAMP(6, 0) }
        // the accumulators into the exchange buffer [f 36][tile 32][channel 32] (over the V buffers: every wave is past the
        // last stage's barrier, the producers' last store was a window earlier)
        if (!(W43_ABLATE & 16)) {
#pragma unroll
            for (int fl = 0; fl < 9; ++fl)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * kh;            // tile held by accumulator register r
                    smem[((f0 + fl) * 32 + row) * 32 + l31] = acc[fl][r];
                }
        }
    }

    // ---------------------------------------------------------------- output transform + epilogue (all eight waves)
    const float sc = ep.scale ? ep.scale[0] : 1.f;
    float *Zs = smem;                                   // [f 36][tile 32][channel 32], written by the consumers above
    if (wave == 0) { W43_STAMP(6, 1) }
    __syncthreads();
    if (wave == 0) { W43_STAMP(6, 2) }
    const int oc = tid & 31, ot = tid >> 5;             // this thread's channel; its tiles are ot and ot + 16
    const int kq = oc & 3;                              // ... and, after the quad transpose, its output row
    const int ch4 = n0 + (oc & ~3);                     // first of the quad's four channels
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!SPLIT && ep.bias) bv = *reinterpret_cast<const float4 *>(ep.bias + ch4);
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int tile = ot + 16 * it;
        float y[16];
        if (W43_ABLATE & 16) {
#pragma unroll
            for (int e = 0; e < 16; ++e) y[e] = 0.f;
        } else {
            float z[24];                                // z[i][b] = sum_j M[i][j] A^T[b][j]
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                float m[6];
#pragma unroll
                for (int j = 0; j < 6; ++j) m[j] = Zs[((i * 6 + j) * 32 + tile) * 32 + oc];
                W43_AT6(m[0], m[1], m[2], m[3], m[4], m[5], z[i * 4 + 0], z[i * 4 + 1], z[i * 4 + 2], z[i * 4 + 3])
            }
#pragma unroll
            for (int b = 0; b < 4; ++b)                 // y[a][b] = sum_i A^T[a][i] z[i][b]
                W43_AT6(z[0 + b], z[4 + b], z[8 + b], z[12 + b], z[16 + b], z[20 + b], y[0 + b], y[4 + b], y[8 + b], y[12 + b])
        }
        // 4 x 4 transpose across the lane quad (the four channels 4q .. 4q + 3 of one tile): afterwards y[c][b] of lane k is
        // channel 4q + c of output pixel (row k, column b).  Two butterfly rounds: rows (0,1),(2,3) with lane ^ 1, then rows
        // (0,2),(1,3) with lane ^ 2
#pragma unroll
        for (int b = 0; b < 4; ++b) {
#pragma unroll
            for (int rnd = 0; rnd < 2; ++rnd) {
                const int bit = rnd ? 2 : 1;
                const bool up = (kq & bit) != 0;
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    const int r0 = rnd ? pr : 2 * pr, r1 = r0 + bit;           // the row pair (r0 has the bit clear)
                    const float cand0 = y[r0 * 4 + b], cand1 = y[r1 * 4 + b];
                    const float give = up ? cand0 : cand1;
                    const float got = __int_as_float(rnd ? __builtin_amdgcn_update_dpp(0, __float_as_int(give), 0x4E, 0xF, 0xF, false)
                                                         : __builtin_amdgcn_update_dpp(0, __float_as_int(give), 0xB1, 0xF, 0xF, false));
                    const float keep0 = y[r0 * 4 + b], keep1 = y[r1 * 4 + b];
                    y[r0 * 4 + b] = up ? got : keep0;
                    y[r1 * 4 + b] = up ? keep1 : got;
                }
            }
        }
        const long ob = obase[tile];
        if (ob >= 0 && !(W43_ABLATE & 32)) {
            float4 v[4];
            long o[4];
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                o[b] = ob + ((long)kq * W + b) * Ko + ch4;
                v[b] = make_float4(y[0 + b], y[4 + b], y[8 + b], y[12 + b]);
            }
            if (SPLIT) {                                // this part's plain sums into its slab; the epilogue follows the slab sum
#pragma unroll
                for (int b = 0; b < 4; ++b) *reinterpret_cast<float4 *>(out + (long)bz * slab_elems + o[b]) = v[b];
                continue;
            }
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                v[b].x = fmaf(v[b].x, sc, bv.x); v[b].y = fmaf(v[b].y, sc, bv.y);
                v[b].z = fmaf(v[b].z, sc, bv.z); v[b].w = fmaf(v[b].w, sc, bv.w);
            }
            if (ep.dact) {
                float4 yv[4];
#pragma unroll
                for (int b = 0; b < 4; ++b) yv[b] = *reinterpret_cast<const float4 *>(ep.dact + ep.dact_index(o[b]));
                if (ep.act == MMDGAN_ACT_LRELU) {
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        v[b].x *= yv[b].x > 0.f ? 1.f : kLreluAlpha; v[b].y *= yv[b].y > 0.f ? 1.f : kLreluAlpha;
                        v[b].z *= yv[b].z > 0.f ? 1.f : kLreluAlpha; v[b].w *= yv[b].w > 0.f ? 1.f : kLreluAlpha;
                    }
                } else if (ep.act != MMDGAN_ACT_LINEAR) {
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        v[b].x *= act_bwd_from_out(yv[b].x, ep.act); v[b].y *= act_bwd_from_out(yv[b].y, ep.act);
                        v[b].z *= act_bwd_from_out(yv[b].z, ep.act); v[b].w *= act_bwd_from_out(yv[b].w, ep.act);
                    }
                }
            } else if (ep.act == MMDGAN_ACT_LRELU) {
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    v[b].x = v[b].x > 0.f ? v[b].x : v[b].x * kLreluAlpha; v[b].y = v[b].y > 0.f ? v[b].y : v[b].y * kLreluAlpha;
                    v[b].z = v[b].z > 0.f ? v[b].z : v[b].z * kLreluAlpha; v[b].w = v[b].w > 0.f ? v[b].w : v[b].w * kLreluAlpha;
                }
            } else if (ep.act != MMDGAN_ACT_LINEAR) {
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    v[b].x = act_fwd(v[b].x, ep.act); v[b].y = act_fwd(v[b].y, ep.act);
                    v[b].z = act_fwd(v[b].z, ep.act); v[b].w = act_fwd(v[b].w, ep.act);
                }
            }
#pragma unroll
            for (int b = 0; b < 4; ++b) *reinterpret_cast<float4 *>(out + o[b]) = ep.add4(v[b], o[b]);
        }
    }
    if (wave == 0) { W43_STAMP(6, 3) }
}
#ifdef W43_TRACE
extern "C" int mmdgan_w43_trace(long *host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_w43_trace), sizeof(long) * (n < 8 * 256 ? n : 8 * 256), 0, hipMemcpyDeviceToHost);
}
#endif

// ------------------------------------------------------------------------------------------------
// MMDGAN_WINO43=0: never; 1 (default): where it measured faster than F(2x2,3x3); 2: every eligible shape (the parity tests)
static int wino43_mode() { return tuning().wino43; }

bool wino43_shape_ok(const ConvDims &d, int cr, int ko) {
    return d.R == 3 && d.stride == 1 && d.pad == 1 && d.N > 1 && d.H % 4 == 0 && d.W % 4 == 0 && cr % w43::BW == 0 && cr >= 32 &&
           ko % 32 == 0 && (long)d.N * d.H * d.W * cr * 4 < w43::kMaxBytes && (long)36 * cr * ko * 4 < 0xffffffffL;
}
bool wino43_eligible(const ConvDims &d, bool dgrad) {
    const int mode = wino43_mode();
    const int cr = dgrad ? d.K : d.C, ko = dgrad ? d.C : d.K;
    if (mode == 0 || !wino43_shape_ok(d, cr, ko)) return false;
    const long T = (long)d.N * (d.H / 4) * (d.W / 4);
    return mode == 2 ? T >= 32 : T >= tuning().wino43_min_tiles;
}

int wino43_transform(const ConvDims &d, const float *w, bool flip, float *U, hipStream_t st) {
    if ((flip ? d.K : d.C) % 8 || (flip ? d.C : d.K) % 32) {
        set_error("wino43_transform: reduction-side channels (%d) must be a multiple of 8 (16 to run), output-side (%d) of 32", flip ? d.K : d.C,
                  flip ? d.C : d.K);
        return MMDGAN_E_ARG;
    }
    const dim3 wg((d.K + 31) / 32, (d.C + 31) / 32);
    if (flip) hipLaunchKernelGGL(wino43_weight_kernel<true>, wg, dim3(256), 0, st, w, U, d.C, d.K);
    else hipLaunchKernelGGL(wino43_weight_kernel<false>, wg, dim3(256), 0, st, w, U, d.C, d.K);
    return check_launch("wino43_transform");
}

// U = the weights transformed by wino43_transform (the caller's tensor, or nullptr: transformed into the workspace here)
static int wino43_launch(const ConvDims &d, const ConvEpilogue &ep, const float *in, const float *w, const float *U, float *out,
                         bool flip, hipStream_t st) {
    const int cr = flip ? d.K : d.C, ko = flip ? d.C : d.K;
    const bool own_u = U != nullptr;
    if (!U) {
        float *ws = (float *)workspace_acquire(sizeof(float) * 36 * (size_t)cr * ko, st);
        if (!ws) { set_error("conv2d (winograd 4x4): no workspace for the transformed weights"); return MMDGAN_E_ARG; }
        if (int rc = wino43_transform(d, w, flip, ws, st)) return rc;
        U = ws;
    }
    const long T = (long)d.N * (d.H / 4) * (d.W / 4);
    const int ntb = (int)((T + 31) / 32), ncb = ko / 32;
    const long wgs = (long)ntb * ncb;
    const int nstages = cr / w43::BW;                   // windows of 16 channels
    // too few tiles for one workgroup per CU and the workspace free (weights transformed by the caller): split the channel
    // reduction over workspace slabs; each part keeps >= 8 stages
    int split = 1;
    if (own_u && wgs < tuning().wino43_ksplit_below)
        while (split < 8 && wgs * split < 256 && nstages % (2 * split) == 0 && nstages / (2 * split) >= 4) split *= 2;
    const long total = (long)d.N * d.H * d.W * ko;
    float *slabs = nullptr;
    if (split > 1) slabs = (float *)workspace_acquire(sizeof(float) * (size_t)split * total, st);
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(wino43_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)w43::LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(wino43_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)w43::LDS_BYTES);
        attr_done = true;
    }
    if (slabs) {
        ConvEpilogue plain{};
        plain.wrap_from = kNoWrap;
        hipLaunchKernelGGL((wino43_kernel<true>), dim3((unsigned)(wgs * split)), dim3(w43::NT), w43::LDS_BYTES, st, d.N, d.H, d.W, cr, ko,
                           plain, in, U, slabs, ntb, ncb, nstages / split, total);
        if (int rc = check_launch(flip ? "conv2d_dgrad(winograd 4x4 split)" : "conv2d_fwd(winograd 4x4 split)")) return rc;
        return slab_epilogue(slabs, split, total, ko, ep, out, st);
    }
    hipLaunchKernelGGL((wino43_kernel<false>), dim3((unsigned)wgs), dim3(w43::NT), w43::LDS_BYTES, st, d.N, d.H, d.W, cr, ko, ep, in, U,
                       out, ntb, ncb, nstages, 0L);
    addend_applied();
    return check_launch(flip ? "conv2d_dgrad(winograd 4x4)" : "conv2d_fwd(winograd 4x4)");
}

int wino43_fwd(const ConvDims &d, const ConvEpilogue &ep, const float *x, const float *w, const float *U, float *y, hipStream_t st) {
    return wino43_launch(d, ep, x, w, U, y, false, st);
}
int wino43_dgrad(const ConvDims &d, const ConvEpilogue &ep, const float *dy, const float *w, const float *U, float *dx, hipStream_t st) {
    return wino43_launch(d, ep, dy, w, U, dx, true, st);
}

}  // namespace mmdgan
