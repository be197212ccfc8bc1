// Winograd F(4x4, 3x3) WEIGHT GRADIENT of the 3x3 / stride-1 layers (layer_func.py:912-916, op 'c'): the third kernel of
// the family in conv_wino43.hip (same points 0, 1, -1, 1/2, -2, infinity, same matrices):
//
//   dW = G^T [ sum_tiles (B^T d B) (.) (A dY A^T) ] G       d: 6x6 input patch, dY: the tile's 4x4 output gradients
//
// i.e. per frequency f = 6 i + j a GEMM  dU_f[c][k] = sum_tiles V_f[tile][c] * M_f[tile][k]  whose reduction runs over the
// 4x4 output tiles: 36 multiplies per tile and (c, k) instead of 144 direct, or 64 as F(2x2,3x3) tiles (conv_wino.hip).
//   * one workgroup = 32 input channels x 32 output channels x ALL 36 frequencies x a slice of the tile range (one round of
//     workgroups, XCD-grouped by tile range), EIGHT waves, a consumer and a producer on every SIMD:
//       waves 0-3 (consumers): 9 frequencies each = 9 accumulators of v_mfma_f32_32x32x2_f32; per window of 8 tiles (4 MFMA
//                 k-pairs) 36 ds_read2_b32 (both operands come from LDS - neither is a weight) and 36 MFMAs - nothing else;
//       waves 4-7 (producers): two PAIRS on alternate windows, lane = (tile, channel pair) for BOTH operands - the 6x6 patch
//                 of two input channels and the 4x4 gradients of two output channels as 8-byte requests (zero padding = the
//                 buffer range check), B^T d B and A dY A^T packed on the two channels, 72 conflict-free 8-byte LDS stores;
//                 the workgroups of channel block 0 also add up the gradients they load: the bias gradient rides along.
//                 (How the producers got this shape - three earlier cuts and what each measured - is at their code below.)
//     One LDS-only barrier per window, both operands double-buffered (2 x 72 KB), operand layout [f][k half][k-pair][channel].
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
constexpr int FS = 2 * 4 * 32;                  // floats per frequency of one operand: [k half][k-pair 4][channel 32]
constexpr int OP_FLOATS = 36 * FS;              // V (or M) of one window
constexpr int WIN_FLOATS = 2 * OP_FLOATS;       // V then M: 73,728 bytes
constexpr int Z_FLOATS = 36 * 32 * 32;          // the epilogue's exchange buffer [f][c][k]
static_assert(Z_FLOATS == 2 * WIN_FLOATS, "the exchange buffer is the double buffer");
constexpr size_t LDS_BYTES = sizeof(float) * (Z_FLOATS + 16 * BK);     // + sixteen partial bias rows
constexpr unsigned kPad = 0x40000000u;          // an out-of-image row / column / tile: keeps the offset out of range
constexpr long kMaxBytes = 0x40000000L;         // ... for tensors below 1 GiB
}  // namespace w43w

// compile-time ablation masks (tools/wino43w_ablate.sh); 0 in the library build
//   1: no patch / dY loads   2: no transforms, no operand stores   4: no MFMAs   8: no operand reads (LDS)   16: no epilogue
//   32: no operand stores (the transforms stay)
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

#ifdef W43W_TRACE     // measurement builds only (tools/wino43w_trace.py): shader-clock stamps of one workgroup's windows
__device__ long g_w43w_trace[8 * 256];
#define W43W_STAMP(ROW, J) \
    if (blockIdx.x == W43W_TRACE && lane == 0 && (J) < 256) g_w43w_trace[(ROW) * 256 + (J)] = (long)__builtin_readcyclecounter();
#else
#define W43W_STAMP(ROW, J)
#endif

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
    if (wave == 0) { W43W_STAMP(7, 0) }
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
    float *dbsum = smem + Z_FLOATS;                                             // [producer wave 4 x tile of the wave 4][k 32]

    if (wave >= 4) {
        // ------------------------------------------------------------ producers: two PAIRS of waves take alternate windows
        // (pair g: windows g, g + 2, ...); within a pair wave h holds tiles 4 h .. 4 h + 3 of the window, lane = (tile, channel
        // PAIR): the tile's 6x6 patch of input channels c0 + 2 cp, + 1 and its 4x4 output gradients of channels k0 + 2 cp, + 1
        // as 8-byte loads (one vector-memory instruction costs the CU's address path ~16 cycles whatever it carries: with a
        // dword per lane the 208 requests of a window took as long as its MFMAs - tools/wino43w_trace.py).  A pair has TWO
        // windows per window of its own, and splits them by what the LDS double buffer allows:
        //   duty window (the consumers multiply window w - 1): A dY A^T, all 72 operand stores of window w (8 bytes each, into
        //                the buffer window w - 2 was read from), then the requests of window w + 2 into the registers just freed;
        //   off window:  B^T d B of window w + 2, in place, as its patches arrive.
        // Every SIMD carries the same transform work beside its consumer wave (VALU from the partner wave costs the MFMA stream
        // about half of its own issue time, tools/mfma_valu_overlap.hip), and no window waits for a request's latency.
        // History (D l3 at batch 128, alone, with reduction + bias gradient; F(2x2,3x3) slab kernel 72 us; profiles/
        // r06_wino43w_ablation.txt): (1) waves 4-5 transform the patches, 6-7 the gradients, dword requests, two windows ahead:
        // 62.6 us - the window waited for the two SIMDs that held the patches; (2) every producer wave one tile of both
        // operands, scalar fp32, three sets in flight: 56.1 (52.2 with the priority below); (3) this shape, requests in a burst
        // behind the stores: 53.9; (4) each request right behind the store that frees its register, as inline assembly tied to
        // that register with hand-placed s_waitcnt: 50.9 - and WRONG under load (below: requests); (5) the same order through
        // the builtin, the loop written without a merge of its two window kinds, the transform pinned in its window: 54.4
        // (a window ~3600 cycles against 2304 of MFMAs, tools/wino43w_trace.py); (6) without the producers' priority: 51.1.
        // (No static priority: s_setprio 1 on the producers won 6 % on cut (2) of the history above and LOSES 4-8 % on the shipped
        // form - D l3 53.1 vs 51.1 us, CelebA D l7 266 vs 244, ResNet 64x64 85.9 vs 79.6 - where the consumers are what a window
        // waits for; -DW43W_PRODUCER_PRIO=n builds the variants, profiles/r06_wino43w_ablation.txt.)
#ifdef W43W_PRODUCER_PRIO
        __builtin_amdgcn_s_setprio(W43W_PRODUCER_PRIO);
#endif
        const int g = (wave - 4) >> 1, hw = (wave - 4) & 1;
        const int tl = 4 * hw + (lane >> 4), cp = lane & 15;        // tile of the window, channel pair
        const int vd = (tl & 1) * 128 + (tl >> 1) * 32 + 2 * cp;     // [k half][k-pair][channel]: + f * FS (+ OP_FLOATS: M)
        int id = (w_begin + g) * BT + tl;                            // this lane's tile of the pair's next window
        int tx = id % TW, ty = (id / TW) % TH, tn = id / (TW * TH);
        const int s_tx = (2 * BT) % TW, s_ty = ((2 * BT) / TW) % TH, s_n = (2 * BT) / (TW * TH);
        // patch pixel (r, j) of a tile = image pixel (4 ty - 1 + r, 4 tx - 1 + j): the resource starts one row and one pixel
        // BEFORE the tensor, so that pixel's offset is tilebase + r * rowbytes + j * pixbytes with a wave-uniform second part
        // (the scalar offset of the load: no address arithmetic per load); the border rows / columns that lie outside the image
        // take an out-of-range tilebase instead (nine variants per window: {top, middle, bottom} x {left, middle, right})
        const unsigned xrow = (unsigned)(W * C * 4), xpix = (unsigned)(C * 4);
        // (The requests go through the compiler's builtin, which places the waits.  A cut with inline-assembly requests tied to the
        // register they replace and hand-placed s_waitcnt was 5 % faster alone and WRONG under load: hipcc, which takes an asm's
        // output for a finished value, copied registers whose request was still in flight - across the wait that named only half
        // of them, and at the loop's merge point - so a cold or contended run multiplied stale values; found by the full GPU
        // suite, reproduced by a stress run with a second stream, round 6.)
        const __amdgpu_buffer_rsrc_t rx = make_rsrc(x - ((long)W * C + C), (long)N * H * W * C * 4 + xrow + xpix);
        const __amdgpu_buffer_rsrc_t rdy = make_rsrc(dy, (long)N * H * W * K * 4);
        const unsigned yrow = (unsigned)(W * K * 4), ypix = (unsigned)(K * 4);
        const unsigned xch = (unsigned)((c0 + 2 * cp) * 4), ych = (unsigned)((k0 + 2 * cp) * 4);
        const bool dosum = DBIAS && c0 == 0;
        f32x2 dbs = {0.f, 0.f};
        f32x2 d[36], e[16];                             // the patch (raw, then B^T d B in place) and the raw output gradients

        unsigned var[3][3], ybase;                      // the offsets of the window being requested
        auto request_begin = [&]() __attribute__((always_inline)) {
            const bool ok = id < t_end;
            const unsigned pix = __umul24(__umul24(tn, H) + 4 * ty, W) + 4 * tx;       // (every factor is below 2^24)
            const unsigned xb = ok ? __umul24(pix, xpix) + xch : kPad;
            ybase = ok ? __umul24(pix, ypix) + ych : kPad;
            const bool top = ty > 0, bot = ty < TH - 1, lef = tx > 0, rig = tx < TW - 1;
#pragma unroll
            for (int rc = 0; rc < 3; ++rc)
#pragma unroll
                for (int cc = 0; cc < 3; ++cc)
                    var[rc][cc] = ((rc == 0 ? top : rc == 2 ? bot : true) && (cc == 0 ? lef : cc == 2 ? rig : true)) ? xb : kPad;
            tx += s_tx;                                 // (advance: the next request's tile)
            const bool c1 = tx >= TW;
            tx -= c1 ? TW : 0;
            ty += s_ty + (c1 ? 1 : 0);
            const bool c2 = ty >= TH;
            ty -= c2 ? TH : 0;
            tn += s_n + (c2 ? 1 : 0);
            id += 2 * BT;
        };
#define W43W_XREQ(Q)                                                                                        \
    if (W43W_ABLATE & 1) { float o = (float)(lane + (Q)); asm volatile("" : "+v"(o)); d[Q] = f32x2{o, o}; }   \
    else d[Q] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(                            \
             rx, var[(Q) / 6 == 0 ? 0 : (Q) / 6 == 5 ? 2 : 1][(Q) % 6 == 0 ? 0 : (Q) % 6 == 5 ? 2 : 1],    \
             (unsigned)((Q) / 6) * xrow + (unsigned)((Q) % 6) * xpix, 0));
#define W43W_YREQ(Q)                                                                                        \
    if (W43W_ABLATE & 1) { float o = (float)(lane - (Q)); asm volatile("" : "+v"(o)); e[Q] = f32x2{o, o}; }   \
    else e[Q] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rdy, ybase, (unsigned)((Q) / 4) * yrow + (unsigned)((Q) % 4) * ypix, 0));
        auto request = [&]() __attribute__((always_inline)) {       // the pair's next window, then advance
            request_begin();
#pragma unroll
            for (int q = 0; q < 36; ++q) { W43W_XREQ(q) }
#pragma unroll
            for (int q = 0; q < 16; ++q) { W43W_YREQ(q) }
        };
        // off window: V = B^T d B in place - along the columns of each patch row, then along the rows of each frequency column
        auto transform_v = [&]() __attribute__((always_inline)) {
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
                d[0 + jj] = v0; d[6 + jj] = v1; d[12 + jj] = v2; d[18 + jj] = v3; d[24 + jj] = v4; d[30 + jj] = v5;
            }
            // (pin the results HERE: arithmetic is no memory access, and hipcc sank the whole transform across the barrier into
            // the duty window - 5600 cycles there, 70 here, tools/wino43w_trace.py.  Volatile statements keep their order.)
            asm volatile("" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]), "+v"(d[8]),
                         "+v"(d[9]), "+v"(d[10]), "+v"(d[11]), "+v"(d[12]), "+v"(d[13]), "+v"(d[14]), "+v"(d[15]), "+v"(d[16]), "+v"(d[17]));
            asm volatile("" : "+v"(d[18]), "+v"(d[19]), "+v"(d[20]), "+v"(d[21]), "+v"(d[22]), "+v"(d[23]), "+v"(d[24]), "+v"(d[25]), "+v"(d[26]),
                         "+v"(d[27]), "+v"(d[28]), "+v"(d[29]), "+v"(d[30]), "+v"(d[31]), "+v"(d[32]), "+v"(d[33]), "+v"(d[34]), "+v"(d[35]));
            __builtin_amdgcn_sched_barrier(0);
        };
        // duty window: both operands of the window to LDS, M = A dY A^T (+ the bias gradient's share of this tile) on the way,
        // and the NEXT request of every register right behind its store: a vector-memory instruction holds the CU's address
        // path for ~16 cycles, so the 52 of a wave go out between the stores and the transform instead of in a burst after them
        auto dump_request = [&](float *buf, bool again) __attribute__((always_inline)) {
            if (W43W_ABLATE & 2) { if (again) request(); return; }
            __builtin_amdgcn_sched_barrier(0);
            if (again) request_begin();
#pragma unroll
            for (int f = 0; f < 36; ++f) {
                if (!(W43W_ABLATE & 32)) *reinterpret_cast<f32x2 *>(buf + f * FS + vd) = d[f];
                else asm volatile("" ::"v"(d[f]));
                if (again) { W43W_XREQ(f) }
            }
            if (dosum) dbs += ((e[0] + e[1]) + (e[2] + e[3])) + ((e[4] + e[5]) + (e[6] + e[7])) + (((e[8] + e[9]) + (e[10] + e[11])) + ((e[12] + e[13]) + (e[14] + e[15])));
            f32x2 t[4][6];                                  // t[a][j] = sum_b dY[a][b] A[j][b]
#pragma unroll
            for (int a = 0; a < 4; ++a)
                W43W_A4(e[a * 4 + 0], e[a * 4 + 1], e[a * 4 + 2], e[a * 4 + 3], t[a][0], t[a][1], t[a][2], t[a][3], t[a][4], t[a][5])
            float *bm = buf + OP_FLOATS + vd;
#pragma unroll
            for (int jj = 0; jj < 6; ++jj) {                // M[i][j] = sum_a A[i][a] t[a][j]
                f32x2 v0, v1, v2, v3, v4, v5;
                W43W_A4(t[0][jj], t[1][jj], t[2][jj], t[3][jj], v0, v1, v2, v3, v4, v5)
                if (W43W_ABLATE & 32) { asm volatile("" ::"v"(v0), "v"(v1), "v"(v2), "v"(v3), "v"(v4), "v"(v5)); continue; }
                *reinterpret_cast<f32x2 *>(bm + (0 + jj) * FS) = v0; *reinterpret_cast<f32x2 *>(bm + (6 + jj) * FS) = v1;
                *reinterpret_cast<f32x2 *>(bm + (12 + jj) * FS) = v2; *reinterpret_cast<f32x2 *>(bm + (18 + jj) * FS) = v3;
                *reinterpret_cast<f32x2 *>(bm + (24 + jj) * FS) = v4; *reinterpret_cast<f32x2 *>(bm + (30 + jj) * FS) = v5;
            }
            // (the gradients' requests LAST, behind a scheduling fence: issued while the old values were still being read they
            // went into other registers and a copy per value - each a wait for its load - stood in front of the barrier)
            __builtin_amdgcn_sched_barrier(0);
            if (again) {
#pragma unroll
                for (int q = 0; q < 16; ++q) { W43W_YREQ(q) }
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        // (beyond the slice's last tile: out-of-range offsets - the loads return zeros without traffic; such a window is
        // stored into the idle buffer and never multiplied)
        request();                                      // window g
        transform_v();
        if (g == 0) dump_request(smem, true);           // (and window 2 requested)
        lds_barrier();
        // the consumers multiply window i.  The two window kinds alternate; pair 1 starts with a duty window (its window 1), pair 0
        // with an off window - written as ONE loop body of (off, duty) with pair 1's first duty peeled off, so that no register
        // of a window in flight has to change place where two paths meet
        int i = 0;
        if (g == 1) {
            if (wave == 6) { W43W_STAMP(2, i) }
            dump_request(smem + WIN_FLOATS, true);
            lds_barrier();
            ++i;
        }
        while (i < S) {
            if (wave == 4) { W43W_STAMP(2, i) }
            transform_v();
            if (wave == 4) { W43W_STAMP(4, i) }
            lds_barrier();
            if (++i >= S) break;
            if (wave == 4) { W43W_STAMP(3, i) }
            dump_request(smem + ((i + 1) & 1) * WIN_FLOATS, true);
            if (wave == 4) { W43W_STAMP(4, i) }
            lds_barrier();
            ++i;
        }
        if (dosum) {
            float *row = dbsum + (((wave - 4) * 4 + (lane >> 4)) * BK + 2 * cp);
            row[0] = dbs.x;
            row[1] = dbs.y;
        }
    } else {
        // ---------------------------------------------------------------- consumers: 9 frequencies x 32 c x 32 k
        f32x16 acc[9];
#pragma unroll
        for (int fl = 0; fl < 9; ++fl)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[fl][r] = 0.f;
        const int f0 = 9 * wave;
        const int obase = f0 * FS + kh * 128 + l31;         // + fl * FS + k-pair * 32 (+ OP_FLOATS: M)
        lds_barrier();
        if (wave == 0) { W43W_STAMP(7, 1) }
        for (int j = 0; j < S; ++j) {
            if (wave == 0) { W43W_STAMP(0, j) }
            const float *cur = smem + (j & 1) * WIN_FLOATS + obase;
            // frequencies in pairs (the last one alone): consecutive MFMAs go to different accumulators, the operands of the
            // next pair are fetched while this one is multiplied
            float2 va[2][2], vb[2][2];                      // [frequency of the pair][k-pair pair]
            auto opload = [&](int fl, int slot) __attribute__((always_inline)) {
                if (W43W_ABLATE & 8) {
                    va[slot][0] = va[slot][1] = vb[slot][0] = vb[slot][1] = make_float2(1.f, 1.f);
                    return;
                }
                const float *p = cur + fl * FS;             // (conflict-free dwords; pairs of them are one ds_read2_b32)
                va[slot][0] = make_float2(p[0], p[32]);
                va[slot][1] = make_float2(p[64], p[96]);
                vb[slot][0] = make_float2(p[OP_FLOATS], p[OP_FLOATS + 32]);
                vb[slot][1] = make_float2(p[OP_FLOATS + 64], p[OP_FLOATS + 96]);
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
            if (wave == 0) { W43W_STAMP(1, j) }
            lds_barrier();
        }
        if (wave == 0) { W43W_STAMP(6, 0) }
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
    if (wave == 0) { W43W_STAMP(6, 1) }
    lds_barrier();
    if (wave == 0) { W43W_STAMP(6, 2) }
    if (DBIAS && c0 == 0 && tid < BK) {
        float t = 0.f;                                  // (a fixed order: the bias gradient is bit-reproducible too)
#pragma unroll
        for (int q = 0; q < 16; ++q) t += dbsum[q * BK + tid];
        dbpart[(long)bz * K + k0 + tid] = t;
    }
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
    if (wave == 0) { W43W_STAMP(6, 3) }
}
#ifdef W43W_TRACE
extern "C" int mmdgan_w43w_trace(long *host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_w43w_trace), sizeof(long) * (n < 8 * 256 ? n : 8 * 256), 0, hipMemcpyDeviceToHost);
}
#endif

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
    const int cus = tuning().wino43_wgrad_cus > 0 ? tuning().wino43_wgrad_cus : wgrad_cus();
    int split = cus / nblk;                                         // one 8-wave workgroup per CU (146 KB of LDS), one round
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
