// Winograd F(2x2, 2x2) for the 4x4 / stride-2 layers (D l2, l4, l6 and the transposed convolutions of
// G): the other half of the conv FLOPs.  A 4x4 stride-2 convolution is a sum of four 2x2 stride-1
// convolutions on the parity sub-images of its input, and its input-gradient (= the forward pass of a
// 4x4 stride-2 transposed convolution) is four independent 2x2 stride-1 convolutions, one per output
// parity phase.  F(2x2,2x2) computes a 2x2 output tile of such a convolution from a 3x3 patch with 9
// multiplies instead of 16:
//
//   y = A^T [ sum (G g G^T) (.) (B^T d B) ] A,   B^T = [1 -1 0; 0 1 0; 0 -1 1], G = [1 0; 1 1; 0 1], A^T = [1 1 0; 0 1 1]
//
// (all constants 0 / +-1).  One kernel serves both forms through a table of "segments": a segment is
// one (sub-image, 2x2 filter) pair, i.e. a patch origin, the steps between patch rows / tiles and a
// slice of the transformed weights U[segment][9][Cr][Ko]:
//   forward      1 phase  x 4 segments (input parities a,b): patch rows 2*(2ty) - 1 + a + 2u of x
//   input-grad   4 phases x 1 segment  (output parities al,be): patch rows 2ty + al - 1 + u of dy
// Structure as conv_wino.hip: 32 tiles x 64 output channels per workgroup, V through LDS, B fragments
// straight from L2 into registers, output transform through LDS in the epilogue.  The 9 frequencies x 2
// column blocks = 18 accumulators are dealt round-robin to the 4 waves (5,5,4,4).
#include <type_traits>

#include "conv_internal.h"
#include "bufload.h"
#include "wino_weight.h"


// Ablation hooks for tools/wino2_ablate.sh (what each part of wino2_kernel costs): a bit mask, 0 in the library build - every
// test below is a compile-time constant, the shipped code object is the one without them.  1: patch loads of the main loop,
// 2: B-fragment loads, 4: A-fragment reads, 8: input transform + V stores, 16: stage barrier, 32: epilogue operand loads,
// 64: epilogue stores, 128: the whole epilogue (the accumulators are summed into a store that never happens).
#ifndef W2_ABLATE
#define W2_ABLATE 0
#endif

namespace mmdgan {

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace wino2 {
constexpr int BC = 32;                  // reduction channels per stage: one 128-byte line of every patch pixel = 4 groups of 8
constexpr int ROW = 32 * 4 + 4;         // V: floats per (channel group, k half) row = [32 tiles][4 k-pairs] + 4 (write banks, below)
constexpr int FSV = 8 * ROW;            // V: floats per frequency = 4 channel groups x 2 k halves
constexpr int V_FLOATS = 9 * FSV;
constexpr int MS_FLOATS = 9 * 32 * 32;       // epilogue exchange buffer (one column block at a time), Ms[f][tile][k 32]
static_assert(MS_FLOATS <= V_FLOATS, "the epilogue buffer is the V buffer the last stage released");
constexpr int SMEM_FLOATS = 2 * V_FLOATS;
constexpr int OB_SLOTS = 4;             // ring of items whose output offsets are kept (two are live at once, see the kernel)
constexpr size_t LDS_BYTES = sizeof(float) * SMEM_FLOATS + sizeof(unsigned) * 32 * OB_SLOTS + sizeof(int) * 4;
constexpr int NM = 5;                   // frequencies per wave (the waves with the odd frequencies use 4)

struct Params {
    int N, TH, TW;          // tiles per image (rows, cols)
    int IH, IW, Cr;         // input image, reduction channels per segment
    int nseg;               // segments per phase (1 or 4)
    int r0[4], c0[4];       // patch origin (input row / col of patch element (0,0) for tile (0,0)) per global segment
    int tstep, pstep;       // input rows (cols) between consecutive tiles / consecutive patch rows (cols)
    int OH, OW, Ko;         // output image, channels
    int otile, ostep;       // output row = ty * otile + a * ostep + o0r[phase]
    int o0r[4], o0c[4];
    int ntb, nkb, nph;      // work items = tile blocks x column blocks x phases x reduction parts
    int ksplit, spp;        // the reduction of an output tile cut into ksplit parts of spp stages each: part k writes its partial
    unsigned slab_bytes;    // result (plain sums, no epilogue) into slab k of `out` (= the workspace then), slab_bytes apart
    int contiguous;         // A/B: a workgroup's items as one contiguous run instead of round-robin over the XCD's workgroups
};
}  // namespace wino2

template <bool DGRAD>
__global__ __launch_bounds__(256) void wino2_weight_kernel(const float *__restrict__ w, float *__restrict__ U, int C, int K) {
    __shared__ float tile[9][32][33];                    // (layout of U: wino_weight.h)
    wino2_weight_block<DGRAD>(tile, blockIdx.x, blockIdx.y, blockIdx.z, w, U, C, K);
}

// PERSISTENT and CONTINUOUS: the grid is (at most) two workgroups per CU, every workgroup walks a contiguous run of work
// items (tile block, phase, column block), and the stages of ALL its items form one software pipeline: while stage L is
// multiplied, stage L+1 goes global -> registers (under the first MFMA group) -> LDS (under the last two), across item
// boundaries (until round 4 stage L+2 was in registers as well: a stage deeper, four registers more, 0-1.3 % slower).  An item of the
// launch this was built for (the 3B-row input-gradient of D l2: 1536 items) is only four stages long; as one workgroup per
// item (round 2) each paid a serial prologue (patch load -> transform -> barrier), and its epilogue's stores sat in front
// of the next workgroup's first loads.  Now the only thing between two items is the epilogue (accumulators -> LDS -> output
// transform -> stores) - into the ONE V buffer that is free then, a column block at a time - and the other workgroup of
// the CU, which is somewhere else in its own item, has the MFMA pipes to itself meanwhile.  Measured on that launch
// (tools/wino2_probe.hip history, in-kernel stamps): 97 us as one workgroup per item -> 80 us persistent with the vector
// fragments below (95 with the epilogue's activation-derivative operands, which the first probe forgot) -> 79 us continuous,
// operands included.  Per item now: main loop 17 us against 15.5 us of MFMA issue per pair of workgroups at the 2.37 GHz the
// part holds in this kernel, epilogue 6 us - 4 of them the output stores and operand loads of ALL workgroups of the chip
// arriving together (they run in lock step); fetching the operands earlier (last stage, or touching their lines in the
// first) and starting the second workgroup of every CU late did not shorten the item (23.3 us in every variant).
//
// A stage is 32 reduction channels of one segment = 4 groups of 4 MFMA k-pairs, 80 (64) MFMAs per wave between barriers:
//   producer  thread = (tile, channel quad): the 9 pixels of its 3x3 patch as float4 (8 neighbouring lanes read one full
//             128-byte line of each pixel), the whole B^T d B in registers, 18 ds_write_b64 into V[f][group][k half][tile][4]
//             (row stride 132 floats: the four groups of a wave's store land on disjoint banks);
//   consumer  B fragments of a group = ONE 16-byte load per frequency straight from L2 (layout above), a group ahead; A
//             fragments = one ds_read_b64 per frequency and k-pair PAIR, two k-pairs ahead: a quarter / half of the vector-
//             memory / LDS instructions of the one-dword-per-MFMA form this kernel had before.
// Workgroup -> wave roles: 18 accumulators over 4 waves = 5,5,4,4.  The roles are rotated by where the workgroup's first
// wave sits (hardware slot and SIMD, read once) so that, with the placement observed on gfx950 (tools/wg_census.hip: the
// two workgroups of a CU on slots 0 / 1, waves on SIMDs in cyclic order), every SIMD carries a 5-wave of one workgroup and
// a 4-wave of the other.  A speed assumption only: any placement computes the same result.
__global__ __launch_bounds__(256, 2) void wino2_kernel(wino2::Params P, ConvEpilogue ep, const float *__restrict__ x,
                                                       const float *__restrict__ U, float *__restrict__ out) {
    using namespace wino2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // [item % OB_SLOTS][tile]: output byte offset of the tile.  The load cursor is one stage ahead of the multiply: it enters
    // item i+1 during item i's last stage, before item i's epilogue has read its offsets - two slots are in use at a time.
    // The ring stays at FOUR: with the two-stage-deep pipeline of rounds 3-4 the cursor of a two-stage item (64 reduction
    // channels, e.g. the 64 -> 64 first block of the ResNet-SN discriminator) was a whole item further, and a ring of two
    // stored item i's tiles at item i+2's addresses (found by the parity test at the ResNet config's own batch,
    // tests/test_production_gpu.py); the spare slots cost 256 bytes.
    unsigned *obase = reinterpret_cast<unsigned *>(smem + SMEM_FLOATS);
    int *wgctl = reinterpret_cast<int *>(obase + 32 * OB_SLOTS);
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, kh = lane >> 5;
    // wave-uniform by construction, but hipcc only knows once it sits in an SGPR: without this every B load with a
    // wave-dependent scalar offset became a waterfall loop and every `m < nm` an exec-mask branch (2x slower)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (tid == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));       // HW_ID: wave slot [3:0], SIMD [5:4]
        const int simd = (hw >> 4) & 3;
        wgctl[0] = hw & 1;                                                               // second workgroup of the CU: odd slot
        wgctl[1] = (simd & 1) * 2 + (simd >> 1);                                         // position in the cycle 0,2,1,3
    }
    __syncthreads();
    const int second = __builtin_amdgcn_readfirstlane(wgctl[0]), pos0 = __builtin_amdgcn_readfirstlane(wgctl[1]);
    const int wrole = (wave + pos0 + 2 * second) & 3;
    // Static priority for the SECOND workgroup of the CU: its wave is the younger one on every SIMD and loses the VALU / MFMA
    // arbitration against the first workgroup's on every segment (MI355X_MICROARCH.md, "two waves per SIMD", item 4).  One
    // s_setprio for the whole kernel, no per-segment flips.  Round 4, the 3B-row D l2 launch, 100 launches each, twice: none
    // 80.87 / 80.80 us, second workgroup 80.57 / 80.61, first workgroup 80.87 / 81.02, the five-accumulator waves 83.18 / 83.06;
    // CIFAR step 1.899 / 1.902 -> 1.883 / 1.899 ms.
    if (second) __builtin_amdgcn_s_setprio(1);
#ifdef W2_SECOND_DELAY   // measurement builds only (tools/wino2_ablate.sh delay): the CU's second workgroup starts this many 10 ns ticks late
    if (second) {
        const long t0_ = (long)wall_clock64();
        while ((long)wall_clock64() - t0_ < W2_SECOND_DELAY) __builtin_amdgcn_s_sleep(16);
    }
#endif
    const int cb = wrole & 1, fq = wrole >> 1;           // this wave: column block cb, frequencies fq + 2m
    const int nm = fq == 0 ? 5 : 4;
    // ---- this workgroup's run of items: XCD x (speed assumption: workgroup b runs on XCD b % 8) gets a contiguous eighth of
    // the items, ordered (tile block, phase, column block) so that the re-reads of a patch by the other phases / column
    // blocks hit that XCD's L2 ...
    const int nitems = P.ntb * P.nkb * P.nph * P.ksplit, nwg = gridDim.x;
    // ... and deals them ROUND-ROBIN to its workgroups (workgroup j takes items j, j + wx, j + 2 wx, ...): at any time the XCD's
    // workgroups are on neighbouring items, i.e. the four phases / the column blocks of one tile block run side by side and
    // share its patches through the L2.  (Contiguous runs per workgroup put the phases of a tile block one after the other in
    // ONE workgroup, 20 us apart, and the 4 MB L2 had turned over by then: 147 MB fetched per launch of the 3B-row D l2
    // input-gradient against 77 MB algorithmic.)
    int first, count, istep;
    {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        const int nx = min(8, nwg);                                   // XCDs in use
        const int wx = nwg / nx + (xcd < nwg % nx ? 1 : 0);           // workgroups on this XCD
        const int iq = nitems / nx, ir = nitems % nx;
        const int xfirst = xcd * iq + min(xcd, ir), xcount = iq + (xcd < ir ? 1 : 0);
        if (P.contiguous) {
            const int q = xcount / wx, r = xcount % wx;
            first = xfirst + j * q + min(j, r);
            count = q + (j < r ? 1 : 0);
            istep = 1;
        } else {
            first = xfirst + j;
            count = j < xcount ? (xcount - j + wx - 1) / wx : 0;
            istep = wx;
        }
    }
    if (count <= 0) return;
    const unsigned T = (unsigned)P.N * P.TH * P.TW;
    const int spc = P.Cr / BC;                           // stages per segment
    const int nstages = P.spp;                           // stages per ITEM; >= 2 (the launcher checks): the pipeline reaches
    const int per_tb = P.nkb * P.nph * P.ksplit;         // one item ahead at most
    const int pt = tid >> 3, cq = tid & 7;               // producer: thread = (tile pt, channel quad cq)
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(x, (long)P.N * P.IH * P.IW * P.Cr * 4);
    const __amdgpu_buffer_rsrc_t ru = make_rsrc(U, (long)4 * 9 * P.Cr * P.Ko * 4);
    // V[f][group g][k half][tile][4 k-pairs]: channel 4 cq + e of the stage -> group cq >> 1, k half e & 1, k-pair 2 (cq & 1) + (e >> 1)
    const int vdst = (cq >> 1) * 2 * ROW + pt * 4 + (cq & 1) * 2;
    const int abase = fq * FSV + kh * ROW + l31 * 4;
    const unsigned ufreq = (unsigned)((long)P.Cr * P.Ko * 4), ugrp = (unsigned)(2 * P.Ko * 16), ustage = 4u * ugrp;
    const unsigned useg = 9u * ufreq;
    const unsigned ubase = (unsigned)(((long)kh * P.Ko + cb * 32 + l31) * 16) + (unsigned)fq * ufreq;   // + per-item / per-stage scalar part
    const float sc = ep.scale ? ep.scale[0] : 1.f;
    const int kq = tid & 7, tb = tid >> 3;               // epilogue: thread = (output item tb, channel quad kq)
    // output and activation-derivative operands go through buffer resources with 32-bit byte offsets (every tensor is
    // < 2 GiB): a tile beyond the ragged end carries offset kOOB - its loads return 0, its stores are dropped, no branch
    const unsigned arow = (unsigned)(P.ostep * P.OW * P.Ko * 4), bcol = (unsigned)(P.ostep * P.Ko * 4);
    const __amdgpu_buffer_rsrc_t ro = make_rsrc(out, P.ksplit > 1 ? (long)P.ksplit * P.slab_bytes : (long)P.N * P.OH * P.OW * P.Ko * 4);
    const bool wraps = ep.wrap_from < 0x20000000L;       // the operand tensor holds fewer images than the output (ConvEpilogue)
    const unsigned wrap_from = wraps ? (unsigned)(ep.wrap_from * 4) : 0xffffffffu, wrap_sub = (unsigned)(ep.wrap_sub * 4);
    const __amdgpu_buffer_rsrc_t rd = make_rsrc(ep.dact ? ep.dact : x, wraps ? ep.wrap_from * 4 : (long)P.N * P.OH * P.OW * P.Ko * 4);
    auto dact_off = [&](unsigned o) { return (o >= wrap_from ? o - wrap_sub : o) | (o & kOOB); };   // (a missing tile stays out of range)

    // ---- the LOAD cursor (stage L+1: global -> registers): item, segment, channel block; per thread the tile it loads
    int l_it = 0, l_seg = 0, l_cs = 0, l_phase = 0;
    int ty = 0, tx = 0, tn = 0;
    bool tile_ok = false;
    unsigned xbase = 0, xmask = 0;                       // byte offset of patch pixel (0,0), channel 4 cq; bit 3u+v: pixel (u,v) exists
    const unsigned xrow = (unsigned)(P.pstep * P.IW * P.Cr * 4), xcol = (unsigned)(P.pstep * P.Cr * 4);
    auto enter_item = [&]() {                            // the load cursor has reached item l_it (< count)
        const int item = first + l_it * istep;
        const int tblk = item / per_tb, rem0 = item - tblk * per_tb;
        const int rem = rem0 / P.ksplit, chunk = rem0 - rem * P.ksplit;
        const int s0 = chunk * P.spp;                    // the item's first stage of the tile's reduction
        l_seg = s0 / spc;
        l_cs = s0 - l_seg * spc;
        l_phase = rem / P.nkb;
        const unsigned id = (unsigned)tblk * 32u + pt;
        tile_ok = id < T;
        const unsigned ii = tile_ok ? id : 0u;
        const unsigned q1 = ii / (unsigned)P.TW;
        tx = (int)(ii - q1 * P.TW);
        tn = (int)(q1 / (unsigned)P.TH);
        ty = (int)(q1 - (unsigned)tn * P.TH);
        if (cq == 0)             // byte offset of this tile's output pixel (a=0, b=0), channel 0 (kOOB: no such tile)
            obase[(l_it & (OB_SLOTS - 1)) * 32 + pt] =
                tile_ok ? (unsigned)((((long)tn * P.OH + ty * P.otile + P.o0r[l_phase]) * P.OW + tx * P.otile + P.o0c[l_phase]) * P.Ko * 4) +
                              (unsigned)chunk * P.slab_bytes : kOOB;
    };
    auto set_segment = [&]() {                           // the 9 patch pixels of segment l_seg: (possibly wrapped) offset of pixel
        const int gs = l_phase * P.nseg + l_seg;         // (0,0) + which of them lie inside the image (tensors are < 2 GiB)
        const int row0 = ty * P.tstep + P.r0[gs], col0 = tx * P.tstep + P.c0[gs];
        xbase = (unsigned)(((((long)tn * P.IH + row0) * P.IW + col0) * P.Cr + 4 * cq) * 4);
        xmask = 0;
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int row = row0 + u * P.pstep;
            const bool rowok = tile_ok && row >= 0 && row < P.IH;
#pragma unroll
            for (int v = 0; v < 3; ++v) {
                const int col = col0 + v * P.pstep;
                xmask |= (rowok && col >= 0 && col < P.IW) ? 1u << (3 * u + v) : 0u;
            }
        }
    };
    int l_left = P.spp;                                  // stages the load cursor's item still has
    auto advance_load = [&]() {
        if (--l_left == 0) {
            l_left = P.spp;
            if (++l_it < count) enter_item();
            else tile_ok = false;                        // past the last item: every patch pixel "outside", the loads fetch nothing
        } else if (++l_cs == spc) {
            l_cs = 0;
            ++l_seg;
        }
    };
    float4 rin[3][3];
    // global -> registers: patch row U_ of the load cursor's stage (padded taps: an offset beyond the buffer reads zeros)
#define W2_XLOAD_ROW(U_)                                                                                 \
    {                                                                                                    \
        if ((U_) == 0 && (l_cs == 0 || l_left == P.spp)) set_segment();   /* a new segment, or a new item's first stage */ \
        const unsigned sx = xbase + (unsigned)(U_) * xrow + (unsigned)(l_cs * BC * 4);                   \
        _Pragma("unroll") for (int v = 0; v < 3; ++v)                                                    \
            rin[U_][v] = bufld4(rx, (xmask >> (3 * (U_) + v)) & 1u ? sx + (unsigned)v * xcol : kOOB);    \
    }
    // registers -> LDS: B^T d B for the channel pair (e, e + 2) of the quad: k-pairs 2 (cq & 1), 2 (cq & 1) + 1 of k half e
#define W2_VSTORE_PAIR(BUF, E_, CA, CB)                                                                  \
    {                                                                                                    \
        float XA[3][3], XB[3][3];                                                                        \
        _Pragma("unroll") for (int u = 0; u < 3; ++u) {                                                  \
            XA[u][0] = rin[u][0].CA - rin[u][1].CA; XA[u][1] = rin[u][1].CA; XA[u][2] = rin[u][2].CA - rin[u][1].CA; \
            XB[u][0] = rin[u][0].CB - rin[u][1].CB; XB[u][1] = rin[u][1].CB; XB[u][2] = rin[u][2].CB - rin[u][1].CB; \
        }                                                                                                \
        float *d_ = (BUF) + vdst + (E_) * ROW;                                                           \
        _Pragma("unroll") for (int j = 0; j < 3; ++j) {                                                  \
            *reinterpret_cast<float2 *>(d_ + (0 * 3 + j) * FSV) = make_float2(XA[0][j] - XA[1][j], XB[0][j] - XB[1][j]); \
            *reinterpret_cast<float2 *>(d_ + (1 * 3 + j) * FSV) = make_float2(XA[1][j], XB[1][j]);         \
            *reinterpret_cast<float2 *>(d_ + (2 * 3 + j) * FSV) = make_float2(XA[2][j] - XA[1][j], XB[2][j] - XB[1][j]); \
        }                                                                                                \
    }
    // ---- the B cursor (stage L+1): scalar byte offset of its (item, segment, channel block) slice of U
    int b_it = 0, b_seg = 0, b_cs = 0, b_left = P.spp;
    unsigned b_off;
    auto b_item = [&]() {
        const int item = first + b_it * istep;
        const int tblk = item / per_tb, rem0 = item - tblk * per_tb;
        const int rem = rem0 / P.ksplit, chunk = rem0 - rem * P.ksplit;
        const int phase = rem / P.nkb, n0 = (rem - phase * P.nkb) * 64;
        const int s0 = chunk * P.spp;
        b_seg = s0 / spc;
        b_cs = s0 - b_seg * spc;
        b_off = (unsigned)(phase * P.nseg) * useg + (unsigned)(n0 * 16);
    };
    auto advance_b = [&]() {                             // (only called while a next stage exists)
        if (--b_left == 0) {
            b_left = P.spp;
            ++b_it;
            b_item();
        } else if (++b_cs == spc) {
            b_cs = 0;
            ++b_seg;
        }
    };
    auto stage_boff = [&]() { return b_off + (unsigned)b_seg * useg + (unsigned)b_cs * ustage; };
    float4 fb[2][NM];
    float2 fa[2][NM];
    // B fragments of group G (4 k-pairs) of the stage whose scalar offset is OFF, frequency fq + 2*M, into slot SL
#define W2_BLOAD(SL, G, M, OFF) fb[SL][M] = bufld4s(ru, ubase, (OFF) + (unsigned)(2 * (M)) * ufreq + (unsigned)(G) * ugrp);
    // A fragments of k-pairs 2H, 2H+1 of group G from buffer BUF into slot SL
#define W2_ALOAD(SL, BUF, G, H, M) fa[SL][M] = *reinterpret_cast<const float2 *>((BUF) + abase + 2 * (M) * FSV + (G) * 2 * ROW + 2 * (H));

    // ---- fill the pipeline: stage 0 -> LDS, B fragments of stage 0 / group 0
    enter_item();
    b_item();
    unsigned boff_cur = stage_boff();
    W2_XLOAD_ROW(0) W2_XLOAD_ROW(1) W2_XLOAD_ROW(2)
    advance_load();
#pragma unroll
    for (int m = 0; m < NM; ++m)
        if (m < nm) W2_BLOAD(0, 0, m, boff_cur)
    W2_VSTORE_PAIR(smem, 0, x, z) W2_VSTORE_PAIR(smem, 1, y, w)
    __syncthreads();
    int par = 0;

    for (int it = 0; it < count; ++it) {
        const int item = first + it * istep;
        const int c_tblk = item / per_tb, c_rem = (item - c_tblk * per_tb) / P.ksplit;
        const int c_phase = c_rem / P.nkb;
        const int n0 = (c_rem - c_phase * P.nkb) * 64;
        const unsigned *ob_it = obase + (it & (OB_SLOTS - 1)) * 32;
        const bool last_item = it + 1 == count;
        f32x16 acc[NM];
#pragma unroll
        for (int m = 0; m < NM; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
        float4 dv[2][2][2];                              // activation-derivative operands of this thread's 8 output float4s

        for (int s = 0; s < nstages; ++s) {
            const float *cur = smem + par * V_FLOATS;
            float *nxt = smem + (par ^ 1) * V_FLOATS;
            // -> stage L+1 (its group 0 is loaded under group 3 below).  After the very last stage there is none: the loads,
            // the transform and the stores below then run once more on stale operands into the idle buffer (no branches here)
            unsigned boff_nxt = boff_cur;
            if (!(last_item && s + 1 == nstages)) { advance_b(); boff_nxt = stage_boff(); }
#pragma unroll
            for (int m = 0; m < NM; ++m)
                if (m < nm) W2_ALOAD(0, cur, 0, 0, m)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    // operands that fly under this half-group's MFMAs: the A fragments of the next two k-pairs, and (once per
                    // group) the B fragments of the next group
#pragma unroll
                    for (int m = 0; m < NM; ++m)
                        if (m < nm) {
                            if (W2_ABLATE & 4) {}
                            else if (h == 0) W2_ALOAD(1, cur, g, 1, m)
                            else if (g + 1 < 4) W2_ALOAD(0, cur, g + 1, 0, m)
                        }
                    if (h == 0) {
#pragma unroll
                        for (int m = 0; m < NM; ++m)
                            if (m < nm) {
                                if (W2_ABLATE & 2) {}
                                else if (g + 1 < 4) W2_BLOAD((g + 1) & 1, g + 1, m, boff_cur)
                                else W2_BLOAD(0, 0, m, boff_nxt)
                            }
                    }
#pragma unroll
                    for (int qq = 0; qq < 2; ++qq) {
                        const int q = 2 * h + qq;
#pragma unroll
                        for (int m = 0; m < NM; ++m)
                            if (m < nm) {
                                const float a = qq == 0 ? fa[h][m].x : fa[h][m].y;
                                const float b = q == 0 ? fb[g & 1][m].x : q == 1 ? fb[g & 1][m].y : q == 2 ? fb[g & 1][m].z : fb[g & 1][m].w;
                                acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m], 0, 0, 0);
                            }
                        // stage L+1: global -> registers under group 0 (one patch row per step), registers -> LDS under groups 2 and 3
                        // (one channel pair each): nothing of it lives in registers across the stage barrier or the epilogue
                        if (g == 0 && q == 0) { if (!(W2_ABLATE & 1)) W2_XLOAD_ROW(0) }
                        else if (g == 0 && q == 1) { if (!(W2_ABLATE & 1)) W2_XLOAD_ROW(1) }
                        else if (g == 0 && q == 2) { if (!(W2_ABLATE & 1)) W2_XLOAD_ROW(2) advance_load(); }
                        else if (g == 2 && q == 1) { if (!(W2_ABLATE & 8)) W2_VSTORE_PAIR(nxt, 0, x, z) }
                        else if (g == 3 && q == 1) { if (!(W2_ABLATE & 8)) W2_VSTORE_PAIR(nxt, 1, y, w) }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            if (!(W2_ABLATE & 16)) __syncthreads();
            par ^= 1;
            boff_cur = boff_nxt;
        }
        // ---- output transform + epilogue in the V buffer the last stage has just released (the other one holds the next
        // item's first stage): Ms[f][tile][k 32], one column block at a time
        float *Ms = smem + (par ^ 1) * V_FLOATS;
        if (ep.dact && !(W2_ABLATE & 32)) {
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int i2 = 0; i2 < 2; ++i2) {
                    const int e = tb + 32 * i2;
                    const unsigned o = ob_it[e >> 1] + (e & 1) * bcol + (unsigned)((n0 + c * 32 + kq * 4) * 4);
                    dv[c][i2][0] = bufld4(rd, dact_off(o));
                    dv[c][i2][1] = bufld4(rd, dact_off(o + arow));
                }
        }
        if (W2_ABLATE & 128) {
            float sacc = 0.f;
#pragma unroll
            for (int m = 0; m < NM; ++m)
                if (m < nm) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) sacc += acc[m][r];
                }
            if (sacc == 1.2345e33f) bufst4(ro, ob_it[0], make_float4(sacc, 0.f, 0.f, 0.f));
        } else
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            if (cb == c) {
#pragma unroll
                for (int m = 0; m < NM; ++m)
                    if (m < nm) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) Ms[((fq + 2 * m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh) * 32 + l31] = acc[m][r];
                    }
            }
            __syncthreads();
            const int ch = n0 + c * 32 + kq * 4;
            float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ep.bias) bv = *reinterpret_cast<const float4 *>(ep.bias + ch);
#pragma unroll
            for (int i2 = 0; i2 < 2; ++i2) {
                const int e = tb + 32 * i2, tile = e >> 1, b = e & 1;
                const unsigned ob = ob_it[tile];
                {
                    float4 z[3];                           // Z[i][b] = M[i][b] + M[i][b+1]
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        const float4 p0 = *reinterpret_cast<const float4 *>(Ms + ((3 * i + b) * 32 + tile) * 32 + kq * 4);
                        const float4 p1 = *reinterpret_cast<const float4 *>(Ms + ((3 * i + b + 1) * 32 + tile) * 32 + kq * 4);
                        z[i] = make_float4(p0.x + p1.x, p0.y + p1.y, p0.z + p1.z, p0.w + p1.w);
                    }
#pragma unroll
                    for (int a = 0; a < 2; ++a) {          // Y[a][b] = Z[a][b] + Z[a+1][b]
                        float4 v = make_float4(z[a].x + z[a + 1].x, z[a].y + z[a + 1].y, z[a].z + z[a + 1].z, z[a].w + z[a + 1].w);
                        v.x = v.x * sc + bv.x; v.y = v.y * sc + bv.y; v.z = v.z * sc + bv.z; v.w = v.w * sc + bv.w;
                        if (ep.dact) {
                            const float4 yv = (W2_ABLATE & 32) ? make_float4(1.f, -1.f, 1.f, -1.f) : dv[c][i2][a];
                            v.x *= act_bwd_from_out(yv.x, ep.act); v.y *= act_bwd_from_out(yv.y, ep.act);
                            v.z *= act_bwd_from_out(yv.z, ep.act); v.w *= act_bwd_from_out(yv.w, ep.act);
                        } else {
                            v.x = act_fwd(v.x, ep.act); v.y = act_fwd(v.y, ep.act);
                            v.z = act_fwd(v.z, ep.act); v.w = act_fwd(v.w, ep.act);
                        }
                        if (!(W2_ABLATE & 64) || v.x == 1.2345e33f) bufst4(ro, ob + a * arow + b * bcol + (unsigned)(ch * 4), v);
                    }
                }
            }
            __syncthreads();                             // Ms is rewritten by the next column block / the next item's stage 1
        }
    }
#undef W2_XLOAD_ROW
#undef W2_VSTORE_PAIR
#undef W2_BLOAD
#undef W2_ALOAD
}

// ------------------------------------------------------------------------------------------------
// Measured against the direct kernels (CIFAR batch 64, tools/bench_conv.py with MMDGAN_WINO2=2, weight transform
// included): D l2 forward 70 vs 94 us (512 workgroups), 3B-row dgrad 115 vs 146 (1536); D l4 73 vs 81 (256) and
// 100 vs 120 (768); D l6 3B dgrad 109 vs 113 (384); it loses below that (D l6 forward, 128 workgroups: 128 vs 84;
// G's top layers at batch 64).  Default (MMDGAN_WINO2=1): forward with >= 256 workgroups, input-gradient with
// >= 384; =0 never; =2 every eligible shape (what the parity tests run).
static int wino2_mode() { return tuning().wino2; }

// d describes the CONV (4x4, stride 2, pad 1): x [N,H,W,C] -> y [N,P,Q,K]
static int wino2_ksplit(long base_items, int nstages);
static bool wino2_split_enabled() { return tuning().wino2_ksplit != 0; }       // MMDGAN_WINO2_KSPLIT=0: no reduction split
static bool wino2_shape_ok(const ConvDims &d, bool dgrad, bool split_ok = false) {
    const int mode = wino2_mode();
    if (mode == 0 || d.R != 4 || d.stride != 2 || d.pad != 1 || d.H % 4 || d.W % 4) return false;
    const int cr = dgrad ? d.K : d.C, ko = dgrad ? d.C : d.K;
    if (cr % wino2::BC || cr < 32 || ko % 64) return false;
    if (dgrad && cr < 2 * wino2::BC) return false;                  // the kernel's pipeline needs >= 2 stages per item (forward: 4 segments)
    if (mode >= 2) return true;
    const long tiles = (long)d.N * (d.P / 2) * (d.Q / 2);          // per phase
    const long wgs = ((tiles + 31) / 32) * (ko / 64) * (dgrad ? 4 : 1);
    // one round of two workgroups per CU at least (measured against the direct kernels, header above); with weights the
    // caller has transformed, the reduction split multiplies the grid
    return wgs * (split_ok && wino2_split_enabled() && d.N > 1 ? wino2_ksplit(wgs, (dgrad ? 1 : 4) * (cr / wino2::BC)) : 1) >= 256;
}
bool wino2_eligible(const ConvDims &d, bool dgrad) { return wino2_shape_ok(d, dgrad, true); }
static size_t wino2_bytes(const ConvDims &d) { return sizeof(float) * 36 * (size_t)d.C * d.K; }
bool wino2_fwd_ok(const ConvDims &d) { return d.N > 1 && wino2_shape_ok(d, false) && workspace(wino2_bytes(d)) != nullptr; }
bool wino2_dgrad_ok(const ConvDims &d) { return d.N > 1 && wino2_shape_ok(d, true) && workspace(wino2_bytes(d)) != nullptr; }

int wino2_transform(const ConvDims &d, const float *w, bool dgrad, float *U, hipStream_t st) {
    if (d.C % 32 || d.K % 32) { set_error("wino_transform (4x4 stride 2): C and K must be multiples of 32 (got %d, %d)", d.C, d.K); return MMDGAN_E_ARG; }
    const dim3 wg(d.K / 32, d.C / 32, 4);
    if (dgrad) hipLaunchKernelGGL(wino2_weight_kernel<true>, wg, dim3(256), 0, st, w, U, d.C, d.K);
    else hipLaunchKernelGGL(wino2_weight_kernel<false>, wg, dim3(256), 0, st, w, U, d.C, d.K);
    return check_launch("wino2_transform");
}

// the second half of a reduction-split launch: out = epilogue(sum of the ksplit partial results), float4 per thread
__global__ __launch_bounds__(256) void slab_epilogue_kernel(const float4 *__restrict__ part, int ksplit, long n4, int Ko,
                                                                  ConvEpilogue ep, float4 *__restrict__ out) {
    const float sc = ep.scale ? ep.scale[0] : 1.f;
    const long stride = (long)gridDim.x * 256;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n4; e += stride) {
        float4 v = part[e];
        for (int k = 1; k < ksplit; ++k) {
            const float4 b = part[(long)k * n4 + e];
            v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
        }
        const int ch = (int)((e * 4) % Ko);
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ep.bias) bv = *reinterpret_cast<const float4 *>(ep.bias + ch);
        v.x = v.x * sc + bv.x; v.y = v.y * sc + bv.y; v.z = v.z * sc + bv.z; v.w = v.w * sc + bv.w;
        if (ep.dact) {
            const float4 yv = *reinterpret_cast<const float4 *>(ep.dact + ep.dact_index(e * 4));
            v.x *= act_bwd_from_out(yv.x, ep.act); v.y *= act_bwd_from_out(yv.y, ep.act);
            v.z *= act_bwd_from_out(yv.z, ep.act); v.w *= act_bwd_from_out(yv.w, ep.act);
        } else {
            v.x = act_fwd(v.x, ep.act); v.y = act_fwd(v.y, ep.act);
            v.z = act_fwd(v.z, ep.act); v.w = act_fwd(v.w, ep.act);
        }
        out[e] = ep.add4(v, e * 4);
    }
}

// The same pass with batch-norm statistics on the way (mmdgan_conv2d_*_stats): out = epilogue(sum of the slabs) AND
// totals += [sum out, sum out^2] per channel - the tensor a batch norm normalises next is summed while it is written instead
// of being read again by a statistics launch.  The workgroup shape is bn.hip's (64 channels x a run of rows, sixteen row
// lanes, fp64 sums, one atomic per channel and workgroup into slot blockIdx.y % slots); U rows of a thread in flight.
template <int U>
__global__ __launch_bounds__(256) void slab_epilogue_bn_kernel(const float *__restrict__ part, int ksplit, long total, long rows, int C,
                                                              long rows_per_split, ConvEpilogue ep, float *__restrict__ out,
                                                              double *__restrict__ totals, int slots) {
    __shared__ double red[2][16][65];
    const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int c = blockIdx.x * 64 + cl * 4;
    const long r0 = (long)blockIdx.y * rows_per_split;
    long r1 = r0 + rows_per_split;
    if (r1 > rows) r1 = rows;
    const float sc = ep.scale ? ep.scale[0] : 1.f;
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ep.bias) bv = *reinterpret_cast<const float4 *>(ep.bias + c);
    double sa[4] = {0, 0, 0, 0}, sb[4] = {0, 0, 0, 0};
    for (long r = r0 + rl; r < r1; r += 16 * U) {
        float4 v[U], p[U][7];                              // every slab of the thread's rows in flight at once (ksplit <= 8)
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long rr = r + u * 16;
            const bool ok = rr < r1;
            v[u] = ok ? *reinterpret_cast<const float4 *>(part + rr * C + c) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int k = 1; k < 8; ++k)
                if (k < ksplit) p[u][k - 1] = ok ? *reinterpret_cast<const float4 *>(part + (long)k * total + rr * C + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int k = 1; k < 8; ++k)
                if (k < ksplit) { v[u].x += p[u][k - 1].x; v[u].y += p[u][k - 1].y; v[u].z += p[u][k - 1].z; v[u].w += p[u][k - 1].w; }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long rr = r + u * 16;
            if (rr >= r1) continue;
            float4 o;
            o.x = act_fwd(v[u].x * sc + bv.x, ep.act); o.y = act_fwd(v[u].y * sc + bv.y, ep.act);
            o.z = act_fwd(v[u].z * sc + bv.z, ep.act); o.w = act_fwd(v[u].w * sc + bv.w, ep.act);
            *reinterpret_cast<float4 *>(out + rr * C + c) = o;
            const float ov[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) { const double t = (double)ov[j]; sa[j] += t; sb[j] += t * t; }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { red[0][rl][cl * 4 + j] = sa[j]; red[1][rl][cl * 4 + j] = sb[j]; }
    __syncthreads();
    if (threadIdx.x < 128) {
        const int which = threadIdx.x >> 6, ch = threadIdx.x & 63;
        double t = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[which][k][ch];
        atomicAdd(totals + (size_t)(blockIdx.y % slots) * 2 * C + (size_t)which * C + blockIdx.x * 64 + ch, t);
    }
}

int slab_epilogue(const float *slabs, int nslabs, long total, int Ko, const ConvEpilogue &ep, float *out, hipStream_t st) {
    if (double *totals = bn_stats_request()) {
        if (Ko % 64 == 0 && !ep.dact && !ep.addend && nslabs <= 8) {
            const long rows = total / Ko;
            const int cblocks = Ko / 64;
            long splits = 512 / cblocks;                  // two workgroups per CU
            if (splits < 1) splits = 1;
            long rps = (rows + splits - 1) / splits;
            if (rps < 16) rps = 16;
            splits = (rows + rps - 1) / rps;
            const dim3 grid((unsigned)cblocks, (unsigned)splits);
            if (rps >= 32)
                hipLaunchKernelGGL(slab_epilogue_bn_kernel<2>, grid, dim3(256), 0, st, slabs, nslabs, total, rows, Ko, rps, ep, out, totals,
                                   bn_slot_count(Ko));
            else
                hipLaunchKernelGGL(slab_epilogue_bn_kernel<1>, grid, dim3(256), 0, st, slabs, nslabs, total, rows, Ko, rps, ep, out, totals,
                                   bn_slot_count(Ko));
            bn_stats_applied();
            return check_launch("conv slab epilogue (+ batch-norm statistics)");
        }
    }
    const long n4 = total / 4;
    long blocks = (n4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(slab_epilogue_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (const float4 *)slabs, nslabs, n4, Ko, ep,
                       (float4 *)out);
    addend_applied();
    return check_launch("conv slab epilogue");
}

// reduction parts a launch with pre-transformed weights is cut into: enough to reach two workgroups per CU, each part
// keeping >= 4 stages (one stage = 16 channels of one segment) so the extra epilogues stay a fraction of the work.
// Only grids below 3/4 of the 512 slots: alone every split launch wins (CIFAR batch 64, us, direct kernel / unsplit / split:
// D l6 forward, 128 items, 82 / 119 / 61; G l2 tc forward 48 / 64 / 39; G l2 tc input-gradient 47 / 119 / 40; G l3 tc
// input-gradient, 256 items, 45 / 62 / 38; D l6 3B input-gradient, 384 items, 111 / 90 / 86), but inside the step the idle
// slots of a 3/4-full grid are already used by the weight-gradient stream, and the second pass then costs more than the
// split gains: ms per CIFAR / STL step with the split applied below 0 / 129 / 257 / 385 / 512 items:
// 2.061 / 2.057 / 2.034 / 2.071 / 2.084 and 4.027 / 4.050 / 3.969 / 3.932 / 3.939 (STL's 288-item launches sit in between).
static int wino2_ksplit(long base_items, int nstages) {
    if (base_items >= tuning().wino2_ksplit_below) return 1;
    int k = 1;
    while (k < 8 && base_items * k < 512 && nstages % (2 * k) == 0 && nstages / (2 * k) >= 4) k *= 2;
    return k;
}

static int wino2_launch(const ConvDims &d, const ConvEpilogue &ep, const float *in, const float *w, const float *U, float *out,
                        bool dgrad, hipStream_t st) {
    const bool own_u = U != nullptr;
    if (!U) {
        float *ws = (float *)workspace_acquire(wino2_bytes(d), st);
        if (!ws) { set_error("conv2d (winograd 2x2): no workspace for the transformed weights"); return MMDGAN_E_ARG; }
        if (int rc = wino2_transform(d, w, dgrad, ws, st)) return rc;
        U = ws;
    }
    wino2::Params P;
    P.N = d.N; P.TH = d.P / 2; P.TW = d.Q / 2;
    if (!dgrad) {            // y tile (2ty.., 2tx..) <- x rows 4ty - 1 + a + 2u
        P.IH = d.H; P.IW = d.W; P.Cr = d.C; P.nseg = 4;
        for (int s = 0; s < 4; ++s) { P.r0[s] = -1 + (s >> 1); P.c0[s] = -1 + (s & 1); P.o0r[s] = 0; P.o0c[s] = 0; }
        P.tstep = 4; P.pstep = 2;
        P.OH = d.P; P.OW = d.Q; P.Ko = d.K; P.otile = 2; P.ostep = 1;
    } else {                 // dx phase (al,be), tile (h' = 2ty.., w' = 2tx..) <- dy rows 2ty + al - 1 + u
        P.IH = d.P; P.IW = d.Q; P.Cr = d.K; P.nseg = 1;
        for (int s = 0; s < 4; ++s) { P.r0[s] = (s >> 1) - 1; P.c0[s] = (s & 1) - 1; P.o0r[s] = s >> 1; P.o0c[s] = s & 1; }
        P.tstep = 2; P.pstep = 1;
        P.OH = d.H; P.OW = d.W; P.Ko = d.C; P.otile = 4; P.ostep = 2;
    }
    const long T = (long)d.N * P.TH * P.TW;
    P.ntb = (int)((T + 31) / 32); P.nkb = P.Ko / 64; P.nph = dgrad ? 4 : 1;
    const int nstages = P.nseg * (P.Cr / wino2::BC);
    // small grids with weights transformed by the caller (the workspace is free then): split the reduction, partial
    // results into workspace slabs, one pass sums them and applies the epilogue
    P.ksplit = own_u && wino2_split_enabled() && d.N > 1 ? wino2_ksplit((long)P.ntb * P.nkb * P.nph, nstages) : 1;
    const size_t out_bytes = sizeof(float) * (size_t)d.N * P.OH * P.OW * P.Ko;
    float *slabs = nullptr;
    if (P.ksplit > 1 && (size_t)P.ksplit * out_bytes < (1ul << 31)) slabs = (float *)workspace_acquire((size_t)P.ksplit * out_bytes, st);
    if (!slabs) P.ksplit = 1;
    P.spp = nstages / P.ksplit;
    P.contiguous = 0;                                   // (contiguous runs per workgroup re-fetch a tile block's patches: 1.55x the bytes)
    P.slab_bytes = (unsigned)out_bytes;
    const long nitems = (long)P.ntb * P.nkb * P.nph * P.ksplit;
    static int ncu = 0;
    if (!ncu) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ncu = prop.multiProcessorCount;
        if (ncu <= 0) ncu = 256;
    }
    const long slots = 2L * ncu;                        // two workgroups per CU (76 KB of LDS each)
    const dim3 grid((unsigned)(nitems < slots ? nitems : slots), 1, 1);
    static bool cap_raised = false;                     // 76 KB of dynamic LDS: above the 64 KB default cap
    if (!cap_raised) {
        (void)hipFuncSetAttribute((const void *)wino2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)wino2::LDS_BYTES);
        cap_raised = true;
    }
    if (P.ksplit > 1) {
        ConvEpilogue plain{};
        plain.wrap_from = kNoWrap;
        hipLaunchKernelGGL(wino2_kernel, grid, dim3(256), wino2::LDS_BYTES, st, P, plain, in, U, slabs);
        if (int rc = check_launch(dgrad ? "conv2d_dgrad(winograd 2x2 split)" : "conv2d_fwd(winograd 2x2 split)")) return rc;
        return slab_epilogue(slabs, P.ksplit, (long)(out_bytes / 4), P.Ko, ep, out, st);
    } else
        hipLaunchKernelGGL(wino2_kernel, grid, dim3(256), wino2::LDS_BYTES, st, P, ep, in, U, out);
    return check_launch(dgrad ? "conv2d_dgrad(winograd 2x2)" : "conv2d_fwd(winograd 2x2)");
}

int wino2_fwd(const ConvDims &d, const ConvEpilogue &ep, const float *x, const float *w, const float *U, float *y, hipStream_t st) {
    return wino2_launch(d, ep, x, w, U, y, false, st);
}
int wino2_dgrad(const ConvDims &d, const ConvEpilogue &ep, const float *dy, const float *w, const float *U, float *dx, hipStream_t st) {
    return wino2_launch(d, ep, dy, w, U, dx, true, st);
}



// ================================================================================================
// Weight gradient of the 4x4 / stride-2 layers in the F(2x2,2x2) domain.  The 16 taps split into four 2x2 filters
// (tap parities a,b), each the gradient of a 2x2 stride-1 correlation on a parity sub-image of x:
//   dW^{ab} = G^T [ sum_tiles (B^T d^{ab} B) (.) (A dY A^T) ] G        - 9 multiplies per tile and (c,k) instead of 16
// Workgroup = 64 input channels x 128 output channels x one parity x a split of the tile range, eight waves; wave
// (c half, k quarter) keeps all nine 32x32 frequency accumulators of its sub-block (144 registers), so
//   * V = B^T d B: a stage is 16 tiles; thread = (tile of the stage, channel quad) loads the 9 patch pixels as float4, does the
//     transform in registers and writes V[f][k-pair][c half][tile parity][32 c] (the A fragment of a k-pair is 64 consecutive
//     words: conflict-free);
//   * the 2x2 dY pixels of the stage's tiles go through LDS as well, coalesced float4 loads by all threads;
//   * dM = A dY A^T (nine values) is built in registers from four LDS words, 13 LDS words per 9 MFMAs;
//   * the epilogue G^T dU G is register-only (every wave owns all frequencies of its outputs) and stores the workgroup's
//     4 taps x 64 x 128 partial result into its slab of the library workspace [split][16][C][K]; one reduction pass sums the
//     slabs (no zeroing, no atomics, deterministic).  Without a workspace: fp32 atomics into the zeroed dw.
// History (D l2 / l4 / l6 at batch 128, direct implicit-GEMM kernel 92 / 86 / 78 us): 32c x 64k tiles with every lane
// fetching its own dY words from global memory 114 / 110 / 99 us; dY through LDS, six waves, operand prefetch 100 / 89 / 90 us
// - a 32 x 64 tile moves 17 flop per byte of L2 traffic and spends as many VALU cycles producing V as MFMA cycles using
// it; 64 x 128 doubles both ratios.
// -DWG_ABLATE=<mask> (tools/wgrad_ablate.hip; 0 in the library): leave parts of wino2_wgrad_kernel out to see what they cost -
// 1 patch loads, 2 input transform + V stores, 4 dY loads + stores, 8 operand reads from LDS, 16 the stage barrier, 32 the slab stores
#ifndef WG_ABLATE
#define WG_ABLATE 0
#endif
namespace wino2w {
constexpr int BT = 16;                        // tiles per stage = 8 MFMA k-pairs
constexpr int BC = 64, BK = 128;              // workgroup tile
constexpr int NT = 512;                       // 8 waves
constexpr int FS = BT * BC;                   // V: floats per frequency, [k-pair 8][c half 2][tile parity 2][32]
constexpr int V_FLOATS = 9 * FS;
constexpr int DYT = 4 * BK + 32;              // dY stage tile: floats per tile ([4 pixels][128 channels]; +32: the two half-waves of
                                              // a read (tiles 2kp, 2kp+1) land on disjoint banks)
constexpr int DY_FLOATS = BT * DYT;
constexpr int SMEM_FLOATS = 2 * V_FLOATS + 2 * DY_FLOATS;
constexpr size_t LDS_BYTES = sizeof(float) * SMEM_FLOATS + sizeof(unsigned) * 4 * BT;
constexpr int DYV = BT * 4 * (BK / 4) / NT;   // float4 items of the dY stage tile per thread (4)
static_assert(BT * 4 * (BK / 4) % NT == 0, "dY items");
static_assert(LDS_BYTES <= 160 * 1024, "LDS");
}  // namespace wino2w

template <bool PART, bool DBIAS>
__global__ __launch_bounds__(wino2w::NT) void wino2_wgrad_kernel(int N, int H, int W, int C, int K, const float *__restrict__ x,
                                                                 const float *__restrict__ dy, float *__restrict__ dw,
                                                                 float *__restrict__ dbpart, int stages_per_split, SlabReduceArgs prev) {
    using namespace wino2w;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, kh = lane >> 5;
    // the previous weight-gradient launch of this stream left its slabs un-summed (mmdgan_wgrad_defer): this workgroup's share first
    slab_reduce_share(prev, blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), gridDim.x * gridDim.y * gridDim.z, tid, NT,
                      reinterpret_cast<double *>(smem));
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wv & 1, wk = wv >> 1;                                   // 32-channel half of the 64, 32-column quarter of the 128
    const int P = H >> 1, Q = W >> 1, TH = P >> 1, TW = Q >> 1;
    const long T = (long)N * TH * TW;
    const int nst_all = (int)((T + BT - 1) / BT);
    const int c0 = blockIdx.x * BC, n0 = blockIdx.y * BK;
    const int par = blockIdx.z & 3, pa = par >> 1, pb = par & 1;
    const int s0 = (blockIdx.z >> 2) * stages_per_split, s1 = min(nst_all, s0 + stages_per_split);
    if (s0 >= s1) return;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(x, (long)N * H * W * C * 4);
    const __amdgpu_buffer_rsrc_t rdy = make_rsrc(dy, (long)N * P * Q * K * 4);
    float *Vs = smem, *DYs = smem + 2 * V_FLOATS;
    unsigned *dyoff = reinterpret_cast<unsigned *>(smem + SMEM_FLOATS);      // [stage & 3][tile of the stage]
    // ---- producer of V: thread = (tile of the stage pt, channel quad cq), threads 0 .. 16*BT-1 (waves 0..3)
    const int pt = tid >> 4, cq = tid & 15;
    const bool xprod = tid < 16 * BT;
    int pstage = s0;
    // tile coordinates of this thread's tile, advanced by BT tiles per stage without divisions
    const int adv_tx = BT % TW, adv_ty = (BT / TW) % TH, adv_n = (BT / TW) / TH;
    int tx, ty, n;
    {
        const long ii = (long)s0 * BT + pt;
        tx = ii % TW, ty = (ii / TW) % TH, n = ii / ((long)TW * TH);
    }
    float4 rin[3][3];
    auto xload = [&]() {
        if ((WG_ABLATE & 1) && pstage > s0 + 1) { ++pstage; return; }
        if (xprod) {
            const bool ok = n < N && pstage < s1;          // beyond the batch (ragged last stage) or the split: no traffic
            if (cq == 0)          // byte offset of dY pixel (2ty, 2tx), channel 0, of this tile (the dY producers add the rest)
                dyoff[(pstage & 3) * BT + pt] = ok ? (unsigned)((((n * P + 2 * ty) * Q + 2 * tx) * K) * 4) : kOOB;
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int row = 4 * ty - 1 + pa + 2 * u;
                const bool rowok = ok && row >= 0 && row < H;
#pragma unroll
                for (int v = 0; v < 3; ++v) {
                    const int col = 4 * tx - 1 + pb + 2 * v;
                    rin[u][v] = bufld4(rx, (rowok && col >= 0 && col < W) ? (unsigned)((((n * H + row) * W + col) * C + c0 + 4 * cq) * 4) : kOOB);
                }
            }
            tx += adv_tx;
            const int cx = tx >= TW;
            tx -= cx ? TW : 0;
            ty += adv_ty + cx;
            const int cy = ty >= TH;
            ty -= cy ? TH : 0;
            n += adv_n + cy;
        }
        ++pstage;
    };
    auto vstore = [&](float *buf) {           // V[f = 3i + j][k-pair][c half][tile parity][4 channels of the quad]
        if (!xprod) return;
        if ((WG_ABLATE & 2) && pstage > s0 + 2) return;
        float4 X[3][3];
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const float4 d0 = rin[u][0], d1 = rin[u][1], d2 = rin[u][2];
            X[u][0] = make_float4(d0.x - d1.x, d0.y - d1.y, d0.z - d1.z, d0.w - d1.w);
            X[u][1] = d1;
            X[u][2] = make_float4(d2.x - d1.x, d2.y - d1.y, d2.z - d1.z, d2.w - d1.w);
        }
        float *dst = buf + (pt >> 1) * 128 + (cq >> 3) * 64 + (pt & 1) * 32 + (cq & 7) * 4;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const float4 m = X[1][j];
            *reinterpret_cast<float4 *>(dst + (0 * 3 + j) * FS) = make_float4(X[0][j].x - m.x, X[0][j].y - m.y, X[0][j].z - m.z, X[0][j].w - m.w);
            *reinterpret_cast<float4 *>(dst + (1 * 3 + j) * FS) = m;
            *reinterpret_cast<float4 *>(dst + (2 * 3 + j) * FS) = make_float4(X[2][j].x - m.x, X[2][j].y - m.y, X[2][j].z - m.z, X[2][j].w - m.w);
        }
    };
    // ---- producer of the dY stage tile: item = (tile, pixel of its 2x2, channel quad of the 128), all threads
    const unsigned dyrow = (unsigned)(Q * K * 4), dypix = (unsigned)(K * 4);
    float4 rdyv[DYV];
    auto dyload = [&](int stage) {            // needs dyoff[stage], written at least one barrier ago
        if ((WG_ABLATE & 4) && stage > s0 + 1) return;
#pragma unroll
        for (int it = 0; it < DYV; ++it) {
            const int e = tid + it * NT;
            const int tile = e >> 7, px = (e >> 5) & 3, kq = e & 31;
            const unsigned off = dyoff[(stage & 3) * BT + tile];
            rdyv[it] = bufld4(rdy, off == kOOB ? kOOB : off + (px >> 1) * dyrow + (px & 1) * dypix + (unsigned)((n0 + 4 * kq) * 4));
        }
    };
    // column sums of dY (the bias gradient) ride along in the workgroups of channel block 0 / parity 0: every dY pixel passes
    // through exactly one of them per column block; thread's items all have the same channel quad (tid & 31)
    const bool dosum = DBIAS && blockIdx.x == 0 && par == 0;
    float4 dbs = make_float4(0.f, 0.f, 0.f, 0.f);
    auto dystore = [&](float *buf) {          // (tiles beyond the split were not loaded: zeros)
        if ((WG_ABLATE & 4) && pstage > s0 + 3) return;
#pragma unroll
        for (int it = 0; it < DYV; ++it) {
            const int e = tid + it * NT;
            const int tile = e >> 7, px = (e >> 5) & 3, kq = e & 31;
            *reinterpret_cast<float4 *>(buf + tile * DYT + px * BK + 4 * kq) = rdyv[it];
            if (dosum) { dbs.x += rdyv[it].x; dbs.y += rdyv[it].y; dbs.z += rdyv[it].z; dbs.w += rdyv[it].w; }
        }
    };

    f32x16 acc[9];
#pragma unroll
    for (int f = 0; f < 9; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;

    xload();
    vstore(Vs);
    xload();
    __syncthreads();                          // dyoff[s0], dyoff[s0 + 1] visible
    dyload(s0);
    dystore(DYs);
    __syncthreads();
    const int abase = wc * 64 + lane;                        // + f * FS + kp * 128
    const int dbase = kh * DYT + wk * 32 + l31;              // + 2 kp * DYT + pixel * BK
    float fa[9], dv[4];
    auto opload = [&](const float *cur, const float *dcur, int kp) {     // operands of k-pair kp: 9 + 4 LDS words
        if ((WG_ABLATE & 8) && (cur != Vs || kp)) return;
#pragma unroll
        for (int f = 0; f < 9; ++f) fa[f] = cur[abase + f * FS + kp * 128];
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
            // A dY A^T, A = [1 0; 1 1; 0 1]
            const float m00 = dv[0], m02 = dv[1], m20 = dv[2], m22 = dv[3];
            const float m01 = m00 + m02, m21 = m20 + m22, m10 = m00 + m20, m12 = m02 + m22, m11 = m01 + m21;
            const float bm[9] = {m00, m01, m02, m10, m11, m12, m20, m21, m22};
            float a[9];
#pragma unroll
            for (int f = 0; f < 9; ++f) a[f] = fa[f];
            if (kp + 1 < BT / 2) opload(cur, dcur, kp + 1);           // next k-pair's operands fly under this one's MFMAs
#pragma unroll
            for (int f = 0; f < 9; ++f) acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[f], bm[f], acc[f], 0, 0, 0);
            if (kp == 0) { vstore(nxt); dyload(s + 1); }      // tile s+1: V -> LDS, its dY -> registers
            else if (kp == 4) xload();                        // tile s+2 -> registers (and its dY offsets)
            else if (kp == 6) dystore(dnxt);
        }
        if (!(WG_ABLATE & 16)) __syncthreads();
    }

    if (DBIAS && dosum) {                      // (workgroup-uniform) 16 threads per channel quad -> one partial row of the split
        float *red = smem;                     // the last stage ended with a barrier
        *reinterpret_cast<float4 *>(red + (tid >> 5) * BK + 4 * (tid & 31)) = dbs;
        __syncthreads();
        if (tid < BK) {
            float a = 0.f;
#pragma unroll
            for (int q = 0; q < NT / 32; ++q) a += red[q * BK + tid];
            dbpart[(long)(blockIdx.z >> 2) * K + n0 + tid] = a;
        }
    }
    // ---- G^T dU G in registers (G = [1 0; 1 1; 0 1]): tap (row 2u + pa, col 2v + pb) = sum_{i in {u,u+1}, j in {v,v+1}} dU[i][j]
    const long CK = (long)C * K;
    float *dst0 = dw + (long)(c0 + wc * 32 + 4 * kh) * K + n0 + wk * 32 + l31;
    if (PART) dst0 += (long)(blockIdx.z >> 2) * 16 * CK;                // this split's slab of the workspace
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            float *dt = dst0 + (long)((2 * u + pa) * 4 + 2 * v + pb) * CK;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float val = (acc[3 * u + v][r] + acc[3 * u + v + 1][r]) + (acc[3 * u + 3 + v][r] + acc[3 * u + 4 + v][r]);
                float *d1 = dt + (long)((r & 3) + 8 * (r >> 2)) * K;
                if ((WG_ABLATE & 32) && val != 1.2345e33f) continue;
                if (PART) *d1 = val;
                else atomicAdd(d1, val);
            }
        }
}

// dw[e] = sum over the splits' slabs in a fixed order (slab_reduce.h: deterministic, and the same bits as the prologue form).
// Elements n4 .. n4 + k4 - 1 are the bias gradient: partial rows [split][K] -> dbias.
// wdot (optional): dot[0] += <dw, wdot> on the way - the scalar the spectral-norm fix-up of this gradient needs
// (mmdgan_conv2d_wgrad_sn): one atomic per workgroup instead of a separate pass over dw and the kernel.
__global__ __launch_bounds__(256) void slab_reduce_kernel(SlabReduceArgs a) {
    __shared__ double red[4];
    slab_reduce_share<2>(a, blockIdx.x, gridDim.x, threadIdx.x, 256, red);   // (a small register footprint: it runs beside MFMA kernels)
}

void slab_reduce_launch(const SlabReduceArgs &a, hipStream_t st) {
    const long total = a.n4 + a.k4;
    long blocks = (total + 63) / 64;                    // >= one 1 KB run per workgroup
    // at most 512 workgroups: this pass runs on the weight-gradient stream beside the main stream's launches, and more of its
    // small workgroups cost those launches more than they save here (CIFAR step, cap 2048 / 1024 / 512 / 256 / 128: 1.931 / 1.919 /
    // 1.910 / 1.922 / 1.966 ms; CelebA flat down to 512)
    if (blocks > 512) blocks = 512;
    hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a);
}

static SlabReduceArgs slab_args(const float *part, int nsplit, size_t n, float *dw, const float *dbpart, int k, float *dbias,
                                const float *wdot, float *dot) {
    SlabReduceArgs a;
    a.part = (const float4 *)part; a.nsplit = nsplit; a.n4 = (long)(n / 4); a.dw = (float4 *)dw;
    a.dbpart = (const float4 *)dbpart; a.k4 = k / 4; a.dbias = (float4 *)dbias; a.wdot = (const float4 *)wdot; a.dot = dot;
    return a;
}
// what a slab weight-gradient launch leaves behind: summed right away by the stand-alone pass, or - mmdgan_wgrad_defer - by
// the prologue of the next weight-gradient launch on this stream (core.hip)
int slab_reduce(const float *part, int nsplit, size_t n, float *dw, const float *dbpart, int k, float *dbias, hipStream_t st,
                const float *wdot, float *dot) {
    return wgrad_slabs_release(slab_args(part, nsplit, n, dw, dbpart, k, dbias, wdot, dot), st);
}

// MMDGAN_WINO2_WGRAD=0 keeps the stride-2 weight gradients on the direct implicit-GEMM kernel, =1 uses this one;
// MMDGAN_WINO2=2 (the parity tests) always.
bool wino2_wgrad_ok(const ConvDims &d) {
    const int en = tuning().wino2_wgrad;
    if ((!en && wino2_mode() < 2) || wino2_mode() == 0) return false;
    if (d.N == 1) return false;     // batch-1 = a power iteration's launch, on a chain concurrent with others: no workspace slabs
    // the batch-1 weight gradients of the power iteration (64 tiles) stay direct: 7.7 us against 7 + the 6 us reduction pass
    // (64-255 tiles with enough channel blocks for a full round of workgroups - the ResNet generator's first up-sampling
    // block, 512 x 1024 channels at 128 tiles - run here as well; MMDGAN_WINO2_WGRAD_MIN_TILES)
    const long min_tiles = tuning().wino2_wgrad_min_tiles;
    const long tiles = (long)d.N * (d.P / 2) * (d.Q / 2), blocks = (long)(d.C / wino2w::BC) * (d.K / wino2w::BK) * 4;
    if (wino2_mode() < 2 && tiles < min_tiles && !(tiles >= 64 && blocks >= 192 && min_tiles <= 256)) return false;
    return d.R == 4 && d.stride == 2 && d.pad == 1 && d.H % 4 == 0 && d.W % 4 == 0 && d.C % wino2w::BC == 0 && d.K % wino2w::BK == 0;
}

// dbias (optional): the column sums of dy.  Returns 0 with *dbias_done = whether dbias was produced here (workspace path);
// otherwise the caller sums dy itself.
int wino2_wgrad(const ConvDims &d, const float *x, const float *dy, float *dw, float *dbias, bool *dbias_done, hipStream_t st,
                const float *wdot, float *dot, bool *dot_done) {
    if (dot_done) *dot_done = false;
    const long T = (long)d.N * (d.P / 2) * (d.Q / 2);
    const int nst = (int)((T + wino2w::BT - 1) / wino2w::BT);
    const long base = (long)(d.C / wino2w::BC) * (d.K / wino2w::BK) * 4;
    int split = (int)(wgrad_cus() / base);                                  // one 8-wave workgroup per CU (140 KB of LDS), one round
    if (split > nst / 2) split = nst / 2;                           // >= 2 stages (144 MFMAs per wave) per workgroup
    if (split < 1) split = 1;
    const int sps = (nst + split - 1) / split;
    split = (nst + sps - 1) / sps;                                  // every slab gets written
    const size_t n = 16 * (size_t)d.C * d.K;
    static bool cap_raised = false;
    if (!cap_raised) {
        (void)hipFuncSetAttribute((const void *)wino2_wgrad_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)wino2w::LDS_BYTES);
        (void)hipFuncSetAttribute((const void *)wino2_wgrad_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)wino2w::LDS_BYTES);
        (void)hipFuncSetAttribute((const void *)wino2_wgrad_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)wino2w::LDS_BYTES);
        cap_raised = true;
    }
    if (dbias_done) *dbias_done = false;
    const dim3 grid(d.C / wino2w::BC, d.K / wino2w::BK, 4 * split);
    SlabReduceArgs prev{};
    if (float *part = (float *)wgrad_slabs_acquire(sizeof(float) * (n + d.K) * split, st, &prev)) {
        float *dbpart = part + n * split;
        if (dbias)
            hipLaunchKernelGGL((wino2_wgrad_kernel<true, true>), grid, dim3(wino2w::NT), wino2w::LDS_BYTES, st, d.N, d.H, d.W, d.C, d.K, x, dy, part,
                               dbpart, sps, prev);
        else
            hipLaunchKernelGGL((wino2_wgrad_kernel<true, false>), grid, dim3(wino2w::NT), wino2w::LDS_BYTES, st, d.N, d.H, d.W, d.C, d.K, x, dy, part,
                               dbpart, sps, prev);
        if (int rc = slab_reduce(part, split, n, dw, dbpart, dbias ? d.K : 0, dbias, st, wdot, dot)) return rc;
        if (dbias_done) *dbias_done = dbias != nullptr;
        if (dot_done) *dot_done = wdot != nullptr;
        return check_launch("conv2d_wgrad(winograd 2x2)");
    }
    if (split == 1) {                           // one slab: straight into dw
        hipLaunchKernelGGL((wino2_wgrad_kernel<true, false>), grid, dim3(wino2w::NT), wino2w::LDS_BYTES, st, d.N, d.H, d.W, d.C, d.K, x, dy, dw,
                           (float *)nullptr, sps, SlabReduceArgs{});
        return check_launch("conv2d_wgrad(winograd 2x2)");
    }
    if (zero_output(dw, sizeof(float) * n, st) != hipSuccess) return check_launch("conv2d_wgrad memset");
    hipLaunchKernelGGL((wino2_wgrad_kernel<false, false>), grid, dim3(wino2w::NT), wino2w::LDS_BYTES, st, d.N, d.H, d.W, d.C, d.K, x, dy, dw,
                       (float *)nullptr, sps, SlabReduceArgs{});
    return check_launch("conv2d_wgrad(winograd 2x2)");
}

}  // namespace mmdgan
