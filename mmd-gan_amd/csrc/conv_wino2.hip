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
#include "conv_internal.h"
#include "bufload.h"

#ifndef W2W_DEFAULT
#define W2W_DEFAULT 1
#endif
#ifndef W2_FUSE_MIN_WGS
#define W2_FUSE_MIN_WGS 0
#endif

namespace mmdgan {

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace wino2 {
constexpr int BC = 32;                  // reduction channels per stage: one 128-byte line of every patch pixel
constexpr int LDT = 33;                 // V: floats per channel row (32 tiles + 1)
constexpr int FSV = BC * LDT + 2;       // V: floats per frequency
constexpr int V_FLOATS = 9 * FSV;
constexpr int MS_FLOATS = 9 * 32 * 32;  // epilogue exchange buffer (one column block at a time)
constexpr int SMEM_FLOATS = 2 * V_FLOATS > MS_FLOATS ? 2 * V_FLOATS : MS_FLOATS;
constexpr size_t LDS_BYTES = sizeof(float) * SMEM_FLOATS + sizeof(long) * 32;
constexpr int NM = 5;                   // frequencies per wave (waves 2, 3 use 4)

struct Params {
    int N, TH, TW;          // tiles per image (rows, cols)
    int IH, IW, Cr;         // input image, reduction channels per segment
    int nseg;               // segments per phase (1 or 4); phase = blockIdx.z
    int r0[4], c0[4];       // patch origin (input row / col of patch element (0,0) for tile (0,0)) per global segment
    int tstep, pstep;       // input rows (cols) between consecutive tiles / consecutive patch rows (cols)
    int OH, OW, Ko;         // output image, channels
    int otile, ostep;       // output row = ty * otile + a * ostep + o0r[phase]
    int o0r[4], o0c[4];
    int ntb, nkb, nph;      // workgroups = tile blocks x column blocks x phases, launched as a 1-D grid
    int xcd_remap;          // 1: workgroup -> (tile block, phase, column block) through the XCD-aware map below
};
}  // namespace wino2

// U[seg][f][cr][ko] = (G g G^T)[f], f = 3i + j, g = the 2x2 filter of the segment
//   FWD  : seg = (a,b);  g[u][v] = w[2u + a][2v + b][c][k];            cr = c, ko = k
//   DGRAD: seg = (al,be); g[u][v] = w[rho(al,1-u)][rho(be,1-v)][c][k], rho(0,r') = 1 + 2r', rho(1,r') = 2r';  cr = k, ko = c
template <bool DGRAD>
__global__ __launch_bounds__(256) void wino2_weight_kernel(const float *__restrict__ w, float *__restrict__ U, int C, int K) {
    __shared__ float tile[9][32][33];
    const int c0 = blockIdx.y * 32, k0 = blockIdx.x * 32, seg = blockIdx.z, sa = seg >> 1, sb = seg & 1;
    const int tk = threadIdx.x & 31, tq = threadIdx.x >> 5;
    for (int cc = tq; cc < 32; cc += 8) {
        const int c = c0 + cc, k = k0 + tk;
        const bool ok = c < C && k < K;
        float g[2][2];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int v = 0; v < 2; ++v) {
                const int r = DGRAD ? (sa == 0 ? 1 + 2 * (1 - u) : 2 * (1 - u)) : 2 * u + sa;
                const int t = DGRAD ? (sb == 0 ? 1 + 2 * (1 - v) : 2 * (1 - v)) : 2 * v + sb;
                g[u][v] = ok ? w[((size_t)(r * 4 + t) * C + c) * K + k] : 0.f;
            }
        float gg[3][2], uu[3][3];
#pragma unroll
        for (int v = 0; v < 2; ++v) { gg[0][v] = g[0][v]; gg[1][v] = g[0][v] + g[1][v]; gg[2][v] = g[1][v]; }
#pragma unroll
        for (int i = 0; i < 3; ++i) { uu[i][0] = gg[i][0]; uu[i][1] = gg[i][0] + gg[i][1]; uu[i][2] = gg[i][1]; }
        if (!DGRAD) {
            if (ok) {
#pragma unroll
                for (int f = 0; f < 9; ++f) U[(((size_t)seg * 9 + f) * C + c) * K + k] = uu[f / 3][f % 3];
            }
        } else {
#pragma unroll
            for (int f = 0; f < 9; ++f) tile[f][cc][tk] = uu[f / 3][f % 3];
        }
    }
    if (DGRAD) {             // transposed write: U[seg][f][k][c], threads along c
        __syncthreads();
        for (int kk = tq; kk < 32; kk += 8) {
            const int k = k0 + kk, c = c0 + tk;
            if (c < C && k < K) {
#pragma unroll
                for (int f = 0; f < 9; ++f) U[(((size_t)seg * 9 + f) * K + k) * C + c] = tile[f][tk][kk];
            }
        }
    }
}

// A stage is 32 reduction channels of one segment = 80 (64) MFMAs per wave between barriers:
//   producer  thread = (tile, channel quad): the 9 pixels of its 3x3 patch as float4 (8 neighbouring lanes read one
//             full 128-byte line of each pixel; the first version read 32 bytes per pixel and stage and paid a third of
//             its time re-fetching lines from L2), the whole B^T d B in registers, 36 ds_write_b32;
//   consumer  A fragments from LDS one k-pair ahead, B fragments straight from L2, every register refilled for 4
//             k-pairs later right after the MFMA that consumed it.
// FUSE: two neighbouring tile blocks of the same (phase, column block) run as ONE 512-thread workgroup - two groups of four
// waves with the same instruction stream, their own halves of LDS and shared barriers.  Both groups read the SAME B fragments
// at about the same time, so the second read is an L1 hit (or merges with the pending miss): the B traffic between L2 and the
// CU - the vector-memory path this kernel is bound by - halves, with no more registers or LDS per CU than two workgroups.
template <bool FUSE>
__global__ __launch_bounds__(FUSE ? 512 : 256, FUSE ? 1 : 2) void wino2_kernel(wino2::Params P, ConvEpilogue ep, const float *__restrict__ x,
                                                                               const float *__restrict__ U, float *__restrict__ out) {
    using namespace wino2;
    extern __shared__ __attribute__((aligned(16))) float smem_all[];
    const int grp = FUSE ? __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8) : 0;
    float *smem = smem_all + grp * (int)(LDS_BYTES / sizeof(float));
    const int tid = threadIdx.x & 255, lane = tid & 63, l31 = lane & 31, kh = lane >> 5;
    // wave-uniform by construction, but hipcc only knows once it sits in an SGPR: without this every B load with a
    // wave-dependent scalar offset became a waterfall loop and every `m < nm` an exec-mask branch (2x slower)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long T = (long)P.N * P.TH * P.TW;
    // XCD-aware placement: the 4 parity phases (and the column blocks) of one tile block read the SAME 3x3 input patches.
    // Launched phase-major they run hundreds of microseconds apart on all 8 XCDs and every phase fetches its patches
    // from HBM again (FETCH_SIZE 2.4x the algorithmic reads, profiles/r01_dominant_kernel_pmc.json).  The dispatcher
    // deals workgroup b to XCD b % 8 (observed, a speed assumption only): give each XCD a contiguous range of tile
    // blocks and walk (phase, column block) innermost, so the re-reads of a patch hit that XCD's 4 MB L2.
    int wg = blockIdx.x;
    if (P.xcd_remap) {
        const int nwg = gridDim.x, xcd = wg & 7, q = nwg >> 3, r = nwg & 7;      // bijective for any nwg
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (wg >> 3);
    }
    const int per_tb = P.nkb * P.nph;
    const int tblk = wg / per_tb, rem = wg - tblk * per_tb;
    const int phase = P.xcd_remap ? rem / P.nkb : wg / (P.ntb * P.nkb);
    const int t0 = ((P.xcd_remap ? tblk : wg % P.ntb) * (FUSE ? 2 : 1) + grp) * 32;   // (P.ntb counts workgroups)
    const int n0 = (P.xcd_remap ? rem % P.nkb : (wg / P.ntb) % P.nkb) * 64;
    const int spc = P.Cr / BC;                           // stages per segment
    const int nstages = P.nseg * spc;
    // ---- producer: thread = (tile pt, channel quad cq)
    const int pt = tid >> 3, cq = tid & 7;
    int ty, tx, tn;
    bool tile_ok;
    {
        const long id = (long)t0 + pt;
        tile_ok = id < T;
        const long ii = tile_ok ? id : 0;
        tx = ii % P.TW; ty = (ii / P.TW) % P.TH; tn = ii / ((long)P.TW * P.TH);
        if (cq == 0)         // element offset of this tile's output pixel (a=0, b=0), channel 0 (-1: no such tile)
            reinterpret_cast<long *>(smem + SMEM_FLOATS)[pt] =
                tile_ok ? (((long)tn * P.OH + ty * P.otile + P.o0r[phase]) * P.OW + tx * P.otile + P.o0c[phase]) * P.Ko : -1;
    }
    unsigned xoff[3][3];
    auto set_segment = [&](int sidx) {                  // byte offsets of the 9 patch pixels, channel 4*cq
        const int gs = phase * P.nseg + sidx;
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int row = ty * P.tstep + P.r0[gs] + u * P.pstep;
            const bool rowok = tile_ok && row >= 0 && row < P.IH;
#pragma unroll
            for (int v = 0; v < 3; ++v) {
                const int col = tx * P.tstep + P.c0[gs] + v * P.pstep;
                xoff[u][v] = (rowok && col >= 0 && col < P.IW) ? (unsigned)(((((long)tn * P.IH + row) * P.IW + col) * P.Cr + 4 * cq) * 4) : kOOB;
            }
        }
    };
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(x, (long)P.N * P.IH * P.IW * P.Cr * 4);
    const __amdgpu_buffer_rsrc_t ru = make_rsrc(U, (long)4 * 9 * P.Cr * P.Ko * 4);
    const int vdst = (4 * cq) * LDT + pt;
    // ---- consumer: wave owns column block cb and frequencies fq + 2m
    // 18 accumulators over 4 waves = 5,5,4,4: odd workgroups rotate the roles so that, with two workgroups per CU,
    // every SIMD carries 9 of them
    const int wrole = (wave + (((FUSE ? grp : wg) & 1) << 1)) & 3;
    const int cb = wrole & 1, fq = wrole >> 1;
    const unsigned ubase = (unsigned)(((long)kh * P.Ko + n0 + cb * 32 + l31) * 4);
    const unsigned ufreq = (unsigned)((long)P.Cr * P.Ko * 4), ukp = (unsigned)(2 * P.Ko * 4), ustage = (unsigned)(BC * P.Ko * 4);
    const unsigned useg = 9u * ufreq;
    const int nm = fq == 0 ? 5 : 4;
#ifndef W2_BD
#define W2_BD 4
#endif
    constexpr int NKP = BC / 2, BD = W2_BD;              // k-pairs per stage, B prefetch distance in k-pairs

    f32x16 acc[NM];
#pragma unroll
    for (int m = 0; m < NM; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;

#define W2_XL(E) (E)
#define W2_VS(DST, VAL) DST = VAL
#define W2_BL
    float4 rin[3][3];
    float fb[BD][NM];
    // global -> registers: patch row U_ of stage S
#define W2_XLOAD_ROW(S, U_)                                                                              \
    {                                                                                                    \
        const int sidx_ = (S) / spc, cs_ = (S) - sidx_ * spc;                                            \
        if ((U_) == 0 && cs_ == 0 && sidx_ < P.nseg) set_segment(sidx_);                                 \
        const unsigned sx = (unsigned)(cs_ * BC * 4);          /* padded taps: kOOB + sx stays out of range */ \
        _Pragma("unroll") for (int v = 0; v < 3; ++v) rin[U_][v] = W2_XL(bufld4(rx, xoff[U_][v] + sx));  \
    }
    // registers -> LDS: B^T d B for channel E_ of the quad (row pass over the 3 columns, column pass over the 3 rows)
#define W2_VSTORE_CH(BUF, E_, C_)                                                                        \
    {                                                                                                    \
        float X[3][3];                                                                                   \
        _Pragma("unroll") for (int u = 0; u < 3; ++u) {                                                  \
            X[u][0] = rin[u][0].C_ - rin[u][1].C_; X[u][1] = rin[u][1].C_; X[u][2] = rin[u][2].C_ - rin[u][1].C_; \
        }                                                                                                \
        _Pragma("unroll") for (int j = 0; j < 3; ++j) {                                                  \
            W2_VS((BUF)[vdst + (0 * 3 + j) * FSV + (E_) * LDT], X[0][j] - X[1][j]);                       \
            W2_VS((BUF)[vdst + (1 * 3 + j) * FSV + (E_) * LDT], X[1][j]);                                 \
            W2_VS((BUF)[vdst + (2 * 3 + j) * FSV + (E_) * LDT], X[2][j] - X[1][j]);                       \
        }                                                                                                \
    }
    // B fragment (k-pair KP of stage S, frequency fq + 2*M) into register slot KP % BD
#define W2_BLOAD(KP, M, S)                                                                               \
    {                                                                                                    \
        const int sidx_ = (S) / spc, cs_ = (S) - sidx_ * spc;                                            \
        fb[(KP) % BD][M] = W2_BL bufld1s(ru, ubase, (unsigned)(phase * P.nseg + sidx_) * useg + (unsigned)(fq + 2 * (M)) * ufreq + \
                                                    (unsigned)cs_ * ustage + (KP) * ukp);                \
    }

    W2_XLOAD_ROW(0, 0) W2_XLOAD_ROW(0, 1) W2_XLOAD_ROW(0, 2)
#pragma unroll
    for (int kp = 0; kp < BD; ++kp)
#pragma unroll
        for (int m = 0; m < NM; ++m)
            if (m < nm) W2_BLOAD(kp, m, 0)
    W2_VSTORE_CH(smem, 0, x) W2_VSTORE_CH(smem, 1, y) W2_VSTORE_CH(smem, 2, z) W2_VSTORE_CH(smem, 3, w)
    W2_XLOAD_ROW(1, 0) W2_XLOAD_ROW(1, 1) W2_XLOAD_ROW(1, 2)
    __syncthreads();
    const int abase = fq * FSV + kh * LDT + l31;
    for (int s = 0; s < nstages; ++s) {
        const float *cur = smem + (s & 1) * V_FLOATS;
        float *nxt = smem + ((s + 1) & 1) * V_FLOATS;
        const int sn = s + 1 < nstages ? s + 1 : s;           // the last refills re-read the last stage (unused)
        float fa[2][NM];                                   // A fragments of k-pair kp+1 are read while kp's MFMAs issue
#pragma unroll
        for (int m = 0; m < NM; ++m)
            if (m < nm) fa[0][m] = cur[abase + 2 * m * FSV];
#pragma unroll
        for (int kp = 0; kp < NKP; ++kp) {
            if (kp + 1 < NKP) {
#pragma unroll
                for (int m = 0; m < NM; ++m)
                    if (m < nm) fa[(kp + 1) & 1][m] = cur[abase + 2 * m * FSV + 2 * (kp + 1) * LDT];
            }
#pragma unroll
            for (int m = 0; m < NM; ++m)
                if (m < nm) {
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[kp & 1][m], fb[kp % BD][m], acc[m], 0, 0, 0);
                    if (kp + BD < NKP) W2_BLOAD(kp + BD, m, s) else W2_BLOAD(kp + BD - NKP, m, sn)
                }
            // tile s+1 -> LDS (one channel of the quad per k-pair), then tile s+2 -> registers (one patch row per k-pair)
            if (kp == 0) W2_VSTORE_CH(nxt, 0, x)
            else if (kp == 1) W2_VSTORE_CH(nxt, 1, y)
            else if (kp == 2) W2_VSTORE_CH(nxt, 2, z)
            else if (kp == 3) W2_VSTORE_CH(nxt, 3, w)
            else if (kp == 5) W2_XLOAD_ROW(s + 2, 0)
            else if (kp == 6) W2_XLOAD_ROW(s + 2, 1)
            else if (kp == 7) W2_XLOAD_ROW(s + 2, 2)
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }
#undef W2_XLOAD_ROW
#undef W2_VSTORE_CH
#undef W2_BLOAD

    // ---- output transform + epilogue: Ms[f][tile][k 32], one column block at a time
    const float sc = ep.scale ? ep.scale[0] : 1.f;
    float *Ms = smem;
    const long *obase = reinterpret_cast<const long *>(smem + SMEM_FLOATS);
    const int kq = tid & 7, tb = tid >> 3;
    const long arow = (long)P.ostep * P.OW * P.Ko, bcol = (long)P.ostep * P.Ko;
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
        for (int it = 0; it < 2; ++it) {
            const int item = tb + 32 * it, tile = item >> 1, b = item & 1;
            const long ob = obase[tile];
            if (ob >= 0) {
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
                    const long o = ob + a * arow + b * bcol + ch;
                    v.x = v.x * sc + bv.x; v.y = v.y * sc + bv.y; v.z = v.z * sc + bv.z; v.w = v.w * sc + bv.w;
                    if (ep.dact) {
                        const float4 yv = *reinterpret_cast<const float4 *>(ep.dact + ep.dact_index(o));
                        v.x *= act_bwd_from_out(yv.x, ep.act); v.y *= act_bwd_from_out(yv.y, ep.act);
                        v.z *= act_bwd_from_out(yv.z, ep.act); v.w *= act_bwd_from_out(yv.w, ep.act);
                    } else {
                        v.x = act_fwd(v.x, ep.act); v.y = act_fwd(v.y, ep.act);
                        v.z = act_fwd(v.z, ep.act); v.w = act_fwd(v.w, ep.act);
                    }
                    *reinterpret_cast<float4 *>(out + o) = v;
                }
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// Measured against the direct kernels (CIFAR batch 64, tools/bench_conv.py with MMDGAN_WINO2=2, weight transform
// included): D l2 forward 70 vs 94 us (512 workgroups), 3B-row dgrad 115 vs 146 (1536); D l4 73 vs 81 (256) and
// 100 vs 120 (768); D l6 3B dgrad 109 vs 113 (384); it loses below that (D l6 forward, 128 workgroups: 128 vs 84;
// G's top layers at batch 64).  Default (MMDGAN_WINO2=1): forward with >= 256 workgroups, input-gradient with
// >= 384; =0 never; =2 every eligible shape (what the parity tests run).
static int wino2_mode() {
    static int v = -1;
    if (v < 0) { const char *e = getenv("MMDGAN_WINO2"); v = e ? atoi(e) : 1; }
    return v;
}

// d describes the CONV (4x4, stride 2, pad 1): x [N,H,W,C] -> y [N,P,Q,K]
static bool wino2_shape_ok(const ConvDims &d, bool dgrad) {
    const int mode = wino2_mode();
    if (mode == 0 || d.R != 4 || d.stride != 2 || d.pad != 1 || d.H % 4 || d.W % 4) return false;
    const int cr = dgrad ? d.K : d.C, ko = dgrad ? d.C : d.K;
    if (cr % wino2::BC || cr < 32 || ko % 64) return false;
    if (mode >= 2) return true;
    const long tiles = (long)d.N * (d.P / 2) * (d.Q / 2);          // per phase
    const long wgs = ((tiles + 31) / 32) * (ko / 64) * (dgrad ? 4 : 1);
    return wgs >= (dgrad ? 384 : 256);
}
bool wino2_eligible(const ConvDims &d, bool dgrad) { return wino2_shape_ok(d, dgrad); }
static size_t wino2_bytes(const ConvDims &d) { return sizeof(float) * 36 * (size_t)d.C * d.K; }
bool wino2_fwd_ok(const ConvDims &d) { return d.N > 1 && wino2_shape_ok(d, false) && workspace(wino2_bytes(d)) != nullptr; }
bool wino2_dgrad_ok(const ConvDims &d) { return d.N > 1 && wino2_shape_ok(d, true) && workspace(wino2_bytes(d)) != nullptr; }

int wino2_transform(const ConvDims &d, const float *w, bool dgrad, float *U, hipStream_t st) {
    const dim3 wg((d.K + 31) / 32, (d.C + 31) / 32, 4);
    if (dgrad) hipLaunchKernelGGL(wino2_weight_kernel<true>, wg, dim3(256), 0, st, w, U, d.C, d.K);
    else hipLaunchKernelGGL(wino2_weight_kernel<false>, wg, dim3(256), 0, st, w, U, d.C, d.K);
    return check_launch("wino2_transform");
}

static int wino2_launch(const ConvDims &d, const ConvEpilogue &ep, const float *in, const float *w, const float *U, float *out,
                        bool dgrad, hipStream_t st) {
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
    static int remap = -1, fuse_min = -1;
    if (remap < 0) { const char *e = getenv("MMDGAN_XCD_REMAP"); remap = (e && e[0] == '0') ? 0 : 1; }
    if (fuse_min < 0) { const char *e = getenv("MMDGAN_WINO2_FUSE"); fuse_min = e ? atoi(e) : W2_FUSE_MIN_WGS; }   // 0: never
    P.xcd_remap = remap;
    // pairs of tile blocks as one workgroup where that still leaves every CU its workgroup
    const bool fuse = fuse_min > 0 && (long)P.ntb * P.nkb * P.nph >= fuse_min;
    if (fuse) P.ntb = (P.ntb + 1) / 2;
    const dim3 grid((unsigned)((long)P.ntb * P.nkb * P.nph), 1, 1);
    static bool cap_raised = false;                     // 76 KB of dynamic LDS: above the 64 KB default cap
    if (!cap_raised) {
        (void)hipFuncSetAttribute((const void *)wino2_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)wino2::LDS_BYTES);
        (void)hipFuncSetAttribute((const void *)wino2_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * wino2::LDS_BYTES));
        cap_raised = true;
    }
    if (fuse) hipLaunchKernelGGL(wino2_kernel<true>, grid, dim3(512), 2 * wino2::LDS_BYTES, st, P, ep, in, U, out);
    else hipLaunchKernelGGL(wino2_kernel<false>, grid, dim3(256), wino2::LDS_BYTES, st, P, ep, in, U, out);
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
                                                                 float *__restrict__ dbpart, int stages_per_split) {
    using namespace wino2w;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, kh = lane >> 5;
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
        __syncthreads();
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
                if (PART) *d1 = val;
                else atomicAdd(d1, val);
            }
        }
}

// dw[e] = sum over the splits' slabs (fixed order: deterministic).  One float4 per thread, eight slab loads in flight.
// Elements n4 .. n4 + k4 - 1 are the bias gradient: partial rows [split][K] -> dbias.
__global__ __launch_bounds__(64) void slab_reduce_kernel(const float4 *__restrict__ part, int nsplit, long n4, float4 *__restrict__ dw,
                                                                const float4 *__restrict__ dbpart, long k4, float4 *__restrict__ dbias) {
    long e = (long)blockIdx.x * 64 + threadIdx.x;
    if (e >= n4 + k4) return;
    long stride = n4;
    if (e >= n4) { e -= n4; part = dbpart; stride = k4; dw = dbias; }
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    int s = 0;
    for (; s + 8 <= nsplit; s += 8) {
        float4 v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = part[(long)(s + q) * stride + e];
#pragma unroll
        for (int q = 0; q < 8; ++q) { a.x += v[q].x; a.y += v[q].y; a.z += v[q].z; a.w += v[q].w; }
    }
    for (; s < nsplit; ++s) {
        const float4 b = part[(long)s * stride + e];
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    dw[e] = a;
}

void slab_reduce(const float *part, int nsplit, size_t n, float *dw, const float *dbpart, int k, float *dbias, hipStream_t st) {
    const long n4 = (long)(n / 4), k4 = k / 4;
    hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)((n4 + k4 + 63) / 64)), dim3(64), 0, st, (const float4 *)part, nsplit, n4, (float4 *)dw,
                       (const float4 *)dbpart, k4, (float4 *)dbias);
}

// MMDGAN_WINO2_WGRAD=0 keeps the stride-2 weight gradients on the direct implicit-GEMM kernel, =1 uses this one;
// MMDGAN_WINO2=2 (the parity tests) always.
bool wino2_wgrad_ok(const ConvDims &d) {
    static int en = -1;
    if (en < 0) { const char *e = getenv("MMDGAN_WINO2_WGRAD"); en = e ? atoi(e) : W2W_DEFAULT; }
    if ((!en && wino2_mode() < 2) || wino2_mode() == 0) return false;
    if (d.N == 1) return false;     // batch-1 = a power iteration's launch, on a chain concurrent with others: no workspace slabs
    // the batch-1 weight gradients of the power iteration (64 tiles) stay direct: 7.7 us against 7 + the 6 us reduction pass
    if (wino2_mode() < 2 && (long)d.N * (d.P / 2) * (d.Q / 2) < 256) return false;
    return d.R == 4 && d.stride == 2 && d.pad == 1 && d.H % 4 == 0 && d.W % 4 == 0 && d.C % wino2w::BC == 0 && d.K % wino2w::BK == 0;
}

// dbias (optional): the column sums of dy.  Returns 0 with *dbias_done = whether dbias was produced here (workspace path);
// otherwise the caller sums dy itself.
int wino2_wgrad(const ConvDims &d, const float *x, const float *dy, float *dw, float *dbias, bool *dbias_done, hipStream_t st) {
    const long T = (long)d.N * (d.P / 2) * (d.Q / 2);
    const int nst = (int)((T + wino2w::BT - 1) / wino2w::BT);
    const long base = (long)(d.C / wino2w::BC) * (d.K / wino2w::BK) * 4;
    int split = (int)(256 / base);                                  // one 8-wave workgroup per CU (140 KB of LDS), one round
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
    if (float *part = (float *)workspace_acquire(sizeof(float) * (n + d.K) * split, st)) {
        float *dbpart = part + n * split;
        if (dbias)
            hipLaunchKernelGGL((wino2_wgrad_kernel<true, true>), grid, dim3(wino2w::NT), wino2w::LDS_BYTES, st, d.N, d.H, d.W, d.C, d.K, x, dy, part,
                               dbpart, sps);
        else
            hipLaunchKernelGGL((wino2_wgrad_kernel<true, false>), grid, dim3(wino2w::NT), wino2w::LDS_BYTES, st, d.N, d.H, d.W, d.C, d.K, x, dy, part,
                               dbpart, sps);
        slab_reduce(part, split, n, dw, dbpart, dbias ? d.K : 0, dbias, st);
        if (dbias_done) *dbias_done = dbias != nullptr;
        return check_launch("conv2d_wgrad(winograd 2x2)");
    }
    if (split == 1) {                           // one slab: straight into dw
        hipLaunchKernelGGL((wino2_wgrad_kernel<true, false>), grid, dim3(wino2w::NT), wino2w::LDS_BYTES, st, d.N, d.H, d.W, d.C, d.K, x, dy, dw,
                           (float *)nullptr, sps);
        return check_launch("conv2d_wgrad(winograd 2x2)");
    }
    if (zero_output(dw, sizeof(float) * n, st) != hipSuccess) return check_launch("conv2d_wgrad memset");
    hipLaunchKernelGGL((wino2_wgrad_kernel<false, false>), grid, dim3(wino2w::NT), wino2w::LDS_BYTES, st, d.N, d.H, d.W, d.C, d.K, x, dy, dw,
                       (float *)nullptr, sps);
    return check_launch("conv2d_wgrad(winograd 2x2)");
}

}  // namespace mmdgan
