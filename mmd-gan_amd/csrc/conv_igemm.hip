// MFMA implicit-GEMM convolution kernels for gfx950: forward, input-gradient (also the
// transposed-conv forward) and weight-gradient, NHWC activations, HWIO kernels, fp32 in / fp32
// accumulate on v_mfma_f32_32x32x2_f32.
//
// Why fp32 MFMA: the parity bar is 1e-4 relative on conv activations (BASELINE.json), bf16/fp16
// inputs give ~1e-3, and gfx950 has no xf32/TF32 form; the f32-input MFMA is exact fp32
// (bitwise an fmaf chain) at the 157 TFLOP/s vector rate - that is the roofline these kernels
// are priced against.
//
// Structure (one 256-thread workgroup = 4 waves in a 2x2 grid over a BM x BN output tile):
//   * im2col is never materialised: each thread owns fixed rows of the A (activation) tile,
//     decomposes the pixel index once, and per K-stage (BK = 32 consecutive channels of ONE
//     filter tap, which is why C % 32 == 0 is required) turns the tap into an address or a zero.
//   * global -> registers (float4, coalesced along the channel axis) for stage s+1 is issued
//     before the MFMAs of stage s; registers -> LDS after them; LDS is double-buffered so there
//     is one barrier per stage.
//   * LDS tiles are stored [k][m] (+4 pad): both MFMA operands are then conflict-free
//     ds_read_b32 (lane l reads row k0 + l/32, column l%32), k-contiguous sources are
//     transposed on the way in (4 ds_write_b32, 2-way at worst = free), m-contiguous sources go
//     in as one ds_write_b128.
//   * the epilogue applies the spectral-norm scale (a device scalar: act_k/sigma), bias and the
//     activation - or, in backward form, the activation derivative of the layer below - so no
//     elementwise pass ever re-reads the output.  Lanes 0..31 of an accumulator register hold 32
//     consecutive output channels of one pixel: 128-byte contiguous stores.
//   * tiles: 128x128, 128x64 or 64x64 picked per layer so that the grid covers the 256 CUs;
//     reductions that are long but narrow (batch-1 spectral-norm convs, weight gradients) are
//     split over blockIdx.z and combined with fp32 atomics into a zeroed output.
#include "conv_internal.h"
#include "bufload.h"

namespace mmdgan {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 32;             // K depth of one LDS stage: 16 MFMA k-pairs between barriers
constexpr int KQ = BK / 4;         // float4 per k-contiguous row
// LDS row stride of a [k][m] tile.  m-contiguous sources are stored with ds_write_b128, so the row
// must stay 16-byte aligned (+4).  k-contiguous sources are transposed with 4 ds_write_b32 per
// float4: lanes of a wave differ in (row, k-chunk) and land on bank (k*LD + row) % 32, which is
// conflict-free exactly when LD == 1 (mod 8) - measured: +4 padding there cost a 4-way conflict
// on every store (SQ_LDS_BANK_CONFLICT 1.5e8 cycles per launch vs 0).
constexpr int ld_of(int cols, bool kc) { return kc ? cols + 1 : cols + 4; }

// KG = number of K-groups: with KG == 2 the workgroup has 8 waves and the two waves that share an
// output sub-tile each take every other MFMA k-pair of a stage (their accumulators are summed
// through LDS before the epilogue).  That puts 2 waves on every SIMD even when the grid is only one
// 64x64 tile per CU (M = 2048 layers at batch 128), where a lone wave cannot hide LDS / barrier
// latency behind its own MFMAs.
template <int BM, int BN, bool AKC = false, bool BKC = false, int KG = 1>
struct TileCfg {
    static constexpr int NT = 256 * KG;
    static constexpr int LDA = ld_of(BM, AKC), LDB = ld_of(BN, BKC);
    static constexpr int WM = BM / 2, WN = BN / 2;
    static constexpr int TM = WM / 32, TN = WN / 32;
    static constexpr int A_F4 = BM * BK / 4 / NT, B_F4 = BN * BK / 4 / NT;
    static constexpr int SMEM_FLOATS = 2 * BK * (LDA + LDB);
};

// k-contiguous source element f of a ROWS x BK tile: row = f/KQ, kq = f%KQ; transposing store
__device__ __forceinline__ void sts_kc(float *S, int LD, int f, const float4 &v) {
    const int row = f / KQ, k = (f % KQ) * 4;
    S[(k + 0) * LD + row] = v.x;
    S[(k + 1) * LD + row] = v.y;
    S[(k + 2) * LD + row] = v.z;
    S[(k + 3) * LD + row] = v.w;
}
// m-contiguous source element f of a BK x COLS tile: k = f/(COLS/4), c4 = f%(COLS/4)
template <int COLS>
__device__ __forceinline__ void sts_mc(float *S, int LD, int f, const float4 &v) {
    const int k = f / (COLS / 4), c4 = f % (COLS / 4);
    *reinterpret_cast<float4 *>(&S[k * LD + c4 * 4]) = v;
}

// Main loop, software-pipelined by hand.  Per K-stage s (BK = 32 deep, NKP = 16 MFMA k-pairs):
//   C(s)    MFMAs on LDS buffer s&1, operand fragments read PF k-pairs ahead of their use
//   W(s+1)  registers -> LDS buffer (s+1)&1 of the tile that was fetched during stage s-1
//   G(s+2)  global -> registers (buffer loads) of the tile after that
// CDNA issues a wave's instructions in order and a v_mfma_f32_32x32x2_f32 holds the matrix pipe
// for 64 cycles, so everything that is not an MFMA is cut into 2*(A_F4+B_F4) small pieces and one
// piece is placed behind the MFMAs of each k-pair, where it issues in the shadow of the pipe;
// __builtin_amdgcn_sched_barrier(0) after every k-pair keeps hipcc from re-clumping them (left
// alone it moves all address arithmetic / loads / LDS stores outside the MFMA run and the pipe
// idles: measured 111 us -> see DESIGN.md for the ablation).  One barrier per stage.
// P supplies load_a1/load_b1 (one float4 of the stage tile, global -> register) and says whether
// each operand is k-contiguous in memory (transposing LDS store) or m-contiguous (ds_write_b128).
// `smem` must be the kernel's own __shared__ array: buffers are addressed by arithmetic on that
// base (never through an array of pointers) so the LDS address space survives and ds_read/ds_write
// are emitted - a pointer table degraded every access to flat_load/flat_store.
template <int BM, int BN, int KG, class P>
__device__ __forceinline__ void mainloop(P &p, int s_begin, int s_end, float *smem,
                                         f32x16 (&acc)[TileCfg<BM, BN>::TM][TileCfg<BM, BN>::TN]) {
    using T = TileCfg<BM, BN, P::A_KC, P::B_KC, KG>;
    constexpr int A_BUF = BK * T::LDA, B_BUF = BK * T::LDB, B_OFF = 2 * A_BUF;
    constexpr int NKP = BK / 2 / KG, PF = 4, NP = T::A_F4 + T::B_F4, NT = T::NT;
    const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) & 3, kgroup = tid >> 8;
    const int wm = wave >> 1, wn = wave & 1;
    const int kh = lane >> 5, l31 = lane & 31;
    float4 ra[T::A_F4], rb[T::B_F4];
    if (s_begin >= s_end) return;
#define MMDGAN_W_A(DST, I) { if (P::A_KC) sts_kc(DST, T::LDA, tid + NT * (I), ra[I]); else sts_mc<BM>(DST, T::LDA, tid + NT * (I), ra[I]); }
#define MMDGAN_W_B(DST, I) { if (P::B_KC) sts_kc(DST, T::LDB, tid + NT * (I), rb[I]); else sts_mc<BN>(DST, T::LDB, tid + NT * (I), rb[I]); }
#pragma unroll
    for (int i = 0; i < T::A_F4; ++i) ra[i] = p.load_a1(s_begin, i);
#pragma unroll
    for (int i = 0; i < T::B_F4; ++i) rb[i] = p.load_b1(s_begin, i);
#pragma unroll
    for (int i = 0; i < T::A_F4; ++i) MMDGAN_W_A(smem, i)
#pragma unroll
    for (int i = 0; i < T::B_F4; ++i) MMDGAN_W_B(smem + B_OFF, i)
#pragma unroll
    for (int i = 0; i < T::A_F4; ++i) ra[i] = p.load_a1(s_begin + 1, i);
#pragma unroll
    for (int i = 0; i < T::B_F4; ++i) rb[i] = p.load_b1(s_begin + 1, i);
    __syncthreads();
    // K-group g takes the k-pairs g, g+KG, g+2KG ... of every stage
    const float *ap0 = smem + (kh + 2 * kgroup) * T::LDA + wm * T::WM + l31;
    const float *bp0 = smem + B_OFF + (kh + 2 * kgroup) * T::LDB + wn * T::WN + l31;
    for (int s = s_begin; s < s_end; ++s) {
        const int buf = (s - s_begin) & 1;
        const float *ap = ap0 + buf * A_BUF, *bp = bp0 + buf * B_BUF;
        float *An = smem + (buf ^ 1) * A_BUF, *Bn = smem + B_OFF + (buf ^ 1) * B_BUF;
        float fa[NKP][T::TM], fb[NKP][T::TN];
#define MMDGAN_FRAG(J)                                                                                  \
    {                                                                                                   \
        _Pragma("unroll") for (int mi = 0; mi < T::TM; ++mi) fa[J][mi] = ap[2 * KG * (J) * T::LDA + mi * 32]; \
        _Pragma("unroll") for (int ni = 0; ni < T::TN; ++ni) fb[J][ni] = bp[2 * KG * (J) * T::LDB + ni * 32]; \
    }
#pragma unroll
        for (int j = 0; j < PF; ++j) MMDGAN_FRAG(j)
#pragma unroll
        for (int j = 0; j < NKP; ++j) {
            if (j + PF < NKP) MMDGAN_FRAG(j + PF)
#pragma unroll
            for (int mi = 0; mi < T::TM; ++mi)
#pragma unroll
                for (int ni = 0; ni < T::TN; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[j][mi], fb[j][ni], acc[mi][ni], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 2 * NP; ++q) {
                if ((q * NKP) / (2 * NP) != j) continue;
                // store register i to LDS (tile s+1), then immediately refill it from global (tile s+2):
                // every load has a full stage of MFMAs between its issue and the ds_write that consumes it
                const int i = q >> 1;
                if ((q & 1) == 0) {
                    if (i < T::A_F4) MMDGAN_W_A(An, i)
                    else MMDGAN_W_B(Bn, i - T::A_F4)
                } else {
                    if (i < T::A_F4) ra[i] = p.load_a1(s + 2, i);
                    else rb[i - T::A_F4] = p.load_b1(s + 2, i - T::A_F4);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#undef MMDGAN_FRAG
        __syncthreads();
    }
#undef MMDGAN_W_A
#undef MMDGAN_W_B
}

// Epilogue.  C/D layout of v_mfma_f32_32x32x2_f32: col = lane&31, row = (reg&3) + 8*(reg>>2) +
// 4*(lane>>5) - a lane holds single floats of 16 different rows, so writing straight from the
// accumulators means 64 scalar stores per lane and one address computation per element (measured:
// 10 us of a 108 us launch).  Instead each wave transposes a 32-row slab of its tile through LDS
// (the main loop's buffers are free after its last barrier) and every lane then owns 4 consecutive
// channels of one pixel: one row-offset computation, float4 bias / dact loads, one float4 store,
// 16 lanes x 16 B = 256 contiguous bytes per row.
// ROWOFF(row) -> element offset of output row `row` (tile-local 0..BM-1) or -1 if out of range.
template <int BM, int BN, bool ATOMIC, int KG, class RowOff>
__device__ __forceinline__ void epilogue_store(float *smem, f32x16 (&acc)[TileCfg<BM, BN>::TM][TileCfg<BM, BN>::TN],
                                               RowOff rowoff, int ch0, const ConvEpilogue &ep, float sc, float *out,
                                               bool add_bias, bool raw) {
    using T = TileCfg<BM, BN>;
    bool active = true;
    if (KG > 1) {
        // sum the K-groups: group 1 parks its accumulators in LDS (lane-major: conflict-free), group 0 adds
        const int kgroup = threadIdx.x >> 8, w4 = (threadIdx.x >> 6) & 3, ln = threadIdx.x & 63;
        float *park = smem + w4 * (T::TM * T::TN * 16 * 64);
        if (kgroup == 1) {
#pragma unroll
            for (int mi = 0; mi < T::TM; ++mi)
#pragma unroll
                for (int ni = 0; ni < T::TN; ++ni)
#pragma unroll
                    for (int r = 0; r < 16; ++r) park[((mi * T::TN + ni) * 16 + r) * 64 + ln] = acc[mi][ni][r];
        }
        __syncthreads();
        if (kgroup == 0) {
#pragma unroll
            for (int mi = 0; mi < T::TM; ++mi)
#pragma unroll
                for (int ni = 0; ni < T::TN; ++ni)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mi][ni][r] += park[((mi * T::TN + ni) * 16 + r) * 64 + ln];
        }
        __syncthreads();
        active = kgroup == 0;
    }
    constexpr int LDE = T::WN + 4;                       // slab row stride (16-byte aligned rows)
    constexpr int LPR = T::WN / 4;                       // lanes per row
    constexpr int RPP = 64 / LPR;                        // rows per pass
    const int lane = threadIdx.x & 63, wave = (threadIdx.x >> 6) & 3, wm = wave >> 1, wn = wave & 1;
    const int kh = lane >> 5, l31 = lane & 31;
    float *slab = smem + wave * (32 * LDE);
    const int c4 = lane % LPR, rsub = lane / LPR;
    const int ch = ch0 + wn * T::WN + c4 * 4;
    if (ATOMIC) {
        if (!active) return;
        // split reductions (a compile-time variant: keeping both paths in one kernel cost scratch): accumulate straight from the MFMA layout - lanes 0..31 of a register are
        // 32 consecutive channels of one row, so each atomic instruction covers two 128-byte runs
        // (going through the float4 path made every lane issue 4 strided atomics: measured slower)
#pragma unroll
        for (int mi = 0; mi < T::TM; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long base = rowoff(wm * T::WM + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh);
#pragma unroll
                for (int ni = 0; ni < T::TN; ++ni) {
                    const int cl = wn * T::WN + ni * 32 + l31;
                    float v = acc[mi][ni][r] * sc;
                    if (ep.bias && add_bias) v += ep.bias[ch0 + cl];
                    if (base >= 0) atomicAdd(out + base + cl, v);
                }
            }
        return;
    }
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ep.bias && add_bias) bv = *reinterpret_cast<const float4 *>(ep.bias + ch);
#pragma unroll
    for (int mi = 0; mi < T::TM; ++mi) {
        if (active) {
#pragma unroll
            for (int ni = 0; ni < T::TN; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    slab[((r & 3) + 8 * (r >> 2) + 4 * kh) * LDE + ni * 32 + l31] = acc[mi][ni][r];
        }
        __syncthreads();
#pragma unroll
        for (int ps = 0; ps < 32 / RPP; ++ps) {
            const int rl = ps * RPP + rsub;
            const long base = active ? rowoff(wm * T::WM + mi * 32 + rl) : -1;
            if (base >= 0) {
                float4 v = *reinterpret_cast<const float4 *>(slab + rl * LDE + c4 * 4);
                const long o = base + wn * T::WN + c4 * 4;
                v.x = v.x * sc + bv.x; v.y = v.y * sc + bv.y; v.z = v.z * sc + bv.z; v.w = v.w * sc + bv.w;
                {
                    if (!raw) {
                        if (ep.dact) {
                            const float4 y = *reinterpret_cast<const float4 *>(ep.dact + ep.dact_index(o));
                            v.x *= act_bwd_from_out(y.x, ep.act); v.y *= act_bwd_from_out(y.y, ep.act);
                            v.z *= act_bwd_from_out(y.z, ep.act); v.w *= act_bwd_from_out(y.w, ep.act);
                        } else {
                            v.x = act_fwd(v.x, ep.act); v.y = act_fwd(v.y, ep.act);
                            v.z = act_fwd(v.z, ep.act); v.w = act_fwd(v.w, ep.act);
                        }
                    }
                    *reinterpret_cast<float4 *>(out + o) = raw ? v : ep.add4(v, o);
                }
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// forward: y[m = (n,p,q)][k] = sum_{tap,c} x[n, p*s-pad+r, q*s-pad+t, c] * w[tap][c][k]
// ------------------------------------------------------------------------------------------------
// Address generation is kept off the vector ALU as far as possible: VALU instructions issue through
// the same port as the MFMAs, integer multiplies are quarter rate, and with four waves per SIMD the
// gather arithmetic of the first version (two 64-bit mads, four compares per load; divisions in the
// weight-gradient kernel) took a third to four fifths as many issue cycles as the MFMAs themselves
// (rocprofv3: SQ_VALU_MFMA_BUSY 54 % on D l7 wgrad).  Now every row of the A tile is decomposed ONCE
// into a byte offset plus a bit mask over the filter taps ("does this tap land inside the image"),
// and a K-stage contributes only wave-uniform scalars: one v_add, one bit test, one select per load.
__device__ __forceinline__ int mad24(int a, int b, int c) { return __mul24(a, b) + c; }

template <int BM, int BN, int KG>
struct FwdProblem {
    static constexpr bool A_KC = true, B_KC = false;
    using T = TileCfg<BM, BN, false, false, KG>;
    __amdgpu_buffer_rsrc_t rx, rw;
    int C, R, RR, WC4, stageK4;
    unsigned apix[T::A_F4];           // byte offset of (n, p*s-pad, q*s-pad, kq*4): tap (0,0), may wrap below 0
    unsigned amask[T::A_F4];          // bit r*R+t: tap (r,t) of this output pixel reads inside the image
    unsigned bbase[T::B_F4];
    __device__ void init(const ConvDims &d, const float *x_, const float *w_, int m0, int n0, long M) {
        C = d.C; R = d.R; RR = d.R * d.R; WC4 = d.W * d.C * 4; stageK4 = BK * d.K * 4;
        rx = make_rsrc(x_, (long)d.N * d.H * d.W * d.C * 4);
        rw = make_rsrc(w_, (long)d.R * d.R * d.C * d.K * 4);
#pragma unroll
        for (int i = 0; i < T::A_F4; ++i) {
            const int f = threadIdx.x + T::NT * i;
            const long m = (long)m0 + f / KQ;
            const bool ok = m < M;
            const long mm = ok ? m : 0;
            const int q = mm % d.Q;
            const long t = mm / d.Q;
            const int p = t % d.P;
            const int n = t / d.P;
            const int h0 = p * d.stride - d.pad, w0 = q * d.stride - d.pad;
            apix[i] = (unsigned)((((n * d.H + h0) * d.W + w0) * d.C + (f % KQ) * 4) * 4);
            unsigned cols = 0, mask = 0;
            for (int tt = 0; tt < d.R; ++tt) cols |= (unsigned)(w0 + tt >= 0 && w0 + tt < d.W) << tt;
            for (int r = 0; r < d.R; ++r) mask |= (h0 + r >= 0 && h0 + r < d.H) ? cols << (r * d.R) : 0u;
            amask[i] = ok ? mask : 0u;
        }
#pragma unroll
        for (int i = 0; i < T::B_F4; ++i) {
            const int f = threadIdx.x + T::NT * i;
            bbase[i] = (unsigned)((((f / (BN / 4)) * d.K) + n0 + (f % (BN / 4)) * 4) * 4);
        }
    }
    __device__ __forceinline__ float4 load_a1(int s, int i) const {
        const int k0 = s * BK;
        const int tap = k0 / C, c0 = k0 - tap * C;           // wave-uniform: scalar ALU
        const int r = tap / R, t = tap - r * R;
        // readfirstlane pins the wave-uniform part in an SGPR (left alone hipcc folds it back into a
        // per-lane 64-bit mad)
        const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane(r * WC4 + (t * C + c0) * 4);
        const unsigned bit = (unsigned)__builtin_amdgcn_readfirstlane(tap < RR ? 1 << tap : 0);
        const bool ok = (amask[i] & bit) != 0;
        return bufld4(rx, ok ? apix[i] + soff : kOOB);
    }
    __device__ __forceinline__ float4 load_b1(int s, int i) const {
        return bufld4(rw, bbase[i] + (unsigned)__builtin_amdgcn_readfirstlane(s * stageK4));  // beyond the last stage: past the buffer -> 0
    }
};

template <int BM, int BN, bool SPLIT, int KG = 1>
__global__ __launch_bounds__(256 * KG) void igemm_fwd_kernel(ConvDims d, ConvEpilogue ep, const float *__restrict__ x,
                                                        const float *__restrict__ w, float *__restrict__ y,
                                                        int stages_per_split) {
    using T = TileCfg<BM, BN>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const long M = (long)d.N * d.P * d.Q;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int nstages = d.R * d.R * d.C / BK;
    const int s0 = blockIdx.z * stages_per_split;
    const int s1 = min(nstages, s0 + stages_per_split);
    f32x16 acc[T::TM][T::TN];
#pragma unroll
    for (int mi = 0; mi < T::TM; ++mi)
#pragma unroll
        for (int ni = 0; ni < T::TN; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
    FwdProblem<BM, BN, KG> p;
    p.init(d, x, w, m0, n0, M);
    mainloop<BM, BN, KG>(p, s0, s1, smem, acc);
    const float sc = ep.scale ? ep.scale[0] : 1.f;
    const int Kc = d.K;
    auto rowoff = [=](int row) -> long {
        const long m = (long)m0 + row;
        return m < M ? m * Kc + n0 : -1;
    };
    epilogue_store<BM, BN, SPLIT, KG>(smem, acc, rowoff, n0, ep, sc, y, !SPLIT || blockIdx.z == 0, false);
}

// ------------------------------------------------------------------------------------------------
// input gradient / transposed conv: dx[n,h,w,c] = sum_{r,t,k} dy[n,p,q,k] * w[r,t,c,k],
// p*stride - pad + r = h.  For stride s the output pixels split into s*s parity phases, each a
// dense GEMM over the (R/s)^2 taps that can reach it (blockIdx.z = phase).
// ------------------------------------------------------------------------------------------------
template <int BM, int BN, int KG>
struct DgradProblem {
    static constexpr bool A_KC = true, B_KC = true;
    using T = TileCfg<BM, BN, false, false, KG>;
    __amdgpu_buffer_rsrc_t rdy, rw;
    int K, Q, R, C, stride, TT, rbase, tbase;
    unsigned apix[T::A_F4];           // byte offset of dy(n, hh+pbase, ww+qbase, kq*4): tap (0,0)
    unsigned amask[T::A_F4];          // bit jr*TT+jt: dy(n, hh+pbase-jr, ww+qbase-jt) exists
    unsigned bbase[T::B_F4];
    __device__ void init(const ConvDims &d, const float *dy_, const float *w_, int m0, int n0, int ph, int pw, int Hh,
                         int Ww, long M) {
        K = d.K; Q = d.Q; R = d.R; C = d.C; stride = d.stride;
        rdy = make_rsrc(dy_, (long)d.N * d.P * d.Q * d.K * 4);
        rw = make_rsrc(w_, (long)d.R * d.R * d.C * d.K * 4);
        TT = d.R / d.stride;
        rbase = (ph + d.pad) % d.stride; tbase = (pw + d.pad) % d.stride;
        const int pbase = (ph + d.pad) / d.stride, qbase = (pw + d.pad) / d.stride;
#pragma unroll
        for (int i = 0; i < T::A_F4; ++i) {
            const int f = threadIdx.x + T::NT * i;
            const long m = (long)m0 + f / KQ;
            const bool ok = m < M;
            const long mm = ok ? m : 0;
            const int qq = (int)(mm % Ww) + qbase;
            const long t = mm / Ww;
            const int pp = (int)(t % Hh) + pbase;
            const int n = t / Hh;
            apix[i] = (unsigned)((((n * d.P + pp) * d.Q + qq) * d.K + (f % KQ) * 4) * 4);
            unsigned cols = 0, mask = 0;
            for (int jt = 0; jt < TT; ++jt) cols |= (unsigned)(qq - jt >= 0 && qq - jt < d.Q) << jt;
            for (int jr = 0; jr < TT; ++jr) mask |= (pp - jr >= 0 && pp - jr < d.P) ? cols << (jr * TT) : 0u;
            amask[i] = ok ? mask : 0u;
        }
#pragma unroll
        for (int i = 0; i < T::B_F4; ++i) {
            const int f = threadIdx.x + T::NT * i;
            bbase[i] = (unsigned)(((long)(n0 + f / KQ) * d.K + (f % KQ) * 4) * 4);     // + tap*C*K + co0 per stage
        }
    }
    __device__ __forceinline__ float4 load_a1(int s, int i) const {
        const int k0 = s * BK;
        const int tap = k0 / K, co0 = k0 - tap * K;          // wave-uniform: scalar ALU
        const int jr = tap / TT, jt = tap - jr * TT;
        const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane((co0 - (jr * Q + jt) * K) * 4);
        const unsigned bit = (unsigned)__builtin_amdgcn_readfirstlane(jr < TT ? 1 << tap : 0);
        const bool ok = (amask[i] & bit) != 0;
        return bufld4(rdy, ok ? apix[i] + soff : kOOB);
    }
    __device__ __forceinline__ float4 load_b1(int s, int i) const {
        const int k0 = s * BK;
        const int tap = k0 / K, co0 = k0 - tap * K;
        const int jr = tap / TT, jt = tap - jr * TT;
        const int r = rbase + jr * stride, t = tbase + jt * stride;
        const unsigned off = (unsigned)__builtin_amdgcn_readfirstlane(jr < TT ? (((r * R + t) * C) * K + co0) * 4 : (int)kOOB);
        return bufld4(rw, bbase[i] + off);                   // bbase < 2^31: adding kOOB stays out of range
    }
};

template <int BM, int BN, bool SPLIT, int KG = 1>
__global__ __launch_bounds__(256 * KG) void igemm_dgrad_kernel(ConvDims d, ConvEpilogue ep, const float *__restrict__ dy,
                                                          const float *__restrict__ w, float *__restrict__ dx,
                                                          int nsplit, int stages_per_split) {
    using T = TileCfg<BM, BN>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int phase = blockIdx.z / nsplit, split = blockIdx.z - phase * nsplit;
    const int ph = phase / d.stride, pw = phase - ph * d.stride;
    const int Hh = (d.H - ph + d.stride - 1) / d.stride, Ww = (d.W - pw + d.stride - 1) / d.stride;
    const long M = (long)d.N * Hh * Ww;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    if (m0 >= M) return;
    const int TT = d.R / d.stride;
    const int nstages = TT * TT * d.K / BK;
    const int s0 = split * stages_per_split;
    const int s1 = min(nstages, s0 + stages_per_split);
    f32x16 acc[T::TM][T::TN];
#pragma unroll
    for (int mi = 0; mi < T::TM; ++mi)
#pragma unroll
        for (int ni = 0; ni < T::TN; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
    DgradProblem<BM, BN, KG> p;
    p.init(d, dy, w, m0, n0, ph, pw, Hh, Ww, M);
    mainloop<BM, BN, KG>(p, s0, s1, smem, acc);
    const float sc = ep.scale ? ep.scale[0] : 1.f;
    const int Mi = (int)M;
    auto rowoff = [=](int row) -> long {
        const int m = m0 + row;
        if (m >= Mi) return -1;
        const int ww = m % Ww, t = m / Ww;
        const int hh = t % Hh, n = t / Hh;
        return (((long)n * d.H + (hh * d.stride + ph)) * d.W + (ww * d.stride + pw)) * d.C + n0;
    };
    epilogue_store<BM, BN, SPLIT, KG>(smem, acc, rowoff, n0, ep, sc, dx, !SPLIT || split == 0, false);
}

// ------------------------------------------------------------------------------------------------
// weight gradient: dw[(tap,c)][k] = sum_{pixels m} x[m shifted by tap][c] * dy[m][k]
// GEMM rows i = (tap, c) (a BM-row tile sits inside one tap: C % BM == 0), reduction over pixels.
// ------------------------------------------------------------------------------------------------
// The reduction index of the weight gradient is the output pixel m = (n,p,q), 32 consecutive pixels
// per stage, and every thread keeps the (q, p, image offset) of its rows as running state that is
// advanced by 32 pixels per call with compare/select carries - no division or 32-bit multiply in the
// loop (load_a1 / load_b1 must therefore be called exactly once per row and stage, in stage order,
// which is how the main loop uses them; their `s` argument is ignored).  Pixels beyond the batch
// walk off the end of the buffer and read 0.
template <int BM, int BN, int KG>
struct WgradProblem {
    static constexpr bool A_KC = false, B_KC = false;
    using T = TileCfg<BM, BN, false, false, KG>;
    __amdgpu_buffer_rsrc_t rx, rdy;
    // running state per A row, all in BYTES so that the loop has no multiply at all:
    //   hq = w * C*4   (w = q*stride - pad + t, negative / >= W*C*4 when the tap leaves the row)
    //   hp = h * W*C*4 (h = p*stride - pad + r)
    //   noff = n * H*W*C*4 + channel offset;   address = noff + hp + hq
    int hq[T::A_F4], hp[T::A_F4];
    unsigned noff[T::A_F4], boff[T::B_F4];
    int dhq, qwrap, qlim, dhp, prow, pwrap, plim;
    unsigned wlim, hlim, HWC4, dnoff, stageK4;
    // bias gradient = column sums of dy: the row-tile-0 workgroups add up the dy tiles they stream anyway
    bool want_bsum;
    int bsum_stages;                  // stages still to count (the two prefetched past the split are not this block's)
    float4 bsum;
    __device__ void init(const ConvDims &d, const float *x_, const float *dy_, int i0, int n0, int s0, int s1, bool sum_dy) {
        want_bsum = sum_dy; bsum_stages = (s1 - s0) * T::B_F4; bsum = make_float4(0.f, 0.f, 0.f, 0.f);
        rx = make_rsrc(x_, (long)d.N * d.H * d.W * d.C * 4);
        rdy = make_rsrc(dy_, (long)d.N * d.P * d.Q * d.K * 4);
        const int tap = i0 / d.C, c0 = i0 - tap * d.C;
        const int r = tap / d.R, t = tap - r * d.R;
        const int C4 = d.C * 4, WC4 = d.W * C4;
        const int hoff = r - d.pad, woff = t - d.pad;
        HWC4 = (unsigned)(d.H * WC4);
        wlim = (unsigned)WC4; hlim = HWC4;
        dhq = (BK % d.Q) * d.stride * C4;               // q += BK % Q
        qwrap = d.Q * d.stride * C4;                    // q -= Q on carry
        qlim = qwrap + woff * C4;                       // q >= Q  <=>  hq >= qlim
        prow = d.stride * WC4;                          // p += 1
        dhp = ((BK / d.Q) % d.P) * prow;
        pwrap = d.P * prow;
        plim = pwrap + hoff * WC4;
        dnoff = (unsigned)(BK / (d.P * d.Q)) * HWC4;
        stageK4 = (unsigned)(BK * d.K * 4);
#pragma unroll
        for (int i = 0; i < T::A_F4; ++i) {
            const int f = threadIdx.x + T::NT * i;
            const int m = s0 * BK + f / (BM / 4);
            const int q = m % d.Q, u = m / d.Q;
            const int p = u % d.P;
            hq[i] = (q * d.stride + woff) * C4;
            hp[i] = (p * d.stride + hoff) * WC4;
            noff[i] = (unsigned)(u / d.P) * HWC4 + (unsigned)((c0 + (f % (BM / 4)) * 4) * 4);
        }
#pragma unroll
        for (int i = 0; i < T::B_F4; ++i) {
            const int f = threadIdx.x + T::NT * i;
            boff[i] = (unsigned)((((s0 * BK + f / (BN / 4)) * d.K) + n0 + (f % (BN / 4)) * 4) * 4);
        }
    }
    __device__ __forceinline__ float4 load_a1(int, int i) {
        const bool ok = (unsigned)hp[i] < hlim && (unsigned)hq[i] < wlim;
        const float4 v = bufld4(rx, ok ? noff[i] + (unsigned)hp[i] + (unsigned)hq[i] : kOOB);
        hq[i] += dhq;
        const bool cq = hq[i] >= qlim;
        hq[i] -= cq ? qwrap : 0;
        hp[i] += dhp + (cq ? prow : 0);
        const bool cp = hp[i] >= plim;
        hp[i] -= cp ? pwrap : 0;
        noff[i] += dnoff + (cp ? HWC4 : 0u);
        return v;
    }
    __device__ __forceinline__ float4 load_b1(int, int i) {
        const float4 v = bufld4(rdy, boff[i]);
        boff[i] += stageK4;
        if (want_bsum) {                               // wave-uniform
            if (bsum_stages > 0) { bsum.x += v.x; bsum.y += v.y; bsum.z += v.z; bsum.w += v.w; }
            --bsum_stages;
        }
        return v;
    }
};

template <int BM, int BN, bool SPLIT, int KG = 1>
__global__ __launch_bounds__(256 * KG) void igemm_wgrad_kernel(ConvDims d, const float *__restrict__ x,
                                                          const float *__restrict__ dy, float *__restrict__ dw,
                                                          int stages_per_split, float *__restrict__ dbias) {
    using T = TileCfg<BM, BN>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const long M = (long)d.N * d.P * d.Q;
    const int i0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int nstages = (int)((M + BK - 1) / BK);
    const int s0 = blockIdx.z * stages_per_split;
    const int s1 = min(nstages, s0 + stages_per_split);
    f32x16 acc[T::TM][T::TN];
#pragma unroll
    for (int mi = 0; mi < T::TM; ++mi)
#pragma unroll
        for (int ni = 0; ni < T::TN; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
    WgradProblem<BM, BN, KG> p;
    const bool sum_dy = dbias != nullptr && blockIdx.x == 0;
    p.init(d, x, dy, i0, n0, s0, s1, sum_dy);
    mainloop<BM, BN, KG>(p, s0, s1, smem, acc);
    if (sum_dy) {                                      // threads tid % (BN/4) share a channel quad: combine through LDS
        constexpr int NTH = 256 * KG, Q = BN / 4;
        *reinterpret_cast<float4 *>(smem + threadIdx.x * 4) = p.bsum;
        __syncthreads();
        if (threadIdx.x < BN) {
            const int c4 = threadIdx.x >> 2, e = threadIdx.x & 3;
            float t = 0.f;
            for (int g = 0; g < NTH / Q; ++g) t += smem[(g * Q + c4) * 4 + e];
            atomicAdd(dbias + n0 + threadIdx.x, t);
        }
        __syncthreads();
    }
    const int Kc = d.K;
    auto rowoff = [=](int row) -> long { return (long)(i0 + row) * Kc + n0; };
    const ConvEpilogue none{nullptr, nullptr, nullptr, MMDGAN_ACT_LINEAR, kNoWrap, 0, false};
    epilogue_store<BM, BN, SPLIT, KG>(smem, acc, rowoff, n0, none, 1.f, dw, false, true);
}

// ------------------------------------------------------------------------------------------------
// host side: eligibility, tile and split selection
// ------------------------------------------------------------------------------------------------
constexpr int kTargetBlocks = 512;      // two workgroups per CU: a second wave per SIMD hides LDS/barrier latency

template <int BM, int BN>
constexpr size_t smem_bytes() { return sizeof(float) * TileCfg<BM, BN>::SMEM_FLOATS; }

// the 128x128 tile needs 67.6 KB of LDS (> the 64 KB default cap): raise the cap once per kernel
static void raise_lds_caps() {
    static bool done = false;
    if (done) return;
    done = true;
    const int cap = (int)smem_bytes<128, 128>();
    (void)hipFuncSetAttribute((const void *)igemm_fwd_kernel<128, 128, false>, hipFuncAttributeMaxDynamicSharedMemorySize, cap);
    (void)hipFuncSetAttribute((const void *)igemm_fwd_kernel<128, 128, true>, hipFuncAttributeMaxDynamicSharedMemorySize, cap);
    (void)hipFuncSetAttribute((const void *)igemm_dgrad_kernel<128, 128, false>, hipFuncAttributeMaxDynamicSharedMemorySize, cap);
    (void)hipFuncSetAttribute((const void *)igemm_dgrad_kernel<128, 128, true>, hipFuncAttributeMaxDynamicSharedMemorySize, cap);
    (void)hipFuncSetAttribute((const void *)igemm_wgrad_kernel<128, 128, false>, hipFuncAttributeMaxDynamicSharedMemorySize, cap);
    (void)hipFuncSetAttribute((const void *)igemm_wgrad_kernel<128, 128, true>, hipFuncAttributeMaxDynamicSharedMemorySize, cap);
}

bool igemm_fwd_ok(const ConvDims &d) { return d.C % BK == 0 && d.K % 64 == 0 && d.R * d.R * d.C >= 64 && d.R <= 5; }   // 25 tap bits
bool igemm_dgrad_ok(const ConvDims &d) { return d.K % BK == 0 && d.C % 64 == 0 && d.R % d.stride == 0 && d.R / d.stride <= 5; }
bool igemm_wgrad_ok(const ConvDims &d) { return d.C % 64 == 0 && d.K % 64 == 0; }

static void pick_tile(long M, int N, int &bm, int &bn) {
    // measured on every layer shape of the step (round 1, each tile forced in turn): the 64x64
    // tile at 4 workgroups per CU beats 128x64 / 128x128 at 1-2 per CU by 3-12 % whenever it gives no
    // more than a few rounds of workgroups - four independent barrier domains per CU hide each other's
    // stalls better than one large tile reuses LDS traffic.  Larger tiles only for very large grids.
    const long t64 = ((M + 63) / 64) * (N / 64);
    if (t64 <= 16 * 1024 || N % 128) { bm = 64; bn = 64; return; }
    if (t64 <= 64 * 1024) { bm = 128; bn = 64; return; }
    bm = 128; bn = 128;
}

static int pick_split(long tiles, int nstages, bool allowed) {
    if (!allowed || tiles >= kTargetBlocks / 2) return 1;
    int s = (int)(kTargetBlocks / tiles);
    const int maxs = nstages / 4;                // keep >= 4 stages (128 deep) per split
    if (s > maxs) s = maxs;
    return s < 1 ? 1 : s;
}

int igemm_fwd(const ConvDims &d, const ConvEpilogue &ep, const float *x, const float *w, float *y, hipStream_t st) {
    raise_lds_caps();
    const long M = (long)d.N * d.P * d.Q;
    int bm, bn;
    pick_tile(M, d.K, bm, bn);
    const int nstages = d.R * d.R * d.C / BK;
    const long tiles = ((M + bm - 1) / bm) * (d.K / bn);
    // with caller-side zeroing only the batch-1 (spectral-norm power iteration) launches may split
    // (an addend rides on the plain store of an unsplit launch only: a split one zeroes and accumulates into the output)
    const bool may_split = ep.act == MMDGAN_ACT_LINEAR && !ep.dact && !ep.addend && (!outputs_prezeroed() || d.N == 1 || ep.out_zeroed);
    int split = pick_split(tiles, nstages, may_split);
    int sps = (nstages + split - 1) / split;
    split = (nstages + sps - 1) / sps;
    if (split > 1 && !ep.out_zeroed && zero_output(y, sizeof(float) * M * d.K, st) != hipSuccess) return check_launch("conv2d_fwd memset");
    const dim3 grid((unsigned)((M + bm - 1) / bm), d.K / bn, split);
    if (bm == 128 && bn == 128) { if (split > 1) hipLaunchKernelGGL((igemm_fwd_kernel<128, 128, true>), grid, dim3(256), (smem_bytes<128, 128>()), st, d, ep, x, w, y, sps); else hipLaunchKernelGGL((igemm_fwd_kernel<128, 128, false>), grid, dim3(256), (smem_bytes<128, 128>()), st, d, ep, x, w, y, sps); }
    else if (bm == 128) { if (split > 1) hipLaunchKernelGGL((igemm_fwd_kernel<128, 64, true>), grid, dim3(256), (smem_bytes<128, 64>()), st, d, ep, x, w, y, sps); else hipLaunchKernelGGL((igemm_fwd_kernel<128, 64, false>), grid, dim3(256), (smem_bytes<128, 64>()), st, d, ep, x, w, y, sps); }
    else if ((long)grid.x * grid.y * grid.z < kTargetBlocks) {
        // a single 64x64 workgroup per CU: use the 8-wave K-group variant (2 waves per SIMD)
        if (split > 1) hipLaunchKernelGGL((igemm_fwd_kernel<64, 64, true, 2>), grid, dim3(512), (smem_bytes<64, 64>()), st, d, ep, x, w, y, sps);
        else hipLaunchKernelGGL((igemm_fwd_kernel<64, 64, false, 2>), grid, dim3(512), (smem_bytes<64, 64>()), st, d, ep, x, w, y, sps);
    }
    else { if (split > 1) hipLaunchKernelGGL((igemm_fwd_kernel<64, 64, true>), grid, dim3(256), (smem_bytes<64, 64>()), st, d, ep, x, w, y, sps); else hipLaunchKernelGGL((igemm_fwd_kernel<64, 64, false>), grid, dim3(256), (smem_bytes<64, 64>()), st, d, ep, x, w, y, sps); }
    if (split == 1) addend_applied();
    return check_launch("conv2d_fwd(igemm)");
}

int igemm_dgrad(const ConvDims &d, const ConvEpilogue &ep, const float *dy, const float *w, float *dx, hipStream_t st) {
    raise_lds_caps();
    const int s = d.stride;
    const int Hh = (d.H + s - 1) / s, Ww = (d.W + s - 1) / s;       // largest phase
    const long M = (long)d.N * Hh * Ww;
    int bm, bn;
    pick_tile(M * s * s, d.C, bm, bn);
    // pick_tile counted all phases together; recompute with per-phase rows
    const long tiles = ((M + bm - 1) / bm) * (d.C / bn) * s * s;
    const int TT = d.R / s;
    const int nstages = TT * TT * d.K / BK;
    // (an addend rides on the plain store of an unsplit launch only: a split one zeroes and accumulates into the output)
    const bool may_split = ep.act == MMDGAN_ACT_LINEAR && !ep.dact && !ep.addend && (!outputs_prezeroed() || d.N == 1 || ep.out_zeroed);
    int split = pick_split(tiles, nstages, may_split);
    int sps = (nstages + split - 1) / split;
    split = (nstages + sps - 1) / sps;
    if (split > 1 && !ep.out_zeroed && zero_output(dx, sizeof(float) * (long)d.N * d.H * d.W * d.C, st) != hipSuccess)
        return check_launch("conv2d_dgrad memset");
    const dim3 grid((unsigned)((M + bm - 1) / bm), d.C / bn, s * s * split);
    if (bm == 128 && bn == 128) { if (split > 1) hipLaunchKernelGGL((igemm_dgrad_kernel<128, 128, true>), grid, dim3(256), (smem_bytes<128, 128>()), st, d, ep, dy, w, dx, split, sps); else hipLaunchKernelGGL((igemm_dgrad_kernel<128, 128, false>), grid, dim3(256), (smem_bytes<128, 128>()), st, d, ep, dy, w, dx, split, sps); }
    else if (bm == 128) { if (split > 1) hipLaunchKernelGGL((igemm_dgrad_kernel<128, 64, true>), grid, dim3(256), (smem_bytes<128, 64>()), st, d, ep, dy, w, dx, split, sps); else hipLaunchKernelGGL((igemm_dgrad_kernel<128, 64, false>), grid, dim3(256), (smem_bytes<128, 64>()), st, d, ep, dy, w, dx, split, sps); }
    else if ((long)grid.x * grid.y * grid.z < kTargetBlocks) {
        // a single 64x64 workgroup per CU: use the 8-wave K-group variant (2 waves per SIMD)
        if (split > 1) hipLaunchKernelGGL((igemm_dgrad_kernel<64, 64, true, 2>), grid, dim3(512), (smem_bytes<64, 64>()), st, d, ep, dy, w, dx, split, sps);
        else hipLaunchKernelGGL((igemm_dgrad_kernel<64, 64, false, 2>), grid, dim3(512), (smem_bytes<64, 64>()), st, d, ep, dy, w, dx, split, sps);
    }
    else { if (split > 1) hipLaunchKernelGGL((igemm_dgrad_kernel<64, 64, true>), grid, dim3(256), (smem_bytes<64, 64>()), st, d, ep, dy, w, dx, split, sps); else hipLaunchKernelGGL((igemm_dgrad_kernel<64, 64, false>), grid, dim3(256), (smem_bytes<64, 64>()), st, d, ep, dy, w, dx, split, sps); }
    if (split == 1) addend_applied();
    return check_launch("conv2d_dgrad(igemm)");
}

int igemm_wgrad(const ConvDims &d, const float *x, const float *dy, float *dw, float *dbias, hipStream_t st) {
    raise_lds_caps();
    const long M = (long)d.N * d.P * d.Q;
    const int rows = d.R * d.R * d.C;
    int bm = 64, bn = 64;
    if (d.C % 128 == 0 && d.K % 128 == 0 && (long)(rows / 128) * (d.K / 128) >= kTargetBlocks) { bm = 128; bn = 128; }
    const long tiles = (long)(rows / bm) * (d.K / bn);
    const int nstages = (int)((M + BK - 1) / BK);
    // Workgroups are equal-cost and spread round-robin over 256 CUs, so a launch runs as long as
    // its most loaded CU: efficiency = (blocks/256) / ceil(blocks/256).  576 tiles (D l7) is 75 %;
    // splitting the pixel reduction 4 ways makes it 2304 blocks = 100 % at the price of atomics.
    // Pick the split with the best balance (ties -> fewer splits), >= 8 stages per block.
    int split = 1;
    {
        double best = 0;
        const int maxs = nstages / 8 > 0 ? nstages / 8 : 1;
        for (int sp = 1; sp <= maxs && sp <= 64; ++sp) {
            const double blocks = (double)tiles * sp;
            double eff = (blocks / 256.0) / (double)((long)((blocks + 255) / 256));
            if (blocks < 256) eff = blocks / 256.0;
            if (blocks < 512) eff *= 0.9;                  // one workgroup per CU = one wave per SIMD: nothing hides latency
            if (sp > 1) eff *= 0.98 - 0.004 * sp;          // atomics + zeroing are not free
            if (eff > best + 1e-9) { best = eff; split = sp; }
        }
    }
    int sps = (nstages + split - 1) / split;
    split = (nstages + sps - 1) / sps;
    if (split > 1 && zero_output(dw, sizeof(float) * (long)rows * d.K, st) != hipSuccess)
        return check_launch("conv2d_wgrad memset");
    if (dbias && zero_output(dbias, sizeof(float) * d.K, st) != hipSuccess) return check_launch("conv2d_wgrad memset");
    const dim3 grid(rows / bm, d.K / bn, split);
    if (bm == 128) { if (split > 1) hipLaunchKernelGGL((igemm_wgrad_kernel<128, 128, true>), grid, dim3(256), (smem_bytes<128, 128>()), st, d, x, dy, dw, sps, dbias); else hipLaunchKernelGGL((igemm_wgrad_kernel<128, 128, false>), grid, dim3(256), (smem_bytes<128, 128>()), st, d, x, dy, dw, sps, dbias); }
    else if ((long)grid.x * grid.y * grid.z < kTargetBlocks) {
        // a single 64x64 workgroup per CU: use the 8-wave K-group variant (2 waves per SIMD)
        if (split > 1) hipLaunchKernelGGL((igemm_wgrad_kernel<64, 64, true, 2>), grid, dim3(512), (smem_bytes<64, 64>()), st, d, x, dy, dw, sps, dbias);
        else hipLaunchKernelGGL((igemm_wgrad_kernel<64, 64, false, 2>), grid, dim3(512), (smem_bytes<64, 64>()), st, d, x, dy, dw, sps, dbias);
    }
    else { if (split > 1) hipLaunchKernelGGL((igemm_wgrad_kernel<64, 64, true>), grid, dim3(256), (smem_bytes<64, 64>()), st, d, x, dy, dw, sps, dbias); else hipLaunchKernelGGL((igemm_wgrad_kernel<64, 64, false>), grid, dim3(256), (smem_bytes<64, 64>()), st, d, x, dy, dw, sps, dbias); }
    return check_launch("conv2d_wgrad(igemm)");
}

}  // namespace mmdgan
