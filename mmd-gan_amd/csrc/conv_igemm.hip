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
//     decomposes the pixel index once, and per K-stage (BK = 16 consecutive channels of ONE
//     filter tap, which is why C % 16 == 0 is required) turns the tap into an address or a zero.
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

namespace mmdgan {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 16;
constexpr int PADL = 4;

template <int BM, int BN>
struct TileCfg {
    static constexpr int LDA = BM + PADL, LDB = BN + PADL;
    static constexpr int WM = BM / 2, WN = BN / 2;
    static constexpr int TM = WM / 32, TN = WN / 32;
    static constexpr int A_F4 = BM * BK / 4 / 256, B_F4 = BN * BK / 4 / 256;
    static constexpr int SMEM_FLOATS = 2 * BK * (LDA + LDB);
};

// Gathers go through buffer loads: the hardware range check returns 0 for an offset beyond
// num_records, so zero padding ('SAME' borders, ragged last tile) is an offset select instead of a
// divergent branch around the load.  The descriptor is built from kernel arguments only
// (wave-uniform, so no waterfall loop is generated).
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned kOOB = 0x80000000u;          // every tensor here is < 2 GiB

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float *base, long bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float4 bufld4(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

// k-contiguous source element f of a ROWS x BK tile: row = f/4, kq = f%4; transposing store
__device__ __forceinline__ void sts_kc(float *S, int LD, int f, const float4 &v) {
    const int row = f >> 2, k = (f & 3) * 4;
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

template <int BM, int BN>
__device__ __forceinline__ void mma_stage(const float *As, const float *Bs, f32x16 (&acc)[TileCfg<BM, BN>::TM][TileCfg<BM, BN>::TN],
                                          int wm, int wn, int lane) {
    using T = TileCfg<BM, BN>;
    const int kh = lane >> 5, l31 = lane & 31;
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
        float a[T::TM], b[T::TN];
#pragma unroll
        for (int mi = 0; mi < T::TM; ++mi) a[mi] = As[(kk + kh) * T::LDA + wm * T::WM + mi * 32 + l31];
#pragma unroll
        for (int ni = 0; ni < T::TN; ++ni) b[ni] = Bs[(kk + kh) * T::LDB + wn * T::WN + ni * 32 + l31];
#pragma unroll
        for (int mi = 0; mi < T::TM; ++mi)
#pragma unroll
            for (int ni = 0; ni < T::TN; ++ni)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
    }
}

// generic main loop: P supplies load_a/load_b (global -> registers for one stage) and says
// whether each operand is k-contiguous (transposing LDS store) or m-contiguous.
// `smem` must be the kernel's own __shared__ array: buffers are addressed by arithmetic on that
// base (never through an array of pointers), so the compiler keeps the LDS address space and
// emits ds_read/ds_write - a pointer table degrades every access to flat_load/flat_store.
template <int BM, int BN, class P>
__device__ __forceinline__ void mainloop(P &p, int s_begin, int s_end, float *smem,
                                         f32x16 (&acc)[TileCfg<BM, BN>::TM][TileCfg<BM, BN>::TN]) {
    using T = TileCfg<BM, BN>;
    constexpr int A_BUF = BK * T::LDA, B_BUF = BK * T::LDB, B_OFF = 2 * A_BUF;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    float4 ra[T::A_F4], rb[T::B_F4];
    if (s_begin >= s_end) return;
    p.load_a(s_begin, ra);
    p.load_b(s_begin, rb);
#define MMDGAN_STORE_STAGE(BUF)                                                                          \
    {                                                                                                    \
        float *As_ = smem + (BUF) * A_BUF;                                                               \
        float *Bs_ = smem + B_OFF + (BUF) * B_BUF;                                                       \
        _Pragma("unroll") for (int i = 0; i < T::A_F4; ++i) {                                            \
            if (P::A_KC) sts_kc(As_, T::LDA, tid + 256 * i, ra[i]);                                      \
            else sts_mc<BM>(As_, T::LDA, tid + 256 * i, ra[i]);                                          \
        }                                                                                                \
        _Pragma("unroll") for (int i = 0; i < T::B_F4; ++i) {                                            \
            if (P::B_KC) sts_kc(Bs_, T::LDB, tid + 256 * i, rb[i]);                                      \
            else sts_mc<BN>(Bs_, T::LDB, tid + 256 * i, rb[i]);                                          \
        }                                                                                                \
    }
    MMDGAN_STORE_STAGE(0)
    __syncthreads();
    for (int s = s_begin; s < s_end; ++s) {
        const int buf = (s - s_begin) & 1;
        const bool more = s + 1 < s_end;
        if (more) { p.load_a(s + 1, ra); p.load_b(s + 1, rb); }
        mma_stage<BM, BN>(smem + buf * A_BUF, smem + B_OFF + buf * B_BUF, acc, wm, wn, lane);
        if (more) MMDGAN_STORE_STAGE(buf ^ 1)
        __syncthreads();
    }
#undef MMDGAN_STORE_STAGE
}

// C/D layout of v_mfma_f32_32x32x2_f32: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#define MMDGAN_FOR_EACH_ACC(T, BODY)                                                            \
    {                                                                                           \
        const int lane_ = threadIdx.x & 63, wave_ = threadIdx.x >> 6;                           \
        const int wm_ = wave_ >> 1, wn_ = wave_ & 1, kh_ = lane_ >> 5, l31_ = lane_ & 31;       \
        _Pragma("unroll") for (int mi = 0; mi < T::TM; ++mi)                                    \
        _Pragma("unroll") for (int ni = 0; ni < T::TN; ++ni)                                    \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                        \
            const int row = wm_ * T::WM + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh_;           \
            const int col = wn_ * T::WN + ni * 32 + l31_;                                       \
            const float v = acc[mi][ni][r];                                                     \
            BODY                                                                                \
        }                                                                                       \
    }

// ------------------------------------------------------------------------------------------------
// forward: y[m = (n,p,q)][k] = sum_{tap,c} x[n, p*s-pad+r, q*s-pad+t, c] * w[tap][c][k]
// ------------------------------------------------------------------------------------------------
template <int BM, int BN>
struct FwdProblem {
    static constexpr bool A_KC = true, B_KC = false;
    using T = TileCfg<BM, BN>;
    ConvDims d;
    __amdgpu_buffer_rsrc_t rx, rw;
    int n0;
    unsigned abase[T::A_F4];          // byte offset of (n, 0, 0, kq*4)
    int ah0[T::A_F4], aw0[T::A_F4];   // ah0 = INT_MIN/2 marks a row beyond M
    __device__ void init(const ConvDims &dd, const float *x_, const float *w_, int m0, int n0_, long M) {
        d = dd; n0 = n0_;
        rx = make_rsrc(x_, (long)d.N * d.H * d.W * d.C * 4);
        rw = make_rsrc(w_, (long)d.R * d.R * d.C * d.K * 4);
#pragma unroll
        for (int i = 0; i < T::A_F4; ++i) {
            const int f = threadIdx.x + 256 * i;
            const long m = (long)m0 + (f >> 2);
            const bool ok = m < M;
            const long mm = ok ? m : 0;
            const int q = mm % d.Q;
            const long t = mm / d.Q;
            const int p = t % d.P;
            const int n = t / d.P;
            ah0[i] = ok ? p * d.stride - d.pad : -(1 << 28);
            aw0[i] = q * d.stride - d.pad;
            abase[i] = (unsigned)(((long)n * d.H * d.W * d.C + (f & 3) * 4) * 4);
        }
    }
    __device__ __forceinline__ void load_a(int s, float4 (&ra)[T::A_F4]) const {
        const int k0 = s * BK;
        const int tap = k0 / d.C, c0 = k0 - tap * d.C;
        const int r = tap / d.R, t = tap - r * d.R;
#pragma unroll
        for (int i = 0; i < T::A_F4; ++i) {
            const int h = ah0[i] + r, ww = aw0[i] + t;
            const bool ok = h >= 0 && h < d.H && ww >= 0 && ww < d.W;
            const unsigned off = abase[i] + (unsigned)(((h * d.W + ww) * d.C + c0) * 4);
            ra[i] = bufld4(rx, ok ? off : kOOB);
        }
    }
    __device__ __forceinline__ void load_b(int s, float4 (&rb)[T::B_F4]) const {
#pragma unroll
        for (int i = 0; i < T::B_F4; ++i) {
            const int f = threadIdx.x + 256 * i;
            const int k = f / (BN / 4), c4 = f % (BN / 4);
            rb[i] = bufld4(rw, (unsigned)((((s * BK + k) * d.K) + n0 + c4 * 4) * 4));
        }
    }
};

template <int BM, int BN>
__global__ __launch_bounds__(256) void igemm_fwd_kernel(ConvDims d, ConvEpilogue ep, const float *__restrict__ x,
                                                        const float *__restrict__ w, float *__restrict__ y,
                                                        int stages_per_split) {
    using T = TileCfg<BM, BN>;
    __shared__ __attribute__((aligned(16))) float smem[T::SMEM_FLOATS];
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
    FwdProblem<BM, BN> p;
    p.init(d, x, w, m0, n0, M);
    mainloop<BM, BN>(p, s0, s1, smem, acc);
    const float sc = ep.scale ? ep.scale[0] : 1.f;
    const bool split = gridDim.z > 1;
    MMDGAN_FOR_EACH_ACC(T, {
        const long m = (long)m0 + row;
        if (m < M) {
            const int ch = n0 + col;
            const long o = m * d.K + ch;
            if (split) atomicAdd(y + o, v * sc + ((ep.bias && blockIdx.z == 0) ? ep.bias[ch] : 0.f));
            else y[o] = ep.apply(v * sc, ch, o);
        }
    })
}

// ------------------------------------------------------------------------------------------------
// input gradient / transposed conv: dx[n,h,w,c] = sum_{r,t,k} dy[n,p,q,k] * w[r,t,c,k],
// p*stride - pad + r = h.  For stride s the output pixels split into s*s parity phases, each a
// dense GEMM over the (R/s)^2 taps that can reach it (blockIdx.z = phase).
// ------------------------------------------------------------------------------------------------
template <int BM, int BN>
struct DgradProblem {
    static constexpr bool A_KC = true, B_KC = true;
    using T = TileCfg<BM, BN>;
    ConvDims d;
    __amdgpu_buffer_rsrc_t rdy, rw;
    int TT, rbase, tbase, pbase, qbase;
    unsigned abase[T::A_F4];
    int ahh[T::A_F4], aww[T::A_F4];
    unsigned bbase[T::B_F4];
    __device__ void init(const ConvDims &dd, const float *dy_, const float *w_, int m0, int n0, int ph, int pw, int Hh,
                         int Ww, long M) {
        d = dd;
        rdy = make_rsrc(dy_, (long)d.N * d.P * d.Q * d.K * 4);
        rw = make_rsrc(w_, (long)d.R * d.R * d.C * d.K * 4);
        TT = d.R / d.stride;
        rbase = (ph + d.pad) % d.stride; tbase = (pw + d.pad) % d.stride;
        pbase = (ph + d.pad) / d.stride; qbase = (pw + d.pad) / d.stride;
#pragma unroll
        for (int i = 0; i < T::A_F4; ++i) {
            const int f = threadIdx.x + 256 * i;
            const long m = (long)m0 + (f >> 2);
            const bool ok = m < M;
            const long mm = ok ? m : 0;
            aww[i] = mm % Ww;
            const long t = mm / Ww;
            ahh[i] = ok ? (int)(t % Hh) : -(1 << 28);
            const int n = t / Hh;
            abase[i] = (unsigned)(((long)n * d.P * d.Q * d.K + (f & 3) * 4) * 4);
        }
#pragma unroll
        for (int i = 0; i < T::B_F4; ++i) {
            const int f = threadIdx.x + 256 * i;
            bbase[i] = (unsigned)(((long)(n0 + (f >> 2)) * d.K + (f & 3) * 4) * 4);     // + tap*C*K + co0 per stage
        }
    }
    __device__ __forceinline__ void load_a(int s, float4 (&ra)[T::A_F4]) const {
        const int k0 = s * BK;
        const int tap = k0 / d.K, co0 = k0 - tap * d.K;
        const int jr = tap / TT, jt = tap - jr * TT;
#pragma unroll
        for (int i = 0; i < T::A_F4; ++i) {
            const int p = ahh[i] + pbase - jr, q = aww[i] + qbase - jt;
            const bool ok = p >= 0 && p < d.P && q >= 0 && q < d.Q;
            const unsigned off = abase[i] + (unsigned)(((p * d.Q + q) * d.K + co0) * 4);
            ra[i] = bufld4(rdy, ok ? off : kOOB);
        }
    }
    __device__ __forceinline__ void load_b(int s, float4 (&rb)[T::B_F4]) const {
        const int k0 = s * BK;
        const int tap = k0 / d.K, co0 = k0 - tap * d.K;
        const int jr = tap / TT, jt = tap - jr * TT;
        const int r = rbase + jr * d.stride, t = tbase + jt * d.stride;
        const unsigned off = (unsigned)((((r * d.R + t) * d.C) * d.K + co0) * 4);
#pragma unroll
        for (int i = 0; i < T::B_F4; ++i) rb[i] = bufld4(rw, bbase[i] + off);
    }
};

template <int BM, int BN>
__global__ __launch_bounds__(256) void igemm_dgrad_kernel(ConvDims d, ConvEpilogue ep, const float *__restrict__ dy,
                                                          const float *__restrict__ w, float *__restrict__ dx,
                                                          int nsplit, int stages_per_split) {
    using T = TileCfg<BM, BN>;
    __shared__ __attribute__((aligned(16))) float smem[T::SMEM_FLOATS];
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
    DgradProblem<BM, BN> p;
    p.init(d, dy, w, m0, n0, ph, pw, Hh, Ww, M);
    mainloop<BM, BN>(p, s0, s1, smem, acc);
    const float sc = ep.scale ? ep.scale[0] : 1.f;
    MMDGAN_FOR_EACH_ACC(T, {
        const long m = (long)m0 + row;
        if (m < M) {
            const int ww = m % Ww;
            const long t = m / Ww;
            const int hh = t % Hh;
            const long n = t / Hh;
            const int ch = n0 + col;
            const long o = ((n * d.H + (hh * d.stride + ph)) * d.W + (ww * d.stride + pw)) * d.C + ch;
            if (nsplit > 1) atomicAdd(dx + o, v * sc + ((ep.bias && split == 0) ? ep.bias[ch] : 0.f));
            else dx[o] = ep.apply(v * sc, ch, o);
        }
    })
}

// ------------------------------------------------------------------------------------------------
// weight gradient: dw[(tap,c)][k] = sum_{pixels m} x[m shifted by tap][c] * dy[m][k]
// GEMM rows i = (tap, c) (a BM-row tile sits inside one tap: C % BM == 0), reduction over pixels.
// ------------------------------------------------------------------------------------------------
template <int BM, int BN>
struct WgradProblem {
    static constexpr bool A_KC = false, B_KC = false;
    using T = TileCfg<BM, BN>;
    ConvDims d;
    __amdgpu_buffer_rsrc_t rx, rdy;
    int r, t, c0, n0;
    long M;
    __device__ void init(const ConvDims &dd, const float *x_, const float *dy_, int i0, int n0_, long M_) {
        d = dd; n0 = n0_; M = M_;
        rx = make_rsrc(x_, (long)d.N * d.H * d.W * d.C * 4);
        rdy = make_rsrc(dy_, (long)d.N * d.P * d.Q * d.K * 4);
        const int tap = i0 / d.C;
        c0 = i0 - tap * d.C;
        r = tap / d.R; t = tap - r * d.R;
    }
    __device__ __forceinline__ void load_a(int s, float4 (&ra)[T::A_F4]) const {
#pragma unroll
        for (int i = 0; i < T::A_F4; ++i) {
            const int f = threadIdx.x + 256 * i;
            const int k = f / (BM / 4), c4 = f % (BM / 4);
            const long m = (long)s * BK + k;
            bool ok = m < M;
            const long mm = ok ? m : 0;
            const int q = mm % d.Q;
            const long u = mm / d.Q;
            const int p = u % d.P;
            const long n = u / d.P;
            const int h = p * d.stride - d.pad + r, ww = q * d.stride - d.pad + t;
            ok = ok && h >= 0 && h < d.H && ww >= 0 && ww < d.W;
            const unsigned off = (unsigned)((((n * d.H + h) * d.W + ww) * d.C + c0 + c4 * 4) * 4);
            ra[i] = bufld4(rx, ok ? off : kOOB);
        }
    }
    __device__ __forceinline__ void load_b(int s, float4 (&rb)[T::B_F4]) const {
#pragma unroll
        for (int i = 0; i < T::B_F4; ++i) {
            const int f = threadIdx.x + 256 * i;
            const int k = f / (BN / 4), c4 = f % (BN / 4);
            const long m = (long)s * BK + k;
            rb[i] = bufld4(rdy, m < M ? (unsigned)((m * d.K + n0 + c4 * 4) * 4) : kOOB);
        }
    }
};

template <int BM, int BN>
__global__ __launch_bounds__(256) void igemm_wgrad_kernel(ConvDims d, const float *__restrict__ x,
                                                          const float *__restrict__ dy, float *__restrict__ dw,
                                                          int stages_per_split) {
    using T = TileCfg<BM, BN>;
    __shared__ __attribute__((aligned(16))) float smem[T::SMEM_FLOATS];
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
    WgradProblem<BM, BN> p;
    p.init(d, x, dy, i0, n0, M);
    mainloop<BM, BN>(p, s0, s1, smem, acc);
    const bool split = gridDim.z > 1;
    MMDGAN_FOR_EACH_ACC(T, {
        const long o = (long)(i0 + row) * d.K + n0 + col;
        if (split) atomicAdd(dw + o, v);
        else dw[o] = v;
    })
}

// ------------------------------------------------------------------------------------------------
// host side: eligibility, tile and split selection
// ------------------------------------------------------------------------------------------------
constexpr int kTargetBlocks = 256;      // one workgroup per CU at least

bool igemm_fwd_ok(const ConvDims &d) { return d.C % BK == 0 && d.K % 64 == 0 && d.R * d.R * d.C >= 64; }
bool igemm_dgrad_ok(const ConvDims &d) { return d.K % BK == 0 && d.C % 64 == 0 && d.R % d.stride == 0 && d.K >= 16; }
bool igemm_wgrad_ok(const ConvDims &d) { return d.C % 64 == 0 && d.K % 64 == 0; }

static void pick_tile(long M, int N, int &bm, int &bn) {
    // largest tile that still yields >= kTargetBlocks workgroups, else the smallest
    const int cand[3][2] = {{128, 128}, {128, 64}, {64, 64}};
    for (int i = 0; i < 3; ++i) {
        bm = cand[i][0]; bn = cand[i][1];
        if (N % bn) continue;
        const long tiles = ((M + bm - 1) / bm) * (N / bn);
        if (tiles >= kTargetBlocks) return;
    }
    bm = 64; bn = 64;
}

static int pick_split(long tiles, int nstages, bool allowed) {
    if (!allowed || tiles >= kTargetBlocks / 2) return 1;
    int s = (int)(kTargetBlocks / tiles);
    const int maxs = nstages / 8;                // keep >= 8 stages (128 deep) per split
    if (s > maxs) s = maxs;
    return s < 1 ? 1 : s;
}

int igemm_fwd(const ConvDims &d, const ConvEpilogue &ep, const float *x, const float *w, float *y, hipStream_t st) {
    const long M = (long)d.N * d.P * d.Q;
    int bm, bn;
    pick_tile(M, d.K, bm, bn);
    const int nstages = d.R * d.R * d.C / BK;
    const long tiles = ((M + bm - 1) / bm) * (d.K / bn);
    int split = pick_split(tiles, nstages, ep.act == MMDGAN_ACT_LINEAR && !ep.dact);
    int sps = (nstages + split - 1) / split;
    split = (nstages + sps - 1) / sps;
    if (split > 1 && hipMemsetAsync(y, 0, sizeof(float) * M * d.K, st) != hipSuccess) return check_launch("conv2d_fwd memset");
    const dim3 grid((unsigned)((M + bm - 1) / bm), d.K / bn, split);
    if (bm == 128 && bn == 128) hipLaunchKernelGGL((igemm_fwd_kernel<128, 128>), grid, dim3(256), 0, st, d, ep, x, w, y, sps);
    else if (bm == 128) hipLaunchKernelGGL((igemm_fwd_kernel<128, 64>), grid, dim3(256), 0, st, d, ep, x, w, y, sps);
    else hipLaunchKernelGGL((igemm_fwd_kernel<64, 64>), grid, dim3(256), 0, st, d, ep, x, w, y, sps);
    return check_launch("conv2d_fwd(igemm)");
}

int igemm_dgrad(const ConvDims &d, const ConvEpilogue &ep, const float *dy, const float *w, float *dx, hipStream_t st) {
    const int s = d.stride;
    const int Hh = (d.H + s - 1) / s, Ww = (d.W + s - 1) / s;       // largest phase
    const long M = (long)d.N * Hh * Ww;
    int bm, bn;
    pick_tile(M * s * s, d.C, bm, bn);
    // pick_tile counted all phases together; recompute with per-phase rows
    const long tiles = ((M + bm - 1) / bm) * (d.C / bn) * s * s;
    const int TT = d.R / s;
    const int nstages = TT * TT * d.K / BK;
    int split = pick_split(tiles, nstages, ep.act == MMDGAN_ACT_LINEAR && !ep.dact);
    int sps = (nstages + split - 1) / split;
    split = (nstages + sps - 1) / sps;
    if (split > 1 && hipMemsetAsync(dx, 0, sizeof(float) * (long)d.N * d.H * d.W * d.C, st) != hipSuccess)
        return check_launch("conv2d_dgrad memset");
    const dim3 grid((unsigned)((M + bm - 1) / bm), d.C / bn, s * s * split);
    if (bm == 128 && bn == 128) hipLaunchKernelGGL((igemm_dgrad_kernel<128, 128>), grid, dim3(256), 0, st, d, ep, dy, w, dx, split, sps);
    else if (bm == 128) hipLaunchKernelGGL((igemm_dgrad_kernel<128, 64>), grid, dim3(256), 0, st, d, ep, dy, w, dx, split, sps);
    else hipLaunchKernelGGL((igemm_dgrad_kernel<64, 64>), grid, dim3(256), 0, st, d, ep, dy, w, dx, split, sps);
    return check_launch("conv2d_dgrad(igemm)");
}

int igemm_wgrad(const ConvDims &d, const float *x, const float *dy, float *dw, hipStream_t st) {
    const long M = (long)d.N * d.P * d.Q;
    const int rows = d.R * d.R * d.C;
    int bm = 64, bn = 64;
    if (d.C % 128 == 0 && d.K % 128 == 0 && (long)(rows / 128) * (d.K / 128) >= kTargetBlocks) { bm = 128; bn = 128; }
    const long tiles = (long)(rows / bm) * (d.K / bn);
    const int nstages = (int)((M + BK - 1) / BK);
    int split = 1;
    if (tiles < kTargetBlocks) {
        split = (int)((2 * kTargetBlocks + tiles - 1) / tiles);
        const int maxs = nstages / 8 > 0 ? nstages / 8 : 1;
        if (split > maxs) split = maxs;
    }
    int sps = (nstages + split - 1) / split;
    split = (nstages + sps - 1) / sps;
    if (split > 1 && hipMemsetAsync(dw, 0, sizeof(float) * (long)rows * d.K, st) != hipSuccess)
        return check_launch("conv2d_wgrad memset");
    const dim3 grid(rows / bm, d.K / bn, split);
    if (bm == 128) hipLaunchKernelGGL((igemm_wgrad_kernel<128, 128>), grid, dim3(256), 0, st, d, x, dy, dw, sps);
    else hipLaunchKernelGGL((igemm_wgrad_kernel<64, 64>), grid, dim3(256), 0, st, d, x, dy, dw, sps);
    return check_launch("conv2d_wgrad(igemm)");
}

}  // namespace mmdgan
