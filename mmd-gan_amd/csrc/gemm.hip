// Dense layers (tf.matmul, layer_func.py:909-911) and their gradients, plus the dense power
// iteration of spectral norm (math_func.py:583-602).  These are the skinny GEMMs of the model
// (G l1: [B,128]x[128,8192]; D l8: [2B,8192]x[8192,16]) - under 0.1 % of the step's FLOPs and
// bound by streaming the weight matrix once, so this is an LDS-tiled fp32 VALU kernel with
// split-K for the K >> M*N shapes rather than an MFMA kernel.
// (Round 4, measured and not kept: a variant for the short reductions - G l1's [B,128] x [128,8192] and its weight gradient -
// that fetches a workgroup's whole K extent at once instead of walking 16-deep steps: the CIFAR step 1.907 vs 1.896 ms.)
//   C[M,N] = act(scale * op(A) op(B) + bias)           (or * act'(dact_of) in backward form)
#include <stdint.h>

#include "common.h"
#include "tuning.h"

namespace mmdgan {

constexpr int GT = 64;   // tile M = N
constexpr int GK = 16;   // tile K

struct GemmArgs {
    const float *A, *B, *bias, *scale, *dact;
    float *C;
    int M, N, K, lda, ldb, ldc, transA, transB, act, ksplit, kchunk;
    long wrap_from, wrap_sub;          // rows >= dact_rows read dact shifted back (see ConvEpilogue)
};

__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs g) {
    // double-buffered LDS tiles; the next k-tile's global loads are issued before the FMAs of the
    // current one and parked in LDS after them, so one barrier per k-step and the HBM latency of
    // these short-K launches hides behind the arithmetic
    __shared__ float As[2][GK][GT + 4];
    __shared__ float Bs[2][GK][GT + 4];
    const int tid = threadIdx.x;
    const int m0 = blockIdx.y * GT, n0 = blockIdx.x * GT;
    const int kbeg = blockIdx.z * g.kchunk;
    int kend = kbeg + g.kchunk;
    if (kend > g.K) kend = g.K;
    const int tm = (tid >> 4) * 4, tn = (tid & 15) * 4;
    float acc[4][4] = {};
    float ra[4], rb[4];
    // 64x16 elements per operand = 1024 -> 4 per thread
    auto gload = [&](int k0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int idx = tid + e * 256;
            {   // A tile: element (m, k)
                int m, k;
                if (g.transA) { m = idx & 63; k = idx >> 6; } else { k = idx & 15; m = idx >> 4; }
                const int gm = m0 + m, gk = k0 + k;
                float v = 0.f;
                if (gm < g.M && gk < kend) v = g.transA ? g.A[(size_t)gk * g.lda + gm] : g.A[(size_t)gm * g.lda + gk];
                ra[e] = v;
            }
            {   // B tile: element (k, n)
                int n, k;
                if (g.transB) { k = idx & 15; n = idx >> 4; } else { n = idx & 63; k = idx >> 6; }
                const int gn = n0 + n, gk = k0 + k;
                float v = 0.f;
                if (gn < g.N && gk < kend) v = g.transB ? g.B[(size_t)gn * g.ldb + gk] : g.B[(size_t)gk * g.ldb + gn];
                rb[e] = v;
            }
        }
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int idx = tid + e * 256;
            if (g.transA) As[buf][idx >> 6][idx & 63] = ra[e]; else As[buf][idx & 15][idx >> 4] = ra[e];
            if (g.transB) Bs[buf][idx & 15][idx >> 4] = rb[e]; else Bs[buf][idx >> 6][idx & 63] = rb[e];
        }
    };
    gload(kbeg);
    sstore(0);
    __syncthreads();
    int cur = 0;
    for (int k0 = kbeg; k0 < kend; k0 += GK, cur ^= 1) {
        const bool more = k0 + GK < kend;
        if (more) gload(k0 + GK);
#pragma unroll
        for (int k = 0; k < GK; ++k) {
            const float4 a = *reinterpret_cast<const float4 *>(&As[cur][k][tm]);
            const float4 b = *reinterpret_cast<const float4 *>(&Bs[cur][k][tn]);
            const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(av[i], bv[j], acc[i][j]);
        }
        if (more) sstore(cur ^ 1);
        __syncthreads();
    }
    const float sc = g.scale ? g.scale[0] : 1.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int gm = m0 + tm + i;
        if (gm >= g.M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int gn = n0 + tn + j;
            if (gn >= g.N) continue;
            const size_t o = (size_t)gm * g.ldc + gn;
            float v = acc[i][j] * sc;
            if (g.ksplit > 1) {                       // linear epilogue only: partial sums via atomics
                if (g.bias && blockIdx.z == 0) v += g.bias[gn];
                atomicAdd(g.C + o, v);
            } else {
                if (g.bias) v += g.bias[gn];
                g.C[o] = g.dact ? v * act_bwd_from_out(g.dact[o >= (size_t)g.wrap_from ? o - g.wrap_sub : o], g.act) : act_fwd(v, g.act);
            }
        }
    }
}

// The discriminator's head (D l8: [2B, 8192] x [8192, 16], layer_func.py:909-911) is a skinny-N product on the step's critical
// path between D's forward and backward passes: the tiled kernel above spends 23 us on it (a 64-wide N tile for 16 columns,
// 262 k atomics).  Here: v_mfma_f32_16x16x4_f32, one wave = 16 rows x 16 columns x a slice of K; a lane loads 16 bytes of its
// row per 16-deep block (lanes permute k consistently on both operands), four waves of a workgroup take neighbouring K
// slices and add up through LDS, one atomic per output and workgroup into the zeroed C.   C += scale * A[M,K] B[K,16] (+ bias once)
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void gemm_skinny16_kernel(const float *__restrict__ A, int lda, const float *__restrict__ B, int ldb,
                                                            const float *__restrict__ bias, const float *__restrict__ scale,
                                                            float *__restrict__ C, int ldc, int K, int kchunk) {
    __shared__ float red[4][16][17];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = lane & 15, kq = lane >> 4;
    const int m0 = blockIdx.x * 16;
    const int per_wave = kchunk / 4;                       // multiple of 64
    const int k0 = blockIdx.y * kchunk + wave * per_wave, k1 = min(K, k0 + per_wave);
    const float *ap = A + (size_t)(m0 + row) * lda + 4 * kq;
    const float *bp = B + (size_t)(4 * kq) * ldb + row;    // (row doubles as the column index of the B operand)
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k = k0; k < k1; k += 64) {                     // (slices are multiples of 64: four blocks of loads in flight)
        float4 a[4];
        float b[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            a[u] = *reinterpret_cast<const float4 *>(ap + k + 16 * u);
#pragma unroll
            for (int j = 0; j < 4; ++j) b[u][j] = bp[(size_t)(k + 16 * u + j) * ldb];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].x, b[u][0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].y, b[u][1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].z, b[u][2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].w, b[u][3], acc, 0, 0, 0);
        }
    }
    // accumulator layout of 16x16x4: register r of lane (col = lane & 15, group = lane >> 4) is C[4 * group + r][col]
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][4 * kq + r][row] = acc[r];
    __syncthreads();
    const int t = threadIdx.x;                              // 256 threads = 16 x 16 outputs
    const int om = t >> 4, on = t & 15;
    float v = (red[0][om][on] + red[1][om][on]) + (red[2][om][on] + red[3][om][on]);
    v *= scale ? scale[0] : 1.f;
    if (bias && blockIdx.y == 0) v += bias[on];
    atomicAdd(C + (size_t)(m0 + om) * ldc + on, v);
}

// Short reductions over a long panel (round 5).  G's first layer [B,128] x [128,8192], its weight gradient z^T [128,B] x
// dz [B,8192] and the weight gradient of D's head x^T [8192,2B] x dz [2B,16] have K <= 128: the tiled kernel above walks them
// in 16-deep steps with a barrier and a global-load round trip each (20-26 us for ~0.1 GFLOP and 4 MB).  Here a workgroup
// takes a 32-column (64-row) piece of the long dimension, fetches its WHOLE K extent of both operands at once - every load
// of the launch is in flight together: one memory round trip - and multiplies out of LDS.
//   npanel: C[M,N] = act(scale * op(A) B + bias), M a multiple of 16 (<= 256), K <= 128, B [K,N] row-major; TRANSA: A is [K,M]
//   mpanel: C[M,16] = A^T B, A [K,M] row-major, B [K,16]
constexpr int GP_KMAX = 128;
// npanel: one wave = 16 rows x 16 columns x the whole K on v_mfma_f32_16x16x4_f32, operands straight from global memory into
// the MFMA registers (every load of the wave issued before the first multiply), four waves = 64 rows; grid (N / 16, M / 64).
// (First cut of the round: operands through LDS, a scalar k-loop - 17.8 us alone against 15.8 for the tiled kernel: the loop
// waited for every LDS read.)
template <bool TRANSA>
__global__ __launch_bounds__(256) void gemm_npanel_kernel(const float *__restrict__ A, int lda, const float *__restrict__ B, int ldb,
                                                          const float *__restrict__ bias, const float *__restrict__ scale, int act,
                                                          float *__restrict__ C, int ldc, int M, int K) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = lane & 15, kq = lane >> 4;
    const int m0 = blockIdx.y * 64 + wave * 16, n0 = blockIdx.x * 16;
    if (m0 >= M) return;                                   // (M a multiple of 16: the last workgroup may have idle waves; no barrier below)
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    // lane (row, kq) holds A[m0 + row][16 u + 4 kq + j] and B[16 u + 4 kq + j][n0 + row], j = 0..3, u = 0 .. K/16 - 1 (+ a 4-deep
    // tail when K % 16 != 0: K is a multiple of 4)
    const int nblk = (K + 15) >> 4;
    float a[GP_KMAX / 16][4], b[GP_KMAX / 16][4];
#pragma unroll
    for (int u = 0; u < GP_KMAX / 16; ++u) {
        if (u < nblk) {
            const int k = 16 * u + 4 * kq;
            const bool ok = k < K;                         // (K % 4 == 0: a quad is inside or outside as a whole)
            if (TRANSA) {
#pragma unroll
                for (int j = 0; j < 4; ++j) a[u][j] = ok ? A[(size_t)(k + j) * lda + m0 + row] : 0.f;
            } else {
                const float4 v = ok ? *reinterpret_cast<const float4 *>(A + (size_t)(m0 + row) * lda + k) : make_float4(0.f, 0.f, 0.f, 0.f);
                a[u][0] = v.x; a[u][1] = v.y; a[u][2] = v.z; a[u][3] = v.w;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) b[u][j] = ok ? B[(size_t)(k + j) * ldb + n0 + row] : 0.f;
        }
    }
#pragma unroll
    for (int u = 0; u < GP_KMAX / 16; ++u) {
        if (u < nblk) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][j], b[u][j], acc, 0, 0, 0);
        }
    }
    // accumulator layout of 16x16x4: register r of lane (col = lane & 15, group = lane >> 4) is C[4 * group + r][col]
    const float sc = scale ? scale[0] : 1.f;
    const float bv = bias ? bias[n0 + row] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) C[(size_t)(m0 + 4 * kq + r) * ldc + n0 + row] = act_fwd(acc[r] * sc + bv, act);
}

__global__ __launch_bounds__(256) void gemm_mpanel16_kernel(const float *__restrict__ A, int lda, const float *__restrict__ B, int ldb,
                                                            float *__restrict__ C, int ldc, int K) {
    __shared__ __attribute__((aligned(16))) float As[GP_KMAX * 64];             // [k][64 rows of M]
    __shared__ __attribute__((aligned(16))) float Bs[GP_KMAX * 16];             // [k][16]
    const int tid = threadIdx.x, m0 = blockIdx.x * 64;
    for (int e = tid; e < K * 16; e += 256) {
        const int k = e >> 4, q = e & 15;
        *reinterpret_cast<float4 *>(&As[k * 64 + 4 * q]) = *reinterpret_cast<const float4 *>(A + (size_t)k * lda + m0 + 4 * q);
    }
    for (int e = tid; e < K * 4; e += 256) {
        const int k = e >> 2, q = e & 3;
        *reinterpret_cast<float4 *>(&Bs[k * 16 + 4 * q]) = *reinterpret_cast<const float4 *>(B + (size_t)k * ldb + 4 * q);
    }
    __syncthreads();
    const int m = tid >> 2, nq = tid & 3;                  // one row, four columns
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = 0; k < K; ++k) {
        const float a = As[k * 64 + m];
        const float4 b = *reinterpret_cast<const float4 *>(&Bs[k * 16 + 4 * nq]);
        acc.x = fmaf(a, b.x, acc.x); acc.y = fmaf(a, b.y, acc.y); acc.z = fmaf(a, b.z, acc.z); acc.w = fmaf(a, b.w, acc.w);
    }
    *reinterpret_cast<float4 *>(C + (size_t)(m0 + m) * ldc + 4 * nq) = acc;
}

}  // namespace mmdgan

using namespace mmdgan;

extern "C" int mmdgan_gemm(int transA, int transB, int M, int N, int K, const float *A, int lda, const float *B, int ldb,
                           const float *bias, const float *scale, int act, const float *dact_of, int dact_rows, float *C,
                           int ldc, void *stream) {
    MMDGAN_REQUIRE(A && B && C, "gemm: null pointer");
    MMDGAN_REQUIRE(M >= 1 && N >= 1 && K >= 1, "gemm: bad shape %dx%dx%d", M, N, K);
    // MMDGAN_ACT_FLAG_OUT_ZEROED: C is zero on entry and the launch may accumulate into it (split-K).  Under
    // mmdgan_set_outputs_prezeroed(1) that flag is the ONLY licence to split: the caller names the outputs it zeroed,
    // the library never guesses from the shape
    const bool out_zeroed = (act & MMDGAN_ACT_FLAG_OUT_ZEROED) != 0;
    act &= 0xff;
    MMDGAN_REQUIRE(act >= MMDGAN_ACT_LINEAR && act <= MMDGAN_ACT_TANH, "gemm: unknown activation %d", act);
    hipStream_t st = (hipStream_t)stream;
    GemmArgs g;
    g.A = A; g.B = B; g.bias = bias; g.scale = scale; g.dact = dact_of; g.C = C;
    g.wrap_from = 0x7fffffffffffffffL; g.wrap_sub = 0;
    if (dact_of && dact_rows != 0 && dact_rows != M) {
        MMDGAN_REQUIRE(dact_rows > 0 && dact_rows < M && M - dact_rows <= dact_rows, "gemm: bad dact_rows %d for M %d", dact_rows, M);
        g.wrap_from = (long)dact_rows * ldc; g.wrap_sub = (long)(M - dact_rows) * ldc;
    }
    if (!transA && !transB && N == 16 && M % 16 == 0 && K % 256 == 0 && K >= 1024 && act == MMDGAN_ACT_LINEAR && !dact_of &&
        lda % 4 == 0 && ((uintptr_t)A & 15) == 0 && ldc == N && (!outputs_prezeroed() || out_zeroed)) {
        if (tuning().gemm_skinny) {
            // ~512 waves: K split so that every wave keeps >= 64 of K (4 blocks of loads in flight)
            int ksplit = 128 / (M / 16);
            if (ksplit < 1) ksplit = 1;
            while (ksplit > 1 && K / (ksplit * 4) < 64) ksplit >>= 1;
            int kchunk = (K + ksplit - 1) / ksplit;
            kchunk = (kchunk + 255) / 256 * 256;
            ksplit = (K + kchunk - 1) / kchunk;
            if (!out_zeroed && zero_output(C, sizeof(float) * (size_t)M * ldc, st) != hipSuccess) return check_launch("gemm memset");
            hipLaunchKernelGGL(gemm_skinny16_kernel, dim3(M / 16, ksplit), dim3(256), 0, st, A, lda, B, ldb, bias, scale, C, ldc, K, kchunk);
            return check_launch("gemm");
        }
    }
    auto al16 = [](const void *p) { return ((uintptr_t)p & 15) == 0; };
    if (tuning().gemm_panel && !transB && !dact_of && K <= GP_KMAX && K % 4 == 0 && al16(A) && al16(B) && al16(C) && lda % 4 == 0 &&
        ldb % 4 == 0 && ldc % 4 == 0) {
        if (M % 16 == 0 && M <= 256 && N % 16 == 0 && N >= 2048) {
            const dim3 grid(N / 16, (M + 63) / 64), block(M >= 64 ? 256 : M * 4);          // a wave per 16 rows
            if (transA)
                hipLaunchKernelGGL(gemm_npanel_kernel<true>, grid, block, 0, st, A, lda, B, ldb, bias, scale, act, C, ldc, M, K);
            else
                hipLaunchKernelGGL(gemm_npanel_kernel<false>, grid, block, 0, st, A, lda, B, ldb, bias, scale, act, C, ldc, M, K);
            return check_launch("gemm");
        }
        if (transA && N == 16 && M % 64 == 0 && M >= 2048 && !bias && !scale && act == MMDGAN_ACT_LINEAR) {
            hipLaunchKernelGGL(gemm_mpanel16_kernel, dim3(M / 64), dim3(256), 0, st, A, lda, B, ldb, C, ldc, K);
            return check_launch("gemm");
        }
    }
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.transA = transA; g.transB = transB; g.act = act;
    const int tiles = ((M + GT - 1) / GT) * ((N + GT - 1) / GT);
    int ksplit = 1;
    if (act == MMDGAN_ACT_LINEAR && !dact_of && tiles < 128 && K >= 512 && ldc == N &&
        (!outputs_prezeroed() || out_zeroed)) {
        ksplit = 512 / tiles;
        const int maxs = K / 64;
        if (ksplit > maxs) ksplit = maxs;
        if (ksplit < 1) ksplit = 1;
    }
    int kchunk = (K + ksplit - 1) / ksplit;
    kchunk = (kchunk + GK - 1) / GK * GK;
    ksplit = (K + kchunk - 1) / kchunk;
    g.ksplit = ksplit; g.kchunk = kchunk;
    if (ksplit > 1 && !out_zeroed && zero_output(C, sizeof(float) * (size_t)M * N, st) != hipSuccess)
        return check_launch("gemm memset");
    hipLaunchKernelGGL(gemm_kernel, dim3((N + GT - 1) / GT, (M + GT - 1) / GT, ksplit), dim3(256), 0, st, g);
    return check_launch("gemm");
}
