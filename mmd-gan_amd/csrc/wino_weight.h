// the Winograd weight transforms G g G^T as workgroup bodies: shared by the one-kernel-per-tensor launches
// (conv_wino.hip, conv_wino2.hip) and the one-launch-for-many-tensors table kernel (wino_multi.hip)
#pragma once
#include "common.h"

namespace mmdgan {

// U[seg][f][cr / 8][cr & 1][ko][(cr & 7) >> 1] = (G g G^T)[f], f = 3i + j, g = the 2x2 filter of the segment
//   FWD  : seg = (a,b);  g[u][v] = w[2u + a][2v + b][c][k];            cr = c, ko = k
//   DGRAD: seg = (al,be); g[u][v] = w[rho(al,1-u)][rho(be,1-v)][c][k], rho(0,r') = 1 + 2r', rho(1,r') = 2r';  cr = k, ko = c
// The innermost four floats are the B operands of four consecutive MFMA k-pairs for one lane (k half = cr & 1, column
// = ko): the convolution kernel fetches them with ONE 16-byte load per lane, 512 contiguous bytes per half-wave.
// C and K are multiples of 32 (every geometry the F(2x2,2x2) kernels accept).
// One workgroup of 256 threads = one 32 x 32 block of (c, k) of one segment; tile: 9 x 32 x 33 floats of LDS.
template <bool DGRAD>
__device__ __forceinline__ void wino2_weight_block(float (*tile)[32][33], int bx, int by, int seg, const float *__restrict__ w,
                                                   float *__restrict__ U, int C, int K) {
    const int c0 = by * 32, k0 = bx * 32, sa = seg >> 1, sb = seg & 1;
    const int tk = threadIdx.x & 31, tq = threadIdx.x >> 5;
    for (int cc = tq; cc < 32; cc += 8) {
        const int c = c0 + cc, k = k0 + tk;
        float g[2][2];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int v = 0; v < 2; ++v) {
                const int r = DGRAD ? (sa == 0 ? 1 + 2 * (1 - u) : 2 * (1 - u)) : 2 * u + sa;
                const int t = DGRAD ? (sb == 0 ? 1 + 2 * (1 - v) : 2 * (1 - v)) : 2 * v + sb;
                g[u][v] = w[((size_t)(r * 4 + t) * C + c) * K + k];
            }
        float gg[3][2], uu[3][3];
#pragma unroll
        for (int v = 0; v < 2; ++v) { gg[0][v] = g[0][v]; gg[1][v] = g[0][v] + g[1][v]; gg[2][v] = g[1][v]; }
#pragma unroll
        for (int i = 0; i < 3; ++i) { uu[i][0] = gg[i][0]; uu[i][1] = gg[i][0] + gg[i][1]; uu[i][2] = gg[i][1]; }
#pragma unroll
        for (int f = 0; f < 9; ++f) tile[f][cc][tk] = uu[f / 3][f % 3];
    }
    __syncthreads();
    // thread = (output channel of the block tk, channel group of the block tq >> 1, k half tq & 1)
    const int Cr = DGRAD ? K : C, Ko = DGRAD ? C : K, cr0 = DGRAD ? k0 : c0, ko0 = DGRAD ? c0 : k0;
    const int g8 = tq >> 1, kh = tq & 1;
#pragma unroll
    for (int f = 0; f < 9; ++f) {
        float4 v;
        float *pv = reinterpret_cast<float *>(&v);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int crl = g8 * 8 + 2 * q + kh;
            pv[q] = DGRAD ? tile[f][tk][crl] : tile[f][crl][tk];
        }
        const size_t row = (((size_t)seg * 9 + f) * (Cr >> 3) + (cr0 >> 3) + g8) * 2 + kh;
        *reinterpret_cast<float4 *>(U + (row * Ko + ko0 + tk) * 4) = v;
    }
}

// U[f][cr / 8][cr & 1][ko][(cr & 7) >> 1] = (G g G^T)[f]  with g = w[.][.][c][k] (cr = c, ko = k)          FLIP = false
//                                                      or g = w[2-.][2-.][c][k] read as (cr = k, ko = c)  FLIP = true
// The innermost four floats are the B operands of the four MFMA k-pairs of one 8-channel stage for one lane (k half = cr & 1,
// column = ko): the convolution kernel fetches them with ONE 16-byte load per lane, 512 contiguous bytes per half-wave.
// The reduction-side channel count is a multiple of 8 (what the F(2x2,3x3) kernels accept); ragged 32-blocks are guarded.
// One workgroup of 256 threads = one 32 x 32 block of (c, k); tile: 8 x 32 x 33 floats of LDS (the first 8 planes of a 9-plane one).
template <bool FLIP>
__device__ __forceinline__ void wino_weight_block(float (*tile)[32][33], int bx, int by, const float *__restrict__ w,
                                                  float *__restrict__ U, int C, int K) {
    const int c0 = by * 32, k0 = bx * 32;
    const int tk = threadIdx.x & 31, tq = threadIdx.x >> 5;
    const int Cr = FLIP ? K : C, Ko = FLIP ? C : K, cr0 = FLIP ? k0 : c0, ko0 = FLIP ? c0 : k0;
    for (int half = 0; half < 2; ++half) {              // 8 of the 16 frequencies at a time (LDS)
        for (int cc = tq; cc < 32; cc += 8) {
            const int c = c0 + cc, k = k0 + tk;
            const bool ok = c < C && k < K;
            float g[3][3];
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int t = 0; t < 3; ++t)
                    g[r][t] = ok ? w[((size_t)((FLIP ? 2 - r : r) * 3 + (FLIP ? 2 - t : t)) * C + c) * K + k] : 0.f;
            float gg[4][3], u[4][4];
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                gg[0][t] = g[0][t];
                gg[1][t] = 0.5f * (g[0][t] + g[1][t] + g[2][t]);
                gg[2][t] = 0.5f * (g[0][t] - g[1][t] + g[2][t]);
                gg[3][t] = g[2][t];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                u[i][0] = gg[i][0];
                u[i][1] = 0.5f * (gg[i][0] + gg[i][1] + gg[i][2]);
                u[i][2] = 0.5f * (gg[i][0] - gg[i][1] + gg[i][2]);
                u[i][3] = gg[i][2];
            }
#pragma unroll
            for (int f = 0; f < 8; ++f) tile[f][cc][tk] = half ? u[2 + (f >> 2)][f & 3] : u[f >> 2][f & 3];
        }
        __syncthreads();
        // thread = (output channel of the block tk, channel group of the block tq >> 1, k half tq & 1)
        const int g8 = tq >> 1, kh = tq & 1;
        const bool st_ok = cr0 + g8 * 8 < Cr && ko0 + tk < Ko;
#pragma unroll
        for (int f = 0; f < 8; ++f) {
            if (!st_ok) break;
            float4 v;
            float *pv = reinterpret_cast<float *>(&v);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int crl = g8 * 8 + 2 * q + kh;
                pv[q] = FLIP ? tile[f][tk][crl] : tile[f][crl][tk];
            }
            const size_t row = (((size_t)(half * 8 + f)) * (Cr >> 3) + (cr0 >> 3) + g8) * 2 + kh;
            *reinterpret_cast<float4 *>(U + (row * Ko + ko0 + tk) * 4) = v;
        }
        __syncthreads();
    }
}

// ---- F(4x4,3x3) (conv_wino43.hip)
// U[f][cr / 8][kh][ko][kp] = (G g G^T)[f], f = 6 i + j, with g = w[.][.][c][k] (cr = c, ko = k)            FLIP = false
//                                                       or g = w[2-.][2-.][c][k] read as (cr = k, ko = c)  FLIP = true
// where channel cr of its 8-channel stage sits on k half kh = (cr >> 1) & 1, k-pair kp = (cr & 1) + 2 * ((cr >> 2) & 1): the
// channel pairs (0,1), (2,3), (4,5), (6,7) are the halves of the producers' 16-byte loads and one 8-byte store of V each
// (conv_wino43.hip) - any one-to-one map of a stage's channels onto the MFMA's reduction index works as long as both operands use it.
// G = [1 0 0; 1/3 1/3 1/3; -1/3 1/3 -1/3; -16/15 -8/15 -4/15; 1/15 -2/15 4/15; 0 0 1].
// One workgroup of 256 threads = one 32 x 32 block of (c, k); the 36 frequencies go through 9 LDS planes in four passes.
template <bool FLIP>
__device__ __forceinline__ void wino43_weight_block(float (*tile)[32][33], int bx, int by, const float *__restrict__ w,
                                                    float *__restrict__ U, int C, int K) {
    const int c0 = by * 32, k0 = bx * 32;
    const int tk = threadIdx.x & 31, tq = threadIdx.x >> 5;
    const int Cr = FLIP ? K : C, Ko = FLIP ? C : K, cr0 = FLIP ? k0 : c0, ko0 = FLIP ? c0 : k0;
    for (int pass = 0; pass < 4; ++pass) {
        for (int cc = tq; cc < 32; cc += 8) {
            const int c = c0 + cc, k = k0 + tk;
            const bool ok = c < C && k < K;
            float g[3][3];
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int t = 0; t < 3; ++t)
                    g[r][t] = ok ? w[((size_t)((FLIP ? 2 - r : r) * 3 + (FLIP ? 2 - t : t)) * C + c) * K + k] : 0.f;
            float gg[6][3], u[6][6];
            const float c13 = 1.f / 3.f, c115 = 1.f / 15.f;
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                gg[0][t] = g[0][t];
                gg[1][t] = c13 * (g[0][t] + g[1][t] + g[2][t]);
                gg[2][t] = c13 * (g[1][t] - g[0][t] - g[2][t]);
                gg[3][t] = -4.f * c115 * (4.f * g[0][t] + 2.f * g[1][t] + g[2][t]);
                gg[4][t] = c115 * (g[0][t] - 2.f * g[1][t] + 4.f * g[2][t]);
                gg[5][t] = g[2][t];
            }
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                u[i][0] = gg[i][0];
                u[i][1] = c13 * (gg[i][0] + gg[i][1] + gg[i][2]);
                u[i][2] = c13 * (gg[i][1] - gg[i][0] - gg[i][2]);
                u[i][3] = -4.f * c115 * (4.f * gg[i][0] + 2.f * gg[i][1] + gg[i][2]);
                u[i][4] = c115 * (gg[i][0] - 2.f * gg[i][1] + 4.f * gg[i][2]);
                u[i][5] = gg[i][2];
            }
#pragma unroll
            for (int f = 0; f < 9; ++f) {
                float v = 0.f;
#pragma unroll
                for (int p = 0; p < 4; ++p)
                    if (p == pass) v = u[(9 * p + f) / 6][(9 * p + f) % 6];
                tile[f][cc][tk] = v;
            }
        }
        __syncthreads();
        // thread = (output channel of the block tk, channel group of the block tq >> 1, k half tq & 1)
        const int g8 = tq >> 1, kh = tq & 1;
        const bool st_ok = cr0 + g8 * 8 < Cr && ko0 + tk < Ko;
#pragma unroll
        for (int f = 0; f < 9; ++f) {
            if (!st_ok) break;
            float4 v;
            float *pv = reinterpret_cast<float *>(&v);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int crl = g8 * 8 + (q & 1) + 2 * kh + 4 * (q >> 1);
                pv[q] = FLIP ? tile[f][tk][crl] : tile[f][crl][tk];
            }
            const size_t row = (((size_t)(pass * 9 + f)) * (Cr >> 3) + (cr0 >> 3) + g8) * 2 + kh;
            *reinterpret_cast<float4 *>(U + (row * Ko + ko0 + tk) * 4) = v;
        }
        __syncthreads();
    }
}

}  // namespace mmdgan
