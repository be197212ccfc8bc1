// One power-iteration step of MANY spectrally normalised kernels in eight launches (math_func.py:661-672, 739-744).
//
// The chain of one kernel is  u = F(x) -> sigma = ||u||, y = u / (sigma + eps) -> dsigma/dW = <x, y> outer form,
// x <- normalise(F^T(y)),  F the batch-1 convolution (or matrix product) with the kernel.  Issued per layer that is five
// small launches, ~45 per step for a DCGAN discriminator - batch-1 convolutions of 70 MFLOP that each occupy a few CUs
// for 8-15 us on two side streams.  The chains of different kernels do not depend on each other, so here every STAGE of
// all of them is one launch:
//     0  patches of x (im2col)                       | form 1: P = x W^T                  (+ the sums of squares zeroed)
//     1  u = patches . W                             | form 1: u = fold(P)            (col2im, gather form)
//     2  sum of squares of u                         (4096 elements per workgroup, double, one atomic each)
//     3  sigma, scale = act_k / sigma, y = u / (sigma + eps)        (elementwise.hip:sn_norm_kernel's arithmetic)
//     4  dsigma = patches^T . y ,  P' = y W^T        | form 1: patches of y
//     5  F^T(y) = fold(P')                           | form 1: dsigma = patches(y)^T . x ,  F^T(y) = patches(y) . W
//     6  sum of squares of F^T(y)
//     7  x <- F^T(y) / (||F^T(y)|| + eps)
// A batch-1 convolution IS a small matrix product on its patch matrix [P*Q, R*R*C] (a few MB at most), the HWIO kernel is
// its [R*R*C, K] operand as it lies, and the weight gradient comes out in HWIO layout: one tiled fp32-MFMA product serves
// every stage (operands addressed through two strides each, so transposes cost nothing; long reductions split over
// workgroups with atomics into the zeroed u / F^T(y) / dsigma buffers).  The per-stage work of all kernels fills the chip for
// a few microseconds instead of occupying a corner of it for hundreds.
#include <cstring>

#include "common.h"

namespace mmdgan {

typedef float f32x16 __attribute__((ext_vector_type(16)));

enum { SN_NONE = 0, SN_IM2COL, SN_COL2IM, SN_GEMM, SN_SUMSQ, SN_NORMALISE, SN_ZERO };
constexpr int kSnOpsMax = 16;

struct SnOp {
    int kind;
    unsigned first_block, nblocks;
    const float *a, *b;            // gemm: A, B | im2col: image | col2im: patches | norm: v
    float *c;                      // gemm: C | im2col: patches | col2im: image | norm: v normalised (may alias nothing)
    float *o1, *o2;                // normalise: ||v||, act_k / ||v||
    double *acc;                   // sumsq / normalise: the sum of squares of v (zeroed by the SN_ZERO operation of stage 0)
    int M, N, K, sam, sak, sbk, sbn, ksplit, kchunk;      // gemm: C[M,N] (+)= A'[M,K] B'[K,N], A'(m,k) = a[m*sam + k*sak] ...
    int H, W, C, R, stride, P, Q, pad;                    // im2col / col2im geometry (one image)
    long n;                        // norm: elements
    float act_k;
};
struct SnPhase {
    SnOp ops[kSnOpsMax];
    int n;
};

__device__ __forceinline__ void sn_gemm_block(const SnOp &j, unsigned b) {
    __shared__ float As[16][65], Bs[16][65];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, kh = lane >> 5, wave = tid >> 6, wm = wave & 1, wn = wave >> 1;
    const int tiles_n = (j.N + 63) / 64, tiles_m = (j.M + 63) / 64;
    const unsigned per = (unsigned)tiles_m * tiles_n;
    const int ks = (int)(b / per), t = (int)(b - (unsigned)ks * per), tm = t / tiles_n, tn = t - tm * tiles_n;
    const int m0 = tm * 64, n0 = tn * 64, k0 = ks * j.kchunk, k1 = min(j.K, k0 + j.kchunk);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // a thread's four elements of either 16 x 64 tile: the tile dimension that is contiguous in memory goes along the lanes
    const bool a_kfast = j.sak == 1, b_nfast = j.sbn == 1;
    int ak[4], am[4], bk[4], bn[4];
    long aoff[4], boff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = tid + 256 * i;
        ak[i] = a_kfast ? (e & 15) : (e >> 6);
        am[i] = a_kfast ? (e >> 4) : (e & 63);
        bk[i] = b_nfast ? (e >> 6) : (e & 15);
        bn[i] = b_nfast ? (e & 63) : (e >> 4);
        aoff[i] = (long)(m0 + am[i]) * j.sam + (long)ak[i] * j.sak;
        boff[i] = (long)bk[i] * j.sbk + (long)(n0 + bn[i]) * j.sbn;
    }
    float ra[4], rb[4];
    auto fetch = [&](int kk) {                           // global -> registers, the tile of reduction step kk
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ra[i] = (m0 + am[i] < j.M && kk + ak[i] < k1) ? j.a[aoff[i] + (long)kk * j.sak] : 0.f;
            rb[i] = (n0 + bn[i] < j.N && kk + bk[i] < k1) ? j.b[boff[i] + (long)kk * j.sbk] : 0.f;
        }
    };
    fetch(k0);
    for (int kk = k0; kk < k1; kk += 16) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            As[ak[i]][am[i]] = ra[i];
            Bs[bk[i]][bn[i]] = rb[i];
        }
        __syncthreads();
        if (kk + 16 < k1) fetch(kk + 16);                // the next tile's loads fly under this tile's MFMAs
#pragma unroll
        for (int k2 = 0; k2 < 8; ++k2)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[2 * k2 + kh][wm * 32 + l31], Bs[2 * k2 + kh][wn * 32 + l31], acc, 0, 0, 0);
        __syncthreads();
    }
    const int col = n0 + wn * 32 + l31;
    if (col < j.N) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            if (row < j.M) {
                float *dst = j.c + (long)row * j.N + col;
                if (j.ksplit > 1) atomicAdd(dst, acc[r]);
                else *dst = acc[r];
            }
        }
    }
}

// patches[(p, q)][(r, s, c)] = image[p * stride + r - pad][q * stride + s - pad][c] (0 outside); < 2^30 elements (32-bit index math)
__device__ __forceinline__ void sn_im2col_block(const SnOp &j, unsigned b) {
    const unsigned total = (unsigned)j.P * j.Q * j.R * j.R * j.C, C = j.C, R = j.R, Q = j.Q;
    for (unsigned e = b * 256 + threadIdx.x; e < total; e += j.nblocks * 256) {
        const unsigned t0 = e / C, c = e - t0 * C;
        const unsigned t1 = t0 / R, s = t0 - t1 * R;
        const unsigned t2 = t1 / R, r = t1 - t2 * R;
        const unsigned p = t2 / Q, q = t2 - p * Q;
        const int h = (int)(p * j.stride + r) - j.pad, w = (int)(q * j.stride + s) - j.pad;
        j.c[e] = (h >= 0 && h < j.H && w >= 0 && w < j.W) ? j.a[((unsigned)h * j.W + w) * C + c] : 0.f;
    }
}

// image[h][w][c] = sum over the taps (r, s) that reach it of patches[(p, q)][(r, s, c)]  (the adjoint of the above, as a gather)
__device__ __forceinline__ void sn_col2im_block(const SnOp &j, unsigned b) {
    const unsigned total = (unsigned)j.H * j.W * j.C, C = j.C, W = j.W;
    const unsigned row = (unsigned)j.R * j.R * C;
    for (unsigned e = b * 256 + threadIdx.x; e < total; e += j.nblocks * 256) {
        const unsigned t = e / C, c = e - t * C;
        const unsigned h = t / W, w = t - h * W;
        float acc = 0.f;
        for (int r = 0; r < j.R; ++r) {
            const int hp = (int)h + j.pad - r;
            if (hp < 0) continue;
            const unsigned p = (unsigned)hp / (unsigned)j.stride;
            if (p * j.stride != (unsigned)hp || p >= (unsigned)j.P) continue;
            for (int s2 = 0; s2 < j.R; ++s2) {
                const int wq = (int)w + j.pad - s2;
                if (wq < 0) continue;
                const unsigned q = (unsigned)wq / (unsigned)j.stride;
                if (q * j.stride != (unsigned)wq || q >= (unsigned)j.Q) continue;
                acc += j.a[(p * j.Q + q) * row + ((unsigned)r * j.R + s2) * C + c];
            }
        }
        j.c[e] = acc;
    }
}

// sum of squares of v in double: 4096 elements per workgroup, one atomic each
__device__ __forceinline__ void sn_sumsq_block(const SnOp &j, unsigned b) {
    __shared__ double red[4];
    double acc = 0;
    const long lo = (long)b * 4096, hi = min(j.n, lo + 4096);
    if ((j.n & 3) == 0 && ((uintptr_t)j.a & 15) == 0) {
        const float4 *v4 = reinterpret_cast<const float4 *>(j.a);
        float4 q[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long e = (lo >> 2) + threadIdx.x + 256 * i;
            q[i] = e < (hi >> 2) ? v4[e] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
            acc += (double)q[i].x * (double)q[i].x + (double)q[i].y * (double)q[i].y + (double)q[i].z * (double)q[i].z +
                   (double)q[i].w * (double)q[i].w;
    } else {
        for (long i = lo + threadIdx.x; i < hi; i += 256) acc += (double)j.a[i] * (double)j.a[i];
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(j.acc, red[0] + red[1] + red[2] + red[3]);
}

// ||v||, act_k / ||v||, v / (||v|| + eps) from the finished sum of squares (elementwise.hip:sn_norm_kernel), 4096 elements per workgroup
__device__ __forceinline__ void sn_normalise_block(const SnOp &j, unsigned b) {
    const float s_norm = (float)sqrt(j.acc[0]);
    if (b == 0 && threadIdx.x == 0) {
        if (j.o1) j.o1[0] = s_norm;
        if (j.o2) j.o2[0] = j.act_k / s_norm;                // layer_func.py:886-887
    }
    if (!j.c) return;
    const float inv = 1.0f / (s_norm + kEpsi);
    const long lo = (long)b * 4096, hi = min(j.n, lo + 4096);
    if ((j.n & 3) == 0 && (((uintptr_t)j.a | (uintptr_t)j.c) & 15) == 0) {
        const float4 *v4 = reinterpret_cast<const float4 *>(j.a);
        float4 *o4 = reinterpret_cast<float4 *>(j.c);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long e = (lo >> 2) + threadIdx.x + 256 * i;
            if (e < (hi >> 2)) {
                const float4 q = v4[e];
                o4[e] = make_float4(q.x * inv, q.y * inv, q.z * inv, q.w * inv);
            }
        }
    } else {
        for (long i = lo + threadIdx.x; i < hi; i += 256) j.c[i] = j.a[i] * inv;
    }
}

__global__ __launch_bounds__(256) void sn_phase_kernel(SnPhase ph_by_value) {
    // the operation table is read from the kernel-argument segment where it lies (uniform loads): indexing the by-value
    // argument with the block's operation index would copy all 2.5 KB of it to every thread's scratch first
    const SnPhase &ph = *(const SnPhase *)__builtin_amdgcn_kernarg_segment_ptr();
    int i = 0;
    while (i + 1 < ph.n && blockIdx.x >= ph.ops[i + 1].first_block) ++i;      // (uniform; <= 16 operations)
    const SnOp &j = ph.ops[i];
    const unsigned b = blockIdx.x - j.first_block;
    switch (j.kind) {
    case SN_GEMM: sn_gemm_block(j, b); break;
    case SN_IM2COL: sn_im2col_block(j, b); break;
    case SN_COL2IM: sn_col2im_block(j, b); break;
    case SN_SUMSQ: sn_sumsq_block(j, b); break;
    case SN_NORMALISE: sn_normalise_block(j, b); break;
    case SN_ZERO:
        if (threadIdx.x < j.n) j.acc[threadIdx.x] = 0.0;
        break;
    default: break;
    }
}

namespace {
struct PhaseBuilder {
    SnPhase ph[8];
    SnOp spill;                                            // where an operation beyond kSnOpsMax lands (then `overflow`)
    bool overflow = false;
    PhaseBuilder() { std::memset(ph, 0, sizeof(ph)); }
    SnOp &add(int p, int kind, unsigned nblocks) {
        SnPhase &s = ph[p];
        if (s.n >= kSnOpsMax) {                            // the by-value kernel argument holds kSnOpsMax operations per stage
            overflow = true;
            return spill;
        }
        SnOp &o = s.ops[s.n];
        o.kind = kind;
        o.first_block = s.n ? s.ops[s.n - 1].first_block + s.ops[s.n - 1].nblocks : 0;
        o.nblocks = nblocks;
        ++s.n;
        return o;
    }
    // C[M,N] (+)= A'[M,K] B'[K,N]; split: the output is zero on entry, so a long reduction may be cut over workgroups
    void gemm(int p, const float *A, int sam, int sak, const float *B, int sbk, int sbn, float *C, int M, int N, int K, bool split) {
        const long tiles = (long)((M + 63) / 64) * ((N + 63) / 64);
        int ks = 1;
        if (split && tiles < 512) {
            // parts: ~256 workgroups per product, or up to ~1024 where the OUTPUT is small (every part adds M x N atomics: at most
            // 2^19 of them) - a stage lasts as long as its longest chain of dependent reduction steps, and the products with few
            // output tiles and thousands of reduction elements (u = patches . W of a 4 x 4 image: 8 tiles, K = 4608) were those
            // chains; cutting EVERY product 1024 ways instead cost the ResNet-SN config, with its many mid-sized products, 0.07 ms
            const long by_atomics = (1L << 19) / ((long)M * N);
            const long hi = 1024 / tiles < by_atomics ? 1024 / tiles : by_atomics, lo = tiles < 128 ? 256 / tiles : 1;
            ks = (int)(hi > lo ? hi : lo);
            const int kmax = (K + 63) / 64;                          // >= 64 reduction elements per part
            if (ks > kmax) ks = kmax;
            if (ks < 1) ks = 1;
        }
        int chunk = ((K + ks - 1) / ks + 15) / 16 * 16;
        ks = (K + chunk - 1) / chunk;
        SnOp &o = add(p, SN_GEMM, (unsigned)(tiles * ks));
        o.a = A; o.b = B; o.c = C; o.M = M; o.N = N; o.K = K;
        o.sam = sam; o.sak = sak; o.sbk = sbk; o.sbn = sbn; o.ksplit = ks; o.kchunk = chunk;
    }
    void fold(int p, int kind, const float *src, float *dst, const ConvDims &d) {       // im2col / col2im of one image
        const long total = kind == SN_IM2COL ? (long)d.P * d.Q * d.R * d.R * d.C : (long)d.H * d.W * d.C;
        long blocks = (total + 255) / 256;
        if (blocks > 1024) blocks = 1024;
        SnOp &o = add(p, kind, (unsigned)blocks);
        o.a = src; o.c = dst;
        o.H = d.H; o.W = d.W; o.C = d.C; o.R = d.R; o.stride = d.stride; o.P = d.P; o.Q = d.Q; o.pad = d.pad;
    }
    // stages p and p + 1: the sum of squares of v (into *acc, zero by then), then ||v||, act_k / ||v|| and v / (||v|| + eps)
    void norm(int p, const float *v, long n, float *vn, float *out_norm, float *scale_out, float act_k, double *acc) {
        const unsigned blocks = (unsigned)((n + 4095) / 4096);
        SnOp &o = add(p, SN_SUMSQ, blocks);
        o.a = v; o.n = n; o.acc = acc;
        SnOp &q = add(p + 1, SN_NORMALISE, vn ? blocks : 1);
        q.a = v; q.c = vn; q.n = n; q.o1 = out_norm; q.o2 = scale_out; q.act_k = act_k; q.acc = acc;
    }
};
}  // namespace

}  // namespace mmdgan

using namespace mmdgan;

extern "C" int mmdgan_sn_power_iteration(const mmdgan_sn_layer *layers, int n_layers, int update, void *stream) {
    MMDGAN_REQUIRE(layers || n_layers == 0, "sn_power_iteration: null layer list");
    MMDGAN_REQUIRE(n_layers >= 0, "sn_power_iteration: negative layer count");
    hipStream_t st = (hipStream_t)stream;
    // a group is kSnOpsMax / 2 kernels: a kernel adds at most TWO operations to a stage (the two products of stages 4 / 5),
    // stage 0 holds the group's zeroing operation plus at most one per kernel - PhaseBuilder::add checks it anyway
    static_assert(kSnOpsMax % 2 == 0 && kSnOpsMax / 2 + 1 <= kSnOpsMax, "group size against the operation table");
    for (int i0 = 0; i0 < n_layers; i0 += kSnOpsMax / 2) {
        const int n = n_layers - i0 < kSnOpsMax / 2 ? n_layers - i0 : kSnOpsMax / 2;
        PhaseBuilder pb;
        {                                                  // stage 0 zeroes the two sums of squares of every kernel of the group
            SnOp &zo = pb.add(0, SN_ZERO, 1);
            zo.acc = reinterpret_cast<double *>(layers[i0].norm_acc);
            zo.n = 2 * n;
        }
        for (int i = 0; i < n; ++i) {
            const mmdgan_sn_layer &L = layers[i0 + i];
            MMDGAN_REQUIRE(L.w && L.x && L.u && L.un && L.sigma && L.scale && L.form >= 0 && L.form <= 3 && L.C >= 1 && L.K >= 1,
                           "sn_power_iteration: layer %d: bad arguments", i0 + i);
            MMDGAN_REQUIRE(!update || (L.xb && L.dsigma), "sn_power_iteration: layer %d: update needs xb and dsigma", i0 + i);
            MMDGAN_REQUIRE(L.norm_acc == layers[i0].norm_acc + 4 * i && ((uintptr_t)L.norm_acc & 7) == 0,
                           "sn_power_iteration: layer %d: norm_acc must be 4 floats per layer, consecutive, 8-byte aligned", i0 + i);
            const bool conv = L.form <= 1;
            long nu, nx;
            ConvDims d{};
            long r2c = L.C, pq = 1;
            if (conv) {
                MMDGAN_REQUIRE(L.H >= 1 && L.W >= 1 && L.R >= 1 && L.stride >= 1 && L.col,
                               "sn_power_iteration: layer %d: a convolution needs its geometry and the patch scratch", i0 + i);
                const mmdgan_conv_geom g{1, L.H, L.W, L.C, L.K, L.R, L.stride};
                d = conv_dims(g);
                r2c = (long)L.R * L.R * L.C;
                pq = (long)d.P * d.Q;
                MMDGAN_REQUIRE(pq * r2c < (1L << 30), "sn_power_iteration: layer %d: patch matrix too large", i0 + i);
                nu = L.form == 0 ? pq * L.K : (long)L.H * L.W * L.C;
                nx = L.form == 0 ? (long)L.H * L.W * L.C : pq * L.K;
            } else {
                nu = L.form == 2 ? L.K : L.C;
                nx = L.form == 2 ? L.C : L.K;
            }
            // accumulation targets of the split products: zero on entry (the caller's once-per-step memset in prezeroed mode)
            if (zero_output(L.u, sizeof(float) * nu, st) != hipSuccess) return check_launch("sn_power_iteration memset");
            if (update) {
                if (zero_output(L.dsigma, sizeof(float) * r2c * L.K, st) != hipSuccess ||
                    zero_output(L.xb, sizeof(float) * nx, st) != hipSuccess)
                    return check_launch("sn_power_iteration memset");
            }
            // the patch matrix that is a PRODUCT (y W^T of form 0, x W^T of form 1) is split over workgroups like the others: 16
            // rows x 4608 columns over K = 512 for D's last conv are 72 tiles of 32 dependent reduction steps otherwise, and that
            // one product was the length of its whole stage (48 and 52 us for stages 0 and 4 of CIFAR's discriminator)
            // - for the kernels where that is the case (few tiles, many steps: below 128 tiles, the rule mmdgan_hip.h states):
            // zeroing the 19 MB patch matrix of a 64 x 64 image to cut an 8-step product in two costs more than it saves
            const bool split_patches = conv && ((pq + 63) / 64) * ((r2c + 63) / 64) < 128;
            if (split_patches && (L.form == 1 || update) &&
                zero_output(L.form == 1 ? L.col : L.col + pq * r2c, sizeof(float) * pq * r2c, st) != hipSuccess)
                return check_launch("sn_power_iteration memset");
            const int R2C = (int)r2c, PQ = (int)pq, K = L.K;
            double *acc = reinterpret_cast<double *>(L.norm_acc);
            float *col2 = conv ? L.col + pq * r2c : nullptr;
            switch (L.form) {
            case 0:                                                    // F = conv2d_fwd: x [H,W,C] -> u [P*Q, K]
                pb.fold(0, SN_IM2COL, L.x, L.col, d);
                pb.gemm(1, L.col, R2C, 1, L.w, K, 1, L.u, PQ, K, R2C, true);
                pb.norm(2, L.u, nu, L.un, L.sigma, L.scale, L.act_k, acc);
                if (update) {
                    pb.gemm(4, L.col, 1, R2C, L.un, K, 1, L.dsigma, R2C, K, PQ, true);          // patches^T . y  (HWIO layout)
                    pb.gemm(4, L.un, K, 1, L.w, 1, K, col2, PQ, R2C, K, split_patches);         // y W^T
                    pb.fold(5, SN_COL2IM, col2, L.xb, d);
                    pb.norm(6, L.xb, nx, L.x_out ? L.x_out : L.x, L.xb_norm, nullptr, 0.f, acc + 1);
                }
                break;
            case 1:                                                    // F = conv2d_dgrad: x [P*Q, K] -> u [H,W,C]
                pb.gemm(0, L.x, K, 1, L.w, 1, K, L.col, PQ, R2C, K, split_patches);
                pb.fold(1, SN_COL2IM, L.col, L.u, d);
                pb.norm(2, L.u, nu, L.un, L.sigma, L.scale, L.act_k, acc);
                if (update) {
                    pb.fold(4, SN_IM2COL, L.un, col2, d);
                    pb.gemm(5, col2, 1, R2C, L.x, K, 1, L.dsigma, R2C, K, PQ, true);            // patches(y)^T . x
                    pb.gemm(5, col2, R2C, 1, L.w, K, 1, L.xb, PQ, K, R2C, true);                // conv(y, W)
                    pb.norm(6, L.xb, nx, L.x_out ? L.x_out : L.x, L.xb_norm, nullptr, 0.f, acc + 1);
                }
                break;
            case 2:                                                    // dense: u [1,K] = x [1,C] W
                pb.gemm(1, L.x, R2C, 1, L.w, K, 1, L.u, 1, K, R2C, true);
                pb.norm(2, L.u, nu, L.un, L.sigma, L.scale, L.act_k, acc);
                if (update) {
                    pb.gemm(4, L.x, 1, R2C, L.un, K, 1, L.dsigma, R2C, K, 1, false);            // x^T y
                    pb.gemm(4, L.un, K, 1, L.w, 1, K, L.xb, 1, R2C, K, true);                   // y W^T
                    pb.norm(6, L.xb, nx, L.x_out ? L.x_out : L.x, L.xb_norm, nullptr, 0.f, acc + 1);
                }
                break;
            default:                                                   // dense: u [1,C] = x [1,K] W^T
                pb.gemm(1, L.x, K, 1, L.w, 1, K, L.u, 1, R2C, K, true);
                pb.norm(2, L.u, nu, L.un, L.sigma, L.scale, L.act_k, acc);
                if (update) {
                    pb.gemm(4, L.un, 1, R2C, L.x, K, 1, L.dsigma, R2C, K, 1, false);            // y^T x
                    pb.gemm(4, L.un, R2C, 1, L.w, K, 1, L.xb, 1, K, R2C, true);                 // y W
                    pb.norm(6, L.xb, nx, L.x_out ? L.x_out : L.x, L.xb_norm, nullptr, 0.f, acc + 1);
                }
                break;
            }
        }
        MMDGAN_REQUIRE(!pb.overflow, "sn_power_iteration: more than %d operations in one stage of a group", kSnOpsMax);
        for (int p = 0; p < 8; ++p) {
            const SnPhase &s = pb.ph[p];
            if (!s.n) continue;
            const unsigned blocks = s.ops[s.n - 1].first_block + s.ops[s.n - 1].nblocks;
            hipLaunchKernelGGL(sn_phase_kernel, dim3(blocks), dim3(256), 0, st, s);
            if (int rc = check_launch("sn_power_iteration")) return rc;
        }
    }
    return MMDGAN_OK;
}
