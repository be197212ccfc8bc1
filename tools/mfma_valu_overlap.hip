// Do a consumer wave's MFMAs and a producer wave's VALU work overlap when the two waves share a SIMD (8-wave workgroup, waves
// 0-3 MFMA-only, waves 4-7 VALU-only)?  One workgroup per CU on every CU; times of: MFMAs alone, VALU alone (plain fp32 FMAs /
// packed v_pk_fma_f32 / ds_write_b64), and both.   hipcc -O3 --offload-arch=gfx950 tools/mfma_valu_overlap.hip -o tools/scratch/mvo.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int MODE, int VKIND>   // MODE bit 0: MFMAs, bit 1: VALU work;  VKIND 0 fma, 1 pk_fma, 2 ds_write_b64 + fma mix
__global__ __launch_bounds__(512) void k(float *out, int iters) {
    __shared__ float lds[8192];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave < 4) {
        if (!(MODE & 1)) return;
        f32x16 acc[9];
        for (int f = 0; f < 9; ++f) for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;
        float a = lane * 0.001f, b = lane * 0.002f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int f = 0; f < 9; ++f) acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[f], 0, 0, 0);
        }
        float s = 0;
        for (int f = 0; f < 9; ++f) for (int r = 0; r < 16; ++r) s += acc[f][r];
        if (s == 1.2345f) out[threadIdx.x] = s;
    } else {
        if (!(MODE & 2)) return;
        if (VKIND == 0) {
            float v[16];
            for (int e = 0; e < 16; ++e) v[e] = lane + e;
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int rep = 0; rep < 24; ++rep)          // 384 plain FMAs per iteration (= 192 packed)
#pragma unroll
                    for (int e = 0; e < 16; ++e) v[e] = fmaf(v[e], 1.0001f, 0.5f);
            }
            float s = 0;
            for (int e = 0; e < 16; ++e) s += v[e];
            if (s == 1.2345f) out[threadIdx.x] = s;
        } else if (VKIND == 1) {
            f32x2 v[16];
            for (int e = 0; e < 16; ++e) v[e] = f32x2{(float)lane + e, (float)e};
            const f32x2 c1 = {1.0001f, 1.0001f}, c2 = {0.5f, 0.5f};
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int rep = 0; rep < 12; ++rep)          // 192 packed FMAs per iteration
#pragma unroll
                    for (int e = 0; e < 16; ++e) v[e] = __builtin_elementwise_fma(v[e], c1, c2);
            }
            f32x2 s = {0, 0};
            for (int e = 0; e < 16; ++e) s += v[e];
            if (s.x + s.y == 1.2345f) out[threadIdx.x] = s.x;
        } else if (VKIND >= 3 && VKIND <= 5) {
            constexpr int NCH = VKIND == 3 ? 2 : VKIND == 4 ? 4 : 6;
            float v[NCH];
            for (int e = 0; e < NCH; ++e) v[e] = lane + e;
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int rep = 0; rep < 384 / NCH; ++rep)   // 384 FMAs per iteration in NCH dependent chains
#pragma unroll
                    for (int e = 0; e < NCH; ++e) v[e] = fmaf(v[e], 1.0001f, 0.5f);
            }
            float s = 0;
            for (int e = 0; e < NCH; ++e) s += v[e];
            if (s == 1.2345f) out[threadIdx.x] = s;
        } else {
            f32x2 v = {(float)lane, 1.f};
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int rep = 0; rep < 36; ++rep) {        // 36 ds_write_b64 per iteration
                    *reinterpret_cast<f32x2 *>(lds + ((wave - 4) * 36 + rep) * 32 + lane * 2 % 32) = v;
                    v.x += 1.f;
                }
            }
            __syncthreads();
            if (lds[lane] == 1.2345f) out[threadIdx.x] = v.x;
        }
    }
}
template <int MODE, int VKIND> float run(float *out, int iters) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) k<MODE, VKIND><<<256, 512>>>(out, iters);
    (void)hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) k<MODE, VKIND><<<256, 512>>>(out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 100.f;     // us per launch
}
int main() {
    float *out; (void)hipMalloc(&out, 4096);
    const int iters = 200;   // per iteration: 36 MFMAs (2304 cycles of the matrix pipe) | 384 FMAs | 192 packed FMAs | 36 LDS stores
    printf("per iteration: 36 x v_mfma_f32_32x32x2_f32 on waves 0-3, VALU work on waves 4-7; %d iterations, us per launch (cycles per iteration at 2.4 GHz)\n", iters);
    auto cyc = [&](float us) { return us * 2400.f / iters; };
    float m = run<1, 0>(out, iters);
    printf("MFMAs alone                      %8.1f us (%6.0f)\n", m, cyc(m));
    float a0 = run<2, 0>(out, iters), b0 = run<3, 0>(out, iters);
    printf("384 v_fma_f32: alone %8.1f us (%6.0f)   with the MFMAs %8.1f us (%6.0f)\n", a0, cyc(a0), b0, cyc(b0));
    float a1 = run<2, 1>(out, iters), b1 = run<3, 1>(out, iters);
    printf("192 v_pk_fma_f32: alone %8.1f us (%6.0f)   with the MFMAs %8.1f us (%6.0f)\n", a1, cyc(a1), b1, cyc(b1));
    float a2 = run<2, 2>(out, iters), b2 = run<3, 2>(out, iters);
    printf("36 ds_write_b64: alone %8.1f us (%6.0f)   with the MFMAs %8.1f us (%6.0f)\n", a2, cyc(a2), b2, cyc(b2));
    float a3 = run<2, 3>(out, iters), b3 = run<3, 3>(out, iters);
    printf("384 v_fma_f32 in 2 dependent chains: alone %8.1f us (%6.0f)   with the MFMAs %8.1f us (%6.0f)\n", a3, cyc(a3), b3, cyc(b3));
    float a4 = run<2, 4>(out, iters), b4 = run<3, 4>(out, iters);
    printf("384 v_fma_f32 in 4 dependent chains: alone %8.1f us (%6.0f)   with the MFMAs %8.1f us (%6.0f)\n", a4, cyc(a4), b4, cyc(b4));
    float a5 = run<2, 5>(out, iters), b5 = run<3, 5>(out, iters);
    printf("384 v_fma_f32 in 6 dependent chains: alone %8.1f us (%6.0f)   with the MFMAs %8.1f us (%6.0f)\n", a5, cyc(a5), b5, cyc(b5));
    return 0;
}
