// micro-benchmark: sustained v_mfma_f32_32x32x2_f32 rate with 1 or 2 waves per SIMD
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void peak(float *out, int iters, float a, float b) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float x = a + threadIdx.x, y = b;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC> void run(const char *name, int blocks) {
    float *out; hipMalloc(&out, blocks * 256 * 4);
    int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(peak<NACC>, dim3(blocks), dim3(256), 0, 0, out, 10, 1.f, 2.f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(peak<NACC>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)blocks * 4 * iters * 16 * NACC * (2.0 * 32 * 32 * 2);
    printf("%s blocks=%d: %.3f ms  %.1f TFLOP/s\n", name, blocks, ms, flops / ms / 1e9);
    hipFree(out);
}
int main() {
    run<4>("4 acc", 256); run<4>("4 acc", 512); run<1>("1 acc", 256); run<1>("1 acc", 512); run<2>("2 acc", 256);
    return 0;
}
