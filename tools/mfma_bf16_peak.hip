// micro-benchmark for the next round's decision: issue rate of the bf16 MFMA a split-bf16 fp32
// emulation would use, next to the fp32 MFMA the conv kernels use now - registers only, no memory, random-ish data
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int NACC, bool BF16>
__global__ __launch_bounds__(256) void peak(float *out, int iters, float a, float b) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float x = a + 0.37f * threadIdx.x, y = b + 0.11f * threadIdx.x;
    bf16x8 xa, yb;
    for (int k = 0; k < 8; ++k) { xa[k] = (__bf16)(x + k); yb[k] = (__bf16)(y - k); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
                if (BF16) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa, yb, acc[i], 0, 0, 0);
                else acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[i], 0, 0, 0);
            }
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC, bool BF16> void run(const char *name, int blocks) {
    float *out; (void)hipMalloc(&out, blocks * 256 * 4);
    int iters = 2000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((peak<NACC, BF16>), dim3(blocks), dim3(256), 0, 0, out, 10, 1.f, 2.f);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((peak<NACC, BF16>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double k = BF16 ? 16.0 : 2.0;
    double flops = (double)blocks * 4 * iters * 16 * NACC * (2.0 * 32 * 32 * k);
    printf("%-28s blocks=%d: %.3f ms  %8.1f TFLOP/s%s\n", name, blocks, ms, flops / ms / 1e9,
           BF16 ? "   (/3 products = fp32-equivalent rate of a 2-term split; /6 of a 3-term split)" : "");
    (void)hipFree(out);
}
int main() {
    run<4, false>("fp32 32x32x2, 4 acc", 512);
    run<4, true>("bf16 32x32x16, 4 acc", 512);
    run<4, true>("bf16 32x32x16, 4 acc", 1024);
    run<2, true>("bf16 32x32x16, 2 acc", 512);
    return 0;
}
