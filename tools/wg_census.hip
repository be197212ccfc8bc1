// where do the workgroups of a 2-per-CU persistent grid land?  prints, per (XCC, SE, CU), the wave slots / SIMDs of the first
// wave of every resident workgroup.  hipcc -O3 --offload-arch=gfx950 wg_census.hip -o wg_census.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <map>
#include <vector>
__global__ __launch_bounds__(256, 2) void census(unsigned *out, int spin) {
    extern __shared__ float smem[];
    const unsigned hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));
    const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11));
    if ((threadIdx.x & 63) == 0) { out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = hw; out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = xcc; }
    smem[threadIdx.x] = 1.f;
    for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(100);      // stay resident while the rest of the grid arrives
}
int main() {
    const int grid = 512;
    unsigned *d; (void)hipMalloc(&d, grid * 8 * 4);
    (void)hipFuncSetAttribute((const void *)census, hipFuncAttributeMaxDynamicSharedMemorySize, 77000);
    census<<<grid, 256, 77000>>>(d, 2000);
    std::vector<unsigned> h(grid * 8);
    (void)hipMemcpy(h.data(), d, grid * 8 * 4, hipMemcpyDeviceToHost);
    std::map<unsigned, std::vector<int>> cu;     // key (xcc, se, cu) -> workgroups
    for (int b = 0; b < grid; ++b) {
        const unsigned hw = h[b * 8], xcc = h[b * 8 + 1] & 15;
        const unsigned key = (xcc << 16) | (((hw >> 13) & 7) << 8) | ((hw >> 8) & 15);
        cu[key].push_back(b);
    }
    printf("%zu CUs in use\n", cu.size());
    int shown = 0, same_parity = 0, pairs = 0, distinct_simd = 0;
    for (auto &kv : cu) {
        if (kv.second.size() == 2) {
            ++pairs;
            const unsigned a = h[kv.second[0] * 8], b = h[kv.second[1] * 8];
            if (((a ^ b) & 1) == 0) ++same_parity;
        }
        for (int b : kv.second) {
            unsigned m = 0;
            for (int w = 0; w < 4; ++w) m |= 1u << ((h[(b * 4 + w) * 2] >> 4) & 3);
            if (m == 15) ++distinct_simd;
        }
        if (shown++ < 12) {
            printf("xcc %u se %u cu %2u:", kv.first >> 16, (kv.first >> 8) & 255, kv.first & 255);
            for (int b : kv.second) {
                printf("  wg %3d [", b);
                for (int w = 0; w < 4; ++w) printf(" simd%u/slot%u", (h[(b * 4 + w) * 2] >> 4) & 3, h[(b * 4 + w) * 2] & 15);
                printf(" ]");
            }
            printf("\n");
        }
    }
    printf("pairs %d, of which both first waves on the same slot parity: %d; workgroups with 4 distinct SIMDs: %d of %d\n", pairs, same_parity, distinct_simd, grid);
    return 0;
}
