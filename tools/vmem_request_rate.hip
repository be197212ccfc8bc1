// What bounds a producer wave's requests: the number of vector-memory instructions, or the bytes they carry?  One workgroup of
// four waves per CU (one per SIMD) on every CU; every wave issues REQ loads per iteration into distinct registers, waits for
// them, repeats.  Patterns (a "run" = lanes reading consecutive bytes of one 128-byte line, as a tile's channels do):
//   dword  x 2 runs of 128 B     (csrc/conv_wino43w.hip, second cut)      b64 x 4 runs of 128 B   (shipped)
//   b128   x 8 runs of 128 B                                               b128 x 1 run of 1 KB / dword x 1 run of 256 B (contiguous)
// The working set per workgroup is a few hundred KB walked repeatedly (L2 hits), or `big` = a stride that defeats the L2.
//   hipcc -O3 --offload-arch=gfx950 tools/vmem_request_rate.hip -o tools/scratch/vrr.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
template <int WIDTH, int RUNS>      // WIDTH dwords per lane; RUNS contiguous runs per instruction (lanes / RUNS lanes each)
__global__ __launch_bounds__(256) void k(const float *buf, long bytes, unsigned stride_req, unsigned stride_run, int iters, float *out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(buf), 0, (int)bytes, 0x00020000);
    const int lpr = 64 / RUNS;                                        // lanes per run
    unsigned base = (unsigned)(((blockIdx.x * 4 + wave) * 131072u) % (unsigned)(bytes / 2)) + (unsigned)(lane / lpr) * stride_run +
                    (unsigned)(lane % lpr) * WIDTH * 4;
    float acc = 0.f;
    constexpr int REQ = 32;
    for (int it = 0; it < iters; ++it) {
        if (WIDTH == 1) {
            unsigned v[REQ];
#pragma unroll
            for (int q = 0; q < REQ; ++q) v[q] = __builtin_amdgcn_raw_buffer_load_b32(r, base + q * stride_req, 0, 0);
#pragma unroll
            for (int q = 0; q < REQ; ++q) acc += __uint_as_float(v[q]);
        } else if (WIDTH == 2) {
            u32x2 v[REQ];
#pragma unroll
            for (int q = 0; q < REQ; ++q) v[q] = __builtin_amdgcn_raw_buffer_load_b64(r, base + q * stride_req, 0, 0);
#pragma unroll
            for (int q = 0; q < REQ; ++q) acc += __uint_as_float(v[q].x) + __uint_as_float(v[q].y);
        } else {
            u32x4 v[REQ];
#pragma unroll
            for (int q = 0; q < REQ; ++q) v[q] = __builtin_amdgcn_raw_buffer_load_b128(r, base + q * stride_req, 0, 0);
#pragma unroll
            for (int q = 0; q < REQ; ++q) acc += __uint_as_float(v[q].x) + __uint_as_float(v[q].w);
        }
        base += REQ * stride_req;
        if (base > (unsigned)(bytes - (REQ + 1) * stride_req - 65536)) base -= (unsigned)(bytes / 2);
    }
    if (acc == 1.2345f) out[threadIdx.x] = acc;
}
template <int WIDTH, int RUNS> void run(const char *what, const float *buf, long bytes, unsigned stride_req, unsigned stride_run, float *out) {
    const int iters = 200;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) k<WIDTH, RUNS><<<256, 256>>>(buf, bytes, stride_req, stride_run, iters, out);
    (void)hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) k<WIDTH, RUNS><<<256, 256>>>(buf, bytes, stride_req, stride_run, iters, out);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 200.0, ninstr = 4.0 * 32 * iters;                       // per CU and launch
    const double bpi = 64.0 * WIDTH * 4;
    printf("%-44s %8.1f us  %6.1f ns per instruction and CU  %6.1f B/ns per CU  %6.2f TB/s over the chip\n", what, us, us * 1e3 / ninstr,
           ninstr * bpi / (us * 1e3), 256.0 * ninstr * bpi / (us * 1e-6) / 1e12);
}
int main() {
    const long bytes = 1L << 30;
    float *buf, *out; (void)hipMalloc(&buf, bytes); (void)hipMalloc(&out, 4096); (void)hipMemset(buf, 0, bytes);
    printf("four waves per CU, 32 requests in flight per wave; requests of a wave 2 KB apart (a tensor row), runs 16 KB apart (another tile)\n");
    run<1, 2>("dword, 2 runs of 128 B", buf, bytes, 2048, 16384, out);
    run<2, 4>("8 bytes, 4 runs of 128 B", buf, bytes, 2048, 16384, out);
    run<4, 8>("16 bytes, 8 runs of 128 B", buf, bytes, 2048, 16384, out);
    run<1, 1>("dword, contiguous 256 B", buf, bytes, 2048, 0, out);
    run<2, 1>("8 bytes, contiguous 512 B", buf, bytes, 2048, 0, out);
    run<4, 1>("16 bytes, contiguous 1 KB", buf, bytes, 2048, 0, out);
    run<4, 4>("16 bytes, 4 runs of 256 B", buf, bytes, 2048, 16384, out);
    run<4, 2>("16 bytes, 2 runs of 512 B", buf, bytes, 2048, 16384, out);
    return 0;
}
