// Shared host/device helpers for libmmdgan_hip (gfx950 only; no portability layers).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <functional>
#include <tuple>
#include <type_traits>

#include "../../include/mmdgan_hip.h"

namespace mmdgan {

constexpr int kWave = 64;          // CDNA wavefront
constexpr float kLreluAlpha = 0.1f;   // layer_func.py:112
constexpr float kEpsi = 1e-10f;       // misc_fun.py:29 FLAGS.EPSI

void set_error(const char *fmt, ...);
void *workspace(size_t need);     // caller-registered scratch of the current handle (mmdgan_set_workspace) or nullptr: availability only
// scratch for a launch on `st` that is about to USE it: a half of the buffer (the whole of it if `need` does not fit a half),
// ordered behind that part's previous user if that was another stream (core.hip).  Batch-1 launches (the power iteration's,
// issued on concurrent streams) never take a workspace path at all.
void *workspace_acquire(size_t need, hipStream_t st);
bool outputs_prezeroed();         // mmdgan_set_outputs_prezeroed of the current handle: skip internal zeroing memsets

// ---- launch plans (mmdgan_plan_*): every kernel launch, memset and stream dependency the library issues goes through the
// three functions below.  Normally they just issue; while the calling thread's handle is RECORDING they also append a
// node that re-issues exactly the same work (same kernel, grid, arguments by value, same stream) - a step whose launch
// sequence is static can then be replayed from one C call without its ~200 host-side entry calls.
bool plan_recording();
void plan_push(std::function<void()> &&node);
void plan_note_kernel(const void *host_fn, dim3 grid, dim3 block, hipStream_t st);   // what mmdgan_plan_describe lists
void plan_note_collective();      // the plan being recorded holds a collective of the library's current communicator
hipError_t memset_async(void *p, int value, size_t bytes, hipStream_t st);
inline hipError_t zero_output(void *p, size_t bytes, hipStream_t st) {
    return outputs_prezeroed() ? hipSuccess : memset_async(p, 0, bytes, st);
}

template <typename... KArgs, typename... Args>
inline void launch_k(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t shmem, hipStream_t st, Args &&...args) {
    // convert to the kernel's own parameter types first: a recorded node must hold exactly what the kernel receives
    std::tuple<std::decay_t<KArgs>...> pack{static_cast<std::decay_t<KArgs>>(args)...};
    std::apply([&](auto &...a) { kernel<<<grid, block, shmem, st>>>(a...); }, pack);
    if (plan_recording()) {
        plan_note_kernel((const void *)kernel, grid, block, st);
        plan_push([kernel, grid, block, shmem, st, pack]() mutable {
            std::apply([&](auto &...a) { kernel<<<grid, block, shmem, st>>>(a...); }, pack);
        });
    }
}

inline int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return MMDGAN_E_LAUNCH;
    }
    return MMDGAN_OK;
}

// every launch site of the library is spelled hipLaunchKernelGGL(kernel, grid, block, lds, stream, args...): route them
// through launch_k (this is not a portability shim - there is one backend - but the single choke point plans need)
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    ::mmdgan::launch_k(kernel, dim3(grid), dim3(block), (size_t)(shmem), (hipStream_t)(stream), ##__VA_ARGS__)

#define MMDGAN_REQUIRE(cond, ...)            \
    do {                                     \
        if (!(cond)) {                       \
            mmdgan::set_error(__VA_ARGS__);  \
            return MMDGAN_E_ARG;             \
        }                                    \
    } while (0)

// forward activation (layer_func.py:104-151)
__device__ __forceinline__ float act_fwd(float v, int act) {
    switch (act) {
        case MMDGAN_ACT_RELU: return v > 0.f ? v : 0.f;
        case MMDGAN_ACT_LRELU: return v > 0.f ? v : v * kLreluAlpha;
        case MMDGAN_ACT_TANH: return tanhf(v);
        default: return v;
    }
}
// derivative of the activation expressed through its OUTPUT y (lrelu alpha>0 keeps the sign)
__device__ __forceinline__ float act_bwd_from_out(float y, int act) {
    switch (act) {
        case MMDGAN_ACT_RELU: return y > 0.f ? 1.f : 0.f;
        case MMDGAN_ACT_LRELU: return y > 0.f ? 1.f : kLreluAlpha;
        case MMDGAN_ACT_TANH: return 1.f - y * y;
        default: return 1.f;
    }
}

// 'SAME' geometry of tf.nn.conv2d
struct ConvDims {
    int N, H, W, C, K, R, stride, P, Q, pad;
};
inline ConvDims conv_dims(const mmdgan_conv_geom &g) {
    ConvDims d;
    d.N = g.N; d.H = g.H; d.W = g.W; d.C = g.C; d.K = g.K; d.R = g.R; d.stride = g.stride;
    d.P = (g.H + g.stride - 1) / g.stride;
    d.Q = (g.W + g.stride - 1) / g.stride;
    int total = (d.P - 1) * g.stride + g.R - g.H;
    if (total < 0) total = 0;
    d.pad = total / 2;
    return d;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Workgroup barrier that publishes LDS only.  __syncthreads() is a workgroup-scope release / acquire fence + s_barrier, and on
// gfx9-family parts the fence's release half is `s_waitcnt vmcnt(0)`: every global load a wave has in flight - the operands it
// requested for a LATER stage - must land before the wave may even arrive at the barrier, which turns a prefetch across a
// barrier into a blocking load (conv_wino43w.hip: the window was as long as a request's latency, tools/wino43w_trace.py).
// Where waves of a workgroup hand each other data through LDS only, this is the barrier: the wave's own LDS accesses have
// completed (lgkmcnt(0)), the compiler moves no memory access across it, global loads stay in flight.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

}  // namespace mmdgan
