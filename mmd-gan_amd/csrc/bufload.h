// Buffer-resource loads for gfx950.  Gathers go through buffer loads: the hardware range check
// returns 0 for an offset beyond num_records, so zero padding ('SAME' borders, ragged last tile) is
// an offset select instead of a divergent branch around the load.  The descriptor is built from
// kernel arguments only (wave-uniform, so no waterfall loop is generated).
#pragma once
#include <hip/hip_runtime.h>

namespace mmdgan {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned kOOB = 0x80000000u;          // every tensor here is < 2 GiB

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float *base, long bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float4 bufld4(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float2 bufld2(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, byte_off, 0, 0);
    return make_float2(__uint_as_float(v.x), __uint_as_float(v.y));
}
__device__ __forceinline__ float bufld1(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, 0));
}

// voffset (per lane, range-checked) + soffset (wave-uniform SGPR, not range-checked): no VALU add per load
__device__ __forceinline__ float bufld1s(__amdgpu_buffer_rsrc_t r, unsigned byte_off, unsigned s_off) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, byte_off, s_off, 0));
}

__device__ __forceinline__ float4 bufld4s(__amdgpu_buffer_rsrc_t r, unsigned byte_off, unsigned s_off) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, s_off, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

// 16-byte store through a buffer resource: an offset beyond num_records is dropped by the hardware (no branch for ragged tiles)
__device__ __forceinline__ void bufst4(__amdgpu_buffer_rsrc_t r, unsigned byte_off, float4 v) {
    u32x4 d;
    d.x = __float_as_uint(v.x); d.y = __float_as_uint(v.y); d.z = __float_as_uint(v.z); d.w = __float_as_uint(v.w);
    __builtin_amdgcn_raw_buffer_store_b128(d, r, byte_off, 0, 0);
}

__device__ __forceinline__ void bufst1(__amdgpu_buffer_rsrc_t r, unsigned byte_off, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, byte_off, 0, 0);
}

}  // namespace mmdgan
