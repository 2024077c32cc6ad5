// mos_common.h — shared device helpers for the gfx950 (CDNA4) kernels of libmos_hip.
// Wave = 64 lanes everywhere; MFMA fragment conventions are documented at each use.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mos_hip.h"

typedef _Float16 f16_t;
typedef __bf16 bf16_t;

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

// ---- per-dtype traits: 8-wide / 4-wide MFMA operand vectors and the MFMA builtins ----------
template <typename T> struct MT;

template <> struct MT<f16_t> {
    typedef __attribute__((ext_vector_type(8))) _Float16 v8;
    typedef __attribute__((ext_vector_type(4))) _Float16 v4;
    static __device__ __forceinline__ f32x16 mfma32(v8 a, v8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x4 mfma16(v8 a, v8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x4 mfma16k16(v4 a, v4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0);
    }
};

template <> struct MT<bf16_t> {
    typedef __attribute__((ext_vector_type(8))) __bf16 v8;
    typedef __attribute__((ext_vector_type(4))) __bf16 v4;
    typedef __attribute__((ext_vector_type(4))) short s4;
    static __device__ __forceinline__ f32x16 mfma32(v8 a, v8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x4 mfma16(v8 a, v8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x4 mfma16k16(v4 a, v4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s4, a),
                                                         __builtin_bit_cast(s4, b), c, 0, 0, 0);
    }
};

// 16-byte / 8-byte raw views used for global<->LDS staging.
template <typename T>
__device__ __forceinline__ typename MT<T>::v8 as_v8(u32x4 x) {
    return __builtin_bit_cast(typename MT<T>::v8, x);
}
template <typename T>
__device__ __forceinline__ u32x4 from_v8(typename MT<T>::v8 x) {
    return __builtin_bit_cast(u32x4, x);
}

__device__ __forceinline__ u32x4 ld16(const void* p) { return *reinterpret_cast<const u32x4*>(p); }
__device__ __forceinline__ u32x2 ld8(const void* p) { return *reinterpret_cast<const u32x2*>(p); }
__device__ __forceinline__ void st16(void* p, u32x4 v) { *reinterpret_cast<u32x4*>(p) = v; }
__device__ __forceinline__ void st8(void* p, u32x2 v) { *reinterpret_cast<u32x2*>(p) = v; }

// Pack 4 floats to 4 T (round-to-nearest-even) as 8 bytes.
template <typename T>
__device__ __forceinline__ u32x2 pack4(float a, float b, float c, float d) {
    typename MT<T>::v4 v;
    v[0] = (T)a; v[1] = (T)b; v[2] = (T)c; v[3] = (T)d;
    return __builtin_bit_cast(u32x2, v);
}
template <typename T>
__device__ __forceinline__ typename MT<T>::v8 pack8(const float* p) {
    typename MT<T>::v8 v;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (T)p[i];
    return v;
}

// 16-bit element i of a 16-byte chunk, as raw bits.
__device__ __forceinline__ uint32_t half_of(u32x4 v, int i) {
    uint32_t w = v[i >> 1];
    return (i & 1) ? (w >> 16) : (w & 0xffffu);
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// ---- buffer-resource loads: rows / bytes past the end of the described range read as ZERO in hardware ----------
typedef __amdgpu_buffer_rsrc_t rsrc_t;
// descriptor of `bytes` valid bytes at p; p and bytes are wave-uniform (made provably so: cdna guide T20)
__device__ __forceinline__ rsrc_t make_rsrc(const void* p, uint32_t bytes) {
    const uint64_t a = (uint64_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)hi << 32) | lo), 0,
                                             (int)__builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
__device__ __forceinline__ u32x4 ldbuf16(rsrc_t src, int byte_off) {
    return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(src, byte_off, 0, 0));
}
// LDS-DMA: 16 B per lane straight from the buffer into LDS at (wave-uniform base) + lane * 16 -- no VGPRs, no ds_write.
// Out-of-range lanes deposit zeros. Completion is tracked by vmcnt; a barrier makes it visible to the other waves.
typedef __attribute__((address_space(3))) void lds_void_t;
__device__ __forceinline__ void dma16(rsrc_t src, void* lds_wave_base, int byte_off) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(src, (lds_void_t*)lds_wave_base, 16, byte_off, 0, 0, 0);
}
// the same with a wave-uniform part of the offset in the instruction's scalar-offset operand (no VALU add per piece)
__device__ __forceinline__ void dma16s(rsrc_t src, void* lds_wave_base, int lane_byte_off, int uniform_byte_off) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(src, (lds_void_t*)lds_wave_base, 16, lane_byte_off, uniform_byte_off, 0, 0);
}

// ---- host-side error plumbing (defined in mos_api.hip) --------------------------------------
int mos_set_error(int code, const char* fmt, ...);
int mos_check_launch(const char* what);

#define MOS_REQUIRE(cond, ...)                                   \
    do {                                                         \
        if (!(cond)) return mos_set_error(MOS_ERR_BAD_ARG, __VA_ARGS__); \
    } while (0)

// ---- optional per-kernel timing with HIP events on the launch stream (bench.py roofline) ----------------
// Usage at a launch site:  { MosProfScope p(stream, "kernel_name", "shape key", flops, bytes); launch...; }
// Disabled (default) it costs one relaxed load. Defined in mos_api.hip.
struct MosProfScope {
    MosProfScope(hipStream_t st, const char* kernel, const char* key, double flops, double bytes);
    ~MosProfScope();
    int slot;
    hipStream_t st;
};
