// storage.h -- storage-dtype abstraction of the private C8-planar volumes (device code).
//
// Arithmetic is always fp32 (fp32 MFMA, fp32 accumulators); only what is written to / read from
// HBM between stages is narrowed: mvs_dtype MVS_F32 / MVS_F16 / MVS_BF16 (BASELINE.json configs
// 2 and 4 name bf16 / fp16 volumes).  Conversions are round-to-nearest-even casts (hipcc emits
// v_cvt_f16_f32 / v_cvt_pk_bf16_f32, which keep NaNs NaN).
#pragma once
#include <hip/hip_runtime.h>

#include "mvs_abi.h"

namespace mvs {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

template <int DT>
struct St;

template <>
struct St<MVS_F32> {
    using T = float;
    static __device__ __forceinline__ f32x4 load4(const void* base, size_t idx) {
        return *reinterpret_cast<const f32x4*>(static_cast<const float*>(base) + idx);
    }
    static __device__ __forceinline__ float load1(const void* base, size_t idx) {
        return static_cast<const float*>(base)[idx];
    }
    static __device__ __forceinline__ void store4(void* base, size_t idx, f32x4 v) {
        *reinterpret_cast<f32x4*>(static_cast<float*>(base) + idx) = v;
    }
    static __device__ __forceinline__ void store1(void* base, size_t idx, float v) {
        static_cast<float*>(base)[idx] = v;
    }
};

template <>
struct St<MVS_F16> {
    using T = _Float16;
    static __device__ __forceinline__ f32x4 load4(const void* base, size_t idx) {
        const f16x4 h = *reinterpret_cast<const f16x4*>(static_cast<const _Float16*>(base) + idx);
        return (f32x4){(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
    }
    static __device__ __forceinline__ float load1(const void* base, size_t idx) {
        return (float)static_cast<const _Float16*>(base)[idx];
    }
    static __device__ __forceinline__ void store4(void* base, size_t idx, f32x4 v) {
        const f16x4 h = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
        *reinterpret_cast<f16x4*>(static_cast<_Float16*>(base) + idx) = h;
    }
    static __device__ __forceinline__ void store1(void* base, size_t idx, float v) {
        static_cast<_Float16*>(base)[idx] = (_Float16)v;
    }
};

template <>
struct St<MVS_BF16> {
    using T = __bf16;
    static __device__ __forceinline__ f32x4 load4(const void* base, size_t idx) {
        const bf16x4 h = *reinterpret_cast<const bf16x4*>(static_cast<const __bf16*>(base) + idx);
        return (f32x4){(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
    }
    static __device__ __forceinline__ float load1(const void* base, size_t idx) {
        return (float)static_cast<const __bf16*>(base)[idx];
    }
    static __device__ __forceinline__ void store4(void* base, size_t idx, f32x4 v) {
        const bf16x4 h = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
        *reinterpret_cast<bf16x4*>(static_cast<__bf16*>(base) + idx) = h;
    }
    static __device__ __forceinline__ void store1(void* base, size_t idx, float v) {
        static_cast<__bf16*>(base)[idx] = (__bf16)v;
    }
};

// 8 consecutive 16-bit elements (one voxel of one C8 plane) <-> 8 floats
typedef _Float16 f16x8s __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8s __attribute__((ext_vector_type(8)));
template <int DT>
__device__ __forceinline__ void load8_16(const void* base, size_t idx, float (&o)[8]) {
    if (DT == MVS_F16) {
        const f16x8s h = *reinterpret_cast<const f16x8s*>(static_cast<const _Float16*>(base) + idx);
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = (float)h[i];
    } else {
        const bf16x8s h = *reinterpret_cast<const bf16x8s*>(static_cast<const __bf16*>(base) + idx);
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = (float)h[i];
    }
}
template <int DT>
__device__ __forceinline__ void store8_16(void* base, size_t idx, const float (&v)[8]) {
    if (DT == MVS_F16) {
        f16x8s h;
#pragma unroll
        for (int i = 0; i < 8; ++i) h[i] = (_Float16)v[i];
        *reinterpret_cast<f16x8s*>(static_cast<_Float16*>(base) + idx) = h;
    } else {
        bf16x8s h;
#pragma unroll
        for (int i = 0; i < 8; ++i) h[i] = (__bf16)v[i];
        *reinterpret_cast<bf16x8s*>(static_cast<__bf16*>(base) + idx) = h;
    }
}

// dispatch a launcher template over the runtime dtype
#define MVS_DISPATCH_DTYPE(dtype, CALL)                                   \
    switch (dtype) {                                                      \
        case MVS_F32: { constexpr int DT = MVS_F32; return CALL; }        \
        case MVS_F16: { constexpr int DT = MVS_F16; return CALL; }        \
        case MVS_BF16: { constexpr int DT = MVS_BF16; return CALL; }      \
        default: return fail(MVS_ERR_BAD_DTYPE, "unknown dtype %d", dtype); \
    }

}  // namespace mvs
