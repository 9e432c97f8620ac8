// split_ops.h -- device helpers of the SPLIT-OPERAND kernels (conv3d_mfma16.hip: convgs; conv11_prob.hip: the fused tail):
// an fp32 value as the sum of three bf16 numbers (RNE, exact residuals) and the bf16 MFMA the six cross products run on.
#pragma once
#include <hip/hip_runtime.h>

#include "mvs_internal.h"
#include "storage.h"

namespace mvs {

typedef unsigned int g_u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 g_bf16x8 __attribute__((ext_vector_type(8)));
typedef float g_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 g_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ g_f32x2 gs_stage(const g_f32x2 a, unsigned& packed) {   // packed = bf16x2(a) RNE; returns a - packed
    const g_bf16x2 h = __builtin_convertvector(a, g_bf16x2);
    packed = __builtin_bit_cast(unsigned, h);
    const g_f32x2 w = {__uint_as_float(packed << 16), __uint_as_float(packed & 0xFFFF0000u)};
    return a - w;
}
// 8 fp32 channels of one voxel -> three 16-byte bf16 fragments
__device__ __forceinline__ void gs_split8(const f32x4 lo, const f32x4 hi, g_u32x4& p1, g_u32x4& p2, g_u32x4& p3) {
    const g_f32x2 v[4] = {{lo.x, lo.y}, {lo.z, lo.w}, {hi.x, hi.y}, {hi.z, hi.w}};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        unsigned a, b;
        const g_f32x2 r1 = gs_stage(v[j], a);
        const g_f32x2 r2 = gs_stage(r1, b);
        p1[j] = a;
        p2[j] = b;
        p3[j] = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, g_bf16x2));
    }
}
__device__ __forceinline__ f32x4 gs_mfma(g_u32x4 a, g_u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(g_bf16x8, a), __builtin_bit_cast(g_bf16x8, b), c, 0, 0, 0);
}

}  // namespace mvs
