// The LIST record of a (tile, Gaussian) entry as the three segment kernels evaluate it (alpha_eval, raster_render.hip): the conic and the
// opacity pre-scaled once per Gaussian by k_preprocess (ranking path: into its 32-byte gather record) or by the comparison sort's tile pass,
//     A = -0.5 log2(e) a,  B = -log2(e) b,  Cq = -0.5 log2(e) c,  lo = log2(opacity)
// so that log2(e) * power = dx (A dx + B dy) + Cq dy dy and opacity * G = exp2(that + lo).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

namespace gom_entry {
constexpr float kLog2e = 1.44269504088896340736f;

// (a, b, c, opacity) -> (A, B, Cq, lo); opacity <= 0 -> lo = -inf -> opacity * G = 0
__device__ __forceinline__ float4 entry_record(float a, float b, float c, float o) {
    return make_float4(__fmul_rn(-0.5f * kLog2e, a), __fmul_rn(-kLog2e, b), __fmul_rn(-0.5f * kLog2e, c), o > 0.f ? __log2f(o) : -INFINITY);
}
// ... and back, for the conservative culls (their margins cover the rounding of the round trip)
__device__ __forceinline__ float4 entry_unrecord(float A, float B, float Cq, float lo) {
    return make_float4(A * (-2.0f / kLog2e), B * (-1.0f / kLog2e), Cq * (-2.0f / kLog2e), __builtin_amdgcn_exp2f(lo));
}
}  // namespace gom_entry
