// One pixel of the fused photometric L1 losses (unpack, train.py:53-55, + L1 rgb / L1 mask, train.py:101-111) with its gradient.
// Shared by the stand-alone loss kernel (loss.hip) and by the riders that carry the frame step's loss inside the rasterizer's forward
// (k_combine_fwd's tile workgroups and its empty-tile riders: GomLossRider) -- every rounding is spelled out (contraction off, the fmas
// explicit), so the call sites give the same bits whatever the compiler would contract around them.
#pragma once
#include <hip/hip_runtime.h>

struct GomL1Px {
    float d0, d1, d2, d3;   // dL/d(pred) of the four planes
    float dshade;           // dL/d(shade) (callers without a shade plane ignore it)
    float abs_rgb, abs_mask;
};

__device__ __forceinline__ float gom_sgn(float x) { return (x > 0.f) ? 1.f : ((x < 0.f) ? -1.f : 0.f); }

// a0..a2, m: the rasterizer's planes; s: shade (1 without); b: background of the unpack; g, gm: targets
__device__ __forceinline__ GomL1Px gom_l1_pixel(float a0, float a1, float a2, float m, float s, float b0, float b1, float b2, float g0, float g1, float g2,
                                                float gm, float k_rgb, float k_mask) {
#pragma clang fp contract(off)   // plain operators below are rounded one by one (HIP's __fmul_rn / __fadd_rn are inline functions the compiler may still fuse); the fmas are explicit
    GomL1Px o;
    const float t = 1.f - m;
    const float as0 = a0 * s, as1 = a1 * s, as2 = a2 * s;
    const float r0 = __builtin_fmaf(as0, m, b0 * t) - g0;
    const float r1 = __builtin_fmaf(as1, m, b1 * t) - g1;
    const float r2 = __builtin_fmaf(as2, m, b2 * t) - g2;
    const float rm = m - gm;
    o.abs_rgb = (fabsf(r0) + fabsf(r1)) + fabsf(r2);
    o.abs_mask = fabsf(rm);
    const float s0 = gom_sgn(r0) * k_rgb, s1 = gom_sgn(r1) * k_rgb, s2 = gom_sgn(r2) * k_rgb;
    const float sm = s * m;
    o.d0 = s0 * sm;
    o.d1 = s1 * sm;
    o.d2 = s2 * sm;
    o.d3 = __builtin_fmaf(s0, as0 - b0, __builtin_fmaf(s1, as1 - b1, __builtin_fmaf(s2, as2 - b2, gom_sgn(rm) * k_mask)));
    o.dshade = __builtin_fmaf(s0, a0, __builtin_fmaf(s1, a1, s2 * a2)) * m;
    return o;
}
