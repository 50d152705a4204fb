// One pixel of the fused photometric L1 losses (unpack, train.py:53-55, + L1 rgb / L1 mask, train.py:101-111) with its gradient.
// Shared by the stand-alone loss kernel (loss.hip) and by the riders that carry the frame step's loss inside the rasterizer's forward
// (k_emit's painters for the empty tiles, k_combine_fwd for the others: GomLossRider) -- every rounding is spelled out (explicit fma /
// mul / sub), so the three call sites give the same bits whatever the compiler would contract around them.
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
    GomL1Px o;
    const float t = __fsub_rn(1.f, m);
    const float as0 = __fmul_rn(a0, s), as1 = __fmul_rn(a1, s), as2 = __fmul_rn(a2, s);
    const float r0 = __fsub_rn(__fmaf_rn(as0, m, __fmul_rn(b0, t)), g0);
    const float r1 = __fsub_rn(__fmaf_rn(as1, m, __fmul_rn(b1, t)), g1);
    const float r2 = __fsub_rn(__fmaf_rn(as2, m, __fmul_rn(b2, t)), g2);
    const float rm = __fsub_rn(m, gm);
    o.abs_rgb = __fadd_rn(__fadd_rn(fabsf(r0), fabsf(r1)), fabsf(r2));
    o.abs_mask = fabsf(rm);
    const float s0 = __fmul_rn(gom_sgn(r0), k_rgb), s1 = __fmul_rn(gom_sgn(r1), k_rgb), s2 = __fmul_rn(gom_sgn(r2), k_rgb);
    const float sm = __fmul_rn(s, m);
    o.d0 = __fmul_rn(s0, sm);
    o.d1 = __fmul_rn(s1, sm);
    o.d2 = __fmul_rn(s2, sm);
    o.d3 = __fmaf_rn(s0, __fsub_rn(as0, b0), __fmaf_rn(s1, __fsub_rn(as1, b1), __fmaf_rn(s2, __fsub_rn(as2, b2), __fmul_rn(gom_sgn(rm), k_mask))));
    o.dshade = __fmul_rn(__fmaf_rn(s0, a0, __fmaf_rn(s1, a1, __fmul_rn(s2, a2))), m);
    return o;
}
