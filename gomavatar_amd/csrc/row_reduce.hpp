// Sum of 10 registers over each 16-lane ROW of a wave, the four rows independently (four different work items per wave),
// transposed so that a level of the tree halves the number of live registers:
//   level A  lane ^ 8   (row_ror:8)            banks {0,1} keep the even value of a pair, banks {2,3} the odd one    10 -> 5 registers
//   level B  lane ^ 7   (row_half_mirror)      banks {0,2} keep the first register of a pair, banks {1,3} the second   5 -> 3
//   level C, D          (quad_perm butterflies) every lane of a quad ends with the quad's total = the ROW total        3 registers
// 21 v_add_f32_dpp for 10 values x 4 rows (5.25 per value and row item; the 64-lane tree of k_seg_bwd is 34 for ONE item).
// DPP bank_mask enables the destination write per bank = quad of lanes (ISA: bit i <-> lanes [4i, 4i + 3] of every row); a
// disabled lane keeps the destination's old value, which is how two source registers fold into one.
// Where the totals land (all four lanes of the bank hold the same number):
//   t0: bank 0..3 = values 0, 2, 1, 3      t1: bank 0..3 = values 4, 6, 5, 7      t2: banks 0, 1 = value 8, banks 2, 3 = value 9
// Checked on the hardware by scripts/ubench/row_reduce_probe.hip.  Deterministic: a fixed tree, no atomics.
#pragma once
#include <hip/hip_runtime.h>

__device__ __forceinline__ void row_sum10_t(const float (&w)[10], float &t0, float &t1, float &t2) {
    float a0, a1, a2, a3, a4, b0, b1, b2;
    asm volatile(
        "s_nop 1\n"
        // level A: lane ^ 8
        "v_add_f32_dpp %0, %8, %8 row_ror:8 row_mask:0xf bank_mask:0x3\n"
        "v_add_f32_dpp %1, %10, %10 row_ror:8 row_mask:0xf bank_mask:0x3\n"
        "v_add_f32_dpp %2, %12, %12 row_ror:8 row_mask:0xf bank_mask:0x3\n"
        "v_add_f32_dpp %3, %14, %14 row_ror:8 row_mask:0xf bank_mask:0x3\n"
        "v_add_f32_dpp %4, %16, %16 row_ror:8 row_mask:0xf bank_mask:0x3\n"
        "v_add_f32_dpp %0, %9, %9 row_ror:8 row_mask:0xf bank_mask:0xc\n"
        "v_add_f32_dpp %1, %11, %11 row_ror:8 row_mask:0xf bank_mask:0xc\n"
        "v_add_f32_dpp %2, %13, %13 row_ror:8 row_mask:0xf bank_mask:0xc\n"
        "v_add_f32_dpp %3, %15, %15 row_ror:8 row_mask:0xf bank_mask:0xc\n"
        "v_add_f32_dpp %4, %17, %17 row_ror:8 row_mask:0xf bank_mask:0xc\n"
        "s_nop 1\n"
        // level B: lane ^ 7 inside each half row
        "v_add_f32_dpp %5, %0, %0 row_half_mirror row_mask:0xf bank_mask:0x5\n"
        "v_add_f32_dpp %6, %2, %2 row_half_mirror row_mask:0xf bank_mask:0x5\n"
        "v_add_f32_dpp %7, %4, %4 row_half_mirror row_mask:0xf bank_mask:0xf\n"
        "v_add_f32_dpp %5, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xa\n"
        "v_add_f32_dpp %6, %3, %3 row_half_mirror row_mask:0xf bank_mask:0xa\n"
        "s_nop 1\n"
        // levels C, D: inside the quad
        "v_add_f32_dpp %5, %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
        "v_add_f32_dpp %6, %6, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
        "v_add_f32_dpp %7, %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
        "s_nop 0\n"
        "v_add_f32_dpp %5, %5, %5 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
        "v_add_f32_dpp %6, %6, %6 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
        "v_add_f32_dpp %7, %7, %7 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
        : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(a4), "=&v"(b0), "=&v"(b1), "=&v"(b2)
        : "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]), "v"(w[5]), "v"(w[6]), "v"(w[7]), "v"(w[8]), "v"(w[9]));
    t0 = b0; t1 = b1; t2 = b2;
}

// Which record value (0..9; 10, 11 = padding slots) the lane's share of (t0, t1, t2) is, when lane (l & 3) == j hands in t_j:
// j = 3, and j = 2 in the odd banks, hand in nothing (slots 10 / 11).
__device__ __forceinline__ int row_sum10_slot(int lane) {
    const int l = lane & 15, bank = l >> 2, j = l & 3;
    const int perm[4] = {0, 2, 1, 3};
    if (j == 0) return perm[bank];
    if (j == 1) return 4 + perm[bank];
    if (j == 2) return (bank & 1) ? 10 : 8 + (bank >> 1);
    return 11;
}

// LDS float add without a return value (ds_add_f32).  Lanes of ONE instruction that hit the same address are applied one after the
// other by the LDS pipeline in a fixed order; on a wave-private accumulator the result is therefore a function of the inputs alone.
__device__ __forceinline__ void lds_add_f32(float *p, float v) {
    const uint32_t a = (uint32_t)(uintptr_t)p;   // LDS addresses are 32-bit offsets
    asm volatile("ds_add_f32 %0, %1" : : "v"(a), "v"(v) : "memory");
}
