// The replay loop of the render backward, shared by k_seg_bwd and k_seg_bwd_pair (included inside raster_render.hip's anonymous namespace).
//
// Round 3, second pass.  The survivors of a piece are compacted into PAIR records (as in k_seg_T) and two entries are replayed per trip
// on register pairs: alpha (alpha_eval), the colour dot products, the weights and the six geometry moments of both entries as
// v_pk_mul / v_pk_fma / v_pk_add_f32; one LDS address per pair (five broadcast ds_read_b128 off one register); the 20 values of the two
// entries down ONE transposed tree (15 lane swaps, 8 packed adds, 12 bank-masked DPP adds: after the swaps 20 values are exactly 5
// registers x 4 rows, and the in-row levels fill all four banks); the record address a scalar multiply + one add.
// What it bought, measured ("Round 3, second pass"): 63 M -> 48 M VALU instructions per 8-frame launch and 158 -> 151 us.
// The instruction count is not what bounds this kernel -- scripts/ubench/valu_rate.hip: a packed fp32 operation issues in 4.2 cycles
// against 2.25 for a plain one (no gain per flop), a lane swap in 8, a compare / select / DPP add in ~4; knock-out builds (GOM_KO_REPLAY):
// 60 us without the replay at all, +58 for the alphas, +19 for the gradient terms, +21 for the reduction.  Built on top and measured
// slower, each parity-green: the next pair's evaluation software-pipelined into the current pair's reduction (165 us at 127 registers;
// 185 at 96 with spills in the loop), both pieces' loads issued ahead of the first replay (177), the next task's descriptors fetched
// through LDS a task ahead (165-182), the forward's "some pixel blended this entry" bits narrowing the survivor list (-24 % entries
// evaluated: 148 us, but k_seg_fwd +7 for producing them).
// Per value the summation tree is the one wave_sum10_banks had (lane ^ 32, ^ 16, ^ 8, ^ 7, ^ 1, ^ 2).
#pragma once

#define GOM_BPAIR_F4 (5 * (GOM_SUB_MAX / 2))   // float4 per wave slab: (x0 x1 y0 y1)(a0 a1 b0 b1)(c0 c1 o0 o1)(r0 r1 g0 g1)(b0 b1 m0 m1) per pair of survivors

// Survivors of the piece, compacted in list order into pair records; an odd count is padded with a null entry (opacity 0 -> alpha 0).
// Returns the number of pairs.  (LDS operations of one wave execute in order: the replay reads the slab without a barrier.)
template <int C>
__device__ __forceinline__ uint32_t stage_bwd_pairs(float4 *slab, const EntryRegs<C> &r, unsigned long long mask, int lane) {
    float *f = reinterpret_cast<float *>(slab);
    const uint32_t n = (uint32_t)__popcll(mask);
    const uint32_t pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
    if (r.keep) {
        float *d = f + 20u * (pos >> 1) + (pos & 1u);
        d[0] = r.x; d[2] = r.y; d[4] = r.a; d[6] = r.b; d[8] = r.c; d[10] = r.o;
        d[12] = r.col[0];
        d[14] = C > 1 ? r.col[1 % C] : 0.f;
        d[16] = C > 2 ? r.col[2 % C] : 0.f;
        d[18] = C > 3 ? r.col[3 % C] : 0.f;
    }
    if ((n & 1u) && lane == 0) {
        float *d = f + 20u * (n >> 1) + 1u;
#pragma unroll
        for (int i = 0; i < 10; i++) d[2 * i] = i == 5 ? -INFINITY : 0.f;   // (lo = log2 of opacity 0)
    }
    return (n + 1u) >> 1;
}

__device__ __forceinline__ v2f swap_add32_2(v2f a, v2f b) {
    const auto rx = __builtin_amdgcn_permlane32_swap(__float_as_uint(a.x), __float_as_uint(b.x), false, false);
    const auto ry = __builtin_amdgcn_permlane32_swap(__float_as_uint(a.y), __float_as_uint(b.y), false, false);
    const v2f lo = {__uint_as_float(rx[0]), __uint_as_float(ry[0])}, hi = {__uint_as_float(rx[1]), __uint_as_float(ry[1])};
    return lo + hi;
}
__device__ __forceinline__ v2f swap_add16_2(v2f a, v2f b) {
    const auto rx = __builtin_amdgcn_permlane16_swap(__float_as_uint(a.x), __float_as_uint(b.x), false, false);
    const auto ry = __builtin_amdgcn_permlane16_swap(__float_as_uint(a.y), __float_as_uint(b.y), false, false);
    const v2f lo = {__uint_as_float(rx[0]), __uint_as_float(ry[0])}, hi = {__uint_as_float(rx[1]), __uint_as_float(ry[1])};
    return lo + hi;
}

// Sums of ten register PAIRS (.x = the first entry of the pair record, .y = the second) over the 64 lanes.  Where the 20 totals land:
//   n0, lane (row r, bank b), all four lanes of the quad alike:  entry b >> 1,  value (b & 1 ? {4, 6, 5, 7} : {0, 2, 1, 3})[r]
//   n1, every lane of row r:                                     entry r & 1,   value 8 + (r >> 1)
__device__ __forceinline__ void wave_sum20_banks(const v2f (&z)[10], float &n0_out, float &n1_out) {
    const v2f p0 = swap_add32_2(z[0], z[1]), p1 = swap_add32_2(z[2], z[3]), p2 = swap_add32_2(z[4], z[5]), p3 = swap_add32_2(z[6], z[7]);
    const v2f p4 = swap_add32_2(z[8], z[9]);
    const v2f s0 = swap_add16_2(p0, p1), s1 = swap_add16_2(p2, p3);
    const float s2 = swap_add16(p4.x, p4.y);
    float m0, m1, m2, n0, n1;
    asm("s_nop 1\n"
                 "v_add_f32_dpp %0, %5, %5 row_ror:8 row_mask:0xf bank_mask:0x3\n"           // m0: banks 0,1 <- s0.x (lane ^ 8)
                 "v_add_f32_dpp %1, %7, %7 row_ror:8 row_mask:0xf bank_mask:0x3\n"           // m1: banks 0,1 <- s1.x
                 "v_add_f32_dpp %2, %9, %9 row_ror:8 row_mask:0xf bank_mask:0xf\n"           // m2: s2
                 "v_add_f32_dpp %0, %6, %6 row_ror:8 row_mask:0xf bank_mask:0xc\n"           // m0: banks 2,3 <- s0.y
                 "v_add_f32_dpp %1, %8, %8 row_ror:8 row_mask:0xf bank_mask:0xc\n"           // m1: banks 2,3 <- s1.y
                 "s_nop 1\n"
                 "v_add_f32_dpp %3, %0, %0 row_half_mirror row_mask:0xf bank_mask:0x5\n"     // n0: banks 0,2 <- m0 (lane ^ 7)
                 "v_add_f32_dpp %4, %2, %2 row_half_mirror row_mask:0xf bank_mask:0xf\n"     // n1: m2
                 "v_add_f32_dpp %3, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xa\n"     // n0: banks 1,3 <- m1
                 "s_nop 1\n"
                 "v_add_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                 "v_add_f32_dpp %4, %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                 "s_nop 0\n"
                 "v_add_f32_dpp %3, %3, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
                 "v_add_f32_dpp %4, %4, %4 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
                 : "=&v"(m0), "=&v"(m1), "=&v"(m2), "=&v"(n0), "=&v"(n1)
                 : "v"(s0.x), "v"(s0.y), "v"(s1.x), "v"(s1.y), "v"(s2));
    n0_out = n0;
    n1_out = n1;
}
// Float offset inside the 20-float double record of a pair (entry 0 at 0..9, entry 1 at 10..19) that this lane's share of wave_sum20_banks goes
// to, or -1; `from_n1`: the lane stores n1 (lanes 1 of every row), otherwise n0 (lanes 0, 4, 8, 12).
__device__ __forceinline__ int wave_sum20_slot(int lane, bool &from_n1) {
    const int l = lane & 15, r = lane >> 4;
    from_n1 = l == 1;
    if (l == 1) return 10 * (r & 1) + 8 + (r >> 1);
    if (l & 3) return -1;
    const int b = l >> 2;
    const int v = (((r & 1) << 1) | (r >> 1)) + ((b & 1) ? 4 : 0);   // rows 0..3 -> 0, 2, 1, 3
    return 10 * (b >> 1) + v;
}

// Replays the staged pairs back to front.  `acc`: the piece's record rows, 10 floats per compacted survivor; `pos_limit`: survivors
// 0 .. pos_limit-1 lie in front of this pixel's last contributor.  Returns the bit set of PAIRS whose two rows were written.
template <int C>
__device__ __forceinline__ unsigned long long bwd_replay(const float4 *slab, uint32_t npairs, float *acc, const float (&dpix)[C], float pfx, float pfy,
                                                         float T, float R_acc, float T_final, float bg_dot, uint32_t pos_limit, int slot20, bool from_n1
) {
#pragma clang fp contract(off)
    unsigned long long done = 0ull;
    float U_last = 0.f, last_alpha = 0.f, one_minus_last = 1.f;
    const v2f px = {pfx, pfx}, py = {pfy, pfy};
    const v2f one = {1.f, 1.f};
    const v2f lim = {(float)pos_limit, (float)pos_limit};
    v2f pos = {(float)(2u * npairs), (float)(2u * npairs + 1u)};
    const float ntf = -T_final;
    for (int j = (int)npairs - 1; j >= 0; j--) {
        const float4 *p = slab + 5 * j;
        const float4 p0 = p[0], p1 = p[1], p2 = p[2], p3 = p[3], p4 = p[4];
        // alpha_eval: the forward's alphas, bit for bit; `lim` - position, clamped, is the third 0 / 1 factor (entries at or beyond this
        // pixel's last contributor)
        pos = pos - v2f{2.f, 2.f};   // (2 j, 2 j + 1) as floats, carried: there is no scalar int -> float on gfx950
        const AlphaEval<v2f> e = alpha_eval_lim(v2f{p0.x, p0.y}, v2f{p0.z, p0.w}, v2f{p1.x, p1.y}, v2f{p1.z, p1.w}, v2f{p2.x, p2.y}, v2f{p2.z, p2.w}, px, py, lim, pos);
        const v2f mm = e.mm, al = e.al, dx = e.dx, dy = e.dy;
        if (__ballot(fmaxf(al.x, al.y) > 0.f) == 0ull) continue;     // wave-uniform: nobody blends either entry (their rows are not written)
        // An entry with alpha == 0 is replayed as a zero-alpha layer: the recurrences leave T / accum_rec exactly as skipping would.
        const v2f oma = one - al;
        const v2f inv = {__builtin_amdgcn_rcpf(oma.x), __builtin_amdgcn_rcpf(oma.y)};   // v_rcp_f32 (1 ulp), shared by both divisions
        const float T1 = T * inv.y, T0 = T1 * inv.x;   // T in front of the second entry of the pair (the later one: replayed first), then of the first
        const v2f Tv = {T0, T1};
        T = T0;
        const v2f w = al * Tv;
        v2f z[10];
        const v2f c0 = {p3.x, p3.y}, c1 = {p3.z, p3.w}, c2 = {p4.x, p4.y}, c3 = {p4.z, p4.w};
        v2f U = c0 * v2f{dpix[0], dpix[0]};
        z[0] = w * v2f{dpix[0], dpix[0]};
        if (C > 1) { U = __builtin_elementwise_fma(c1, v2f{dpix[1 % C], dpix[1 % C]}, U); z[1] = w * v2f{dpix[1 % C], dpix[1 % C]}; } else z[1] = v2f{0.f, 0.f};
        if (C > 2) { U = __builtin_elementwise_fma(c2, v2f{dpix[2 % C], dpix[2 % C]}, U); z[2] = w * v2f{dpix[2 % C], dpix[2 % C]}; } else z[2] = v2f{0.f, 0.f};
        if (C > 3) { U = __builtin_elementwise_fma(c3, v2f{dpix[3 % C], dpix[3 % C]}, U); z[3] = w * v2f{dpix[3 % C], dpix[3 % C]}; } else z[3] = v2f{0.f, 0.f};
        // App. A.4's accum_rec / last_color only meet the gradient through their dot product with dL/dpix: carried as R = accum_rec . dpix, U_last
        const float R1 = __fmaf_rn(last_alpha, U_last, one_minus_last * R_acc);
        const float R0 = __fmaf_rn(al.y, U.y, oma.y * R1);
        const v2f Rv = {R0, R1};
        R_acc = R0;
        U_last = U.x;
        last_alpha = al.x;
        one_minus_last = oma.x;
        const v2f dLa = __builtin_elementwise_fma(U - Rv, Tv, (inv * v2f{ntf, ntf}) * v2f{bg_dot, bg_dot});
        // Q = opacity * G * dL/dalpha where the entry blends (the per-Gaussian backward no longer multiplies by the opacity).  An INDEFINITE conic (a
        // caller's non-PSD covariance) can overflow the exponential where the entry is skipped (pw > 0, mm = 0): inf * 0 would poison the sums the
        // reference never touches (`power > 0.f -> continue`), so og is capped at a finite stand-in first (tests: test_fuzz_indefinite_covariances)
        const v2f og_f = {fminf(e.og.x, 3.0e38f), fminf(e.og.y, 3.0e38f)};
        const v2f Q = (og_f * mm) * dLa;
        z[4] = Q;
        z[5] = Q * dx;
        z[6] = Q * dy;
        z[7] = z[5] * dx;
        z[8] = z[5] * dy;
        z[9] = z[6] * dy;
        float n0, n1;
        wave_sum20_banks(z, n0, n1);
        done |= 1ull << j;
        if (slot20 >= 0) acc[20 * j + slot20] = from_n1 ? n1 : n0;
    }
    return done;
}

// How many of the piece's survivors (bits of `mask`, list order) lie in front of list index `lim_k` of the sub-range.
__device__ __forceinline__ uint32_t survivors_before(unsigned long long mask, uint32_t lim_k) {
    const unsigned long long low = lim_k >= 64u ? ~0ull : ((1ull << lim_k) - 1ull);
    return (uint32_t)__popcll(mask & low);
}
