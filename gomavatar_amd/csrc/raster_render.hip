// Per-tile kernels of the splat rasterizer: depth sort, front-to-back alpha
// compositing (forward) and back-to-front replay (backward).
//
// Replaces cub::DeviceRadixSort + identifyTileRanges + renderCUDA (forward and
// backward) of the CUDA extension the reference calls at
// models/modules/renderer/gaussian.py:83-91 (algorithm: SURVEY.md App. A.2-A.4).
//
// MI355X design -- segment-parallel compositing (not a warp-shaped port).
// A GoMAvatar frame touches only ~170 of 1024 tiles and a few of them hold
// >5 000 Gaussians; "one workgroup per tile walks its list" leaves the chip idle
// behind those tiles.  Alpha compositing is associative, so the list of every
// tile is cut into segments of 128 entries (256 in a batched launch), every segment into 4
// sub-ranges of 32, and the unit of work is one wave = (segment, 8x8 pixel
// quadrant, sub-range): ~19 000 independent waves per frame, x B for a batched
// launch (B frames stacked into one tall tile grid, see raster_pre.hip).
//
//   k_sort         per tile: merge-path merge sort of the unique
//                  (depth_bits<<32 | gaussian) keys in registers + LDS ->
//                  identical to the reference's stable (tile, depth) radix order.
//                  Also leaves the entries' geometry in LIST order (ent_geo) so
//                  the compositing kernels stream contiguous 24-byte records.
//   k_seg_T        pass A: prod(1-alpha) of every sub-range and segment for every
//                  pixel of the tile, from alpha alone (no colours, no stop rule).
//   k_seg_fwd      pass B: every wave composites its <=32 entries with the
//                  reference's exact per-pixel rules, starting from the
//                  transmittance pass A implies; the 4 sub-ranges are folded in
//                  LDS (contribution, T behind, last contributor, stop flag).
//   k_combine_fwd  pass C, per tile: folds the segments front to back, writes the
//                  image / final_T / n_contrib and per-segment checkpoints (T
//                  behind the segment, colour still to come) for the backward.
//   k_seg_bwd      per (segment, sub-range), 4 quadrants per workgroup:
//                  back-to-front replay from the checkpoint; the 6+C per-pixel
//                  terms of every entry are reduced over the wave with a
//                  transposed tree (v_permlane32/16_swap halve the live registers,
//                  four in-row v_add_f32_dpp steps finish), the four quadrants are
//                  summed in LDS in a fixed order and one 48-byte record per
//                  (tile, entry) is written.  No float atomics anywhere.
// In a batched launch the three segment kernels run as exactly-resident grids that
// draw their (segment, piece) tasks from a sharded queue (TaskQueue below).
//
// Inside a wave, lane = pixel.  Per sub-range lane = ENTRY first: every lane
// tests one entry against the wave's 8x8 rectangle with a conservative bound on
// the largest alpha it can reach there; a 64-bit ballot then drives a scalar
// loop over the survivors only (~35 %), whose attributes every lane fetches
// with broadcast reads from a wave-private LDS slab (v_readlane, 6.6 issue cycles
// per value, was the first version).  Entries skipped this way are exactly those the
// reference skips for all 64 pixels (`alpha < 1/255 -> continue`).  The blend
// loop is branch-free and keeps its state in VGPRs (float masks) so that the
// serial T chain never round-trips through SALU/VCC logic.
#include "gom_internal.h"
#include "bwd_order.hpp"
#include "sort_util.hpp"
#include "entry_record.hpp"
#include "l1_pixel.hpp"
#include <cstdlib>

namespace {

constexpr float kStopT = 0.0001f;          // App. A.3: stop when T(1-alpha) < 1e-4
constexpr float kMinAlpha = 1.0f / 255.0f;
constexpr float kMaxAlpha = 0.99f;


// Building blocks of the backward's transposed 64-lane reduction (csrc/seg_bwd_replay.hpp: wave_sum20_banks): v_permlane32_swap /
// v_permlane16_swap exchange halves (rows) of two registers, so one swap + one add folds a level of the tree for TWO values and halves the
// number of live registers (layout of the swap levels checked by scripts/ubench/permlane_probe.hip, of the bank-masked in-row levels by
// scripts/ubench/row_reduce_probe.hip).
__device__ __forceinline__ float swap_add16(float a, float b) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t o = __shfl_xor(v, d, 64);
        v = o > v ? o : v;
    }
    return v;
}

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

// Records mode of the render backward (round 4).  The compositing pass leaves ONE record per blending (pixel, entry) pair -- the lanes of
// its serial chain with w = alpha T > 0 -- compacted per entry (ballot + mbcnt), entry-major inside the piece's region:
//     rec_ti  = (T in front of the entry at the pixel, bits: entry of the sub-range << 6 | pixel of the quadrant)
//     rec_acc = colour the PIECE had added to the pixel in front of the entry
// and the backward (k_rec_bwd) is a lane per record: no replay, no 64-lane reduction per entry, every lane alive.  What it needs beside
// the record is per pixel (dL/dpix, the colour from the piece's first entry to the end of the list, T_final, n_contrib) and per entry.
// A piece's region is allocated when k_seg_fwd finds the piece alive, with the number of (lane, surviving entry) pairs with alpha > 0
// that k_seg_T counted as the upper bound (w > 0 implies alpha > 0, and the two passes see bit-identical alphas).
struct GomRecArgs {
    uint32_t *piece_ub;      // null: records off
    uint2 *piece_rec;
    uint8_t *piece_cnt;      // [piece][64] records per entry
    float2 *rec_ti;
    float4 *rec_acc;
    uint32_t *cursor;        // GOM_REC_SHARDS heads, 32 words apart
    uint32_t *overflow;
    uint32_t shard_cap;      // records per shard
};

// True when the entry can be skipped for EVERY pixel centre in
// [x0,x1]x[y0,y1]: the largest alpha it reaches there is provably < 1/255
// (with a rounding margin).  Never culls when unsure.
__device__ __forceinline__ bool cull_entry(float cx, float cy, float a, float b, float cz, float o, float x0, float y0, float x1, float y1) {
    if (!(a > 0.f) || !(cz > 0.f) || !(a * cz - b * b > 0.f)) return false;  // not positive definite: exact path
    if (o <= 0.f) return true;                                                // alpha = o*G <= 0 < 1/255
    if (!(o < 3.0e38f)) return false;
    const float X = cx < x0 ? (x0 - cx) : (cx > x1 ? (x1 - cx) : 0.f);
    const float Y = cy < y0 ? (y0 - cy) : (cy > y1 ? (y1 - cy) : 0.f);
    if (X == 0.f && Y == 0.f) return false;
    float q = 3.0e38f;
    // (v_rcp_f32, 1 ulp: the bound below carries a 1e-5 relative + 1e-2 absolute margin, and dx / dy are clamped anyway)
    if (X != 0.f) {
        float dy = -b * X * __builtin_amdgcn_rcpf(cz);
        dy = fminf(fmaxf(dy, y0 - cy), y1 - cy);
        q = fminf(q, a * X * X + 2.f * b * X * dy + cz * dy * dy);
    }
    if (Y != 0.f) {
        float dx = -b * Y * __builtin_amdgcn_rcpf(a);
        dx = fminf(fmaxf(dx, x0 - cx), x1 - cx);
        q = fminf(q, a * dx * dx + 2.f * b * dx * Y + cz * Y * Y);
    }
    const float DX = fmaxf(fabsf(x0 - cx), fabsf(x1 - cx));
    const float DY = fmaxf(fabsf(y0 - cy), fabsf(y1 - cy));
    const float mag = a * DX * DX + 2.f * fabsf(b) * DX * DY + cz * DY * DY;
    const float lthr = -__logf(255.0f * o);  // alpha >= 1/255  <=>  power >= lthr
    return (-0.5f * q + (1e-5f * mag + 1e-2f)) < lthr;
}

// alpha of an entry at a pixel with the reference's skip rules folded in (0 where the reference would `continue`: power > 0 or
// alpha < 1/255), from the entry's LIST record (x, y, A, B, Cq, lo) -- the tile pass stores the conic and the opacity pre-scaled,
//     A = -0.5 log2(e) a,  B = -log2(e) b,  Cq = -0.5 log2(e) c,  lo = log2(opacity)        (entry_record.hpp)
// so that, with d = centre - pixel,
//     pw = dx (A dx + B dy) + Cq dy dy = log2(e) * power,      og = opacity * G = exp2(pw + lo),      alpha = min(0.99, og)
// in five multiply-adds, one add and the v_exp_f32 (the reference's expression -- forward.cu:330-340, three products per term, the
// -0.5, the log2(e) of __expf and the opacity as separate factors -- took eleven).  The two skip rules are 0 / 1 factors made by
// clamped multiply-adds instead of compare + select pairs:  m1 = clamp(2^64 (og - pred(1/255))) is 1 exactly when og >= 1/255 (the
// difference is then at least an ulp of 1/255 = 2^-31), m2 = clamp(1 - 2^126 pw) is 1 exactly when pw <= 0 and 0 for pw >= 2^-126.
// NOT exact in between: a DENORMAL positive exponent, 0 < pw < 2^-126 (only an indefinite conic makes pw positive at all), gives a
// fractional factor where the reference skips the pixel.  The contribution it lets through is alpha * m2 with alpha = opacity * 2^pw =
// opacity to within 1e-38: a set of measure zero, documented rather than guarded (tests/test_gpu_raster.py::test_fuzz_indefinite_covariances).
// Why: the three segment kernels are bound by VALU issue, the alpha evaluation is half of what they issue, and on gfx950 a compare or a
// select costs 4.2 issue cycles where a multiply-add costs 2.25 (scripts/ubench/valu_rate.hip).  Every operation is written out, nothing
// is left to contraction, and the float and the register-pair version below are the same sequence: the transmittance pre-pass, the
// compositing pass and the backward see bit-identical alphas -- sum_i alpha_i T_i + T_final = 1 only holds to 3e-6 if they do.
// Against the reference's expression the exponent moves by a few ulp (re-association); tests/ hold the result to the float64 oracle.
using gom_entry::entry_record;
using gom_entry::entry_unrecord;
constexpr float kMinAlphaPred = 0.003921568393707275f;   // the float below 1/255 (0x3b808080)
typedef float v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float vfma(float a, float b, float c) { return __fmaf_rn(a, b, c); }
__device__ __forceinline__ v2f vfma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
// The 0 / 1 skip factors: m1 = clamp(2^64 og - 2^64 pred(1/255)), m2 = clamp(1 - 2^126 pw) and, for the backward, m3 = clamp(lim - pos)
// (1 for list positions in front of this pixel's last contributor).  Scalar form: the compiler folds the clamp into v_fma_f32.  Pair form:
// one inline-assembly block (the compiler does not fold a clamp into v_pk_fma_f32), m1 LAST -- `og` comes out of v_exp_f32, and a VALU
// instruction that reads the result of a transcendental one needs two wait states the hazard recognizer does not insert in front of
// inline assembly (without them the multiply-add read the register before the exponential had landed: 43 parity tests).
__device__ __forceinline__ void skip_factors(float og, float pw, float &m1, float &m2) {
    m1 = fminf(fmaxf(__fmaf_rn(og, 0x1p64f, -kMinAlphaPred * 0x1p64f), 0.f), 1.f);
    m2 = fminf(fmaxf(__fmaf_rn(pw, -0x1p126f, 1.f), 0.f), 1.f);
}
__device__ __forceinline__ void skip_factors(v2f og, v2f pw, v2f &m1, v2f &m2) {
    const v2f k1 = {0x1p64f, 0x1p64f}, k2 = {-kMinAlphaPred * 0x1p64f, -kMinAlphaPred * 0x1p64f}, k3 = {-0x1p126f, -0x1p126f}, one = {1.f, 1.f};
    asm("v_pk_fma_f32 %1, %3, %6, %7 clamp\n\ts_nop 0\n\tv_pk_fma_f32 %0, %2, %4, %5 clamp"
        : "=&v"(m1), "=&v"(m2) : "v"(og), "v"(pw), "v"(k1), "v"(k2), "v"(k3), "v"(one));
}
__device__ __forceinline__ void skip_factors(v2f og, v2f pw, v2f lim, v2f pos, v2f &m1, v2f &m2, v2f &m3) {
    const v2f k1 = {0x1p64f, 0x1p64f}, k2 = {-kMinAlphaPred * 0x1p64f, -kMinAlphaPred * 0x1p64f}, k3 = {-0x1p126f, -0x1p126f}, one = {1.f, 1.f};
    asm("v_pk_fma_f32 %1, %4, %7, %8 clamp\n\tv_pk_add_f32 %2, %9, %10 neg_lo:[0,1] neg_hi:[0,1] clamp\n\tv_pk_fma_f32 %0, %3, %5, %6 clamp"
        : "=&v"(m1), "=&v"(m2), "=&v"(m3) : "v"(og), "v"(pw), "v"(k1), "v"(k2), "v"(k3), "v"(one), "v"(lim), "v"(pos));
}
__device__ __forceinline__ float vexp2(float t) { return __builtin_amdgcn_exp2f(t); }
__device__ __forceinline__ v2f vexp2(v2f t) { return v2f{__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)}; }
__device__ __forceinline__ float vmin_alpha(float og) { return fminf(kMaxAlpha, og); }
__device__ __forceinline__ v2f vmin_alpha(v2f og) { return v2f{fminf(kMaxAlpha, og.x), fminf(kMaxAlpha, og.y)}; }
template <typename F> __device__ __forceinline__ F vsplat(float x);
template <> __device__ __forceinline__ float vsplat<float>(float x) { return x; }
template <> __device__ __forceinline__ v2f vsplat<v2f>(float x) { return v2f{x, x}; }

// F = float: one entry; F = v2f: two entries on register pairs (v_pk_mul / v_pk_fma / v_pk_add_f32).
template <typename F>
struct AlphaEval {
    F al, og, mm, dx, dy;   // alpha (skips folded in), opacity * G unclamped, the 0 / 1 skip factor, centre - pixel
};
template <typename F>
__device__ __forceinline__ AlphaEval<F> alpha_eval(F ex, F ey, F A, F B, F Cq, F lo, F px, F py) {
#pragma clang fp contract(off)
    AlphaEval<F> r;
    r.dx = ex - px;
    r.dy = ey - py;
    const F t1 = vfma(A, r.dx, B * r.dy);
    const F pw = vfma(r.dx, t1, (Cq * r.dy) * r.dy);
    r.og = vexp2(pw + lo);
    F m1, m2;
    skip_factors(r.og, pw, m1, m2);
    r.mm = m1 * m2;
    r.al = vmin_alpha(r.og) * r.mm;
    return r;
}
// the backward's: + the position limit (mm and al carry all three factors)
__device__ __forceinline__ AlphaEval<v2f> alpha_eval_lim(v2f ex, v2f ey, v2f A, v2f B, v2f Cq, v2f lo, v2f px, v2f py, v2f lim, v2f pos) {
#pragma clang fp contract(off)
    AlphaEval<v2f> r;
    r.dx = ex - px;
    r.dy = ey - py;
    const v2f t1 = vfma(A, r.dx, B * r.dy);
    const v2f pw = vfma(r.dx, t1, (Cq * r.dy) * r.dy);
    r.og = vexp2(pw + lo);
    v2f m1, m2, m3;
    skip_factors(r.og, pw, lim, pos, m1, m2, m3);
    r.mm = (m1 * m2) * m3;
    r.al = (vmin_alpha(r.og) * (m1 * m2)) * m3;   // (= alpha_eval's alpha times m3, bit for bit)
    return r;
}
__device__ __forceinline__ float entry_alpha(float ex, float ey, float A, float B, float Cq, float lo, float pfx, float pfy) {
    return alpha_eval<float>(ex, ey, A, B, Cq, lo, pfx, pfy).al;
}
struct PairGeo { v2f x, y, a, b, c, o; };   // (a, b, c, o hold A, B, Cq, lo)
__device__ __forceinline__ v2f pair_alpha(const PairGeo &e, float pfx, float pfy) {
    return alpha_eval<v2f>(e.x, e.y, e.a, e.b, e.c, e.o, v2f{pfx, pfx}, v2f{pfy, pfy}).al;
}

// Survivors of a sub-range, compacted in list order into PAIR records in a wave-private LDS slab: pair j = survivors 2j and
// 2j + 1 as (x0 x1 y0 y1)(a0 a1 b0 b1)(c0 c1 o0 o1), so that three broadcast ds_read_b128 deliver both entries already laid
// out as register pairs.  The slab is padded with null entries (opacity 0, lo = -inf -> alpha 0) up to a multiple of four survivors.
#define GOM_PAIR_F4 (3 * (GOM_SUB_MAX / 2 + 2))   // float4 per wave slab
__device__ __forceinline__ uint32_t stage_pairs(float4 *slab, bool keep, unsigned long long mask, int lane, float x, float y, float a, float b, float c,
                                                float o) {
    float *f = reinterpret_cast<float *>(slab);
    const uint32_t n = (uint32_t)__popcll(mask), n4 = (n + 3u) & ~3u;
    const uint32_t pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
    if (keep) {
        float *d = f + 12u * (pos >> 1) + (pos & 1u);
        d[0] = x; d[2] = y; d[4] = a; d[6] = b; d[8] = c; d[10] = o;
    }
    if (lane < 3 && n + (uint32_t)lane < n4) {
        const uint32_t pp = n + (uint32_t)lane;
        float *d = f + 12u * (pp >> 1) + (pp & 1u);
        d[0] = 0.f; d[2] = 0.f; d[4] = 0.f; d[6] = 0.f; d[8] = 0.f; d[10] = -INFINITY;   // (lo = log2 of opacity 0)
    }
    return n4;
}
__device__ __forceinline__ PairGeo load_pair(const float4 *slab, uint32_t j) {
    const float4 p0 = slab[3 * j], p1 = slab[3 * j + 1], p2 = slab[3 * j + 2];
    PairGeo e;
    e.x = v2f{p0.x, p0.y}; e.y = v2f{p0.z, p0.w}; e.a = v2f{p1.x, p1.y}; e.b = v2f{p1.z, p1.w}; e.c = v2f{p2.x, p2.y}; e.o = v2f{p2.z, p2.w};
    return e;
}

// ---------------------------------------------------------------- sort ------
// (merge-sort building blocks: sort_util.hpp)
using namespace gom_sort;

// One workgroup per tile.  Lists of up to 2^log_chunk (= 8 * NT) keys are sorted in registers + LDS.  Longer lists
// (a handful of tiles at the 220k-Gaussian configuration) are cut into chunks of that size, each sorted as above,
// and the chunks are then merged level by level in global memory (ping-pong with `scratch`).
// Two instantiations share the tiles of a batched launch: NT = 256 takes the lists of up to 2048 entries (16 KiB
// of LDS and 4 waves per workgroup, so a CU holds 8 of them instead of 2 mostly idle 16-wave ones), NT = 1024 the
// longer ones.
template <int NT>
__global__ void __launch_bounds__(NT, 8) k_sort(int gx, const uint32_t *__restrict__ tile_base, const uint32_t *__restrict__ seg_base,
                                               uint64_t *__restrict__ keys, uint32_t *__restrict__ point_list,
                                               uint4 *__restrict__ seg_desc, const ushort4 *__restrict__ rect,
                                               const uint32_t *__restrict__ pair_off, uint32_t *__restrict__ pair_pos, uint32_t *__restrict__ ent_slot,
                                               const float2 *__restrict__ xy, const float4 *__restrict__ conic_opacity,
                                               float2 *__restrict__ ent_geo, uint64_t *__restrict__ scratch,
                                               const GomDevStatus *__restrict__ status, uint32_t log_chunk, uint32_t small_max, uint32_t seg_shift, uint32_t bitmap_words) {
    __shared__ __attribute__((aligned(16))) uint64_t s_x[8 * NT];
    __shared__ uint32_t s_wsum[NT / 64];
    if (status->overflow) return;
    const int tile = blockIdx.x;
    const uint32_t base = tile_base[tile];
    const uint32_t n = tile_base[tile + 1] - base;
    if (n == 0 || (n <= small_max) != (NT == 256)) return;  // the other instantiation's tile
    const uint32_t sb = seg_base[tile], nseg = seg_base[tile + 1] - sb;
    // one 16-byte descriptor per segment: the segment kernels start from a single load
    for (uint32_t i = threadIdx.x; i < nseg; i += NT)
        seg_desc[sb + i] = make_uint4((uint32_t)tile, base + (i << seg_shift), min(1u << seg_shift, n - (i << seg_shift)), i);
    const int tx = tile % gx, ty = tile / gx;
    const uint32_t t = threadIdx.x;
    const uint32_t CH = 1u << log_chunk;
    if (n <= CH) {
        uint64_t x[8];
        block_merge_sort<NT>(keys + base, n, s_x, x);
        // blocked -> striped through LDS (thread t: positions t, NT + t, ...): every store of the write-out below is
        // then a contiguous run per wave instead of 64 separate cache lines
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 8; r += 2) *reinterpret_cast<ulonglong2 *>(s_x + 8 * t + r) = make_ulonglong2(x[r], x[r + 1]);
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 8; r++) x[r] = s_x[r * NT + t];
        // write-out, 4 entries per trip: their gathers are issued before the first dependent store
        // (8 at once would push the kernel past 64 VGPRs and halve the workgroups per CU)
#pragma unroll
        for (int r0 = 0; r0 < 8; r0 += 4) {
            if ((uint32_t)r0 * NT >= n) break;  // block-uniform: nothing left in the upper stripes
            ushort4 rc[4];
            uint32_t po[4];
            float2 cxy[4];
            float4 cco[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t i = (uint32_t)(r0 + u) * NT + t;
                const uint32_t g = i < n ? (uint32_t)x[r0 + u] : 0u;
                rc[u] = rect[g];
                po[u] = pair_off[g];
                cxy[u] = xy[g];
                cco[u] = conic_opacity[g];
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t i = (uint32_t)(r0 + u) * NT + t;
                if (i < n) {
                    const uint32_t g = (uint32_t)x[r0 + u];
                    keys[base + i] = x[r0 + u];
                    point_list[base + i] = g;
                    const uint32_t k = (uint32_t)(ty - (int)rc[u].y) * (uint32_t)(rc[u].z - rc[u].x) + (uint32_t)(tx - (int)rc[u].x);
                    pair_pos[po[u] + k] = base + i;
                    ent_slot[base + i] = po[u] + k;
                    // geometry of the entry in LIST order: the compositing kernels read it contiguously
                    float2 *dst = ent_geo + 3 * (size_t)(base + i);
                    const float4 er = entry_record(cco[u].x, cco[u].y, cco[u].z, cco[u].w);   // (A, B, Cq, lo: see alpha_eval)
                    dst[0] = cxy[u]; dst[1] = make_float2(er.x, er.y); dst[2] = make_float2(er.z, er.w);
                }
            }
        }
        return;
    }
    // ---- rare: list longer than one chunk
    uint64_t *src = keys + base, *dst = scratch + base;
    if (bitmap_words) {
        // the keys are INDICES alone (the mesh rasterizer: depth bits zero, a face at most once per tile, bitmap_words x 32 >= the face count): a bitmap of the
        // indices in LDS and a popcount scan give the sorted list in one pass -- the chunk sorts + merge levels below are a ~50 us chain for the
        // few 8 x 8-pixel tiles that hold thousands of faces, beside 4 000 tiles that are done in a fraction of that
        uint32_t *bm = reinterpret_cast<uint32_t *>(s_x);
        for (uint32_t w = t; w < bitmap_words; w += NT) bm[w] = 0u;
        __syncthreads();
        for (uint32_t i = t; i < n; i += NT) { const uint32_t g = (uint32_t)keys[base + i]; atomicOr(&bm[g >> 5], 1u << (g & 31u)); }
        __syncthreads();
        const uint32_t wpt = (bitmap_words + NT - 1u) / NT, w0 = min(bitmap_words, t * wpt), w1 = min(bitmap_words, w0 + wpt);
        uint32_t mine = 0;
        for (uint32_t w = w0; w < w1; w++) mine += __popc(bm[w]);
        uint32_t incl = mine;
        const uint32_t lane = t & 63u, wid = t >> 6;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t y = __shfl_up(incl, d, 64);
            if (lane >= (uint32_t)d) incl += y;
        }
        if (lane == 63) s_wsum[wid] = incl;
        __syncthreads();
        uint32_t pos = incl - mine;
        for (uint32_t w = 0; w < wid; w++) pos += s_wsum[w];
        for (uint32_t w = w0; w < w1; w++) {
            uint32_t bits = bm[w];
            while (bits) {
                const uint32_t bit = __ffs(bits) - 1u;
                bits &= bits - 1u;
                src[pos++] = (uint64_t)((w << 5) | bit);   // (every key of the list has been read: the bitmap holds them)
            }
        }
    } else {
    for (uint32_t c0 = 0; c0 < n; c0 += CH) {
        const uint32_t cn = min(CH, n - c0);
        uint64_t x[8];
        block_merge_sort<NT>(keys + base + c0, cn, s_x, x);
#pragma unroll
        for (int r = 0; r < 8; r++)
            if (8 * t + r < cn) keys[base + c0 + 8 * t + r] = x[r];
        __syncthreads();  // s_x is re-used by the next chunk
    }
    for (uint32_t L = CH; L < n; L <<= 1) {
        __syncthreads();  // the previous level's (or the chunk sorts') global writes are visible to the block
        for (uint32_t o8 = 8 * t; o8 < n; o8 += 8 * NT) {
            const uint32_t pbase = o8 & ~(2 * L - 1);
            const uint32_t la = min(L, n - pbase);
            const uint32_t lb = n > pbase + L ? min(L, n - pbase - L) : 0u;
            uint64_t y[8];
            merge8(src + pbase, L, la, lb, o8 - pbase, y);
#pragma unroll
            for (int r = 0; r < 8; r++)
                if (o8 + r < n) dst[o8 + r] = y[r];
        }
        uint64_t *tmp = src; src = dst; dst = tmp;
    }
    }
    __syncthreads();
    for (uint32_t i = t; i < n; i += NT) {
        const uint64_t key = src[i];
        const uint32_t g = (uint32_t)key;
        keys[base + i] = key;   // (a no-op copy when the last level landed in `keys`)
        point_list[base + i] = g;
        const ushort4 rc = rect[g];
        const uint32_t k = (uint32_t)(ty - (int)rc.y) * (uint32_t)(rc.z - rc.x) + (uint32_t)(tx - (int)rc.x);
        const uint32_t slot = pair_off[g] + k;
        pair_pos[slot] = base + i;
        ent_slot[base + i] = slot;
        const float2 c = xy[g];
        const float4 co = conic_opacity[g];
        float2 *d2 = ent_geo + 3 * (size_t)(base + i);
        const float4 er = entry_record(co.x, co.y, co.z, co.w);
        d2[0] = c; d2[1] = make_float2(er.x, er.y); d2[2] = make_float2(er.z, er.w);
    }
}

// ------------------------------------------------------------- entries -----
// One lane's view of "its" entry of the wave's 32-entry sub-range.
template <int C>
struct EntryRegs {
    float x, y, a, b, c, o, col[C > 0 ? C : 1];   // a, b, c, o: the record's A, B, Cq, lo (alpha_eval)
    bool keep;
};

// The unit of work is one wave = (segment, quadrant q of 8x8 pixels, sub-range j of 32 (or 64) list entries).
// A single wave issues at most one VALU instruction every ~5 cycles on gfx950 (measured,
// scripts/ubench/dpp_bench.hip) and most of a wave's life here is load latency, so the lists are cut into many
// short independent pieces that the CU interleaves 8 per SIMD.  Workgroups are 4 waves: the four sub-ranges of a
// (segment, quadrant) in the forward (they fold their results in LDS), the four quadrants of a (segment,
// sub-range) in the backward (they sum their per-entry reductions in LDS).
// Segment size is a launch parameter (seg_shift = 7 or 8): 128-entry segments / 32-entry sub-ranges give a single frame the
// most independent waves; a batched launch has parallelism to spare and halves the per-task fixed cost (checkpoint loads,
// cull, LDS fold, stores -- measured at 40-60 % of these kernels) with 256 / 64.
#define GOM_NSUB 4
#ifndef GOM_FWD_SKIP_DEAD_ROWS
#define GOM_FWD_SKIP_DEAD_ROWS 1
#endif
#ifndef GOM_FWD_SKIP_IDLE
#define GOM_FWD_SKIP_IDLE 0   // (round 3, measured: 1 = skip the chain of an entry no compositing lane blends -- k_seg_fwd 110 -> 124 us: the branch between the four
#endif                       //  unrolled chain steps costs more than the ~14 instructions it saves on a quarter of the entries)
#ifndef GOM_FWD_EPT
#define GOM_FWD_EPT 4   // entries evaluated per trip of the forward loops
#endif
#define GOM_SUB_MAX 64

// The <=32 entries of this wave's sub-range, read straight from the list-ordered records the sort left behind
// (contiguous 24-byte geometry + 16-byte colour rows: coalesced, no LDS staging, no barrier): lanes 0..31 load
// one entry each and test it against the wave's 8x8 rectangle.
template <int C>
__device__ __forceinline__ EntryRegs<C> load_sub(const float2 *__restrict__ ent_geo, const float *__restrict__ ent_col, uint32_t start,
                                                 uint32_t cnt, int sub, int lane, float qx0, float qy0, float qx1, float qy1, uint32_t sub_sz,
                                                 const unsigned long long *__restrict__ known = nullptr) {
    EntryRegs<C> r;
    const uint32_t e = (uint32_t)sub * sub_sz + (uint32_t)lane;
    const bool valid = (uint32_t)lane < sub_sz && e < cnt;
    r.x = r.y = r.a = r.b = r.c = 0.f;
    r.o = -INFINITY;
#pragma unroll
    for (int ch = 0; ch < (C > 0 ? C : 1); ch++) r.col[ch] = 0.f;
    if (valid) {
        const float2 *g = ent_geo + 3 * (size_t)(start + e);
        const float2 g0 = g[0], g1 = g[1], g2 = g[2];
        r.x = g0.x; r.y = g0.y; r.a = g1.x; r.b = g1.y; r.c = g2.x; r.o = g2.y;
        if (C > 0) {
            const float4 cl = *reinterpret_cast<const float4 *>(ent_col + 4 * (size_t)(start + e));
            r.col[0] = cl.x;
            if (C > 1) r.col[1 % (C > 0 ? C : 1)] = cl.y;
            if (C > 2) r.col[2 % (C > 0 ? C : 1)] = cl.z;
            if (C > 3) r.col[3 % (C > 0 ? C : 1)] = cl.w;
        }
    }
    // `known`: the survivors of this (sub-range, quadrant) as k_seg_T found them (cull_masks) -- the same test on the same numbers, so
    // the compositing pass and the backward take its 64 bits instead of ~55 instructions per lane; null: test here.
    if (known) {
        r.keep = valid && ((*known >> lane) & 1ull) != 0ull;
    } else {
        const float4 u = entry_unrecord(r.a, r.b, r.c, r.o);
        r.keep = valid && !cull_entry(r.x, r.y, u.x, u.y, u.z, u.w, qx0, qy0, qx1, qy1);
    }
    return r;
}

// colours of the list entries in LIST order (D x 4 floats); written by the T pre-pass when it runs, by this
// kernel when a colour pass re-uses an existing binning
template <int C>
__global__ void __launch_bounds__(256) k_gather_colors(const uint32_t *__restrict__ point_list, const float *__restrict__ colors,
                                                       float *__restrict__ ent_col, const GomDevStatus *__restrict__ status, uint32_t *__restrict__ rec_cursor,
                                                       uint32_t *__restrict__ rec_overflow) {
    if (rec_cursor && blockIdx.x == 0) {   // the record allocator of this colour pass (k_seg_T's job when it runs)
        if (threadIdx.x < GOM_REC_SHARDS) rec_cursor[32 * threadIdx.x] = 0;
        if (threadIdx.x == 0) *rec_overflow = 0;
    }
    if (status->overflow) return;
    const uint32_t D = status->num_pairs;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < D; i += gridDim.x * 256) {
        const uint32_t g = point_list[i];
        float4 cl = make_float4(0.f, 0.f, 0.f, 0.f);
        cl.x = colors[(size_t)g * C]; cl.y = colors[(size_t)g * C + 1]; cl.z = colors[(size_t)g * C + 2];
        if (C == 4) cl.w = colors[(size_t)g * C + 3];
        *reinterpret_cast<float4 *>(ent_col + 4 * (size_t)i) = cl;
    }
}

// Per-pixel colour checkpoints (seg_C, seg_Sbehind, sub_C) are rows of 256 float4 (one 16-byte load or store per lane,
// 1 KiB contiguous per wave) rather than 4 planes of 256 floats: the backward reads up to 16 of them per lane and task.
template <int C>
__device__ __forceinline__ void st4(float *base, size_t row, int pxi, const float (&v)[C]) {
    float4 o = make_float4(v[0], v[1], v[2], 0.f);
    if (C == 4) o.w = v[C - 1];
    reinterpret_cast<float4 *>(base)[row * GOM_TPX + pxi] = o;
}
template <int C>
__device__ __forceinline__ void ld4(const float *base, size_t row, int pxi, float (&v)[C]) {
    const float4 o = reinterpret_cast<const float4 *>(base)[row * GOM_TPX + pxi];
    v[0] = o.x; v[1] = o.y; v[2] = o.z;
    if (C == 4) v[C - 1] = o.w;
}


// Dynamic distribution of the (segment, piece) tasks of the segment kernels.  A static grid-stride assignment left a
// third of the chip idle behind the last workgroups to start (timeline: scripts/wg_timeline_T.py); instead exactly as
// many workgroups as the chip holds are launched and each draws tasks from a queue until they run out, so all of
// them finish within one task of each other.
//   * One device-scope counter saturates at ~88 dequeues/us (MI355X_MICROARCH.md "dequeue"): the queue is sharded 8
//     ways (one head per XCD, 128 bytes apart).  Shard x owns the segments with seg % 8 == x; a workgroup starts on
//     shard blockIdx % 8 (the XCD it normally runs on, so a segment's four pieces share an L2) and moves on to the
//     next shard when its own is empty -- placement is only an affinity, any workgroup may take any task.
//   * The first task of every workgroup is implied by its index (no atomic); head values count from there.
//   * request() is issued AFTER the task's own loads (returns are in order: a load behind the atomic would wait for
//     it), publish() hands the result to the other waves through LDS before the task's last __syncthreads() --
//     every path through a task must call both and then pass a barrier.
//   * A workgroup whose shard is drained tries GOM_TQ_STEAL more shards and leaves.  Stealing is only balance between XCDs: every
//     task is drawn by the workgroups of its own shard whatever the others do.  (Walking all eight was ~7 failing device-scope
//     dequeues per workgroup at the moment all shards drain together: 14 000 of them at ~88 per us and address, the last task of
//     every workgroup took 16 us instead of 7 -- scripts/wg_timeline_T.py; k_seg_T 94 -> 87 us.  0, 1, 2, 3 measure the same.)
//   * The last workgroup to leave zeroes the heads for the next launch (graph replays included).
// ctr layout: head of shard x at ctr[32 * x], workgroups that have left at ctr[32 * 8].
#define GOM_TQ_WORDS (32 * GOM_TQ_SHARDS + 32)
#ifndef GOM_TQ_STEAL
#define GOM_TQ_STEAL 2   // shards a workgroup tries after its own before it leaves
#endif
template <int PER_SEG>
struct TaskQueueT {
    uint32_t *ctr;
    uint32_t nsegs, per_shard_wgs, shard, tried, pend, it, next;
    bool have_next;
    __device__ __forceinline__ uint32_t shard_tasks(uint32_t x) const { return (uint32_t)PER_SEG * gom_shard_segments(nsegs, x); }
    __device__ __forceinline__ uint32_t task_of(uint32_t x, uint32_t j) const {
        constexpr uint32_t SH = PER_SEG == 4 ? 2u : (PER_SEG == 2 ? 1u : 0u);   // PER_SEG queue items per segment: (segment << SH) | piece
        return (gom_shard_segment(x, j >> SH) << SH) | (j & ((1u << SH) - 1u));
    }
    // thread 0 only: local index j on the current shard -> queue item, moving to the next shards while the current one is empty
    __device__ __forceinline__ uint32_t resolve(uint32_t j) {
        while (j >= shard_tasks(shard)) {
            if (++tried > GOM_TQ_STEAL) return 0xffffffffu;
            shard = (shard + 1) % GOM_TQ_SHARDS;
            j = per_shard_wgs + atomicAdd(ctr + 32 * shard, 1u);
        }
        return task_of(shard, j);
    }
    __device__ __forceinline__ void init(uint32_t *c, uint32_t n_segs, uint32_t *s_task) {
        ctr = c; nsegs = n_segs; it = 0; tried = 0; pend = 0; next = 0; have_next = false;
        if (!ctr) return;  // static mode: plain grid-stride (single-frame launches, see the launchers)
        per_shard_wgs = gridDim.x / GOM_TQ_SHARDS;  // the launchers round the grid to a multiple of the shard count
        shard = blockIdx.x % GOM_TQ_SHARDS;
        if (threadIdx.x == 0) s_task[0] = resolve(blockIdx.x / GOM_TQ_SHARDS);
        __syncthreads();
    }
    __device__ __forceinline__ uint32_t current(const uint32_t *s_task) const {
        if (!ctr) {   // striding: workgroup (XCD x = blockIdx % 8, rank r) takes items r, r + grid / 8, ... of shard x -- the same XCD as through the queue
            if (gridDim.x % GOM_TQ_SHARDS) {   // (a grid that is not a multiple of the shard count: plain stride)
                const uint32_t t = blockIdx.x + it * gridDim.x;
                return t < nsegs * (uint32_t)PER_SEG ? t : 0xffffffffu;
            }
            const uint32_t x = blockIdx.x % GOM_TQ_SHARDS, j = blockIdx.x / GOM_TQ_SHARDS + it * (gridDim.x / GOM_TQ_SHARDS);
            return j < shard_tasks(x) ? task_of(x, j) : 0xffffffffu;
        }
        return s_task[it & 1];
    }
    __device__ __forceinline__ void request() {
        if (ctr && threadIdx.x == 0) pend = atomicAdd(ctr + 32 * shard, 1u);
    }
    // optional, between request() and publish(): turn the dequeue into the next task now, so that the order-table load it issues has
    // the rest of the task to arrive (publish() would otherwise wait for it in front of the barrier)
    __device__ __forceinline__ void look_ahead() {
        if (ctr && threadIdx.x == 0 && !have_next) { next = resolve(per_shard_wgs + pend); have_next = true; }
    }
    __device__ __forceinline__ void publish(uint32_t *s_task) {
        look_ahead();
        if (ctr && threadIdx.x == 0) { s_task[(it + 1) & 1] = next; have_next = false; }
    }
    __device__ __forceinline__ void advance() { it++; }
    __device__ __forceinline__ void finish() {
        if (ctr && threadIdx.x == 0 && atomicAdd(ctr + 32 * GOM_TQ_SHARDS, 1u) == gridDim.x - 1) {
            for (int x = 0; x <= GOM_TQ_SHARDS; x++) ctr[32 * x] = 0;
        }
    }
};

using TaskQueue = TaskQueueT<4>;

// The queue of k_seg_bwd_pair: the same sharded dequeue, but the tasks are the words of gom_internal.h ((segment << 3) | code) and, when the
// riders of the loss kernel have run, come from the cost-ordered table (its own count per shard) instead of the (segment, pair) grid.
struct PairQueue {
    uint32_t *ctr;
    const uint32_t *order;
    uint32_t nsegs, region, per_shard_wgs, shard, ntasks, tried, pend, it, next;
    bool have_next;
    __device__ __forceinline__ uint32_t shard_tasks(uint32_t x) const {
        if (order) return order[x];
        return 2u * gom_shard_segments(nsegs, x);
    }
    __device__ __forceinline__ uint32_t task_at(uint32_t x, uint32_t j) const {
        if (order) return order[GOM_BWD_ORDER_BASE + (size_t)x * region + j];
        return (gom_shard_segment(x, j >> 1) << 3) | (j & 1u);
    }
    __device__ __forceinline__ uint32_t resolve(uint32_t j) {   // thread 0 only
        while (j >= ntasks) {
            if (++tried > GOM_TQ_STEAL) return 0xffffffffu;
            shard = (shard + 1) % GOM_TQ_SHARDS;
            ntasks = shard_tasks(shard);
            j = per_shard_wgs + atomicAdd(ctr + 32 * shard, 1u);
        }
        return task_at(shard, j);
    }
    __device__ __forceinline__ void init(uint32_t *c, uint32_t n_segs, uint32_t *s_task, const uint32_t *task_order) {
        ctr = c; order = c ? task_order : nullptr; nsegs = n_segs; region = gom_bwd_order_region(n_segs); it = 0; tried = 0; pend = 0; next = 0; have_next = false;
        if (!ctr) return;  // static mode: plain grid-stride over the (segment, pair) grid
        per_shard_wgs = gridDim.x / GOM_TQ_SHARDS;
        shard = blockIdx.x % GOM_TQ_SHARDS;
        ntasks = 0;
        if (threadIdx.x == 0) { ntasks = shard_tasks(shard); s_task[0] = resolve(blockIdx.x / GOM_TQ_SHARDS); }
        __syncthreads();
    }
    __device__ __forceinline__ uint32_t current(const uint32_t *s_task) const {
        if (!ctr) {
            const uint32_t t = blockIdx.x + it * gridDim.x;
            return t < nsegs * 2u ? ((t >> 1) << 3) | (t & 1u) : 0xffffffffu;
        }
        return s_task[it & 1];
    }
    __device__ __forceinline__ void request() {
        if (ctr && threadIdx.x == 0) pend = atomicAdd(ctr + 32 * shard, 1u);
    }
    __device__ __forceinline__ void look_ahead() {   // optional, between request() and publish(): the table loads get the rest of the task to arrive
        if (ctr && threadIdx.x == 0 && !have_next) { next = resolve(per_shard_wgs + pend); have_next = true; }
    }
    __device__ __forceinline__ void publish(uint32_t *s_task) {
        look_ahead();
        if (ctr && threadIdx.x == 0) { s_task[(it + 1) & 1] = next; have_next = false; }
    }
    __device__ __forceinline__ void advance() { it++; }
    __device__ __forceinline__ void finish() {
        if (ctr && threadIdx.x == 0 && atomicAdd(ctr + 32 * GOM_TQ_SHARDS, 1u) == gridDim.x - 1) {
            for (int x = 0; x <= GOM_TQ_SHARDS; x++) ctr[32 * x] = 0;
        }
    }
};

// ------------------------------------------------- forward, pass A (T only) -
// prod(1 - alpha) of every 32-entry sub-range (and of the whole segment) for every pixel of the tile, from
// alpha alone (no colours, no stop rule): lets every later pass know the transmittance at which each piece
// starts without walking the list serially.
template <int C, bool REC>
__global__ void __launch_bounds__(256) k_seg_T(uint32_t seg_shift, int gx, int gy, const uint4 *__restrict__ seg_desc, const uint32_t *__restrict__ point_list,
                                                const float2 *__restrict__ ent_geo, const float *__restrict__ colors,
                                                float *__restrict__ ent_col, float *__restrict__ seg_T, float *__restrict__ sub_T,
                                                const GomDevStatus *__restrict__ status, uint32_t *__restrict__ task_ctr, unsigned long long *__restrict__ cull_masks,
                                                GomRecArgs rec) {
    __shared__ float s_P[2][GOM_NSUB][64];
    __shared__ uint32_t s_task[2];
    // Survivors' attributes reach the lanes through a wave-private LDS slab (uniform addresses = broadcast reads) instead of six
    // v_readlane (~6.6 issue cycles each): k_seg_T 117 -> 100 us; as PAIR records (stage_pairs) for the packed evaluation.
    __shared__ float4 s_pr[GOM_NSUB][GOM_PAIR_F4];
    const uint32_t sub_sz = (1u << seg_shift) >> 2;
    if (REC && blockIdx.x == 0) {   // the record allocator of this forward (k_seg_fwd draws from it)
        if (threadIdx.x < GOM_REC_SHARDS) rec.cursor[32 * threadIdx.x] = 0;
        if (threadIdx.x == 0) *rec.overflow = 0;
    }
    if (status->overflow) return;
    const uint32_t nsegs = status->num_segs;
    const int lane = threadIdx.x & 63, sub = threadIdx.x >> 6;  // the 4 waves of a workgroup = the 4 sub-ranges of one (segment, quadrant)
    TaskQueue tq;
    for (tq.init(task_ctr, nsegs, s_task);; tq.advance()) {
        const uint32_t task = tq.current(s_task);
        if (task == 0xffffffffu) break;
        const uint32_t seg = task >> 2;
        const int q = (int)(task & 3);
        const int pxi = q * 64 + lane;
        const uint4 d = seg_desc[seg];
        const uint32_t tile = d.x, start = d.y, cnt = d.z;
        const int tx = tile % gx, ty = (tile / gx) % gy;  // row inside the tile's own frame (batched launches stack the frames)
        const float qx0 = (float)(tx * 16 + (q & 1) * 8), qy0 = (float)(ty * 16 + (q >> 1) * 8);
        const float qx1 = qx0 + 7.f, qy1 = qy0 + 7.f;
        const float pfx = qx0 + (float)(lane & 7), pfy = qy0 + (float)(lane >> 3);
        // pass-through: colours into list order for the compositing / backward kernels (nobody waits on it here)
        if (q == 0 && threadIdx.x < cnt) {
            const uint32_t g = point_list[start + threadIdx.x];
            float4 cl = make_float4(0.f, 0.f, 0.f, 0.f);
            cl.x = colors[(size_t)g * C]; cl.y = colors[(size_t)g * C + 1]; cl.z = colors[(size_t)g * C + 2];
            if (C == 4) cl.w = colors[(size_t)g * C + 3];
            *reinterpret_cast<float4 *>(ent_col + 4 * (size_t)(start + threadIdx.x)) = cl;
        }
        float T = 1.f;
        uint32_t n_pos = 0;   // (REC) surviving entries with alpha > 0 at this lane's pixel
        {
            const EntryRegs<0> r = load_sub<0>(ent_geo, nullptr, start, cnt, sub, lane, qx0, qy0, qx1, qy1, sub_sz);
            tq.request();
            unsigned long long mask = __ballot(r.keep);
            if (lane == 0) cull_masks[((size_t)seg * GOM_NSUB + sub) * 4 + q] = mask;   // for the two passes that follow
            const uint32_t n4 = stage_pairs(s_pr[sub], r.keep, mask, lane, r.x, r.y, r.a, r.b, r.c, r.o);   // (LDS operations of one wave execute in order: no barrier)
            for (uint32_t j = 0; j < n4 / 2; j += 2) {
                const v2f a0 = pair_alpha(load_pair(s_pr[sub], j), pfx, pfy);
                const v2f a1 = pair_alpha(load_pair(s_pr[sub], j + 1), pfx, pfy);
                T = T * (1.f - a0.x);
                T = T * (1.f - a0.y);
                T = T * (1.f - a1.x);
                T = T * (1.f - a1.y);
                if (REC) n_pos += (uint32_t)(a0.x > 0.f) + (uint32_t)(a0.y > 0.f) + (uint32_t)(a1.x > 0.f) + (uint32_t)(a1.y > 0.f);
            }
        }
        if (REC) {
            const uint32_t tot = wave_sum_u32(n_pos);
            if (lane == 0) rec.piece_ub[((size_t)seg * GOM_NSUB + sub) * 4 + q] = tot;
        }
        sub_T[((size_t)seg * GOM_NSUB + sub) * GOM_TPX + pxi] = T;   // (the last piece's row is never read; leaving it out measured 82 us instead of 76)
        float(*sp)[64] = s_P[tq.it & 1];  // double-buffered: the readers of the previous task use the other half
        sp[sub][lane] = T;
        tq.publish(s_task);
        __syncthreads();
        if (sub == 0) seg_T[(size_t)seg * GOM_TPX + pxi] = ((sp[0][lane] * sp[1][lane]) * sp[2][lane]) * sp[3][lane];
    }
    tq.finish();
}

// ---------------------------------------- forward, pass B (exact, per segment)
// Every wave composites its <=32 entries with the reference's exact per-pixel rules (skip, stop at
// T(1-alpha) < 1e-4), starting from the transmittance the pixel has when it reaches the sub-range (product of
// the earlier pieces' prod(1-alpha)).  The four sub-ranges are then folded in LDS: the segment stores its
// contribution, T after it (negated if the stop rule fired inside) and the last contributor; the per-sub-range
// pieces are kept as checkpoints for the backward.
#ifndef GOM_FWD_WAVES
#define GOM_FWD_WAVES 7
#endif
#ifndef GOM_FWD_WAVES_REC
#define GOM_FWD_WAVES_REC 6   // (records mode: the emission's addresses and the colour in front of the entry cost ~10 registers)
#endif
template <int C, bool REC>
__global__ void __launch_bounds__(256, REC ? GOM_FWD_WAVES_REC : GOM_FWD_WAVES) k_seg_fwd(uint32_t seg_shift, int gx, int gy, const uint4 *__restrict__ seg_desc, const float2 *__restrict__ ent_geo,
                                                  const float *__restrict__ ent_col, const float *__restrict__ seg_T,
                                                  const float *__restrict__ sub_T, float *__restrict__ seg_C, float *__restrict__ seg_Tend,
                                                  uint32_t *__restrict__ seg_last, float *__restrict__ sub_C, float *__restrict__ sub_Tend,
                                                  const GomDevStatus *__restrict__ status, uint32_t *__restrict__ task_ctr, uint32_t *__restrict__ seg_cost, const unsigned long long *__restrict__ cull_masks,
                                                  GomRecArgs rec) {
    __shared__ float s_c[GOM_NSUB][C][64];
    __shared__ float s_t[GOM_NSUB][64];
    __shared__ uint32_t s_l[GOM_NSUB][64];
    __shared__ uint32_t s_task[2];
    __shared__ float4 s_e0[GOM_NSUB][64], s_e2[GOM_NSUB][64];   // wave-private broadcast slabs: (x, y, a, b), colour
    __shared__ float2 s_e1[GOM_NSUB][64];                        //                                (c, opacity)
    const uint32_t sub_sz = (1u << seg_shift) >> 2;
    if (status->overflow) return;
    const uint32_t nsegs = status->num_segs;
    const int lane = threadIdx.x & 63, sub = threadIdx.x >> 6;  // the 4 waves of a workgroup = the 4 sub-ranges of one (segment, quadrant)
    TaskQueue tq;
    for (tq.init(task_ctr ? task_ctr + GOM_TQ_WORDS : nullptr, nsegs, s_task);; tq.advance()) {
        const uint32_t task = tq.current(s_task);
        if (task == 0xffffffffu) break;
        const uint32_t seg = task >> 2;
        const int q = (int)(task & 3);
        const int pxi = q * 64 + lane;
        const uint4 d = seg_desc[seg];
        const uint32_t tile = d.x, start = d.y, cnt = d.z, sb = seg - d.w;
        const uint32_t e0 = d.w << seg_shift;
        const int tx = tile % gx, ty = (tile / gx) % gy;  // row inside the tile's own frame (batched launches stack the frames)
        const float qx0 = (float)(tx * 16 + (q & 1) * 8), qy0 = (float)(ty * 16 + (q >> 1) * 8);
        const float qx1 = qx0 + 7.f, qy1 = qy0 + 7.f;
        const float pfx = qx0 + (float)(lane & 7), pfy = qy0 + (float)(lane >> 3);
#ifndef GOM_FWD_LATE_ENTRIES
#define GOM_FWD_LATE_ENTRIES 1   // 1: the entries are loaded once the task is known to be alive (two thirds are not); 0: with the transmittance prefix
#endif
#if !GOM_FWD_LATE_ENTRIES
        const EntryRegs<C> r = load_sub<C>(ent_geo, ent_col, start, cnt, sub, lane, qx0, qy0, qx1, qy1, sub_sz, cull_masks + ((size_t)seg * GOM_NSUB + sub) * 4 + q);
#endif
        // transmittance at the start of this sub-range; 0 = the pixel has certainly stopped earlier.
        // (In the 1e-5-wide borderline band the pixel stays alive; the fold below / the combine pass honour
        //  the exact stop flag of the earlier piece.)
        float T = 1.f;
        for (uint32_t r0 = sb; r0 < seg; r0 += 8) {
            float p[8];
#pragma unroll
            for (int u = 0; u < 8; u++) p[u] = seg_T[(size_t)min(r0 + u, seg - 1) * GOM_TPX + pxi];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                if (r0 + u < seg) {
                    const float Tn = T * p[u];
                    T = (Tn >= kStopT * 0.99999f) ? Tn : 0.f;
                }
            }
            // (measured, round 6: ending this walk once every pixel of the quadrant has stopped -- a ballot per eight rows -- changes nothing: 108.2 against 107.5 us)
        }
        {
            float p[GOM_NSUB - 1];
#pragma unroll
            for (int u = 0; u < GOM_NSUB - 1; u++) p[u] = sub_T[((size_t)seg * GOM_NSUB + u) * GOM_TPX + pxi];
#pragma unroll
            for (int u = 0; u < GOM_NSUB - 1; u++) {
                if (u < sub) {
                    const float Tn = T * p[u];
                    T = (Tn >= kStopT * 0.99999f) ? Tn : 0.f;
                }
            }
        }
        float wl = T > 0.f ? 1.f : 0.f;  // lane still compositing (float mask: no SALU in the chain)
        tq.request();  // (behind every load of this task)
        const bool any_alive = __syncthreads_or(wl != 0.f ? 1 : 0) != 0;  // also fences the LDS of the previous segment
        if (!any_alive) {  // every pixel of the quadrant stopped before this segment
            if (sub == 0) {   // (the assembly pass needs seg_Tend = 0 only, but loads all three rows of a segment in one batch: with the other two
                              //  left unwritten -- 24 MB of stores less per 8-frame launch -- its loads came from HBM instead of the L2 / MALL: k_combine_fwd 25 -> 29 us)
                const size_t o = (size_t)seg * GOM_TPX + pxi;
                seg_Tend[o] = 0.f;
                seg_last[o] = 0;
                const float zero[C] = {};
                st4<C>(seg_C, seg, pxi, zero);
            }
            tq.publish(s_task);
            __syncthreads();
            continue;
        }
#if GOM_FWD_LATE_ENTRIES
        const EntryRegs<C> r = load_sub<C>(ent_geo, ent_col, start, cnt, sub, lane, qx0, qy0, qx1, qy1, sub_sz, cull_masks + ((size_t)seg * GOM_NSUB + sub) * 4 + q);
#endif
        float acc[C];
#pragma unroll
        for (int ch = 0; ch < C; ch++) acc[ch] = 0.f;
        uint32_t last = 0;
        if (__ballot(wl != 0.f) != 0ull) {
            unsigned long long mask = __ballot(r.keep);
            // (REC) the piece's record region: upper bound from k_seg_T, one dequeue-like atomic per live piece
            const uint32_t piece = ((seg * GOM_NSUB + (uint32_t)sub) << 2) | (uint32_t)q;
            uint32_t rbase = 0, rcur = 0, rub = 0, my_cnt = 0;   // my_cnt: records of entry `lane` of the sub-range (v_writelane, one per entry)
            bool rec_on = false;
            if (REC) {
                rub = __builtin_amdgcn_readfirstlane(rec.piece_ub[piece]);
                const uint32_t shard = piece % GOM_REC_SHARDS;
                uint32_t b = 0;
                if (lane == 0 && rub) b = atomicAdd(rec.cursor + 32 * shard, rub);
                b = __builtin_amdgcn_readfirstlane(b);
                rec_on = rub != 0u && b + rub <= rec.shard_cap;
                if (rub != 0u && !rec_on && lane == 0) atomicOr(rec.overflow, 1u);
                rbase = shard * rec.shard_cap + b;
                rcur = rbase;
            }
            // what the backward will pay for this piece, roughly: the entries that reach alive pixels here (GomBwdOrderRider)
            if (seg_cost && lane == 0 && mask) seg_cost[16 * (size_t)seg + 4 * sub + q] = (uint32_t)__popcll(mask);   // (one writer per word)
            s_e0[sub][lane] = make_float4(r.x, r.y, r.a, r.b);   // (LDS operations of one wave execute in order: no barrier)
            s_e1[sub][lane] = make_float2(r.c, r.o);
            {
                float4 cl = make_float4(r.col[0], 0.f, 0.f, 0.f);
                if (C > 1) cl.y = r.col[1 % C];
                if (C > 2) cl.z = r.col[2 % C];
                if (C > 3) cl.w = r.col[3 % C];
                s_e2[sub][lane] = cl;
            }
            while (mask) {
                int kk[GOM_FWD_EPT];
                float al[GOM_FWD_EPT], ecol[GOM_FWD_EPT][C];
#pragma unroll
                for (int u = 0; u < GOM_FWD_EPT; u++) {  // independent alpha evaluations (ILP)
                    const bool kv = mask != 0ull;
                    const int k = kv ? __builtin_ctzll(mask) : 0;
                    mask &= mask - 1;
                    kk[u] = k;
                    const float4 e0 = s_e0[sub][k], e2 = s_e2[sub][k];
                    const float2 e1 = s_e1[sub][k];
                    const float eo = kv ? e1.y : -INFINITY;  // lo of opacity 0 -> alpha 0 -> "skip"
                    al[u] = entry_alpha(e0.x, e0.y, e0.z, e0.w, e1.x, eo, pfx, pfy);
                    const float cv[4] = {e2.x, e2.y, e2.z, e2.w};
#pragma unroll
                    for (int ch = 0; ch < C; ch++) ecol[u][ch] = cv[ch];
                }
#pragma unroll
                for (int u = 0; u < GOM_FWD_EPT; u++) {  // the serial chain: T -> test_T -> select
                    const float a = al[u] * wl;
                    // no lane blends this entry (it misses every pixel that is still compositing: a quarter of the survivors of the
                    // conservative cull): the chain below would leave T, wl, acc and last exactly as they are (wave-uniform skip)
                    if (GOM_FWD_SKIP_IDLE && __ballot(a > 0.f) == 0ull) continue;
                    const float test_T = T * (1.f - a);
                    const bool cont = test_T >= kStopT;  // reference: `test_T < 0.0001f -> done`
                    const float w = cont ? a * T : 0.f;
                    if (REC) {   // one record per blending lane, compacted: T and the piece's colour IN FRONT of the entry
                        const bool em = w > 0.f;
                        const unsigned long long bm = __ballot(em);
                        if (bm != 0ull && rec_on) {
                            if (em) {
                                const uint32_t idx = rcur + __builtin_amdgcn_mbcnt_hi((uint32_t)(bm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bm, 0u));
                                rec.rec_acc[idx] = make_float4(acc[0], C > 1 ? acc[1 % C] : 0.f, C > 2 ? acc[2 % C] : 0.f, C > 3 ? acc[3 % C] : 0.f);
                                rec.rec_ti[idx] = make_float2(T, __uint_as_float(((uint32_t)kk[u] << 6) | (uint32_t)lane));
                            }
                            const uint32_t nb = (uint32_t)__popcll(bm);
                            rcur += nb;
                            asm("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(my_cnt) : "s"(nb), "s"((uint32_t)kk[u]) : "m0");   // lane kk[u] <- nb (both wave-uniform; one SGPR per VALU operand set: the lane select goes through M0)
                        }
                    }
#pragma unroll
                    for (int ch = 0; ch < C; ch++) acc[ch] += ecol[u][ch] * w;
                    T = cont ? test_T : T;
                    wl = cont ? wl : 0.f;
                    last = (w > 0.f) ? (e0 + (uint32_t)sub * sub_sz + (uint32_t)kk[u] + 1u) : last;
                }
                if (__ballot(wl != 0.f) == 0ull) break;
            }
            if (REC) {
                rec.piece_cnt[(size_t)piece * 64 + lane] = (uint8_t)my_cnt;
                if (lane == 0) rec.piece_rec[piece] = make_uint2(rbase, rec_on ? rcur - rbase : (rub ? 0xffffffffu : 0u));
            }
        }
        // dead on arrival: T = 0.  stopped inside: -T.  still going: +T.
        s_t[sub][lane] = (T > 0.f && wl == 0.f) ? -T : T;
        s_l[sub][lane] = last;
#pragma unroll
        for (int ch = 0; ch < C; ch++) s_c[sub][ch][lane] = acc[ch];
        tq.publish(s_task);
        __syncthreads();
        if (sub == 0) {  // fold the four pieces of this segment for pixel pxi
            float Tc = 0.f, tot[C];
#pragma unroll
            for (int ch = 0; ch < C; ch++) tot[ch] = 0.f;
            uint32_t lastc = 0;
            float cadd[GOM_NSUB][C];
            bool going = true, stopped = false, any = false;
            // Pieces every pixel of the quadrant has stopped in front of: no pixel's last contributor lies in or behind them, so the
            // backward (which starts a (sub-range, quadrant) only when the quadrant's largest n_contrib reaches into it) never reads
            // their checkpoint rows -- they are not written (GOM_FWD_SKIP_DEAD_ROWS; ~40 of this kernel's 158 MB per 8-frame launch).
            int n_live = GOM_NSUB;
#pragma unroll
            for (int u = 0; u < GOM_NSUB; u++) {
                const float te = s_t[u][lane];
                const bool counts = going && te != 0.f;
                if (going && te == 0.f) going = false;  // had stopped before this piece
                if (GOM_FWD_SKIP_DEAD_ROWS && n_live == GOM_NSUB && __ballot(counts) == 0ull) n_live = u;   // (wave-uniform; dead once = dead behind)
                if (counts) {
                    any = true;
#pragma unroll
                    for (int ch = 0; ch < C; ch++) tot[ch] += s_c[u][ch][lane];
                    lastc = s_l[u][lane] ? s_l[u][lane] : lastc;
                    Tc = fabsf(te);
                    if (te < 0.f) { going = false; stopped = true; }
                }
                // checkpoints for the backward: T behind this piece, and (below) the colour the LATER pieces of the segment really added
                if (!REC && u < n_live) sub_Tend[((size_t)seg * GOM_NSUB + u) * GOM_TPX + pxi] = Tc;   // (REC: T travels in the records)
#pragma unroll
                for (int ch = 0; ch < C; ch++) cadd[u][ch] = counts ? s_c[u][ch][lane] : 0.f;
            }
            {   // sub_C[u] = sum of the pieces behind piece u, last piece first (small terms first): with seg_Sbehind the backward has
                // the colour still to come behind ITS piece from two loads (it used to add up to three rows itself)
                float run[C];
#pragma unroll
                for (int ch = 0; ch < C; ch++) run[ch] = 0.f;
#pragma unroll
                for (int u = GOM_NSUB - 1; u >= 0; u--) {
                    if (!REC && u < GOM_NSUB - 1 && u < n_live) st4<C>(sub_C, (size_t)seg * GOM_NSUB + u, pxi, run);
#pragma unroll
                    for (int ch = 0; ch < C; ch++) run[ch] += cadd[u][ch];
                    if (REC && u < n_live) st4<C>(sub_C, (size_t)seg * GOM_NSUB + u, pxi, run);   // (REC) INCLUSIVE: from the piece's first entry to the end of the segment
                }
            }
            const size_t o = (size_t)seg * GOM_TPX + pxi;
            seg_Tend[o] = !any ? 0.f : (stopped ? -Tc : Tc);
            seg_last[o] = lastc;
            st4<C>(seg_C, seg, pxi, tot);
        }
    }
    tq.finish();
}

// ----------------------------------------------- forward, pass C (assembly) -
template <int C>
__device__ __forceinline__ void combine_tile(const int tile, int H, int W, int gx, int gy, float bg0, float bg1, float bg2, float bg3,
                                                     const GomCamera *__restrict__ cams,
                                                     const uint32_t *__restrict__ seg_base, const float *__restrict__ seg_C,
                                                     const uint32_t *__restrict__ seg_last, float *__restrict__ seg_Tend,
                                                     float *__restrict__ seg_Sbehind, float *__restrict__ out_color,
                                                     float *__restrict__ final_T, uint32_t *__restrict__ n_contrib,
                                                     uint32_t *__restrict__ tile_nmax, uint4 *__restrict__ seg_qmax,
                                                     const GomDevStatus *__restrict__ status, const uint32_t *__restrict__ tile_base,
                                                     const uint32_t *__restrict__ point_list, const uint32_t *__restrict__ rank_of,
                                                     uint32_t *__restrict__ tile_qlim, int skip_empty, const GomLossRider &lr) {
    __shared__ uint32_t s_nmax[4], s_qmax[4];
    __shared__ float s_lsum[2][4];
    const int tx = tile % gx, fr = (tile / gx) / gy, ty = (tile / gx) % gy;  // fr: frame of a batched launch
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int px = tx * 16 + (wave & 1) * 8 + (lane & 7);
    const int py = ty * 16 + (wave >> 1) * 8 + (lane >> 3);
    const bool inside = px < W && py < H;
    const size_t HW = (size_t)H * W;
    const size_t pix = (size_t)py * W + px;
    out_color += (size_t)fr * C * HW;
    final_T += (size_t)fr * HW;
    n_contrib += (size_t)fr * HW;
    float bg[4] = {bg0, bg1, bg2, bg3};
    if (cams) {
#pragma unroll
        for (int ch = 0; ch < 4; ch++) bg[ch] = cams[fr].bg[ch];
    }
    if (status->overflow) {  // pair buffers too small: poison loudly
        if (inside) {
            const float nanv = __uint_as_float(0x7fc00000u);
#pragma unroll
            for (int ch = 0; ch < C; ch++) out_color[ch * HW + pix] = nanv;
            final_T[pix] = nanv;
            n_contrib[pix] = 0;
        }
        if (threadIdx.x == 0) {
            tile_nmax[tile] = 0;
            if (rank_of) tile_qlim[tile] = 0;
            if (C == 4 && lr.gt_rgb) *reinterpret_cast<float2 *>(lr.partials + 2 * ((size_t)fr * lr.slots + (tile - fr * gx * gy))) = make_float2(__uint_as_float(0x7fc00000u), __uint_as_float(0x7fc00000u));
        }
        return;
    }
    const uint32_t sb = seg_base[tile], nseg = seg_base[tile + 1] - sb;
    if (nseg == 0 && skip_empty) return;   // this forward's k_emit painted the tile (background, T = 1, no contributor; tile_nmax zeroed by the scan)
    // GomLossRider: the pixel's targets, loaded here -- in the shadow of the walk over the segments -- for the loss at the end
    float3 lr_g = make_float3(0.f, 0.f, 0.f);
    float lr_gm = 0.f, lr_b[3] = {0.f, 0.f, 0.f};
    if (C == 4 && lr.gt_rgb && inside) {
        const size_t fp = (size_t)fr * HW + pix;
        lr_g = *reinterpret_cast<const float3 *>(lr.gt_rgb + 3 * fp);
        lr_gm = lr.gt_mask[fp];
        lr_b[0] = lr.bgcolor[3 * fr]; lr_b[1] = lr.bgcolor[3 * fr + 1]; lr_b[2] = lr.bgcolor[3 * fr + 2];
    }
    float T = 1.f, acc[C];
#pragma unroll
    for (int ch = 0; ch < C; ch++) acc[ch] = 0.f;
    uint32_t last = 0;
    int s_stop = (int)nseg - 1;  // last segment that contributed to this pixel
    bool going = true;
    // 8 segments per trip: all loads of a trip are issued before the (cheap, serial) fold, so the list of
    // segments costs one memory latency per 8 instead of one per segment.
    float cs0[8][C];   // the colours of the first trip: tiles of up to 8 segments (nearly all) make their second pass from registers
    for (uint32_t s0 = 0; s0 < nseg; s0 += 8) {
        float te[8], cs[8][C];
        uint32_t ll[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const uint32_t sl = min(s0 + u, nseg - 1);
            const size_t o = (size_t)(sb + sl) * GOM_TPX + threadIdx.x;
            te[u] = seg_Tend[o];
            ll[u] = seg_last[o];
            ld4<C>(seg_C, sb + sl, threadIdx.x, cs[u]);
        }
        if (s0 == 0) {
#pragma unroll
            for (int u = 0; u < 8; u++)
#pragma unroll
                for (int ch = 0; ch < C; ch++) cs0[u][ch] = cs[u][ch];
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const uint32_t sl = s0 + u;
            if (sl < nseg) {
                if (going) {
                    if (te[u] == 0.f) {  // the pixel had stopped before this segment
                        going = false;
                        s_stop = (int)sl - 1;
                    } else {
#pragma unroll
                        for (int ch = 0; ch < C; ch++) acc[ch] += cs[u][ch];
                        last = ll[u] ? ll[u] : last;
                        T = fabsf(te[u]);
                        if (te[u] < 0.f) {  // stop rule fired inside this segment
                            going = false;
                            s_stop = (int)sl;
                        }
                    }
                }
                // checkpoint for the backward: T behind the segment (0 once the pixel is finished)
                seg_Tend[(size_t)(sb + sl) * GOM_TPX + threadIdx.x] = (going || (int)sl <= s_stop) ? T : 0.f;
            }
        }
    }
    // depth ranking: 1 + rank of the pixel's last contributor.  The tile's maximum (= the rank of its last contributing entry: ranks grow
    // along the list) is what the per-Gaussian backward compares a Gaussian's own rank with.  Every lane fetches its own -- two dependent
    // gathers issued HERE, in the shadow of the second pass -- where thread 0 used to walk tile_base -> point_list -> rank_of alone
    // behind the barrier at the end.
    uint32_t qrank = 0;
    if (rank_of && inside && last) qrank = rank_of[point_list[tile_base[tile] + last - 1u]] + 1u;
    // colour still to come behind each segment (small terms first: accurate suffix sums)
    {
        float S[C];
#pragma unroll
        for (int ch = 0; ch < C; ch++) S[ch] = 0.f;
        if (nseg <= 8) {
#pragma unroll
            for (int u = 7; u >= 0; u--) {
                if (u < (int)nseg) {
                    st4<C>(seg_Sbehind, (size_t)(sb + u), threadIdx.x, S);
#pragma unroll
                    for (int ch = 0; ch < C; ch++)
                        if (u <= s_stop) S[ch] += cs0[u][ch];
                }
            }
        } else {
            for (int s1 = (int)nseg; s1 > 0; s1 -= 8) {
                float cs[8][C];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int sl = max(s1 - 1 - u, 0);
                    ld4<C>(seg_C, (size_t)(sb + sl), threadIdx.x, cs[u]);
                }
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int sl = s1 - 1 - u;
                    if (sl >= 0) {
                        st4<C>(seg_Sbehind, (size_t)(sb + sl), threadIdx.x, S);
#pragma unroll
                        for (int ch = 0; ch < C; ch++)
                            if (sl <= s_stop) S[ch] += cs[u][ch];
                    }
                }
            }
        }
    }
    float l_rgb = 0.f, l_mask = 0.f;   // GomLossRider: this pixel's |residuals|
    if (inside) {
        final_T[pix] = T;
        n_contrib[pix] = last;
        float v[C];
#pragma unroll
        for (int ch = 0; ch < C; ch++) { v[ch] = acc[ch] + T * bg[ch]; out_color[ch * HW + pix] = v[ch]; }
        if (C == 4 && lr.gt_rgb) {   // the frame step's loss on the pixel just assembled: its gradient goes out next to the image (l1_pixel.hpp)
            const GomL1Px o = gom_l1_pixel(v[0], v[1], v[2], v[C - 1], 1.f, lr_b[0], lr_b[1], lr_b[2], lr_g.x, lr_g.y, lr_g.z, lr_gm, lr.k_rgb, lr.k_mask);
            float *dp = lr.dpred + (size_t)fr * 4 * HW + pix;
            dp[0] = o.d0; dp[HW] = o.d1; dp[2 * HW] = o.d2; dp[3 * HW] = o.d3;
            l_rgb = o.abs_rgb; l_mask = o.abs_mask;
        }
    }
    const uint32_t wmax = wave_max_u32(inside ? last : 0u);
    const uint32_t wq = rank_of ? wave_max_u32(qrank) : 0u;
    if (C == 4 && lr.gt_rgb) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { l_rgb += __shfl_xor(l_rgb, d, 64); l_mask += __shfl_xor(l_mask, d, 64); }
    }
    if (lane == 0) { s_nmax[wave] = wmax; s_qmax[wave] = wq; s_lsum[0][wave] = l_rgb; s_lsum[1][wave] = l_mask; }
    __syncthreads();
    if (threadIdx.x == 0) {
        tile_nmax[tile] = max(max(s_nmax[0], s_nmax[1]), max(s_nmax[2], s_nmax[3]));
        if (rank_of) tile_qlim[tile] = max(max(s_qmax[0], s_qmax[1]), max(s_qmax[2], s_qmax[3]));
        if (C == 4 && lr.gt_rgb) *reinterpret_cast<float2 *>(lr.partials + 2 * ((size_t)fr * lr.slots + (tile - fr * gx * gy))) = make_float2((s_lsum[0][0] + s_lsum[0][1]) + (s_lsum[0][2] + s_lsum[0][3]), (s_lsum[1][0] + s_lsum[1][1]) + (s_lsum[1][2] + s_lsum[1][3]));
    }
    // the four quadrant maxima once per SEGMENT of the tile: the backward finds them with the segment index alone
    for (uint32_t i = threadIdx.x; i < nseg; i += 256) seg_qmax[sb + i] = make_uint4(s_nmax[0], s_nmax[1], s_nmax[2], s_nmax[3]);
}

// GomLossRider, empty tiles: rider workgroups of k_combine_fwd stride over the tiles in groups of four; an empty tile's prediction is the
// background k_emit painted, so its loss needs the targets alone.  All loads of a group are issued before the sums (one latency per group).
#define GOM_LOSS_RIDER_BLOCKS 2048
__device__ __forceinline__ void loss_empty_tiles_rider(const GomLossRider &lr, const uint32_t rid, const uint32_t n_rid, int H, int W, int gx, int gy, int n_tiles,
                                                       float bg0, float bg1, float bg2, float bg3, const GomCamera *__restrict__ cams,
                                                       const uint32_t *__restrict__ seg_base) {
    __shared__ float s_fill[4][4][2];
    const int per = gx * gy;
    const size_t HW = (size_t)H * W;
    for (int t0 = (int)rid * 4; t0 < n_tiles; t0 += (int)n_rid * 4) {
        bool emp[4], ins[4];
        size_t pixs[4];
        int frs[4];
        float3 g[4];
        float gm[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int t = t0 + k;
            emp[k] = false; ins[k] = false; pixs[k] = 0; frs[k] = 0; g[k] = make_float3(0.f, 0.f, 0.f); gm[k] = 0.f;
            if (t < n_tiles) {
                emp[k] = seg_base[t + 1] == seg_base[t];
                const int fr = t / per, tl = t - fr * per;
                const int px = (tl % gx) * 16 + (threadIdx.x & 15), py = (tl / gx) * 16 + (threadIdx.x >> 4);
                ins[k] = px < W && py < H;
                frs[k] = fr;
                pixs[k] = (size_t)fr * HW + (size_t)py * W + px;
            }
        }
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (emp[k] && ins[k]) { g[k] = *reinterpret_cast<const float3 *>(lr.gt_rgb + 3 * pixs[k]); gm[k] = lr.gt_mask[pixs[k]]; }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (!emp[k]) continue;   // (uniform over the workgroup)
            float sr = 0.f, sm = 0.f;
            if (ins[k]) {
                const int fr = frs[k];
                float bg[4] = {bg0, bg1, bg2, bg3};
                if (cams) {
#pragma unroll
                    for (int ch = 0; ch < 4; ch++) bg[ch] = cams[fr].bg[ch];
                }
                const GomL1Px o = gom_l1_pixel(bg[0], bg[1], bg[2], bg[3], 1.f, lr.bgcolor[3 * fr], lr.bgcolor[3 * fr + 1], lr.bgcolor[3 * fr + 2], g[k].x, g[k].y, g[k].z, gm[k],
                                               lr.k_rgb, lr.k_mask);
                sr = o.abs_rgb; sm = o.abs_mask;
                if (lr.zero_empty) {   // (split frame call: the caller reads the gradient image itself)
                    const size_t pix = pixs[k] - (size_t)fr * HW;
#pragma unroll
                    for (int ch = 0; ch < 4; ch++) lr.dpred[((size_t)fr * 4 + ch) * HW + pix] = 0.f;
                }
            }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) { sr += __shfl_xor(sr, d, 64); sm += __shfl_xor(sm, d, 64); }
            if ((threadIdx.x & 63) == 0) { s_fill[k][threadIdx.x >> 6][0] = sr; s_fill[k][threadIdx.x >> 6][1] = sm; }
        }
        __syncthreads();
        if (threadIdx.x < 4 && t0 + (int)threadIdx.x < n_tiles && seg_base[t0 + threadIdx.x + 1] == seg_base[t0 + threadIdx.x]) {
            const int t = t0 + (int)threadIdx.x, fr = t / per;
            *reinterpret_cast<float2 *>(lr.partials + 2 * ((size_t)fr * lr.slots + (t - fr * per))) =
                make_float2((s_fill[threadIdx.x][0][0] + s_fill[threadIdx.x][1][0]) + (s_fill[threadIdx.x][2][0] + s_fill[threadIdx.x][3][0]),
                            (s_fill[threadIdx.x][0][1] + s_fill[threadIdx.x][1][1]) + (s_fill[threadIdx.x][2][1] + s_fill[threadIdx.x][3][1]));
        }
        __syncthreads();
    }
    // (a frame of fewer tiles than slots: the slots behind its tiles are zero)
    if (rid == 0 && per < lr.slots) {
        const int B = n_tiles / per;
        for (int i = (int)threadIdx.x; i < B * (lr.slots - per); i += 256) {
            const int fr = i / (lr.slots - per), j = per + i % (lr.slots - per);
            *reinterpret_cast<float2 *>(lr.partials + 2 * ((size_t)fr * lr.slots + j)) = make_float2(0.f, 0.f);
        }
    }
}

// One workgroup per tile -- or, when this forward's k_emit has painted the empty tiles (five in six on a body) and the scan kernel has
// listed the others (`work`: the items of the tile pass, one per tile and window of list positions), a grid that strides over the list:
// 6 800 workgroups that start only to find their tile empty were most of this launch's dispatch.
template <int C>
__global__ void __launch_bounds__(256) k_combine_fwd(int H, int W, int gx, int gy, float bg0, float bg1, float bg2, float bg3,
                                                     const GomCamera *__restrict__ cams,
                                                     const uint32_t *__restrict__ seg_base, const float *__restrict__ seg_C,
                                                     const uint32_t *__restrict__ seg_last, float *__restrict__ seg_Tend,
                                                     float *__restrict__ seg_Sbehind, float *__restrict__ out_color,
                                                     float *__restrict__ final_T, uint32_t *__restrict__ n_contrib,
                                                     uint32_t *__restrict__ tile_nmax, uint4 *__restrict__ seg_qmax,
                                                     const GomDevStatus *__restrict__ status, const uint32_t *__restrict__ tile_base,
                                                     const uint32_t *__restrict__ point_list, const uint32_t *__restrict__ rank_of,
                                                     uint32_t *__restrict__ tile_qlim, int skip_empty, const uint32_t *__restrict__ work, int n_tiles,
                                                     GomBwdOrderRider rider, GomLossRider lr, int n_loss_riders) {
    // (frame step, batched) the first eight workgroups order the backward's task queue: bwd_order.hpp
    const uint32_t n_rid = rider.status ? 8u : 0u;
    if (blockIdx.x < n_rid) { gom_bwd_order_rider(rider, blockIdx.x); return; }
    // (frame step) the LAST workgroups sum the loss of the empty tiles (GomLossRider): bandwidth work in the slots the tiles' latency chains leave free
    const uint32_t n_lr = (C == 4 && lr.gt_rgb) ? (uint32_t)n_loss_riders : 0u;
    if (blockIdx.x >= gridDim.x - n_lr) {
        if (!status->overflow) loss_empty_tiles_rider(lr, blockIdx.x - (gridDim.x - n_lr), n_lr, H, W, gx, gy, n_tiles, bg0, bg1, bg2, bg3, cams, seg_base);
        return;
    }
    const uint32_t bx = blockIdx.x - n_rid, nbx = gridDim.x - n_rid - n_lr;
#define GOM_COMBINE_TILE(T) combine_tile<C>((T), H, W, gx, gy, bg0, bg1, bg2, bg3, cams, seg_base, seg_C, seg_last, seg_Tend, seg_Sbehind, out_color, final_T, \
                                            n_contrib, tile_nmax, seg_qmax, status, tile_base, point_list, rank_of, tile_qlim, skip_empty, lr)
    if (!work) { GOM_COMBINE_TILE((int)bx); }
    else if (status->overflow) {   // (no work items then: every tile is poisoned)
        for (int tile = (int)bx; tile < n_tiles; tile += (int)nbx) { GOM_COMBINE_TILE(tile); __syncthreads(); }
    } else {
        const uint32_t n_work = status->n_work_items;
        for (uint32_t wi = bx; wi < n_work; wi += nbx) {
            const uint32_t item = work[wi];
            if (item >> 24) continue;   // (further windows of a long list: the tile has been taken with window 0)
            GOM_COMBINE_TILE((int)(item & 0xffffffu));
            __syncthreads();            // s_nmax / s_qmax of this tile have been read
        }
    }
#undef GOM_COMBINE_TILE
}

// ---------------------------------------------------------------- backward -
#include "seg_bwd_replay.hpp"
#ifndef GOM_BWD_WAVES
#define GOM_BWD_WAVES 6
#endif
template <int C>
__global__ void __launch_bounds__(256, GOM_BWD_WAVES) k_seg_bwd(uint32_t seg_shift, int H, int W, int gx, int gy, float bg0, float bg1, float bg2, float bg3,
                                                  const GomCamera *__restrict__ cams,
                                                  const uint4 *__restrict__ seg_desc, const uint4 *__restrict__ seg_qmax,
                                                  const float2 *__restrict__ ent_geo, const float *__restrict__ ent_col,
                                                  const float *__restrict__ final_T, const uint32_t *__restrict__ n_contrib,
                                                  const float *__restrict__ dL_dpix, const float *__restrict__ sub_Tend,
                                                  const float *__restrict__ sub_C, const float *__restrict__ seg_Sbehind,
                                                  const uint32_t *__restrict__ ent_slot, float *__restrict__ partial, const GomDevStatus *__restrict__ status,
                                                  uint32_t *__restrict__ task_ctr, const unsigned long long *__restrict__ cull_masks) {
    // [task parity][quadrant][entry of the sub-range][value]; s_done = which entries the quadrant's wave really wrote.
    // Double-buffered by task parity and never cleared: the flush reads only the rows s_done names, and the next task
    // writes the other half, so one barrier per task (before the flush) is all the synchronisation there is.
    __shared__ float s_acc[2][4][GOM_SUB_MAX][10];   // rows by COMPACTED survivor position (seg_bwd_replay.hpp)
    __shared__ unsigned long long s_done[2][4], s_mask[2][4];   // pairs of rows written; the survivors (list order) they belong to
    __shared__ uint32_t s_task[2];
    __shared__ float4 s_slab[4][GOM_BPAIR_F4];       // wave-private pair records of the survivors, geometry and colours
    const uint32_t sub_sz = (1u << seg_shift) >> 2;
    if (status->overflow) return;
    const uint32_t nsegs = status->num_segs;
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;  // the 4 waves of a workgroup = the 4 quadrants of one (segment, sub-range)
    const int pxi = q * 64 + lane;
    bool from_n1;
    const int slot20 = wave_sum20_slot(lane, from_n1);   // where this lane's share of a reduced pair goes (or -1)
    const size_t HW = (size_t)H * W;
    TaskQueue tq;
    for (tq.init(task_ctr ? task_ctr + 2 * GOM_TQ_WORDS : nullptr, nsegs, s_task);; tq.advance()) {
        const uint32_t task = tq.current(s_task);
        if (task == 0xffffffffu) break;
        const uint32_t seg = task >> 2;
        const int sub = (int)(task & 3);
        const int buf = (int)(tq.it & 1u);
        // Both per-segment records are indexed by the segment alone: one memory round trip tells the workgroup whether
        // the sub-range is dead (every pixel of the tile finished before it) and each wave whether its quadrant is.
        const uint4 d = seg_desc[seg];
        const uint4 qm4 = seg_qmax[seg];
        const uint32_t tile = d.x, start = d.y, cnt = d.z;
        const uint32_t e0 = d.w << seg_shift;
        const uint32_t s0 = e0 + (uint32_t)sub * sub_sz;  // list index of the first entry of the sub-range
        const bool empty = (uint32_t)sub * sub_sz >= cnt;
        const uint32_t scnt = empty ? 0u : min(sub_sz, cnt - (uint32_t)sub * sub_sz);
        const uint32_t tmax = max(max(qm4.x, qm4.y), max(qm4.z, qm4.w));
        if (empty || s0 >= tmax) {
            // No entries, or every pixel of the tile stopped before this sub-range.  Nothing is written: the per-Gaussian backward
            // skips entries at or beyond the tile's last contributor (tile_nmax) without reading their record.
            tq.request();
            tq.publish(s_task);
            __syncthreads();
            continue;
        }
        const uint32_t wmax = q == 0 ? qm4.x : (q == 1 ? qm4.y : (q == 2 ? qm4.z : qm4.w));  // max n_contrib over this wave's 8x8 pixels
        // record slot of "my" entry (Gaussian-major: the per-Gaussian backward reads a Gaussian's records as one contiguous run)
        const uint32_t my_slot = threadIdx.x < scnt ? ent_slot[start + (uint32_t)sub * sub_sz + threadIdx.x] : 0u;
        unsigned long long done = 0ull, smask = 0ull;
        bool requested = false;
        if (wmax > s0) {
            const int tx = tile % gx, fr = (tile / gx) / gy, ty = (tile / gx) % gy;  // fr: frame of a batched launch
            const int px = tx * 16 + (q & 1) * 8 + (lane & 7);
            const int py = ty * 16 + (q >> 1) * 8 + (lane >> 3);
            const bool inside = px < W && py < H;
            const size_t pix = (size_t)py * W + px;
            const size_t fpix = (size_t)fr * HW + pix;  // per-pixel state of the stacked frames
            const float pfx = (float)px, pfy = (float)py;
            const float qx0 = (float)(tx * 16 + (q & 1) * 8), qy0 = (float)(ty * 16 + (q >> 1) * 8);
            const float qx1 = qx0 + 7.f, qy1 = qy0 + 7.f;
            // every load of the task is issued here, back to back
            const uint32_t my_last = inside ? n_contrib[fpix] : 0u;
            const float T_final = final_T[inside ? fpix : 0];
            float dpix[C], bg_dot = 0.f;
            {
                float bg[4] = {bg0, bg1, bg2, bg3};
                if (cams) {
#pragma unroll
                    for (int ch = 0; ch < 4; ch++) bg[ch] = cams[fr].bg[ch];
                }
#pragma unroll
                for (int ch = 0; ch < C; ch++) {
                    dpix[ch] = inside ? dL_dpix[((size_t)fr * C + ch) * HW + pix] : 0.f;
                    bg_dot += bg[ch] * dpix[ch];
                }
            }
            // checkpoint: state just behind this sub-range (T there; colour still to come = behind the segment
            // + the later pieces of this segment, smallest terms first)
            float T = sub_Tend[((size_t)seg * GOM_NSUB + sub) * GOM_TPX + pxi];
            // App. A.4 keeps accum_rec[ch] / last_color[ch] per channel, but they only ever meet the gradient through their dot product
            // with this pixel's dL/dpix: the recurrence is linear, so it is carried as two scalars R = accum_rec . dpix and
            // U_last = last_color . dpix (half the instructions of the serial chain, six registers fewer).
            float R_acc;
            float S[C], cu[C];
            ld4<C>(seg_Sbehind, seg, pxi, S);
            if (sub < GOM_NSUB - 1) ld4<C>(sub_C, (size_t)seg * GOM_NSUB + sub, pxi, cu);   // what the later pieces of the segment added (wave-uniform condition)
            const uint32_t lim = min(cnt, wmax - e0);  // entries at or beyond wmax are dead for this wave
            const EntryRegs<C> r = load_sub<C>(ent_geo, ent_col, start, lim, sub, lane, qx0, qy0, qx1, qy1, sub_sz, cull_masks + ((size_t)seg * GOM_NSUB + sub) * 4 + q);
            tq.request();  // (behind every load of this task)
            requested = true;
            const float invT = T > 0.f ? 1.f / T : 0.f;
            if (sub < GOM_NSUB - 1) {
#pragma unroll
                for (int ch = 0; ch < C; ch++) S[ch] += cu[ch];
            }
            {
                float sd = 0.f;
#pragma unroll
                for (int ch = 0; ch < C; ch++) sd += S[ch] * dpix[ch];
                R_acc = sd * invT;
            }
            const unsigned long long mask = __ballot(r.keep);
            const uint32_t npairs = stage_bwd_pairs<C>(s_slab[q], r, mask, lane);
            const uint32_t pos_limit = survivors_before(mask, my_last > s0 ? my_last - s0 : 0u);
            done = bwd_replay<C>(s_slab[q], npairs, &s_acc[buf][q][0][0], dpix, pfx, pfy, T, R_acc, T_final, bg_dot, pos_limit, slot20, from_n1);
            smask = mask;
        }
        if (lane == 0) { s_done[buf][q] = done; s_mask[buf][q] = smask; }
        if (!requested) tq.request();
        tq.publish(s_task);
        __syncthreads();
        if (threadIdx.x < scnt) {  // one 48-byte record per entry, quadrants summed in a fixed order
            float rr[10];
#pragma unroll
            for (int qq = 0; qq < 10; qq++) rr[qq] = 0.f;
#pragma unroll
            for (int w4 = 0; w4 < 4; w4++) {
                const unsigned long long sm = s_mask[buf][w4];
                const uint32_t pos = (uint32_t)__popcll(sm & ((1ull << threadIdx.x) - 1ull));   // the entry's row in quadrant w4, if it survived there
                const bool have = ((sm >> threadIdx.x) & 1ull) && ((s_done[buf][w4] >> (pos >> 1)) & 1ull);
#pragma unroll
                for (int qq = 0; qq < 10; qq++) rr[qq] += have ? s_acc[buf][w4][pos][qq] : 0.f;
            }
            float4 *rec = reinterpret_cast<float4 *>(partial + (size_t)my_slot * GOM_PARTIAL_STRIDE);
            rec[0] = make_float4(rr[0], rr[1], rr[2], rr[3]);
            rec[1] = make_float4(rr[4], rr[5], rr[6], rr[7]);
            rec[2] = make_float4(rr[8], rr[9], 0.f, 0.f);
        }
    }
    tq.finish();
}

// ------------------------------------------- backward, two sub-ranges between barriers -
// Same replay and the same four-quadrant fold as k_seg_bwd, but a queue item is a PAIR of consecutive sub-ranges of a segment and
// wave w takes quadrant w of the first and quadrant 3 - w -- the diagonally opposite one -- of the second before the workgroup
// meets at the barrier.  Why: the four waves of k_seg_bwd wait for the busiest quadrant of every task, and on a body frame that
// quadrant has twice the mean quadrant's survivors (oracle-side count over the live tasks: sum of max / sum of mean = 2.04;
// SQ_WAIT_ANY: waves parked 64 % of their cycles).  Which quadrant is busy is a property of the TILE (where the silhouette crosses
// it), so it is the same one in both sub-ranges: giving the wave that had it the opposite quadrant next evens the four waves out.
//   (Measured and dropped on the way here: one task per SEGMENT with the next piece's entries prefetched -- 276 us instead of 205:
//    a quarter of the tasks, and these kernels live on many short independent chains; one WAVE per task walking its four quadrants
//    without any barrier -- 218-259 us: it needs ~100 VGPRs, and at five waves per SIMD with spills the gain is gone.)
#ifndef GOM_BWDP_WAVES
#define GOM_BWDP_WAVES 4   // (127 registers, none spilled: 152 us; at 5 waves per SIMD = 96 registers a handful spill inside the task: 158; both pieces' loads issued ahead of the first replay: 177)
#endif
template <int C>
__global__ void __launch_bounds__(256, GOM_BWDP_WAVES) k_seg_bwd_pair(uint32_t seg_shift, int H, int W, int gx, int gy, float bg0, float bg1, float bg2, float bg3,
                                                  const GomCamera *__restrict__ cams,
                                                  const uint4 *__restrict__ seg_desc, const uint4 *__restrict__ seg_qmax,
                                                  const float2 *__restrict__ ent_geo, const float *__restrict__ ent_col,
                                                  const float *__restrict__ final_T, const uint32_t *__restrict__ n_contrib,
                                                  const float *__restrict__ dL_dpix, const float *__restrict__ sub_Tend,
                                                  const float *__restrict__ sub_C, const float *__restrict__ seg_Sbehind,
                                                  const uint32_t *__restrict__ ent_slot, float *__restrict__ partial, const GomDevStatus *__restrict__ status,
                                                  uint32_t *__restrict__ task_ctr, const uint32_t *__restrict__ task_order, const unsigned long long *__restrict__ cull_masks) {
    // [half of the pair][quadrant][entry of the sub-range][value]; s_done = which entries the quadrant's wave really wrote
    __shared__ float s_acc[2][4][GOM_SUB_MAX][10];   // rows by COMPACTED survivor position (seg_bwd_replay.hpp)
    __shared__ unsigned long long s_done[2][4], s_mask[2][4];   // pairs of rows written; the survivors (list order) they belong to
    __shared__ uint32_t s_task[2];
    __shared__ float4 s_slab[4][GOM_BPAIR_F4];       // wave-private pair records of the survivors, geometry and colours
    const uint32_t sub_sz = (1u << seg_shift) >> 2;
    if (status->overflow) return;
    const uint32_t nsegs = status->num_segs;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    bool from_n1;
    const int slot20 = wave_sum20_slot(lane, from_n1);
    const size_t HW = (size_t)H * W;
    PairQueue tq;
    for (tq.init(task_ctr ? task_ctr + 2 * GOM_TQ_WORDS : nullptr, nsegs, s_task, task_order);; tq.advance()) {
        const uint32_t task = tq.current(s_task);
        if (task == 0xffffffffu) break;
        const uint32_t seg = task >> 3, code = task & 7u;
        const int nhalf = code < 4u ? 2 : 1;                                   // a pair of sub-ranges, or one alone (split by the riders)
        const int sub_a = code < 4u ? (int)code * 2 : (int)code - 4;
        const uint4 d = seg_desc[seg];
        const uint4 qm4 = seg_qmax[seg];
        const uint32_t tile = d.x, start = d.y, cnt = d.z;
        const uint32_t e0 = d.w << seg_shift;
        const uint32_t tmax = max(max(qm4.x, qm4.y), max(qm4.z, qm4.w));
        const uint32_t s0a = e0 + (uint32_t)sub_a * sub_sz;
        if ((uint32_t)sub_a * sub_sz >= cnt || s0a >= tmax) {
            tq.request();
            tq.publish(s_task);
            __syncthreads();
            continue;
        }
        const int tx = tile % gx, fr = (tile / gx) / gy, ty = (tile / gx) % gy;
        float bg[4] = {bg0, bg1, bg2, bg3};
        if (cams) {
#pragma unroll
            for (int ch = 0; ch < 4; ch++) bg[ch] = cams[fr].bg[ch];
        }
        bool live_h[2];
        uint32_t scnt_h[2];
#pragma unroll
        for (int half = 0; half < 2; half++) {
            const int sub = sub_a + half;
            const bool empty = half >= nhalf || (uint32_t)sub * sub_sz >= cnt;
            scnt_h[half] = empty ? 0u : min(sub_sz, cnt - (uint32_t)sub * sub_sz);
            live_h[half] = !empty && e0 + (uint32_t)sub * sub_sz < tmax;
        }
        const bool my_rec = (threadIdx.x >> 6) < 2 && live_h[(threadIdx.x >> 6) & 1] && (uint32_t)(threadIdx.x & 63) < scnt_h[(threadIdx.x >> 6) & 1];

        bool requested = false;
#pragma unroll
        for (int half = 0; half < 2; half++) {
            const int sub = sub_a + half;
            const uint32_t s0 = e0 + (uint32_t)sub * sub_sz;
            const int q = half == 0 ? wv : 3 - wv;   // the diagonally opposite quadrant in the second sub-range
            const uint32_t wmax = q == 0 ? qm4.x : (q == 1 ? qm4.y : (q == 2 ? qm4.z : qm4.w));
            unsigned long long done = 0ull, smask = 0ull;
            if (live_h[half] && wmax > s0) {
                const int pxi = q * 64 + lane;
                const int px = tx * 16 + (q & 1) * 8 + (lane & 7);
                const int py = ty * 16 + (q >> 1) * 8 + (lane >> 3);
                const bool inside = px < W && py < H;
                const size_t pix = (size_t)py * W + px;
                const size_t fpix = (size_t)fr * HW + pix;
                const float pfx = (float)px, pfy = (float)py;
                const float qx0 = (float)(tx * 16 + (q & 1) * 8), qy0 = (float)(ty * 16 + (q >> 1) * 8);
                const uint32_t my_last = inside ? n_contrib[fpix] : 0u;
                const float T_final = final_T[inside ? fpix : 0];
                float dpix[C], bg_dot = 0.f;
#pragma unroll
                for (int ch = 0; ch < C; ch++) {
                    dpix[ch] = inside ? dL_dpix[((size_t)fr * C + ch) * HW + pix] : 0.f;
                    bg_dot += bg[ch] * dpix[ch];
                }
                float T = sub_Tend[((size_t)seg * GOM_NSUB + sub) * GOM_TPX + pxi];
                float S[C], cu[C];
                ld4<C>(seg_Sbehind, seg, pxi, S);
                if (sub < GOM_NSUB - 1) ld4<C>(sub_C, (size_t)seg * GOM_NSUB + sub, pxi, cu);
                const uint32_t lim = min(cnt, wmax - e0);
                const EntryRegs<C> r = load_sub<C>(ent_geo, ent_col, start, lim, sub, lane, qx0, qy0, qx0 + 7.f, qy0 + 7.f, sub_sz, cull_masks + ((size_t)seg * GOM_NSUB + sub) * 4 + q);
                if (!requested) { tq.request(); requested = true; }   // (behind the loads of this piece)
                if (sub < GOM_NSUB - 1) {
#pragma unroll
                    for (int ch = 0; ch < C; ch++) S[ch] += cu[ch];
                }
                float R_acc;
                {
                    float sd = 0.f;
#pragma unroll
                    for (int ch = 0; ch < C; ch++) sd += S[ch] * dpix[ch];
                    R_acc = sd * (T > 0.f ? 1.f / T : 0.f);   // (same operations as k_seg_bwd: the two kernels agree bitwise)
                }
                const unsigned long long mask = __ballot(r.keep);
                const uint32_t npairs = stage_bwd_pairs<C>(s_slab[wv], r, mask, lane);
                const uint32_t pos_limit = survivors_before(mask, my_last > s0 ? my_last - s0 : 0u);
                done = bwd_replay<C>(s_slab[wv], npairs, &s_acc[half][q][0][0], dpix, pfx, pfy, T, R_acc, T_final, bg_dot, pos_limit, slot20, from_n1);
                smask = mask;
            }
            if (lane == 0) { s_done[half][q] = done; s_mask[half][q] = smask; }
            if (half == 0 && requested) tq.look_ahead();   // (the dequeue went out with the first half's loads: it is back)
        }
        if (!requested) tq.request();
        tq.publish(s_task);
        __syncthreads();
        {   // one 48-byte record per entry of the two sub-ranges (threads 0..127), quadrants summed in a fixed order
            const int half = (threadIdx.x >> 6) & 1, e = threadIdx.x & 63;
            if (my_rec) {
                const uint32_t slot = ent_slot[start + (uint32_t)(sub_a + half) * sub_sz + (uint32_t)e];
                float rr[10];
#pragma unroll
                for (int qq = 0; qq < 10; qq++) rr[qq] = 0.f;
#pragma unroll
                for (int w4 = 0; w4 < 4; w4++) {
                    const unsigned long long sm = s_mask[half][w4];
                    const uint32_t pos = (uint32_t)__popcll(sm & ((1ull << e) - 1ull));   // the entry's row in quadrant w4, if it survived there
                    const bool have = ((sm >> e) & 1ull) && ((s_done[half][w4] >> (pos >> 1)) & 1ull);
#pragma unroll
                    for (int qq = 0; qq < 10; qq++) rr[qq] += have ? s_acc[half][w4][pos][qq] : 0.f;
                }
                float4 *rec = reinterpret_cast<float4 *>(partial + (size_t)slot * GOM_PARTIAL_STRIDE);
                rec[0] = make_float4(rr[0], rr[1], rr[2], rr[3]);
                rec[1] = make_float4(rr[4], rr[5], rr[6], rr[7]);
                rec[2] = make_float4(rr[8], rr[9], 0.f, 0.f);
            }
        }
        __syncthreads();   // the next pair overwrites s_acc / s_done
    }
    tq.finish();
}

#ifdef GOM_LAB   // (include/gom_hip_lab.h: built, measured, not adopted)
#include "lab/seg_bwd_blk.hpp"
#include "lab/rec_bwd.hpp"
#endif

}  // namespace


int gom_launch_sort(GomState *s, hipStream_t st) {
    const int n_tiles = s->gx * s->gy * s->B;
    if (n_tiles == 0) return 0;
    GomKernelTimer timer(s, GOM_K_SORT, st);
    const uint32_t lc = (uint32_t)(31 - __builtin_clz((unsigned)s->sortCap));  // log2 of the chunk (13 unless a test lowered it)
    // A single frame is latency-bound by its longest list: one launch.  A batch is throughput-bound: the short lists
    // (the vast majority) go to 4-wave workgroups that pack 8 per CU.
    // sortSplit (the mesh rasterizer: 4 096 tiles of 8 x 8 pixels, ~40 faces each, keys = face indices alone): ONE launch of the 4-wave instantiation takes
    // every tile -- eight workgroups per CU instead of two 16-wave ones -- and the few lists beyond its 2 048-entry chunk are sorted through an index
    // bitmap in LDS (sort_tile), not through the global-memory merge levels (57 us for the mesh's lists through the 16-wave kernel, 80 through this one's merges)
    const uint32_t bm_words = s->sortSplit && s->P <= 8 * 256 * 2 * 32 ? ((uint32_t)s->P + 31u) >> 5 : 0u;   // (the bitmap shares s_x: 16 NT words)
    const uint32_t small_max = bm_words ? 0xffffffffu : (s->B > 1 ? GOM_SORT_SMALL : 0u);
    if (small_max) {
        hipLaunchKernelGGL(k_sort<256>, dim3(n_tiles), dim3(256), 0, st, s->gx, s->tile_base, s->seg_base, s->keys, s->point_list, s->seg_desc,
                           s->rect, s->pair_off, s->pair_pos, s->ent_slot, s->xy, s->conic_opacity, s->ent_geo, reinterpret_cast<uint64_t *>(s->partial),
                           s->status, lc < 11u ? lc : 11u, small_max, (uint32_t)s->segShift, bm_words);
        GOM_LAUNCH_CHECK();
        if (bm_words) return 0;
    }
    hipLaunchKernelGGL(k_sort<1024>, dim3(n_tiles), dim3(1024), 0, st, s->gx, s->tile_base, s->seg_base, s->keys, s->point_list, s->seg_desc,
                       s->rect, s->pair_off, s->pair_pos, s->ent_slot, s->xy, s->conic_opacity, s->ent_geo, reinterpret_cast<uint64_t *>(s->partial), s->status,
                       lc, small_max, (uint32_t)s->segShift, 0u);
    GOM_LAUNCH_CHECK();
    return 0;
}

// Workgroups the chip holds at once for a segment kernel (the task queue needs no more than that).
template <typename K>
static int resident_grid(K kernel, int device) {
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, 0) != hipSuccess || per_cu < 1) per_cu = 4;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus < 1) cus = 256;
    return per_cu * cus;
}
static int task_grid(int resident, int pct) {   // GOM_OPT_TASK_GRID_PCT of the resident grid, a multiple of the shard count
    const int g = (int)((long long)resident * pct / 100);
    return g < GOM_TQ_SHARDS ? GOM_TQ_SHARDS : g / GOM_TQ_SHARDS * GOM_TQ_SHARDS;
}
// A batched launch (tens of thousands of tasks) uses the task queue on a grid that just fills the chip; a single frame
// has fewer tasks than two rounds of that grid, where one task per workgroup and no queue is the shorter path.
#define GOM_RESIDENT(KERNEL) (s->B > 1 ? task_grid([&]() { static const int g = resident_grid(KERNEL, s->device); return g; }(), s->taskGridPct) : GOM_SEG_GRID * 4)
#define GOM_TASK_CTR (s->B > 1 ? s->task_ctr : nullptr)
// 1 = the kernel strides over its tasks (no dequeue) also in a batched launch.  The transmittance pre-pass's tasks are alike (every piece is
// evaluated): striding measures 75 us against 78 through the queue.  The compositing pass's are not (two thirds are dead): 130 against 110.
#ifndef GOM_STATIC_T
#define GOM_STATIC_T 1
#endif
#ifndef GOM_STATIC_FWD
#define GOM_STATIC_FWD 0
#endif

static GomRecArgs rec_args(GomState *s) {
    return GomRecArgs{s->piece_ub, s->piece_rec, s->piece_cnt, s->rec_ti, s->rec_acc, &s->status->rec_cursor[0][0], &s->status->rec_overflow, (uint32_t)(s->capRec / GOM_REC_SHARDS)};
}
// which render backward a forward prepares for: records (GOM_OPT_BWD_MODE 3) exist in -DGOM_LAB builds only -- measured slower than the replay on
// the metric workload (298 us + 55 us of forward overhead against 153: profiles/r04_records_backward.txt), closed in round 5;
// the product library neither instantiates the REC = true kernels nor allocates their buffers
#ifdef GOM_LAB
static bool records_mode(const GomState *s) { return s->bwdMode == 3; }
#else
static bool records_mode(const GomState *) { return false; }
#endif

int gom_launch_render_forward(GomState *s, const GomCamera &cam, int C, const float *colors, float *out_color, bool reuse_T,
                              hipStream_t st) {
    const int n_tiles = s->gx * s->gy * s->B;
    if (n_tiles == 0) return 0;
    // Records mode: the transmittance pre-pass counts every piece's (lane, entry) pairs with alpha > 0 (a binning made in another mode has no
    // counts: the colour pass over it keeps that mode's checkpoints)
    const bool rec = records_mode(s) && (!reuse_T || s->recCounts);
    const GomRecArgs ra = rec ? rec_args(s) : GomRecArgs{};
    {
        GomKernelTimer timer(s, GOM_K_SEG_T, st);
        if (!reuse_T) {  // transmittances depend on geometry only: shared by every colour pass over the same binning
#define GOM_ST(CC, RR)                                                                                                    \
    hipLaunchKernelGGL((k_seg_T<CC, RR>), dim3(GOM_RESIDENT((k_seg_T<CC, RR>))), dim3(256), 0, st, (uint32_t)s->segShift, s->gx, s->gy, s->seg_desc, s->point_list, s->ent_geo, colors, \
                       s->ent_col, s->seg_T, s->sub_T, s->status, GOM_STATIC_T ? nullptr : GOM_TASK_CTR, s->cull_masks, ra)
#ifdef GOM_LAB
            if (rec) { if (C == 3) GOM_ST(3, true); else GOM_ST(4, true); } else
#endif
            { if (C == 3) GOM_ST(3, false); else GOM_ST(4, false); }
#undef GOM_ST
            s->recCounts = rec;
        } else {  // only the colours changed: bring them into list order
            if (C == 3) hipLaunchKernelGGL((k_gather_colors<3>), dim3(1024), dim3(256), 0, st, s->point_list, colors, s->ent_col, s->status, rec ? ra.cursor : nullptr, ra.overflow);
            else hipLaunchKernelGGL((k_gather_colors<4>), dim3(1024), dim3(256), 0, st, s->point_list, colors, s->ent_col, s->status, rec ? ra.cursor : nullptr, ra.overflow);
        }
    }
    GOM_LAUNCH_CHECK();
    {
        GomKernelTimer timer(s, GOM_K_SEG_FWD, st);
#define GOM_SF(CC, RR)                                                                                                    \
    hipLaunchKernelGGL((k_seg_fwd<CC, RR>), dim3(GOM_RESIDENT((k_seg_fwd<CC, RR>))), dim3(256), 0, st, (uint32_t)s->segShift, s->gx, s->gy, s->seg_desc, s->ent_geo, s->ent_col, s->seg_T,  \
                       s->sub_T, s->seg_C, s->seg_Tend, s->seg_last, s->sub_C, s->sub_Tend, s->status, GOM_STATIC_FWD ? nullptr : GOM_TASK_CTR, s->B > 1 && s->rankSort && s->bwdOrder ? s->seg_cost : nullptr, s->cull_masks, ra)
#ifdef GOM_LAB
        if (rec) { if (C == 3) GOM_SF(3, true); else GOM_SF(4, true); } else
#endif
        { if (C == 3) GOM_SF(3, false); else GOM_SF(4, false); }
#undef GOM_SF
        s->recForward = rec;
    }
    GOM_LAUNCH_CHECK();
    {
        GomKernelTimer timer(s, GOM_K_COMBINE, st);
        const bool ride = s->rideBwdOrder && s->rankSort && s->B > 1;
        const bool listed = s->emptyFilled && s->rankSort;   // the non-empty tiles are listed (work items of the tile pass) and the others painted
#define GOM_CF(CC)                                                                                                        \
    hipLaunchKernelGGL((k_combine_fwd<CC>), dim3((listed ? (n_tiles < 2048 ? n_tiles : 2048) : n_tiles) + (ride ? 8 : 0) + n_lr), dim3(256), 0, st, s->H, s->W, s->gx, s->gy, cam.bg[0], cam.bg[1], cam.bg[2], \
                       cam.bg[3], s->cams, s->seg_base, s->seg_C, s->seg_last, s->seg_Tend, s->seg_Sbehind, out_color, s->final_T,        \
                       s->n_contrib, s->tile_nmax, s->seg_qmax, s->status, s->tile_base, s->point_list, s->rankSort ? s->rank_of : nullptr, s->tile_qlim, s->emptyFilled ? 1 : 0, \
                       listed ? s->work_items : nullptr, n_tiles, ride ? GomBwdOrderRider{s->status, s->seg_cost, s->bwd_order} : GomBwdOrderRider{}, lr, n_lr)
        GomLossRider lr{};   // the frame step's loss rides here
        if (s->lossRider.gt_rgb && C == 4) lr = s->lossRider;
        const int n_lr = lr.gt_rgb ? ((n_tiles + 3) / 4 < GOM_LOSS_RIDER_BLOCKS ? (n_tiles + 3) / 4 : GOM_LOSS_RIDER_BLOCKS) : 0;   // a group of four tiles each, up to a chip's worth
        if (C == 3) GOM_CF(3); else GOM_CF(4);
#undef GOM_CF
    }
    GOM_LAUNCH_CHECK();
    return 0;
}

int gom_launch_render_backward(GomState *s, const GomCamera &cam, int C, const float *colors, const float *dL_dcolor,
                               hipStream_t st) {
    (void)colors;  // already in list order (ent_col) from the forward / checkpoint re-creation
    const int n_tiles = s->gx * s->gy * s->B;
    if (n_tiles == 0) return 0;
    GomKernelTimer timer(s, GOM_K_SEG_BWD, st);
#ifdef GOM_LAB
    if (s->recForward) {   // the forward left records: a lane per blending (pixel, entry) pair (lab/rec_bwd.hpp)
        const GomRecArgs ra = rec_args(s);
#define GOM_RB(CC)                                                                                                        \
    hipLaunchKernelGGL((k_rec_bwd<CC>), dim3(GOM_RESIDENT(k_rec_bwd<CC>)), dim3(256), 0, st, (uint32_t)s->segShift, s->H, s->W, s->gx, s->gy, cam.bg[0], cam.bg[1], cam.bg[2], \
                       cam.bg[3], s->cams, s->seg_desc, s->seg_qmax, s->ent_geo, s->ent_col, s->final_T, s->n_contrib, dL_dcolor,           \
                       s->sub_C, s->seg_Sbehind, s->ent_slot, s->partial, s->status, ra)
        if (C == 3) GOM_RB(3); else GOM_RB(4);
#undef GOM_RB
        GOM_LAUNCH_CHECK();
        return 0;
    }
    if (s->bwdMode == 2) {   // (sub-range, 4 x 4 block) items, one per DPP row (lab/seg_bwd_blk.hpp)
#define GOM_SBB(CC)                                                                                                       \
    hipLaunchKernelGGL((k_seg_bwd_blk<CC>), dim3(GOM_RESIDENT(k_seg_bwd_blk<CC>)), dim3(256), 0, st, (uint32_t)s->segShift, s->H, s->W, s->gx, s->gy, cam.bg[0], cam.bg[1], cam.bg[2], \
                       cam.bg[3], s->cams, s->seg_desc, s->seg_qmax, s->ent_geo, s->ent_col, s->final_T, s->n_contrib, dL_dcolor,           \
                       s->sub_Tend, s->sub_C, s->seg_Sbehind, s->ent_slot, s->partial, s->status, GOM_TASK_CTR, s->B > 1 && s->bwdOrderReady ? s->bwd_order : nullptr, s->cull_masks)
        if (C == 3) GOM_SBB(3); else GOM_SBB(4);
#undef GOM_SBB
        GOM_LAUNCH_CHECK();
        return 0;
    }
#endif
    if (s->bwdMode == 0 || (s->bwdMode < 0 && s->B > 1)) {   // two sub-ranges between barriers, opposite quadrants per wave (GOM_OPT_BWD_MODE 1: one sub-range per barrier, round 1)
#define GOM_SBW(CC)                                                                                                       \
    hipLaunchKernelGGL((k_seg_bwd_pair<CC>), dim3(GOM_RESIDENT(k_seg_bwd_pair<CC>)), dim3(256), 0, st, (uint32_t)s->segShift, s->H, s->W, s->gx, s->gy, cam.bg[0], cam.bg[1], cam.bg[2], \
                       cam.bg[3], s->cams, s->seg_desc, s->seg_qmax, s->ent_geo, s->ent_col, s->final_T, s->n_contrib, dL_dcolor,           \
                       s->sub_Tend, s->sub_C, s->seg_Sbehind, s->ent_slot, s->partial, s->status, GOM_TASK_CTR, s->B > 1 && s->bwdOrderReady ? s->bwd_order : nullptr, s->cull_masks)
        if (C == 3) GOM_SBW(3); else GOM_SBW(4);
#undef GOM_SBW
        GOM_LAUNCH_CHECK();
        return 0;
    }
#define GOM_SB(CC)                                                                                                        \
    hipLaunchKernelGGL((k_seg_bwd<CC>), dim3(GOM_RESIDENT(k_seg_bwd<CC>)), dim3(256), 0, st, (uint32_t)s->segShift, s->H, s->W, s->gx, s->gy, cam.bg[0], cam.bg[1], cam.bg[2], \
                       cam.bg[3], s->cams, s->seg_desc, s->seg_qmax, s->ent_geo, s->ent_col, s->final_T, s->n_contrib, dL_dcolor,           \
                       s->sub_Tend, s->sub_C, s->seg_Sbehind, s->ent_slot, s->partial, s->status, GOM_TASK_CTR, s->cull_masks)
    if (C == 3) GOM_SB(3); else GOM_SB(4);
#undef GOM_SB
    GOM_LAUNCH_CHECK();
    return 0;
}
