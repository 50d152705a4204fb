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
// tile is cut into segments of 256 entries and the work is spread over
// (tile, segment) pairs -- ~750 workgroups for 256 CUs:
//
//   k_sort         per tile, 1024 threads: normalised bitonic network on the
//                  unique (depth_bits<<32 | gaussian) keys in LDS -> identical to
//                  the reference's stable (tile, depth) radix order.
//   k_seg_fwd      per (tile, segment): the 256 entries are staged in LDS once;
//                  each of the 4 waves composites them over its own 8x8 pixel
//                  quadrant STARTING FROM T = 1 and stores (prod(1-alpha),
//                  colour, last contributor) per pixel.
//   k_combine_fwd  per tile: folds the segment results front to back.  While
//                  T * T_seg stays above the 1e-4 stop threshold the fold is one
//                  fma per channel; the segment in which a pixel crosses the
//                  threshold is re-walked entry by entry with the reference's
//                  exact stop rule (so n_contrib / final_T keep their meaning).
//                  Also emits per-segment checkpoints (T after the segment,
//                  colour still behind it) for the backward.
//   k_seg_bwd      per (tile, segment): back-to-front replay from the
//                  checkpoint; the 6+C per-pixel terms of every entry are reduced
//                  over the wave with DPP row_shr/row_bcast adds, the four waves
//                  are summed in LDS in a fixed order and one 48-byte record per
//                  (tile, entry) is written.  No float atomics anywhere.
//
// Inside a wave, lane = pixel.  Per 64-entry batch lane = ENTRY first: every
// lane tests one entry against the wave's 8x8 rectangle with a conservative
// bound on the largest alpha it can reach there; a 64-bit ballot then drives a
// scalar loop over the survivors only, whose attributes are broadcast with
// v_readlane (SGPR operands).  Entries skipped this way are exactly those the
// reference skips for all 64 pixels (`alpha < 1/255 -> continue`).  The blend
// loop is branch-free and keeps its state in VGPRs (float masks) so that the
// serial T chain never round-trips through SALU/VCC logic.
#include "gom_internal.h"

#ifdef GOM_INSTRUMENT
__device__ unsigned long long g_dbg[8192 * 12];
extern "C" int gom_debug_fetch(unsigned long long *host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_dbg), sizeof(unsigned long long) * n);
}
#endif

namespace {

constexpr float kStopT = 0.0001f;          // App. A.3: stop when T(1-alpha) < 1e-4
constexpr float kMinAlpha = 1.0f / 255.0f;
constexpr float kMaxAlpha = 0.99f;

__device__ __forceinline__ float rl(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
    const int m = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, true);
    return v + __int_as_float(m);
}

// Sum over the 64 lanes; the total lands in lane 63 (other lanes hold prefixes).
__device__ __forceinline__ float wave_sum_lane63(float v) {
    v = dpp_add<0x111, 0xF>(v);  // row_shr:1
    v = dpp_add<0x112, 0xF>(v);  // row_shr:2
    v = dpp_add<0x114, 0xF>(v);  // row_shr:4
    v = dpp_add<0x118, 0xF>(v);  // row_shr:8
    v = dpp_add<0x142, 0xA>(v);  // row_bcast:15 -> rows 1,3
    v = dpp_add<0x143, 0xC>(v);  // row_bcast:31 -> rows 2,3
    return v;
}

// Sums NV registers over the 64 lanes (totals land in lane 63), one fused v_add_f32_dpp per value and step.
// The NV chains are interleaved so that every DPP read is >= NV instructions behind the write it depends on
// (the 2-wait-state VALU->DPP hazard is covered; the leading s_nop covers the producers of the inputs).
template <int NV>
__device__ __forceinline__ void wave_sum_lane63_n(float (&v)[NV]) {
    static_assert(NV == 9 || NV == 10, "6 + C values");
#define GOM_DPP_STEP(CTRL)                                                                                          \
    "v_add_f32_dpp %0, %0, %0 " CTRL "\n v_add_f32_dpp %1, %1, %1 " CTRL "\n v_add_f32_dpp %2, %2, %2 " CTRL "\n"     \
    "v_add_f32_dpp %3, %3, %3 " CTRL "\n v_add_f32_dpp %4, %4, %4 " CTRL "\n v_add_f32_dpp %5, %5, %5 " CTRL "\n"     \
    "v_add_f32_dpp %6, %6, %6 " CTRL "\n v_add_f32_dpp %7, %7, %7 " CTRL "\n v_add_f32_dpp %8, %8, %8 " CTRL "\n"
#define GOM_DPP_STEP10(CTRL) GOM_DPP_STEP(CTRL) "v_add_f32_dpp %9, %9, %9 " CTRL "\n"
#define GOM_DPP_ALL(STEP)                                                                                           \
    "s_nop 1\n" STEP("row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0") STEP("row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0") \
    STEP("row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0") STEP("row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:0")             \
    STEP("row_bcast:15 row_mask:0xa bank_mask:0xf") STEP("row_bcast:31 row_mask:0xc bank_mask:0xf")
    if constexpr (NV == 10) {
        asm volatile(GOM_DPP_ALL(GOM_DPP_STEP10)
                     : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]));
    } else {
        asm volatile(GOM_DPP_ALL(GOM_DPP_STEP)
                     : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]));
    }
#undef GOM_DPP_ALL
#undef GOM_DPP_STEP10
#undef GOM_DPP_STEP
}

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t o = __shfl_xor(v, d, 64);
        v = o > v ? o : v;
    }
    return v;
}

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
    if (X != 0.f) {
        float dy = -b * X / cz;
        dy = fminf(fmaxf(dy, y0 - cy), y1 - cy);
        q = fminf(q, a * X * X + 2.f * b * X * dy + cz * dy * dy);
    }
    if (Y != 0.f) {
        float dx = -b * Y / a;
        dx = fminf(fmaxf(dx, x0 - cx), x1 - cx);
        q = fminf(q, a * dx * dx + 2.f * b * dx * Y + cz * Y * Y);
    }
    const float DX = fmaxf(fabsf(x0 - cx), fabsf(x1 - cx));
    const float DY = fmaxf(fabsf(y0 - cy), fabsf(y1 - cy));
    const float mag = a * DX * DX + 2.f * fabsf(b) * DX * DY + cz * DY * DY;
    const float lthr = -__logf(255.0f * o);  // alpha >= 1/255  <=>  power >= lthr
    return (-0.5f * q + (1e-5f * mag + 1e-2f)) < lthr;
}

// alpha of one entry at one pixel with the reference's skip rules folded in:
// returns 0 when the reference would `continue` (power > 0 or alpha < 1/255).
__device__ __forceinline__ float entry_alpha(float ex, float ey, float ea, float eb, float ec, float eo, float pfx, float pfy) {
    const float dx = ex - pfx, dy = ey - pfy;
    const float power = -0.5f * (ea * dx * dx + ec * dy * dy) - eb * dx * dy;
    float a = fminf(kMaxAlpha, eo * __expf(power));
    a = (power <= 0.f) ? a : 0.f;
    a = (a >= kMinAlpha) ? a : 0.f;
    return a;
}

// ---------------------------------------------------------------- sort ------
// Normalised bitonic network on n unique 64-bit keys: every comparator orders
// ascending, so indices >= n behave as +inf padding and are simply skipped.
// 4 comparators per thread per trip with all loads issued first.
template <int NT, typename PTR>
__device__ __forceinline__ void bitonic_sort_u64(PTR keys, uint32_t n) {
    for (uint32_t m = 1; (1u << (m - 1)) < n; m++) {
        const uint32_t k = 1u << m, half = k >> 1;
        {
            const uint32_t total = ((n + k - 1) >> m) << (m - 1);
            for (uint32_t i0 = threadIdx.x; i0 < total; i0 += NT * 4) {
                uint32_t lo[4], hi[4];
                uint64_t a[4], b[4];
                bool ok[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const uint32_t i = i0 + NT * u;
                    const uint32_t blk = i >> (m - 1), off = i & (half - 1);
                    lo[u] = (blk << m) + off;
                    hi[u] = (blk << m) + (k - 1 - off);
                    ok[u] = i < total && hi[u] < n;
                    if (ok[u]) { a[u] = keys[lo[u]]; b[u] = keys[hi[u]]; }
                }
#pragma unroll
                for (int u = 0; u < 4; u++)
                    if (ok[u] && a[u] > b[u]) { keys[lo[u]] = b[u]; keys[hi[u]] = a[u]; }
            }
        }
        __syncthreads();
        for (int q = (int)m - 2; q >= 0; q--) {
            const uint32_t j = 1u << q;
            const uint32_t total = ((n + 2 * j - 1) >> (q + 1)) << q;
            for (uint32_t i0 = threadIdx.x; i0 < total; i0 += NT * 4) {
                uint32_t lo[4];
                uint64_t a[4], b[4];
                bool ok[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const uint32_t i = i0 + NT * u;
                    lo[u] = ((i >> q) << (q + 1)) + (i & (j - 1));
                    ok[u] = i < total && lo[u] + j < n;
                    if (ok[u]) { a[u] = keys[lo[u]]; b[u] = keys[lo[u] + j]; }
                }
#pragma unroll
                for (int u = 0; u < 4; u++)
                    if (ok[u] && a[u] > b[u]) { keys[lo[u]] = b[u]; keys[lo[u] + j] = a[u]; }
            }
            __syncthreads();
        }
    }
}

// Register-resident bitonic sort: 8 keys per thread (element i = 8*thread + r).  In the normalised network every
// step pairs element i with i ^ mask (mask = 2^m - 1 for the mirror step of stage m, a power of two otherwise) and
// leaves the minimum at the lower index.  mask < 8 stays inside a thread, mask < 512 inside a wave (ds_bpermute
// shuffles, no barrier), and only masks >= 512 go through LDS -- 10 barrier-fenced exchanges for 8192 keys
// instead of 91.
__device__ __forceinline__ void cmpswap(uint64_t &lo, uint64_t &hi) {
    const uint64_t a = lo, b = hi;
    const bool sw = a > b;
    lo = sw ? b : a;
    hi = sw ? a : b;
}

template <int MASK>
__device__ __forceinline__ void sort_step_regs(uint64_t (&x)[8]) {
#pragma unroll
    for (int r = 0; r < 8; r++)
        if ((r ^ MASK) > r) cmpswap(x[r], x[r ^ MASK]);
}

// lane ^ MASK exchange of one dword without touching LDS: DPP quad/row permutes for MASK < 16,
// v_permlane16_swap / v_permlane32_swap (gfx950) for the row and half-wave bits.
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_mov(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, false);
}
__device__ __forceinline__ uint32_t xor16(uint32_t v) {
    const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    return ((threadIdx.x >> 4) & 1) ? r[0] : r[1];   // odd rows: vdst now holds the even neighbour, even rows: src0 holds the odd one
}
__device__ __forceinline__ uint32_t xor32(uint32_t v) {
    const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return ((threadIdx.x >> 5) & 1) ? r[0] : r[1];
}
template <int MASK>
__device__ __forceinline__ uint32_t lane_xor(uint32_t v) {
    if constexpr (MASK == 1) return dpp_mov<0xB1>(v);                       // quad_perm [1,0,3,2]
    else if constexpr (MASK == 2) return dpp_mov<0x4E>(v);                  // quad_perm [2,3,0,1]
    else if constexpr (MASK == 3) return dpp_mov<0x1B>(v);                  // quad_perm [3,2,1,0]
    else if constexpr (MASK == 7) return dpp_mov<0x141>(v);                 // row_half_mirror
    else if constexpr (MASK == 15) return dpp_mov<0x140>(v);                // row_mirror
    else if constexpr (MASK == 4) return dpp_mov<0x1B>(dpp_mov<0x141>(v));  // (l^7)^3
    else if constexpr (MASK == 8) return dpp_mov<0x141>(dpp_mov<0x140>(v)); // (l^15)^7
    else if constexpr (MASK == 16) return xor16(v);
    else if constexpr (MASK == 32) return xor32(v);
    else if constexpr (MASK == 31) return xor16(dpp_mov<0x140>(v));
    else if constexpr (MASK == 63) return xor32(xor16(dpp_mov<0x140>(v)));
    else return 0;
}
template <int MASK>
__device__ __forceinline__ void lanes_xor_u64x8(const uint64_t (&x)[8], uint64_t (&y)[8]) {
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const uint32_t lo = lane_xor<MASK>((uint32_t)x[r]);
        const uint32_t hi = lane_xor<MASK>((uint32_t)(x[r] >> 32));
        y[r] = ((uint64_t)hi << 32) | lo;
    }
}

// partner keys arrive in y[]; keep the minimum when this thread owns the lower index
template <bool MIRROR>
__device__ __forceinline__ void sort_step_merge(uint64_t (&x)[8], const uint64_t (&y)[8], bool lower) {
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const uint64_t o = y[MIRROR ? 7 - r : r];
        const bool take = (o < x[r]) == lower;  // keys are unique (or both +inf padding, where either choice is the same)
        x[r] = take ? o : x[r];
    }
}

template <bool MIRROR>
__device__ __forceinline__ void sort_step_cross(uint64_t (&x)[8], uint32_t tmask, uint64_t *s_x, bool wave_active, uint32_t n) {
    // tmask = mask >> 3: partner thread = t ^ tmask; this thread owns the lower index iff the top bit of tmask is clear in t
    const uint32_t t = threadIdx.x;
    const bool lower = (t & (1u << (31 - __builtin_clz(tmask)))) == 0;
    uint64_t y[8];
    if (tmask < 64) {
        if (!wave_active) return;  // this wave holds only +inf padding (wave-uniform)
        switch (tmask) {  // wave-uniform
            case 1: lanes_xor_u64x8<1>(x, y); break;
            case 2: lanes_xor_u64x8<2>(x, y); break;
            case 3: lanes_xor_u64x8<3>(x, y); break;
            case 4: lanes_xor_u64x8<4>(x, y); break;
            case 7: lanes_xor_u64x8<7>(x, y); break;
            case 8: lanes_xor_u64x8<8>(x, y); break;
            case 15: lanes_xor_u64x8<15>(x, y); break;
            case 16: lanes_xor_u64x8<16>(x, y); break;
            case 31: lanes_xor_u64x8<31>(x, y); break;
            case 32: lanes_xor_u64x8<32>(x, y); break;
            default: lanes_xor_u64x8<63>(x, y); break;
        }
    } else {
        __syncthreads();  // previous readers of s_x are done (idle waves only keep the barriers company)
        if (wave_active) {
#pragma unroll
            for (int r = 0; r < 8; r++) s_x[r * 1024 + t] = x[r];  // transposed: conflict-free 8-byte lanes
        }
        __syncthreads();
        if (!wave_active) return;
        const bool partner_wrote = ((t ^ tmask) & ~63u) * 8 < n;  // padding-only waves wrote nothing: their keys are +inf
#pragma unroll
        for (int r = 0; r < 8; r++) y[r] = partner_wrote ? s_x[r * 1024 + (t ^ tmask)] : ~0ull;
    }
    sort_step_merge<MIRROR>(x, y, lower);
}

__global__ void __launch_bounds__(1024) k_sort(int gx, const uint32_t *__restrict__ tile_base, const uint32_t *__restrict__ seg_base,
                                               uint64_t *__restrict__ keys, uint32_t *__restrict__ point_list,
                                               uint32_t *__restrict__ seg_tile, const ushort4 *__restrict__ rect,
                                               const uint32_t *__restrict__ pair_off, uint32_t *__restrict__ pair_pos,
                                               const GomDevStatus *__restrict__ status, uint32_t sort_cap) {
    __shared__ uint64_t s_x[GOM_SORT_CAP_MAX];
    if (status->overflow) return;
    const int tile = blockIdx.x;
    const uint32_t base = tile_base[tile];
    const uint32_t n = tile_base[tile + 1] - base;
    if (n == 0) return;
    const uint32_t sb = seg_base[tile], nseg = seg_base[tile + 1] - sb;
    for (uint32_t i = threadIdx.x; i < nseg; i += 1024) seg_tile[sb + i] = (uint32_t)tile;
    const int tx = tile % gx, ty = tile / gx;
    const uint32_t t = threadIdx.x;
    if (n <= sort_cap) {
        uint32_t logN = 3;
        while ((1u << logN) < n) logN++;
#ifdef GOM_INSTRUMENT
        const unsigned long long d0 = __builtin_readcyclecounter();
#endif
        uint64_t x[8];
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const uint32_t i = 8 * t + r;
            x[r] = i < n ? keys[base + i] : ~0ull;  // +inf padding never moves down: every comparator sorts ascending
        }
#ifdef GOM_INSTRUMENT
        const unsigned long long d1 = __builtin_readcyclecounter();
#endif
        const bool wave_active = (t & ~63u) * 8 < n;  // waves past the list only hold padding
        // stages m = 1..3 live entirely in registers
        if (wave_active) {
            sort_step_regs<1>(x);
            sort_step_regs<3>(x); sort_step_regs<1>(x);
            sort_step_regs<7>(x); sort_step_regs<2>(x); sort_step_regs<1>(x);
        }
        for (uint32_t m = 4; m <= logN; m++) {
            sort_step_cross<true>(x, ((1u << m) - 1) >> 3, s_x, wave_active, n);
            for (int q = (int)m - 2; q >= 3; q--) sort_step_cross<false>(x, (1u << q) >> 3, s_x, wave_active, n);
            if (wave_active) { sort_step_regs<4>(x); sort_step_regs<2>(x); sort_step_regs<1>(x); }
        }
#ifdef GOM_INSTRUMENT
        const unsigned long long d2 = __builtin_readcyclecounter();
#endif
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const uint32_t i = 8 * t + r;
            if (i < n) {
                keys[base + i] = x[r];
                const uint32_t g = (uint32_t)x[r];
                point_list[base + i] = g;
                const ushort4 rc = rect[g];
                const uint32_t k = (uint32_t)(ty - (int)rc.y) * (uint32_t)(rc.z - rc.x) + (uint32_t)(tx - (int)rc.x);
                pair_pos[pair_off[g] + k] = base + i;
            }
        }
#ifdef GOM_INSTRUMENT
        if (t == 0 && tile < 8192) {
            g_dbg[tile * 4 + 0] = d1 - d0; g_dbg[tile * 4 + 1] = d2 - d1; g_dbg[tile * 4 + 2] = __builtin_readcyclecounter() - d2; g_dbg[tile * 4 + 3] = n;
        }
#endif
    } else {
        bitonic_sort_u64<1024>(keys + base, n);  // rare: list longer than the LDS capacity, sorted in global memory
        for (uint32_t i = threadIdx.x; i < n; i += 1024) {
            const uint32_t g = (uint32_t)keys[base + i];
            point_list[base + i] = g;
            const ushort4 rc = rect[g];
            const uint32_t k = (uint32_t)(ty - (int)rc.y) * (uint32_t)(rc.z - rc.x) + (uint32_t)(tx - (int)rc.x);
            pair_pos[pair_off[g] + k] = base + i;
        }
    }
}

// ------------------------------------------------------------- staging -----
// The 256 entries of a segment, SoA in LDS: x y a b c o col[C].
template <int C>
struct SegLds {
    float v[6 + C][GOM_SEG];
};

template <int C>
__device__ __forceinline__ void stage_segment(SegLds<C> &s, uint32_t cnt, const uint32_t *__restrict__ list, const float2 *__restrict__ xy,
                                              const float4 *__restrict__ conic_opacity, const float *__restrict__ colors) {
    const uint32_t i = threadIdx.x;
    if (i < cnt) {
        const uint32_t g = list[i];
        const float2 c = xy[g];
        const float4 co = conic_opacity[g];
        s.v[0][i] = c.x; s.v[1][i] = c.y; s.v[2][i] = co.x; s.v[3][i] = co.y; s.v[4][i] = co.z; s.v[5][i] = co.w;
        if (C == 4) {
            const float4 cl = *reinterpret_cast<const float4 *>(colors + (size_t)g * 4);
            s.v[6][i] = cl.x; s.v[7][i] = cl.y; s.v[8][i] = cl.z; s.v[6 + C - 1][i] = cl.w;
        } else {
#pragma unroll
            for (int ch = 0; ch < C; ch++) s.v[6 + ch][i] = colors[(size_t)g * C + ch];
        }
    }
}

// One lane's view of "its" entry of the current 64-batch.
template <int C>
struct EntryRegs {
    float x, y, a, b, c, o, col[C];
    bool keep;
};

template <int C>
__device__ __forceinline__ EntryRegs<C> fetch_entry(const SegLds<C> &s, uint32_t e, uint32_t cnt, float qx0, float qy0, float qx1, float qy1) {
    EntryRegs<C> r;
    const uint32_t ee = e < cnt ? e : 0;
    r.x = s.v[0][ee]; r.y = s.v[1][ee]; r.a = s.v[2][ee]; r.b = s.v[3][ee]; r.c = s.v[4][ee]; r.o = s.v[5][ee];
#pragma unroll
    for (int ch = 0; ch < C; ch++) r.col[ch] = s.v[6 + ch][ee];
    r.keep = (e < cnt) && !cull_entry(r.x, r.y, r.a, r.b, r.c, r.o, qx0, qy0, qx1, qy1);
    return r;
}

// ------------------------------------------------- forward, pass A (T only) -
// prod(1 - alpha) of every segment for every pixel of its tile, from alpha alone
// (no colours, no stop rule).  Lets every later pass know the transmittance at
// which each segment starts without walking the list serially.
__global__ void __launch_bounds__(256) k_seg_T(int gx, const uint32_t *__restrict__ tile_base, const uint32_t *__restrict__ seg_base,
                                               const uint32_t *__restrict__ seg_tile, const uint32_t *__restrict__ point_list,
                                               const float2 *__restrict__ xy, const float4 *__restrict__ conic_opacity,
                                               float *__restrict__ seg_T, const GomDevStatus *__restrict__ status) {
    __shared__ float s[6][GOM_SEG];
    if (status->overflow) return;
    const uint32_t nsegs = status->num_segs;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t seg = blockIdx.x; seg < nsegs; seg += gridDim.x) {
        const uint32_t tile = seg_tile[seg];
        const uint32_t base = tile_base[tile], n = tile_base[tile + 1] - base;
        const uint32_t e0 = (seg - seg_base[tile]) * GOM_SEG;
        const uint32_t cnt = min((uint32_t)GOM_SEG, n - e0);
        __syncthreads();
        if (threadIdx.x < cnt) {
            const uint32_t g = point_list[base + e0 + threadIdx.x];
            const float2 c = xy[g];
            const float4 co = conic_opacity[g];
            s[0][threadIdx.x] = c.x; s[1][threadIdx.x] = c.y; s[2][threadIdx.x] = co.x;
            s[3][threadIdx.x] = co.y; s[4][threadIdx.x] = co.z; s[5][threadIdx.x] = co.w;
        }
        __syncthreads();
        const int tx = tile % gx, ty = tile / gx;
        const float qx0 = (float)(tx * 16 + (wave & 1) * 8), qy0 = (float)(ty * 16 + (wave >> 1) * 8);
        const float qx1 = qx0 + 7.f, qy1 = qy0 + 7.f;
        const float pfx = qx0 + (float)(lane & 7), pfy = qy0 + (float)(lane >> 3);
        float T = 1.f;
        for (uint32_t b0 = 0; b0 < cnt; b0 += 64) {
            const uint32_t e = b0 + lane, ee = e < cnt ? e : 0;
            const float ex = s[0][ee], ey = s[1][ee], ea = s[2][ee], eb = s[3][ee], ec = s[4][ee], eo = s[5][ee];
            const bool keep = e < cnt && !cull_entry(ex, ey, ea, eb, ec, eo, qx0, qy0, qx1, qy1);
            unsigned long long mask = __ballot(keep);
            while (mask) {
                float al[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const bool kv = mask != 0ull;
                    const int k = kv ? __builtin_ctzll(mask) : 0;
                    mask &= mask - 1;
                    al[u] = entry_alpha(rl(ex, k), rl(ey, k), rl(ea, k), rl(eb, k), rl(ec, k), kv ? rl(eo, k) : 0.f, pfx, pfy);
                }
#pragma unroll
                for (int u = 0; u < 4; u++) T = T * (1.f - al[u]);
            }
        }
        seg_T[(size_t)seg * GOM_TPX + threadIdx.x] = T;
    }
}

// ---------------------------------------- forward, pass B (exact, per segment)
// Every (tile, segment) composites its 256 entries with the reference's exact
// per-pixel rules (skip, stop at T(1-alpha) < 1e-4), starting from the
// transmittance the pixel has when it reaches the segment (product of the
// earlier segments' prod(1-alpha)).  Stores the contribution, T after the
// segment (negated if the stop rule fired inside) and the last contributor.
template <int C>
__global__ void __launch_bounds__(256) k_seg_fwd(int gx, const uint32_t *__restrict__ tile_base, const uint32_t *__restrict__ seg_base,
                                                 const uint32_t *__restrict__ seg_tile, const uint32_t *__restrict__ point_list,
                                                 const float2 *__restrict__ xy, const float4 *__restrict__ conic_opacity,
                                                 const float *__restrict__ colors, const float *__restrict__ seg_T, float *__restrict__ seg_C,
                                                 float *__restrict__ seg_Tend, uint32_t *__restrict__ seg_last,
                                                 const GomDevStatus *__restrict__ status) {
    __shared__ SegLds<C> s;
    if (status->overflow) return;
    const uint32_t nsegs = status->num_segs;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t seg = blockIdx.x; seg < nsegs; seg += gridDim.x) {
        const uint32_t tile = seg_tile[seg];
        const uint32_t base = tile_base[tile], n = tile_base[tile + 1] - base;
        const uint32_t sb = seg_base[tile];
        const uint32_t e0 = (seg - sb) * GOM_SEG;
        const uint32_t cnt = min((uint32_t)GOM_SEG, n - e0);
        // transmittance at the start of this segment; 0 = the pixel has certainly stopped earlier.
        // (In the 1e-5-wide borderline band the pixel stays alive; the combine pass honours the
        //  exact stop flag of the earlier segment.)
        float T = 1.f;
        for (uint32_t r = sb; r < seg; r++) {
            const float Tn = T * seg_T[(size_t)r * GOM_TPX + threadIdx.x];
            T = (Tn >= kStopT * 0.99999f) ? Tn : 0.f;
        }
        float wl = T > 0.f ? 1.f : 0.f;  // lane still compositing (float mask: no SALU in the chain)
        const bool any_alive = __syncthreads_or(wl != 0.f ? 1 : 0) != 0;
        float acc[C];
#pragma unroll
        for (int ch = 0; ch < C; ch++) acc[ch] = 0.f;
        uint32_t last = 0;
        if (any_alive) {
            stage_segment<C>(s, cnt, point_list + base + e0, xy, conic_opacity, colors);
            __syncthreads();
            if (__ballot(wl != 0.f) != 0ull) {
                const int tx = tile % gx, ty = tile / gx;
                const float qx0 = (float)(tx * 16 + (wave & 1) * 8), qy0 = (float)(ty * 16 + (wave >> 1) * 8);
                const float qx1 = qx0 + 7.f, qy1 = qy0 + 7.f;
                const float pfx = qx0 + (float)(lane & 7), pfy = qy0 + (float)(lane >> 3);
                for (uint32_t b0 = 0; b0 < cnt; b0 += 64) {
                    if (__ballot(wl != 0.f) == 0ull) break;
                    const EntryRegs<C> r = fetch_entry<C>(s, b0 + lane, cnt, qx0, qy0, qx1, qy1);
                    unsigned long long mask = __ballot(r.keep);
                    while (mask) {
                        int kk[4];
                        float al[4], ecol[4][C];
#pragma unroll
                        for (int u = 0; u < 4; u++) {  // independent alpha evaluations (ILP)
                            const bool kv = mask != 0ull;
                            const int k = kv ? __builtin_ctzll(mask) : 0;
                            mask &= mask - 1;
                            kk[u] = k;
                            const float eo = kv ? rl(r.o, k) : 0.f;  // opacity 0 -> alpha 0 -> "skip"
                            al[u] = entry_alpha(rl(r.x, k), rl(r.y, k), rl(r.a, k), rl(r.b, k), rl(r.c, k), eo, pfx, pfy);
#pragma unroll
                            for (int ch = 0; ch < C; ch++) ecol[u][ch] = rl(r.col[ch], k);
                        }
#pragma unroll
                        for (int u = 0; u < 4; u++) {  // the serial chain: T -> test_T -> select
                            const float a = al[u] * wl;
                            const float test_T = T * (1.f - a);
                            const bool cont = test_T >= kStopT;  // reference: `test_T < 0.0001f -> done`
                            const float w = cont ? a * T : 0.f;
#pragma unroll
                            for (int ch = 0; ch < C; ch++) acc[ch] += ecol[u][ch] * w;
                            T = cont ? test_T : T;
                            wl = cont ? wl : 0.f;
                            last = (w > 0.f) ? (e0 + b0 + (uint32_t)kk[u] + 1u) : last;
                        }
                        if (__ballot(wl != 0.f) == 0ull) break;
                    }
                }
            }
            __syncthreads();  // LDS is restaged by the next segment of this workgroup
        }
        const size_t o = (size_t)seg * GOM_TPX + threadIdx.x;
        // dead on arrival: T = 0.  stopped inside: -T.  still going: +T.
        seg_Tend[o] = (T > 0.f && wl == 0.f) ? -T : T;
        seg_last[o] = last;
#pragma unroll
        for (int ch = 0; ch < C; ch++) seg_C[((size_t)seg * 4 + ch) * GOM_TPX + threadIdx.x] = acc[ch];
    }
}

// ----------------------------------------------- forward, pass C (assembly) -
template <int C>
__global__ void __launch_bounds__(256) k_combine_fwd(int H, int W, int gx, float bg0, float bg1, float bg2, float bg3,
                                                     const uint32_t *__restrict__ seg_base, const float *__restrict__ seg_C,
                                                     const uint32_t *__restrict__ seg_last, float *__restrict__ seg_Tend,
                                                     float *__restrict__ seg_Sbehind, float *__restrict__ out_color,
                                                     float *__restrict__ final_T, uint32_t *__restrict__ n_contrib,
                                                     uint32_t *__restrict__ tile_nmax, const GomDevStatus *__restrict__ status) {
    __shared__ uint32_t s_nmax[4];
    const int tile = blockIdx.x;
    const int tx = tile % gx, ty = tile / gx;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int px = tx * 16 + (wave & 1) * 8 + (lane & 7);
    const int py = ty * 16 + (wave >> 1) * 8 + (lane >> 3);
    const bool inside = px < W && py < H;
    const size_t HW = (size_t)H * W;
    const size_t pix = (size_t)py * W + px;
    const float bg[4] = {bg0, bg1, bg2, bg3};
    if (status->overflow) {  // pair buffers too small: poison loudly
        if (inside) {
            const float nanv = __uint_as_float(0x7fc00000u);
#pragma unroll
            for (int ch = 0; ch < C; ch++) out_color[ch * HW + pix] = nanv;
            final_T[pix] = nanv;
            n_contrib[pix] = 0;
        }
        if (threadIdx.x == 0) tile_nmax[tile] = 0;
        return;
    }
    const uint32_t sb = seg_base[tile], nseg = seg_base[tile + 1] - sb;
    float T = 1.f, acc[C];
#pragma unroll
    for (int ch = 0; ch < C; ch++) acc[ch] = 0.f;
    uint32_t last = 0;
    int s_stop = (int)nseg - 1;  // last segment that contributed to this pixel
    bool going = true;
    // 8 segments per trip: all loads of a trip are issued before the (cheap, serial) fold, so the list of
    // segments costs one memory latency per 8 instead of one per segment.
    for (uint32_t s0 = 0; s0 < nseg; s0 += 8) {
        float te[8], cs[8][C];
        uint32_t ll[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const uint32_t sl = min(s0 + u, nseg - 1);
            const size_t o = (size_t)(sb + sl) * GOM_TPX + threadIdx.x;
            te[u] = seg_Tend[o];
            ll[u] = seg_last[o];
#pragma unroll
            for (int ch = 0; ch < C; ch++) cs[u][ch] = seg_C[((size_t)(sb + sl) * 4 + ch) * GOM_TPX + threadIdx.x];
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
    // colour still to come behind each segment (small terms first: accurate suffix sums)
    {
        float S[C];
#pragma unroll
        for (int ch = 0; ch < C; ch++) S[ch] = 0.f;
        for (int s1 = (int)nseg; s1 > 0; s1 -= 8) {
            float cs[8][C];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int sl = max(s1 - 1 - u, 0);
#pragma unroll
                for (int ch = 0; ch < C; ch++) cs[u][ch] = seg_C[((size_t)(sb + sl) * 4 + ch) * GOM_TPX + threadIdx.x];
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int sl = s1 - 1 - u;
                if (sl >= 0) {
#pragma unroll
                    for (int ch = 0; ch < C; ch++) {
                        seg_Sbehind[((size_t)(sb + sl) * 4 + ch) * GOM_TPX + threadIdx.x] = S[ch];
                        if (sl <= s_stop) S[ch] += cs[u][ch];
                    }
                }
            }
        }
    }
    if (inside) {
        final_T[pix] = T;
        n_contrib[pix] = last;
#pragma unroll
        for (int ch = 0; ch < C; ch++) out_color[ch * HW + pix] = acc[ch] + T * bg[ch];
    }
    const uint32_t wmax = wave_max_u32(inside ? last : 0u);
    if (lane == 0) s_nmax[wave] = wmax;
    __syncthreads();
    if (threadIdx.x == 0) tile_nmax[tile] = max(max(s_nmax[0], s_nmax[1]), max(s_nmax[2], s_nmax[3]));
}

// ---------------------------------------------------------------- backward -
template <int C>
__global__ void __launch_bounds__(256) k_seg_bwd(int H, int W, int gx, float bg0, float bg1, float bg2, float bg3,
                                                 const uint32_t *__restrict__ tile_base, const uint32_t *__restrict__ seg_base,
                                                 const uint32_t *__restrict__ seg_tile, const uint32_t *__restrict__ tile_nmax,
                                                 const uint32_t *__restrict__ point_list, const float2 *__restrict__ xy,
                                                 const float4 *__restrict__ conic_opacity, const float *__restrict__ colors,
                                                 const float *__restrict__ final_T, const uint32_t *__restrict__ n_contrib,
                                                 const float *__restrict__ dL_dpix, const float *__restrict__ seg_Tend,
                                                 const float *__restrict__ seg_Sbehind, float *__restrict__ partial,
                                                 const GomDevStatus *__restrict__ status) {
    constexpr int NV = 6 + C;  // values reduced per entry
    __shared__ SegLds<C> s;
    __shared__ float s_acc[4][GOM_SEG][10];
    if (status->overflow) return;
    const uint32_t nsegs = status->num_segs;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t HW = (size_t)H * W;
    for (uint32_t seg = blockIdx.x; seg < nsegs; seg += gridDim.x) {
        const uint32_t tile = seg_tile[seg];
        const uint32_t base = tile_base[tile], n = tile_base[tile + 1] - base;
        const uint32_t e0 = (seg - seg_base[tile]) * GOM_SEG;
        const uint32_t cnt = min((uint32_t)GOM_SEG, n - e0);
        float4 *rec = reinterpret_cast<float4 *>(partial + (size_t)(base + e0 + threadIdx.x) * GOM_PARTIAL_STRIDE);
        if (e0 >= tile_nmax[tile]) {  // every pixel of the tile stopped before this segment: all-zero records
            if (threadIdx.x < cnt) {
                const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
                rec[0] = z; rec[1] = z; rec[2] = z;
            }
            continue;
        }
        __syncthreads();
        stage_segment<C>(s, cnt, point_list + base + e0, xy, conic_opacity, colors);
        for (int i = threadIdx.x; i < 4 * GOM_SEG * 10; i += 256) (&s_acc[0][0][0])[i] = 0.f;
        __syncthreads();

        const int tx = tile % gx, ty = tile / gx;
        const int px = tx * 16 + (wave & 1) * 8 + (lane & 7);
        const int py = ty * 16 + (wave >> 1) * 8 + (lane >> 3);
        const bool inside = px < W && py < H;
        const size_t pix = (size_t)py * W + px;
        const float pfx = (float)px, pfy = (float)py;
        const float qx0 = (float)(tx * 16 + (wave & 1) * 8), qy0 = (float)(ty * 16 + (wave >> 1) * 8);
        const float qx1 = qx0 + 7.f, qy1 = qy0 + 7.f;
        const float T_final = inside ? final_T[pix] : 0.f;
        const uint32_t my_last = inside ? n_contrib[pix] : 0u;
        float dpix[C], bg_dot = 0.f;
        {
            const float bg[4] = {bg0, bg1, bg2, bg3};
#pragma unroll
            for (int ch = 0; ch < C; ch++) {
                dpix[ch] = inside ? dL_dpix[ch * HW + pix] : 0.f;
                bg_dot += bg[ch] * dpix[ch];
            }
        }
        // checkpoint written by the combine pass: state just behind this segment
        const size_t o = (size_t)seg * GOM_TPX + threadIdx.x;
        float T = seg_Tend[o];
        float accum_rec[C], last_color[C], last_alpha = 0.f;
        const float invT = T > 0.f ? 1.f / T : 0.f;
#pragma unroll
        for (int ch = 0; ch < C; ch++) {
            accum_rec[ch] = seg_Sbehind[((size_t)seg * 4 + ch) * GOM_TPX + threadIdx.x] * invT;
            last_color[ch] = 0.f;
        }
        const uint32_t wmax = wave_max_u32(my_last);
        if (wmax > e0) {
            const uint32_t lim = min(cnt, wmax - e0);  // entries at or beyond wmax are dead for this wave
            for (int b0 = (int)((lim - 1) & ~63u); b0 >= 0; b0 -= 64) {
                const EntryRegs<C> r = fetch_entry<C>(s, (uint32_t)b0 + lane, lim, qx0, qy0, qx1, qy1);
                unsigned long long mask = __ballot(r.keep);
                while (mask) {
                    const int k = 63 - __builtin_clzll(mask);
                    mask &= ~(1ull << k);
                    const uint32_t el = (uint32_t)b0 + (uint32_t)k;  // index inside the segment
                    const float ex = rl(r.x, k), ey = rl(r.y, k);
                    const float ea = rl(r.a, k), eb = rl(r.b, k), ec = rl(r.c, k), eo = rl(r.o, k);
                    float ecol[C];
#pragma unroll
                    for (int ch = 0; ch < C; ch++) ecol[ch] = rl(r.col[ch], k);
                    const float dx = ex - pfx, dy = ey - pfy;
                    const float power = -0.5f * (ea * dx * dx + ec * dy * dy) - eb * dx * dy;
                    const float G0 = __expf(power);
                    float a = fminf(kMaxAlpha, eo * G0);
                    a = (power <= 0.f) ? a : 0.f;
                    a = (a >= kMinAlpha) ? a : 0.f;
                    a = (e0 + el < my_last) ? a : 0.f;  // beyond this pixel's last contributor
                    if (__ballot(a > 0.f) == 0ull) continue;
                    // An entry with a == 0 is replayed as a zero-alpha layer: the recurrences below then
                    // leave T / accum_rec exactly as skipping would (App. A.4), without divergent branches.
                    const float G = (a > 0.f) ? G0 : 0.f;
                    const float inv1ma = __frcp_rn(1.f - a);  // v_rcp_f32: 1 ulp, shared by both divisions
                    T = T * inv1ma;
                    const float w = a * T;
                    float dL_dalpha = 0.f, v[NV];
#pragma unroll
                    for (int ch = 0; ch < C; ch++) {
                        accum_rec[ch] = last_alpha * last_color[ch] + (1.f - last_alpha) * accum_rec[ch];
                        last_color[ch] = ecol[ch];
                        dL_dalpha += (ecol[ch] - accum_rec[ch]) * dpix[ch];
                        v[ch] = w * dpix[ch];
                    }
                    dL_dalpha *= T;
                    last_alpha = a;
                    dL_dalpha += (-T_final * inv1ma) * bg_dot;
                    const float Q = G * dL_dalpha;
                    v[C + 0] = Q;
                    v[C + 1] = Q * dx;
                    v[C + 2] = Q * dy;
                    v[C + 3] = Q * dx * dx;
                    v[C + 4] = Q * dx * dy;
                    v[C + 5] = Q * dy * dy;
                    wave_sum_lane63_n<NV>(v);
                    if (lane == 63) {
                        float *dst = &s_acc[wave][el][0];
#pragma unroll
                        for (int ch = 0; ch < C; ch++) dst[ch] = v[ch];
#pragma unroll
                        for (int q = 0; q < 6; q++) dst[4 + q] = v[C + q];
                    }
                }
            }
        }
        __syncthreads();
        if (threadIdx.x < cnt) {  // one 48-byte record per entry, waves summed in a fixed order
            float rr[10];
#pragma unroll
            for (int q = 0; q < 10; q++)
                rr[q] = ((s_acc[0][threadIdx.x][q] + s_acc[1][threadIdx.x][q]) + s_acc[2][threadIdx.x][q]) + s_acc[3][threadIdx.x][q];
            rec[0] = make_float4(rr[0], rr[1], rr[2], rr[3]);
            rec[1] = make_float4(rr[4], rr[5], rr[6], rr[7]);
            rec[2] = make_float4(rr[8], rr[9], 0.f, 0.f);
        }
    }
}

}  // namespace

int gom_launch_sort(GomState *s, hipStream_t st) {
    const int n_tiles = s->gx * s->gy;
    if (n_tiles == 0) return 0;
    GomKernelTimer timer(s, GOM_K_SORT, st);
    hipLaunchKernelGGL(k_sort, dim3(n_tiles), dim3(1024), 0, st, s->gx, s->tile_base, s->seg_base, s->keys, s->point_list, s->seg_tile,
                       s->rect, s->pair_off, s->pair_pos, s->status, (uint32_t)s->sortCap);
    GOM_LAUNCH_CHECK();
    return 0;
}

int gom_launch_render_forward(GomState *s, const GomCamera &cam, int C, const float *colors, float *out_color, bool reuse_T,
                              hipStream_t st) {
    const int n_tiles = s->gx * s->gy;
    if (n_tiles == 0) return 0;
    if (!reuse_T) {  // depends on geometry only: shared by every colour pass over the same binning
        GomKernelTimer timer(s, GOM_K_SEG_T, st);
        hipLaunchKernelGGL(k_seg_T, dim3(GOM_SEG_GRID), dim3(256), 0, st, s->gx, s->tile_base, s->seg_base, s->seg_tile, s->point_list,
                           s->xy, s->conic_opacity, s->seg_T, s->status);
    }
    GOM_LAUNCH_CHECK();
    {
        GomKernelTimer timer(s, GOM_K_SEG_FWD, st);
#define GOM_SF(CC)                                                                                                        \
    hipLaunchKernelGGL((k_seg_fwd<CC>), dim3(GOM_SEG_GRID), dim3(256), 0, st, s->gx, s->tile_base, s->seg_base, s->seg_tile,     \
                       s->point_list, s->xy, s->conic_opacity, colors, s->seg_T, s->seg_C, s->seg_Tend, s->seg_last, s->status)
        if (C == 3) GOM_SF(3); else GOM_SF(4);
#undef GOM_SF
    }
    GOM_LAUNCH_CHECK();
    {
        GomKernelTimer timer(s, GOM_K_COMBINE, st);
#define GOM_CF(CC)                                                                                                        \
    hipLaunchKernelGGL((k_combine_fwd<CC>), dim3(n_tiles), dim3(256), 0, st, s->H, s->W, s->gx, cam.bg[0], cam.bg[1], cam.bg[2], \
                       cam.bg[3], s->seg_base, s->seg_C, s->seg_last, s->seg_Tend, s->seg_Sbehind, out_color, s->final_T,        \
                       s->n_contrib, s->tile_nmax, s->status)
        if (C == 3) GOM_CF(3); else GOM_CF(4);
#undef GOM_CF
    }
    GOM_LAUNCH_CHECK();
    return 0;
}

int gom_launch_render_backward(GomState *s, const GomCamera &cam, int C, const float *colors, const float *dL_dcolor,
                               hipStream_t st) {
    const int n_tiles = s->gx * s->gy;
    if (n_tiles == 0) return 0;
    GomKernelTimer timer(s, GOM_K_SEG_BWD, st);
#define GOM_SB(CC)                                                                                                        \
    hipLaunchKernelGGL((k_seg_bwd<CC>), dim3(GOM_SEG_GRID), dim3(256), 0, st, s->H, s->W, s->gx, cam.bg[0], cam.bg[1], cam.bg[2], \
                       cam.bg[3], s->tile_base, s->seg_base, s->seg_tile, s->tile_nmax, s->point_list, s->xy, s->conic_opacity,   \
                       colors, s->final_T, s->n_contrib, dL_dcolor, s->seg_Tend, s->seg_Sbehind, s->partial, s->status)
    if (C == 3) GOM_SB(3); else GOM_SB(4);
#undef GOM_SB
    GOM_LAUNCH_CHECK();
    return 0;
}
