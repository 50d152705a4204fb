// Per-tile kernels of the splat rasterizer: depth sort + front-to-back alpha
// compositing (forward) and back-to-front replay (backward).
//
// Replaces cub::DeviceRadixSort + identifyTileRanges + renderCUDA (forward and
// backward) of the CUDA extension the reference calls at
// models/modules/renderer/gaussian.py:83-91 (algorithm: SURVEY.md App. A.2-A.4).
//
// MI355X design (wave64, not a warp-shaped port):
//  * one 256-thread workgroup per 16x16 tile; it first sorts the tile's
//    (depth_bits<<32 | gaussian) keys in LDS (normalised bitonic network, no
//    padding) -- the keys are unique, so the order equals the reference's
//    stable (tile, depth) radix sort -- and writes the sorted list back.
//  * then each of the 4 waves composites its own 8x8 pixel quadrant
//    independently (no block barriers in the blend loop): lane = pixel.
//    Per 64-entry batch, lane = ENTRY first: every lane fetches one entry
//    (coalesced list read, L2 gathers of the attributes) and tests it against
//    the wave's 8x8 rectangle with a conservative bound on the Gaussian's
//    maximum alpha there; a 64-bit ballot then drives a scalar loop over the
//    surviving entries only, whose attributes are broadcast with v_readlane
//    (SGPR operands) instead of being staged through LDS.  Skipped entries
//    are exactly those the reference skips for all 64 pixels, so results and
//    the per-pixel contributor indices are unchanged.
//  * backward: same structure back-to-front; the 6+C per-pixel terms of each
//    entry are reduced over the wave with DPP row_shr/row_bcast adds and the
//    four waves' sums are combined in LDS in a fixed order, then written as one
//    48-byte record per (tile, entry).  No float atomics: gradients are bitwise
//    reproducible.
#include "gom_internal.h"

namespace {

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
// (with a rounding margin), which is the reference's own skip test
// (App. A.3: `alpha < 1/255 -> continue`).  Never culls when unsure.
__device__ __forceinline__ bool cull_entry(float2 c, float4 co, float x0, float y0, float x1, float y1) {
    const float a = co.x, b = co.y, cz = co.z, o = co.w;
    if (!(a > 0.f) || !(cz > 0.f) || !(a * cz - b * b > 0.f)) return false;  // not positive definite: exact path
    if (o <= 0.f) return true;                                                // alpha = o*G <= 0 < 1/255
    if (!(o < 3.0e38f)) return false;
    const float X = c.x < x0 ? (x0 - c.x) : (c.x > x1 ? (x1 - c.x) : 0.f);
    const float Y = c.y < y0 ? (y0 - c.y) : (c.y > y1 ? (y1 - c.y) : 0.f);
    if (X == 0.f && Y == 0.f) return false;
    float q = 3.0e38f;
    if (X != 0.f) {
        float dy = -b * X / cz;
        dy = fminf(fmaxf(dy, y0 - c.y), y1 - c.y);
        q = fminf(q, a * X * X + 2.f * b * X * dy + cz * dy * dy);
    }
    if (Y != 0.f) {
        float dx = -b * Y / a;
        dx = fminf(fmaxf(dx, x0 - c.x), x1 - c.x);
        q = fminf(q, a * dx * dx + 2.f * b * dx * Y + cz * Y * Y);
    }
    const float DX = fmaxf(fabsf(x0 - c.x), fabsf(x1 - c.x));
    const float DY = fmaxf(fabsf(y0 - c.y), fabsf(y1 - c.y));
    const float mag = a * DX * DX + 2.f * fabsf(b) * DX * DY + cz * DY * DY;
    const float lthr = -__logf(255.0f * o);  // alpha >= 1/255  <=>  power >= lthr
    return (-0.5f * q + (1e-5f * mag + 1e-2f)) < lthr;
}

// Normalised bitonic network on n (<= cap) unique 64-bit keys: every
// comparator orders ascending, so indices >= n behave as +inf padding and are
// simply skipped.  Works on LDS or global memory (workgroup-visible).
template <typename PTR>
__device__ __forceinline__ void bitonic_sort_u64(PTR keys, uint32_t n) {
    for (uint32_t k = 2; (k >> 1) < n; k <<= 1) {
        const uint32_t half = k >> 1;
        for (uint32_t i = threadIdx.x; i < ((n + k - 1) / k) * half; i += 256) {
            const uint32_t blk = i / half, off = i % half;
            const uint32_t lo = blk * k + off;
            const uint32_t hi = blk * k + (k - 1 - off);
            if (hi < n) {
                const uint64_t a = keys[lo], b = keys[hi];
                if (a > b) { keys[lo] = b; keys[hi] = a; }
            }
        }
        __syncthreads();
        for (uint32_t j = half >> 1; j > 0; j >>= 1) {
            for (uint32_t i = threadIdx.x; i < ((n + 2 * j - 1) / (2 * j)) * j; i += 256) {
                const uint32_t lo = (i / j) * 2 * j + (i % j);
                const uint32_t hi = lo + j;
                if (hi < n) {
                    const uint64_t a = keys[lo], b = keys[hi];
                    if (a > b) { keys[lo] = b; keys[hi] = a; }
                }
            }
            __syncthreads();
        }
    }
}

// ---------------------------------------------------------------- forward --
template <int C, bool DO_SORT>
__global__ void __launch_bounds__(256) k_render_fwd(int H, int W, int gx, float bg0, float bg1, float bg2, float bg3,
                                                    const uint32_t *__restrict__ tile_base, uint64_t *__restrict__ keys,
                                                    uint32_t *__restrict__ point_list, const float2 *__restrict__ xy,
                                                    const float4 *__restrict__ conic_opacity, const float *__restrict__ colors,
                                                    float *__restrict__ out_color, float *__restrict__ final_T,
                                                    uint32_t *__restrict__ n_contrib, const GomDevStatus *__restrict__ status,
                                                    uint32_t sort_cap) {
    __shared__ uint64_t s_keys[DO_SORT ? GOM_SORT_CAP_MAX : 1];
    const int tile = blockIdx.x;
    const int tx = tile % gx, ty = tile / gx;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int px = tx * 16 + (wave & 1) * 8 + (lane & 7);
    const int py = ty * 16 + (wave >> 1) * 8 + (lane >> 3);
    const bool inside = px < W && py < H;
    const size_t HW = (size_t)H * W;
    const float bg[4] = {bg0, bg1, bg2, bg3};

    if (status->overflow) {  // pair buffers too small: poison loudly
        if (inside) {
            const float nanv = __uint_as_float(0x7fc00000u);
#pragma unroll
            for (int ch = 0; ch < C; ch++) out_color[ch * HW + (size_t)py * W + px] = nanv;
            final_T[(size_t)py * W + px] = nanv;
            n_contrib[(size_t)py * W + px] = 0;
        }
        return;
    }
    const uint32_t base = tile_base[tile];
    const uint32_t n = tile_base[tile + 1] - base;
    bool list_in_lds = false;
    if (DO_SORT && n > 0) {
        if (n <= sort_cap) {
            for (uint32_t i = threadIdx.x; i < n; i += 256) s_keys[i] = keys[base + i];
            __syncthreads();
            bitonic_sort_u64(s_keys, n);
            for (uint32_t i = threadIdx.x; i < n; i += 256) {
                const uint64_t k = s_keys[i];
                keys[base + i] = k;
                point_list[base + i] = (uint32_t)k;
            }
            list_in_lds = true;
        } else {
            bitonic_sort_u64(keys + base, n);
            for (uint32_t i = threadIdx.x; i < n; i += 256) point_list[base + i] = (uint32_t)keys[base + i];
            __syncthreads();
        }
    }

    // ---- compositing: this wave's 8x8 quadrant, lane = pixel ----
    const float pfx = (float)px, pfy = (float)py;
    const float qx0 = (float)(tx * 16 + (wave & 1) * 8), qy0 = (float)(ty * 16 + (wave >> 1) * 8);
    const float qx1 = qx0 + 7.f, qy1 = qy0 + 7.f;
    float T = 1.f, acc[C];
#pragma unroll
    for (int ch = 0; ch < C; ch++) acc[ch] = 0.f;
    uint32_t last = 0;
    bool done = !inside;

    for (uint32_t b0 = 0; b0 < n; b0 += 64) {
        if (__ballot(!done) == 0ull) break;
        // lane = entry
        const uint32_t e = b0 + lane;
        float2 c = make_float2(0.f, 0.f);
        float4 co = make_float4(0.f, 0.f, 0.f, 0.f);
        float col[C];
#pragma unroll
        for (int ch = 0; ch < C; ch++) col[ch] = 0.f;
        bool keep = false;
        if (e < n) {
            const uint32_t g = (DO_SORT && list_in_lds) ? (uint32_t)s_keys[e] : point_list[base + e];
            c = xy[g];
            co = conic_opacity[g];
            keep = !cull_entry(c, co, qx0, qy0, qx1, qy1);
            if (keep) {
#pragma unroll
                for (int ch = 0; ch < C; ch++) col[ch] = colors[(size_t)g * C + ch];
            }
        }
        unsigned long long mask = __ballot(keep);
        while (mask) {
            const int k = __builtin_ctzll(mask);
            mask &= mask - 1;
            const float ex = rl(c.x, k), ey = rl(c.y, k);
            const float ea = rl(co.x, k), eb = rl(co.y, k), ec = rl(co.z, k), eo = rl(co.w, k);
            float ecol[C];
#pragma unroll
            for (int ch = 0; ch < C; ch++) ecol[ch] = rl(col[ch], k);
            if (!done) {
                const float dx = ex - pfx, dy = ey - pfy;
                const float power = -0.5f * (ea * dx * dx + ec * dy * dy) - eb * dx * dy;
                if (power <= 0.f) {
                    const float alpha = fminf(0.99f, eo * __expf(power));
                    if (alpha >= 1.0f / 255.0f) {
                        const float test_T = T * (1.f - alpha);
                        if (test_T < 0.0001f) {
                            done = true;
                        } else {
                            const float w = alpha * T;
#pragma unroll
                            for (int ch = 0; ch < C; ch++) acc[ch] += ecol[ch] * w;
                            T = test_T;
                            last = b0 + (uint32_t)k + 1u;
                        }
                    }
                }
            }
            if (__ballot(!done) == 0ull) break;
        }
    }
    if (inside) {
        const size_t pix = (size_t)py * W + px;
        final_T[pix] = T;
        n_contrib[pix] = last;
#pragma unroll
        for (int ch = 0; ch < C; ch++) out_color[ch * HW + pix] = acc[ch] + T * bg[ch];
    }
}

// ---------------------------------------------------------------- backward -
template <int C>
__global__ void __launch_bounds__(256) k_render_bwd(int H, int W, int gx, float bg0, float bg1, float bg2, float bg3,
                                                    const uint32_t *__restrict__ tile_base, const uint32_t *__restrict__ point_list,
                                                    const float2 *__restrict__ xy, const float4 *__restrict__ conic_opacity,
                                                    const float *__restrict__ colors, const float *__restrict__ final_T,
                                                    const uint32_t *__restrict__ n_contrib, const float *__restrict__ dL_dpix,
                                                    float *__restrict__ partial, uint32_t *__restrict__ tile_done,
                                                    const GomDevStatus *__restrict__ status) {
    constexpr int NV = 6 + C;  // values reduced per entry
    __shared__ float s_acc[4][GOM_BWD_CHUNK][10];
    __shared__ uint32_t s_nmax[4];
    if (status->overflow) return;
    const int tile = blockIdx.x;
    const int tx = tile % gx, ty = tile / gx;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int px = tx * 16 + (wave & 1) * 8 + (lane & 7);
    const int py = ty * 16 + (wave >> 1) * 8 + (lane >> 3);
    const bool inside = px < W && py < H;
    const size_t HW = (size_t)H * W;
    const size_t pix = (size_t)py * W + px;
    const uint32_t base = tile_base[tile];
    const uint32_t n = tile_base[tile + 1] - base;
    if (n == 0) return;

    const float pfx = (float)px, pfy = (float)py;
    const float qx0 = (float)(tx * 16 + (wave & 1) * 8), qy0 = (float)(ty * 16 + (wave >> 1) * 8);
    const float qx1 = qx0 + 7.f, qy1 = qy0 + 7.f;
    const float T_final = inside ? final_T[pix] : 0.f;
    const uint32_t my_last = inside ? n_contrib[pix] : 0u;
    float dpix[C];
    float bg_dot = 0.f;
    {
        const float bg[4] = {bg0, bg1, bg2, bg3};
#pragma unroll
        for (int ch = 0; ch < C; ch++) {
            dpix[ch] = inside ? dL_dpix[ch * HW + pix] : 0.f;
            bg_dot += bg[ch] * dpix[ch];
        }
    }
    const uint32_t wmax = wave_max_u32(my_last);
    if (lane == 0) s_nmax[wave] = wmax;
    for (int i = threadIdx.x; i < 4 * GOM_BWD_CHUNK * 10; i += 256) (&s_acc[0][0][0])[i] = 0.f;
    __syncthreads();
    const uint32_t nmax = max(max(s_nmax[0], s_nmax[1]), max(s_nmax[2], s_nmax[3]));
    if (nmax == 0) return;  // tile_done stays 0: no records
    const uint32_t n_chunks = (nmax + GOM_BWD_CHUNK - 1) / GOM_BWD_CHUNK;
    if (threadIdx.x == 0) tile_done[tile] = min(n, n_chunks * GOM_BWD_CHUNK);

    float T = T_final, last_alpha = 0.f, accum_rec[C], last_color[C];
#pragma unroll
    for (int ch = 0; ch < C; ch++) { accum_rec[ch] = 0.f; last_color[ch] = 0.f; }

    for (int chunk = (int)n_chunks - 1; chunk >= 0; chunk--) {
        const uint32_t c0 = (uint32_t)chunk * GOM_BWD_CHUNK;
        if (c0 < wmax) {
            for (int bb = GOM_BWD_CHUNK / 64 - 1; bb >= 0; bb--) {
                const uint32_t b0 = c0 + (uint32_t)bb * 64;
                if (b0 >= wmax) continue;
                const uint32_t e = b0 + lane;
                float2 c = make_float2(0.f, 0.f);
                float4 co = make_float4(0.f, 0.f, 0.f, 0.f);
                float col[C];
#pragma unroll
                for (int ch = 0; ch < C; ch++) col[ch] = 0.f;
                bool keep = false;
                if (e < wmax) {
                    const uint32_t g = point_list[base + e];
                    c = xy[g];
                    co = conic_opacity[g];
                    keep = !cull_entry(c, co, qx0, qy0, qx1, qy1);
                    if (keep) {
#pragma unroll
                        for (int ch = 0; ch < C; ch++) col[ch] = colors[(size_t)g * C + ch];
                    }
                }
                unsigned long long mask = __ballot(keep);
                while (mask) {
                    const int k = 63 - __builtin_clzll(mask);
                    mask &= ~(1ull << k);
                    const uint32_t ek = b0 + (uint32_t)k;
                    const float ex = rl(c.x, k), ey = rl(c.y, k);
                    const float ea = rl(co.x, k), eb = rl(co.y, k), ec = rl(co.z, k), eo = rl(co.w, k);
                    float ecol[C];
#pragma unroll
                    for (int ch = 0; ch < C; ch++) ecol[ch] = rl(col[ch], k);
                    const float dx = ex - pfx, dy = ey - pfy;
                    const float power = -0.5f * (ea * dx * dx + ec * dy * dy) - eb * dx * dy;
                    const float G = __expf(power);
                    const float alpha = fminf(0.99f, eo * G);
                    const bool act = (ek < my_last) && (power <= 0.f) && (alpha >= 1.0f / 255.0f);
                    if (__ballot(act) == 0ull) continue;
                    float v[NV];
#pragma unroll
                    for (int q = 0; q < NV; q++) v[q] = 0.f;
                    if (act) {
                        T = T / (1.f - alpha);
                        const float w = alpha * T;
                        float dL_dalpha = 0.f;
#pragma unroll
                        for (int ch = 0; ch < C; ch++) {
                            accum_rec[ch] = last_alpha * last_color[ch] + (1.f - last_alpha) * accum_rec[ch];
                            last_color[ch] = ecol[ch];
                            dL_dalpha += (ecol[ch] - accum_rec[ch]) * dpix[ch];
                            v[ch] = w * dpix[ch];
                        }
                        dL_dalpha *= T;
                        last_alpha = alpha;
                        dL_dalpha += (-T_final / (1.f - alpha)) * bg_dot;
                        const float Q = G * dL_dalpha;
                        v[C + 0] = Q;
                        v[C + 1] = Q * dx;
                        v[C + 2] = Q * dy;
                        v[C + 3] = Q * dx * dx;
                        v[C + 4] = Q * dx * dy;
                        v[C + 5] = Q * dy * dy;
                    }
#pragma unroll
                    for (int q = 0; q < NV; q++) v[q] = wave_sum_lane63(v[q]);
                    if (lane == 63) {
                        float *dst = &s_acc[wave][ek - c0][0];
#pragma unroll
                        for (int ch = 0; ch < C; ch++) dst[ch] = v[ch];
#pragma unroll
                        for (int q = 0; q < 6; q++) dst[4 + q] = v[C + q];
                    }
                }
            }
        }
        __syncthreads();
        {   // flush this chunk: one 48-byte record per entry, waves summed in fixed order
            const uint32_t e = c0 + threadIdx.x;
            if (e < n) {
                float r[10];
#pragma unroll
                for (int q = 0; q < 10; q++) {
                    r[q] = ((s_acc[0][threadIdx.x][q] + s_acc[1][threadIdx.x][q]) + s_acc[2][threadIdx.x][q]) + s_acc[3][threadIdx.x][q];
                }
                float4 *dst = reinterpret_cast<float4 *>(partial + (size_t)(base + e) * GOM_PARTIAL_STRIDE);
                dst[0] = make_float4(r[0], r[1], r[2], r[3]);
                dst[1] = make_float4(r[4], r[5], r[6], r[7]);
                dst[2] = make_float4(r[8], r[9], 0.f, 0.f);
            }
#pragma unroll
            for (int w = 0; w < 4; w++)
#pragma unroll
                for (int q = 0; q < 10; q++) s_acc[w][threadIdx.x][q] = 0.f;
        }
        __syncthreads();
    }
}

}  // namespace

int gom_launch_render_forward(GomState *s, const GomCamera &cam, int C, const float *colors, float *out_color,
                              bool do_sort, hipStream_t st) {
    const int n_tiles = s->gx * s->gy;
    if (n_tiles == 0) return 0;
    GomKernelTimer timer(s, GOM_K_RENDER_FWD, st);
#define GOM_RF(CC, SS)                                                                                                  \
    hipLaunchKernelGGL((k_render_fwd<CC, SS>), dim3(n_tiles), dim3(256), 0, st, s->H, s->W, s->gx, cam.bg[0], cam.bg[1], \
                       cam.bg[2], cam.bg[3], s->tile_base, s->keys, s->point_list, s->xy, s->conic_opacity, colors,      \
                       out_color, s->final_T, s->n_contrib, s->status, (uint32_t)s->sortCap)
    if (C == 3) { if (do_sort) GOM_RF(3, true); else GOM_RF(3, false); }
    else        { if (do_sort) GOM_RF(4, true); else GOM_RF(4, false); }
#undef GOM_RF
    GOM_LAUNCH_CHECK();
    return 0;
}

int gom_launch_render_backward(GomState *s, const GomCamera &cam, int C, const float *colors, const float *dL_dcolor,
                               hipStream_t st) {
    const int n_tiles = s->gx * s->gy;
    if (n_tiles == 0) return 0;
    GomKernelTimer timer(s, GOM_K_RENDER_BWD, st);
#define GOM_RB(CC)                                                                                                       \
    hipLaunchKernelGGL((k_render_bwd<CC>), dim3(n_tiles), dim3(256), 0, st, s->H, s->W, s->gx, cam.bg[0], cam.bg[1],      \
                       cam.bg[2], cam.bg[3], s->tile_base, s->point_list, s->xy, s->conic_opacity, colors, s->final_T,   \
                       s->n_contrib, dL_dcolor, s->partial, s->tile_done, s->status)
    if (C == 3) GOM_RB(3); else GOM_RB(4);
#undef GOM_RB
    GOM_LAUNCH_CHECK();
    return 0;
}
