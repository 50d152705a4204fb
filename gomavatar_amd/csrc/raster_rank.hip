// Depth ranking of the Gaussians + bitmap ranking of the tile lists: the splat path's replacement for "sort every tile
// list" (cub::DeviceRadixSort::SortPairs of the CUDA extension the reference calls at
// models/modules/renderer/gaussian.py:83-91; algorithm: SURVEY.md App. A.2).
//
// The reference orders every tile list by (depth bits, Gaussian index).  That order does not depend on the tile: it is the
// restriction of ONE order of the frame's Gaussians.  So instead of sorting D ~ 150 k (tile, Gaussian) pairs per frame in
// ~170 lists (round 1: a merge sort per tile, 168 us per 8-frame launch, a 6 000-entry list being one workgroup's 50 us
// chain), the P = 55 104 Gaussians of a frame are ranked ONCE and a tile list is sorted by a pass that is linear in its
// length:
//
//   k_depth_hist      per Gaussian: depth -> one of NB equal-width buckets between the frame's min / max depth (a monotone
//   bucket scatter    map: every key of bucket b precedes every key of bucket b + 1); counts, then (depth_bits << 32 | index)
//                     keys scattered into the bucket's range (raster_pre.hip: further workgroups of the scan launch).
//   bucket sort       per bucket (~200 keys): merge sort in registers + LDS -> rank q of every visible Gaussian in the
//                     packed (frame, depth, index) order: order[q] = Gaussian, rank_of[Gaussian] = q (rank_sort.hpp; the
//                     first blocks of the emit launch, beside the emission).
//   k_emit<RANK>      (raster_pre.hip) writes the Gaussian's 32-bit id instead of a 64-bit key into the tile's range, in arbitrary order.
//   k_tile_rank       per tile: the ranks of its entries (rank_of[id]) set bits of a P-bit bitmap in LDS; a popcount scan turns the bitmap
//                     into the sorted list; the entries' 32-byte records (rec_g, left by k_preprocess) are gathered through
//                     order[] and written in LIST order (point_list, ent_geo, ent_slot, segment descriptors).  No comparison,
//                     no log factor, no limit on the list length.
//
// The result is bit-identical to the per-tile sort (tests/test_gpu_raster.py runs both and compares every integer output with
// the oracle).  Buckets longer than one sort chunk (degenerate depth distributions: a plane facing the camera) take the
// chunk + global-merge path of sort_util.hpp: slower, same result.
#include "gom_internal.h"
#include "sort_util.hpp"
#include "rank_map.hpp"

namespace {

using namespace gom_sort;

using namespace gom_rank;   // BucketMap, bucket_map: rank_map.hpp

// ---- counts per (frame, bucket) ------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_depth_hist(int P, uint32_t nb, const float *__restrict__ depth, const int32_t *__restrict__ radii,
                                                    const uint32_t *__restrict__ minmax, int nblk, uint32_t *__restrict__ bucket_count,
                                                    const GomDevStatus *__restrict__ status) {
    extern __shared__ uint32_t s_cnt[];
    __shared__ uint32_t s_red[8];
    const int fr = blockIdx.y;
    for (uint32_t b = threadIdx.x; b < nb; b += 256) s_cnt[b] = 0;
    const BucketMap bm = bucket_map(minmax, fr, nblk, nb, s_red);
    const int il = blockIdx.x * 256 + threadIdx.x;
    const size_t i = (size_t)fr * P + il;
    if (il < P && radii[i] > 0) atomicAdd(&s_cnt[bm(depth[i])], 1u);
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < nb; b += 256) {
        const uint32_t c = s_cnt[b];
        if (c) atomicAdd(&bucket_count[(size_t)fr * nb + b], c);
    }
}

// ---- per tile: bitmap of ranks -> sorted list -> records in list order -------------------------------------------------
// Work item = (non-empty tile, window of GOM_RANK_WIN consecutive list positions), listed by the scan kernel: four fifths of a
// body frame's tiles are empty (a workgroup per tile that only finds that out cost more than the work), and a 5 000-entry list
// becomes three items of the same size as everybody else's instead of one long chain.  Every item of a tile rebuilds the tile's
// whole bitmap (4 bytes per entry from L2) and writes only its own window.  One resident grid strides over the items.
template <int NT>
__global__ void __launch_bounds__(NT) k_tile_rank(int gx, int gy, uint32_t nb, const uint32_t *__restrict__ tile_base, const uint32_t *__restrict__ seg_base,
                                                  const uint32_t *__restrict__ work, const uint32_t *__restrict__ n_work_p,
                                                  const uint32_t *__restrict__ keys32, const uint32_t *__restrict__ bucket_base,
                                                  const uint32_t *__restrict__ order, const float4 *__restrict__ rec_g,
                                                  uint32_t *__restrict__ point_list, uint4 *__restrict__ seg_desc,
                                                  uint32_t *__restrict__ ent_slot, float2 *__restrict__ ent_geo,
                                                  const GomDevStatus *__restrict__ status, uint32_t seg_shift, uint32_t bm_words, uint32_t *__restrict__ seg_cost, const uint32_t *__restrict__ rank_of) {
    extern __shared__ uint32_t s_mem[];
    __shared__ uint32_t s_wsum[NT / 64];
    uint32_t *s_bm = s_mem, *s_stage = s_mem + bm_words;
    if (status->overflow) return;
    const uint32_t n_work = *n_work_p;
    const uint32_t t = threadIdx.x, lane = t & 63u, wid = t >> 6;
    for (uint32_t wi = blockIdx.x; wi < n_work; wi += gridDim.x) {
        const uint32_t item = work[wi];
        const int tile = (int)(item & 0xffffffu);
        const uint32_t c0 = (item >> 24) * GOM_RANK_WIN;
        const uint32_t base = tile_base[tile];
        const uint32_t n = tile_base[tile + 1] - base;
        const uint32_t c1 = min(n, c0 + GOM_RANK_WIN);
        if (c0 == 0) {
            const uint32_t sb = seg_base[tile], nseg = seg_base[tile + 1] - sb;
            for (uint32_t i = t; i < nseg; i += NT)
            {
                seg_desc[sb + i] = make_uint4((uint32_t)tile, base + (i << seg_shift), min(1u << seg_shift, n - (i << seg_shift)), i);
                if (seg_cost) {   // [sub-range][quadrant] survivor counts, filled by the pieces of k_seg_fwd that find pixels alive
                    uint4 *c = reinterpret_cast<uint4 *>(seg_cost + 16 * (size_t)(sb + i));
                    c[0] = c[1] = c[2] = c[3] = make_uint4(0u, 0u, 0u, 0u);
                }
            }
        }
        const int fr = tile / (gx * gy);
        const int tx = tile % gx, ty = tile / gx;                 // ty: row in the STACKED grid (rects carry the same offset)
        const uint32_t fs = bucket_base[(size_t)fr * nb];         // first packed rank of this frame
        const uint32_t nvis = bucket_base[(size_t)(fr + 1) * nb] - fs;
        const uint32_t W = (nvis + 31u) >> 5;                     // (<= bm_words: at most P visible Gaussians per frame)
        for (uint32_t w = t; w < W; w += NT) s_bm[w] = 0u;
        __syncthreads();
        for (uint32_t i0 = t; i0 < n; i0 += 8 * NT) {            // 8 independent loads in flight per thread
            uint32_t r[8];
#pragma unroll
            for (int u = 0; u < 8; u++) r[u] = i0 + u * NT < n ? keys32[base + i0 + u * NT] : 0xffffffffu;   // Gaussian ids (the emission ran beside the bucket sorts)
#pragma unroll
            for (int u = 0; u < 8; u++) r[u] = r[u] != 0xffffffffu ? rank_of[r[u]] - fs : 0xffffffffu;
#pragma unroll
            for (int u = 0; u < 8; u++)
                if (r[u] != 0xffffffffu) atomicOr(&s_bm[r[u] >> 5], 1u << (r[u] & 31u));   // ranks of one frame are unique, a Gaussian is in a tile list once
        }
        __syncthreads();
        // thread t owns the words [w0, w1): how many set bits precede them
        const uint32_t wpt = (W + NT - 1u) / NT;
        const uint32_t w0 = min(W, t * wpt), w1 = min(W, w0 + wpt);
        uint32_t mine = 0;
        for (uint32_t w = w0; w < w1; w++) mine += __popc(s_bm[w]);
        uint32_t incl = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t y = __shfl_up(incl, d, 64);
            if (lane >= (uint32_t)d) incl += y;
        }
        if (lane == 63) s_wsum[wid] = incl;
        __syncthreads();
        uint32_t first = incl - mine;
        for (uint32_t w = 0; w < wid; w++) first += s_wsum[w];
        if (first < c1 && first + mine > c0) {   // some of my bits fall into this item's window
            uint32_t pos = first;
            for (uint32_t w = w0; w < w1 && pos < c1; w++) {
                uint32_t bits = s_bm[w];
                while (bits) {
                    const uint32_t bit = __ffs(bits) - 1u;
                    bits &= bits - 1u;
                    if (pos >= c0 && pos < c1) s_stage[pos - c0] = (w << 5) | bit;
                    pos++;
                }
            }
        }
        __syncthreads();
        const uint32_t cnt = c1 - c0;
        for (uint32_t i0 = t; i0 < cnt; i0 += 4 * NT) {       // 4 entries per trip: their record gathers are issued before the first store
            float4 r0[4], r1[4];
            uint32_t g4[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t i = i0 + u * NT;
                g4[u] = order[fs + s_stage[i < cnt ? i : i0]];
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const float4 *rec = rec_g + 2 * (size_t)g4[u];
                r0[u] = rec[0]; r1[u] = rec[1];
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t i = i0 + u * NT;
                if (i < cnt) {
                    const uint32_t li = base + c0 + i;
                    const uint32_t po = __float_as_uint(r1[u].z), pk = __float_as_uint(r1[u].w);
                    const uint32_t rx0 = pk & 1023u, rw = (pk >> 10) & 1023u, ry0 = pk >> 20;
                    const uint32_t k = ((uint32_t)ty - ry0) * rw + ((uint32_t)tx - rx0);
                    point_list[li] = g4[u];
                    ent_slot[li] = po + k;
                    float2 *dst = ent_geo + 3 * (size_t)li;
                    dst[0] = make_float2(r0[u].x, r0[u].y); dst[1] = make_float2(r0[u].z, r0[u].w); dst[2] = make_float2(r1[u].x, r1[u].y);
                }
            }
        }
        __syncthreads();   // stage and bitmap are re-used by the next item
    }
}

// keys (depth_bits << 32 | gaussian) of the sorted lists, for gom_state_export: the ranking path never materialises them
__global__ void __launch_bounds__(256) k_rebuild_keys(const uint32_t *__restrict__ point_list, const float *__restrict__ depth, uint64_t *__restrict__ keys,
                                                      const GomDevStatus *__restrict__ status) {
    if (status->overflow) return;
    const uint32_t D = status->num_pairs;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < D; i += gridDim.x * 256) {
        const uint32_t g = point_list[i];
        keys[i] = ((uint64_t)__float_as_uint(depth[g]) << 32) | g;
    }
}

}  // namespace

int gom_launch_depth_hist(GomState *s, int P, hipStream_t st) {
    const int blocks = (P + 255) / 256;
    if (blocks == 0) return 0;
    GomKernelTimer timer(s, GOM_K_DEPTH_HIST, st);
    const uint32_t nb = 1u << s->nbShift;
    hipLaunchKernelGGL(k_depth_hist, dim3(blocks, s->B), dim3(256), nb * sizeof(uint32_t), st, P, nb, s->depth, s->radii, s->rank_minmax, s->rank_blocks, s->bucket_count,
                       s->status);
    GOM_LAUNCH_CHECK();
    return 0;
}

int gom_launch_tile_rank(GomState *s, hipStream_t st) {
    const int n_tiles = s->gx * s->gy * s->B;
    if (n_tiles == 0) return 0;
    GomKernelTimer timer(s, GOM_K_SORT, st);
    const uint32_t nb = 1u << s->nbShift;
    const uint32_t bm_words = ((uint32_t)s->P + 31u) >> 5;
    const size_t lds = (bm_words + GOM_RANK_WIN) * sizeof(uint32_t);
    const int cap_items = n_tiles + (int)(s->capPairs / GOM_RANK_WIN < 0x7fffffff ? s->capPairs / GOM_RANK_WIN : 0x7fffffff);
    const int grid = cap_items < 2048 ? cap_items : 2048;   // a resident grid striding over the work items of the scan kernel
    // One frame: 16-wave workgroups.  A single frame's tile pass is the latency chain of its longest list (5 000 entries on a body: every work item of the tile rebuilds the
    // tile's whole bitmap), and four times the threads walk it in a quarter of the trips: 26.7 -> 16.1 us, 197 -> 186 us per frame (5.07 -> 5.37 k frames/s at B = 1; round 6).
    // A batched launch has items to spare and keeps 4-wave workgroups (8 frames: 256 / 512 / 1024 threads = 0.512 / 0.515 / 0.530 ms per step).
    static const bool wide_b1 = !(getenv("GOM_RANK_NT_B1") && atoi(getenv("GOM_RANK_NT_B1")) == 256);   // (development switch: 256 = the 4-wave workgroups for one frame too)
    if (s->B == 1 && wide_b1) {
        hipLaunchKernelGGL((k_tile_rank<1024>), dim3(grid), dim3(1024), lds, st, s->gx, s->gy, nb, s->tile_base, s->seg_base, s->work_items, &s->status->n_work_items,
                           s->keys32, s->bucket_base, s->order, s->rec_g, s->point_list, s->seg_desc, s->ent_slot, s->ent_geo, s->status, (uint32_t)s->segShift, bm_words, s->seg_cost, s->rank_of);
        GOM_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL((k_tile_rank<256>), dim3(grid), dim3(256), lds, st, s->gx, s->gy, nb, s->tile_base, s->seg_base, s->work_items, &s->status->n_work_items,
                       s->keys32, s->bucket_base, s->order, s->rec_g, s->point_list, s->seg_desc, s->ent_slot, s->ent_geo, s->status, (uint32_t)s->segShift, bm_words, s->seg_cost, s->rank_of);
    GOM_LAUNCH_CHECK();
    return 0;
}

int gom_launch_rebuild_keys(GomState *s, hipStream_t st) {
    hipLaunchKernelGGL(k_rebuild_keys, dim3(1024), dim3(256), 0, st, s->point_list, s->depth, s->keys, s->status);
    GOM_LAUNCH_CHECK();
    return 0;
}
