// The per-bucket sort of the depth ranking (raster_rank.hip) as a device function: it runs as rider blocks of the emit launch
// (raster_pre.hip), one workgroup per (frame, bucket).
#pragma once
#include "sort_util.hpp"

namespace gom_rank {

// order[q] = global Gaussian id at packed rank q, rank_of[g] = q.  Nothing else moves here: the tile pass fetches a Gaussian's
// 32-byte record (rec_g, written by k_preprocess in Gaussian order) through order[] -- the entries of one tile are neighbours on
// the mesh, so their records are neighbours in memory, where a copy in rank order (first version: 113 MB of sector traffic to
// build it, 30 us) scattered them by depth.  A bucket is ~200 keys: 8 keys per thread, most of the workgroup idle.
// s_x: 8 * NT keys of LDS, 16-byte aligned.
template <int NT>
__device__ __forceinline__ void bucket_sort_block(int fr, uint32_t bucket, int P, uint32_t nb, const uint32_t *__restrict__ bucket_base,
                                                  uint64_t *__restrict__ bkeys, uint64_t *__restrict__ scratch, uint32_t *__restrict__ order,
                                                  uint32_t *__restrict__ rank_of, uint32_t log_chunk, uint64_t *s_x) {
    const uint32_t base = bucket_base[(size_t)fr * nb + bucket];
    const uint32_t n = bucket_base[(size_t)fr * nb + bucket + 1] - base;
    if (n == 0) return;
    const uint32_t go = (uint32_t)fr * (uint32_t)P;
    uint64_t x[8];
    bool in_regs;
    const uint64_t *sorted = gom_sort::block_sort_any<NT>(bkeys + base, scratch + base, n, s_x, log_chunk, x, in_regs);
    if (in_regs) {   // blocked registers: thread t holds positions 8t .. 8t+7
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const uint32_t i = 8 * threadIdx.x + r;
            if (i < n) {
                const uint32_t g = go + (uint32_t)x[r];
                order[base + i] = g;
                rank_of[g] = base + i;
            }
        }
        return;
    }
    for (uint32_t i = threadIdx.x; i < n; i += NT) {
        const uint32_t g = go + (uint32_t)sorted[i];
        order[base + i] = g;
        rank_of[g] = base + i;
    }
}

}  // namespace gom_rank
