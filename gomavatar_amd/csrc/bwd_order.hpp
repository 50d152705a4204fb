// The riders that order the render backward's task queue (GomBwdOrderRider, gom_internal.h): eight workgroups of 256 threads, one per
// queue shard, in front of the tile workgroups of k_combine_fwd -- the launch behind k_seg_fwd, which counts the costs.
#pragma once
#include "gom_internal.h"

// Rider x orders the tasks of queue shard x (the segments with seg mod 8 = x): a counting sort over 512 cost levels, most
// expensive first; a pair of sub-ranges above GOM_BWD_SPLIT_COST becomes two single-sub-range tasks.
__device__ __forceinline__ void gom_bwd_order_rider(const GomBwdOrderRider &rider, const uint32_t x) {
    if (rider.status->overflow) return;
    __shared__ uint32_t s_lvl[512];
    const uint32_t nsegs = rider.status->num_segs;
    const uint32_t npairs = 2u * gom_shard_segments(nsegs, x);                   // (segment, pair) units of this shard
    const uint32_t region = gom_bwd_order_region(nsegs);
    uint32_t *out = rider.bwd_order + GOM_BWD_ORDER_BASE + (size_t)x * region;
    for (int k = threadIdx.x; k < 512; k += 256) s_lvl[k] = 0u;
    __syncthreads();
    auto level = [](uint32_t c) { return 511u - min(c, 511u); };                  // level 0 = the most expensive
    // what a (segment, pair) unit costs its workgroup: wave w takes quadrant w of the first sub-range, then quadrant 3 - w of the second
    // (k_seg_bwd_pair), and the task lasts as long as its busiest wave; a sub-range alone: its busiest quadrant.
    // -> (.x, .y) = the two single sub-ranges' costs, or (cost of the pair, 0xfffffffe) when it is not split; .x = 0xffffffff: no unit
    auto cost_of = [&](uint32_t j) {
        const uint32_t seg = gom_shard_segment(x, j >> 1);
        uint2 c = make_uint2(0xffffffffu, 0u);
        if (j < npairs && seg < nsegs) {
            const uint4 a = *reinterpret_cast<const uint4 *>(rider.seg_cost + 16 * (size_t)seg + 8 * (j & 1u));
            const uint4 b = *reinterpret_cast<const uint4 *>(rider.seg_cost + 16 * (size_t)seg + 8 * (j & 1u) + 4);
            const uint32_t both = max(max(a.x + b.w, a.y + b.z), max(a.z + b.y, a.w + b.x));
            c = make_uint2(max(max(a.x, a.y), max(a.z, a.w)), max(max(b.x, b.y), max(b.z, b.w)));
            if (both <= GOM_BWD_SPLIT_COST) c = make_uint2(both, 0xfffffffeu);
        }
        return c;
    };
    auto account = [&](int pass, uint32_t j, uint2 c) {
        if (c.x == 0xffffffffu) return;
        const uint32_t seg = gom_shard_segment(x, j >> 1), pair = j & 1u;
        if (c.y != 0xfffffffeu) {
            if (pass == 0) { atomicAdd(&s_lvl[level(c.x)], 1u); atomicAdd(&s_lvl[level(c.y)], 1u); }
            else {
                out[atomicAdd(&s_lvl[level(c.x)], 1u)] = (seg << 3) | (4u + 2u * pair);
                out[atomicAdd(&s_lvl[level(c.y)], 1u)] = (seg << 3) | (5u + 2u * pair);
            }
        } else {
            if (pass == 0) atomicAdd(&s_lvl[level(c.x)], 1u);
            else out[atomicAdd(&s_lvl[level(c.x)], 1u)] = (seg << 3) | pair;    // (order inside a level: any)
        }
    };
    // Up to 8 units per thread (2 048 per shard: the bench's 8-frame step has ~1 700) stay in registers -- ONE round of 16 loads per
    // thread in front of both passes instead of eight dependent rounds: the loss launch lasts as long as its riders do.
    const bool in_regs = npairs <= 8u * 256u;
    uint2 creg[8];
    if (in_regs) {
#pragma unroll
        for (int u = 0; u < 8; u++) creg[u] = cost_of(threadIdx.x + u * 256u);
    }
    for (int pass = 0; pass < 2; pass++) {
        if (in_regs) {
#pragma unroll
            for (int u = 0; u < 8; u++) account(pass, threadIdx.x + u * 256u, creg[u]);
        } else {
            for (uint32_t j0 = threadIdx.x; j0 < npairs; j0 += 2 * 256) {            // 2 units = 4 independent 16-byte loads in flight per thread
                uint2 c[2];
#pragma unroll
                for (int u = 0; u < 2; u++) c[u] = cost_of(j0 + u * 256);
#pragma unroll
                for (int u = 0; u < 2; u++) account(pass, j0 + u * 256, c[u]);
            }
        }
        __syncthreads();
        if (pass == 0) {
            if (threadIdx.x < 64) {   // exclusive scan of the 512 level counts: 8 per lane + a wave scan
                uint32_t c8[8], tot = 0;
#pragma unroll
                for (int k = 0; k < 8; k++) { c8[k] = s_lvl[8 * threadIdx.x + k]; tot += c8[k]; }
                uint32_t y = tot;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) { const uint32_t z = __shfl_up(y, d, 64); if ((int)threadIdx.x >= d) y += z; }
                uint32_t run = y - tot;
#pragma unroll
                for (int k = 0; k < 8; k++) { s_lvl[8 * threadIdx.x + k] = run; run += c8[k]; }
                if (threadIdx.x == 63) rider.bwd_order[x] = run;   // tasks of this shard
            }
            __syncthreads();
        }
    }
}
