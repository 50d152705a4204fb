// Block-level merge sort of unique 64-bit keys in registers + LDS (shared by the per-tile sort of raster_render.hip and the
// per-bucket depth sort of raster_rank.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gom_sort {

// Per-tile merge sort of the unique 64-bit keys (depth_bits << 32 | gaussian): identical to the reference's stable
// radix order on (tile, depth bits) because ties in depth fall back to the Gaussian index.
//
// Every thread owns 8 consecutive list positions.  It sorts its 8 keys in registers (19 compare-exchanges), then
// log2(n/8) merge levels follow: the sorted runs of length L sit in LDS, each thread finds by binary search
// ("merge path") where its 8 outputs start in the two runs being merged and merges 8 elements sequentially.
// O(n log n) work instead of the O(n log^2 n) of a bitonic network -- 4-5x fewer instructions at n = 2048..8192,
// which matters because a batched launch is VALU-bound here (scripts/ubench/dpp_bench.hip: ~2.6 cycles per wave
// instruction per SIMD at best).
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

// normalised bitonic network on 8 registers (every comparator ascending)
__device__ __forceinline__ void sort8_regs(uint64_t (&x)[8]) {
    sort_step_regs<1>(x);
    sort_step_regs<3>(x); sort_step_regs<1>(x);
    sort_step_regs<7>(x); sort_step_regs<2>(x); sort_step_regs<1>(x);
}

// Outputs [o, o+8) of the merge of the sorted runs A = src[0, la) and B = src[L, L + lb)  (la, lb = real lengths;
// positions past la + lb yield +inf).  PTR: LDS or global pointer to uint64_t.
template <typename PTR>
__device__ __forceinline__ void merge8(PTR src, uint32_t L, uint32_t la, uint32_t lb, uint32_t o, uint64_t (&out)[8]) {
    const uint64_t INF = ~0ull;
    if (o >= la + lb) {
#pragma unroll
        for (int k = 0; k < 8; k++) out[k] = INF;
        return;
    }
    // merge path: i = how many of the first o outputs come from A
    uint32_t lo = o > lb ? o - lb : 0u, hi = o < la ? o : la;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        const uint64_t a = src[mid], b = src[L + (o - 1 - mid)];
        if (a < b) lo = mid + 1; else hi = mid;
    }
    uint32_t i = lo, j = o - lo;
    uint64_t a = i < la ? src[i] : INF;
    uint64_t b = j < lb ? src[L + j] : INF;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const bool take = a <= b;   // unique keys; +inf only ever ties with +inf
        out[k] = take ? a : b;
        i += take ? 1u : 0u;
        j += take ? 0u : 1u;
        if (k < 7) {
            const bool ok = take ? (i < la) : (j < lb);
            const uint64_t v = ok ? src[take ? i : L + j] : INF;
            a = take ? v : a;
            b = take ? b : v;
        }
    }
}

// Sorts keys[0, n) (n <= 8 * NT) in place in registers + LDS; on return thread t holds positions 8t .. 8t+7 in x.
template <int NT>
__device__ __forceinline__ void block_merge_sort(const uint64_t *__restrict__ keys, uint32_t n, uint64_t *s_x, uint64_t (&x)[8]) {
    const uint32_t t = threadIdx.x;
    uint32_t n_pad = 8;
    while (n_pad < n) n_pad <<= 1;
    const bool active = 8 * t < n_pad;
    // coalesced (striped) global loads, then blocked ownership (8 consecutive positions per thread) through LDS
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const uint32_t i = (uint32_t)r * NT + t;
        if ((uint32_t)r * NT < n_pad) s_x[i] = i < n ? keys[i] : ~0ull;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 8; r++) x[r] = active ? s_x[8 * t + r] : ~0ull;
    if (active) sort8_regs(x);
    for (uint32_t L = 8; L < n_pad; L <<= 1) {
        __syncthreads();  // readers of the previous level are done
        if (active) {
#pragma unroll
            for (int r = 0; r < 8; r += 2) *reinterpret_cast<ulonglong2 *>(s_x + 8 * t + r) = make_ulonglong2(x[r], x[r + 1]);
        }
        __syncthreads();
        if (active) {
            const uint32_t pbase = (8 * t) & ~(2 * L - 1), o = 8 * t - pbase;
            const uint32_t la = n > pbase ? min(L, n - pbase) : 0u;
            const uint32_t lb = n > pbase + L ? min(L, n - pbase - L) : 0u;
            merge8(s_x + pbase, L, la, lb, o, x);
        }
    }
}


// Sorts keys[0, n) for ANY n with one workgroup of NT threads: chunks of CH = 2^log_chunk <= 8 NT keys in registers + LDS, then
// (n > CH only) the chunks are merged level by level in global memory, ping-ponging between `keys` and `scratch`.
// Returns the array that holds the sorted result (keys or scratch); for n <= CH the result is ALSO left blocked in x
// (thread t: positions 8t..8t+7) and `in_regs` is set.  Ends with a __syncthreads().
template <int NT>
__device__ __forceinline__ uint64_t *block_sort_any(uint64_t *__restrict__ keys, uint64_t *__restrict__ scratch, uint32_t n, uint64_t *s_x,
                                                    uint32_t log_chunk, uint64_t (&x)[8], bool &in_regs) {
    const uint32_t t = threadIdx.x, CH = 1u << log_chunk;
    in_regs = n <= CH;
    if (in_regs) {
        block_merge_sort<NT>(keys, n, s_x, x);
        __syncthreads();
        return keys;
    }
    for (uint32_t c0 = 0; c0 < n; c0 += CH) {
        const uint32_t cn = min(CH, n - c0);
        block_merge_sort<NT>(keys + c0, cn, s_x, x);
#pragma unroll
        for (int r = 0; r < 8; r++)
            if (8 * t + r < cn) keys[c0 + 8 * t + r] = x[r];
        __syncthreads();  // s_x is re-used by the next chunk
    }
    uint64_t *src = keys, *dst = scratch;
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
    __syncthreads();
    return src;
}

}  // namespace gom_sort
