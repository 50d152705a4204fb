// The monotone depth -> bucket map of the depth ranking (raster_rank.hip), shared with the kernels that build its histogram
// (k_depth_hist; k_preprocess in the frame step).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gom_rank {

struct BucketMap {
    float dmin, scale;
    uint32_t nb;
    __device__ __forceinline__ uint32_t operator()(float d) const {
        const float v = (d - dmin) * scale;              // monotone in d (IEEE subtraction / multiplication by a constant >= 0)
        const uint32_t b = v > 0.f ? (uint32_t)v : 0u;   // (truncation is monotone; NaN cannot occur: scale is finite)
        return b < nb ? b : nb - 1u;
    }
};

// The frame's depth range from the per-block (min, max) pairs k_preprocess left (float BIT PATTERNS: depths are > 0.2, unsigned
// order = float order; a block without a visible Gaussian wrote min > max): every workgroup folds the ~200 pairs itself --
// 1.7 KB from L2 -- instead of a reduction kernel or contended atomics.  Ends with a __syncthreads().
__device__ __forceinline__ BucketMap bucket_map(const uint32_t *__restrict__ minmax, int fr, int nblk, uint32_t nb, uint32_t *s_red /* [2 * waves of the workgroup] */) {
    const uint2 *mm = reinterpret_cast<const uint2 *>(minmax) + (size_t)fr * nblk;
    uint32_t lo = 0xffffffffu, hi = 0u;
    for (int k = threadIdx.x; k < nblk; k += blockDim.x) {
        const uint2 v = mm[k];
        lo = min(lo, v.x);
        hi = max(hi, v.y);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        lo = min(lo, (uint32_t)__shfl_xor((int)lo, d, 64));
        hi = max(hi, (uint32_t)__shfl_xor((int)hi, d, 64));
    }
    const uint32_t nw = blockDim.x >> 6;
    if ((threadIdx.x & 63) == 0) { s_red[threadIdx.x >> 6] = lo; s_red[nw + (threadIdx.x >> 6)] = hi; }
    __syncthreads();
    lo = s_red[0]; hi = s_red[nw];
    for (uint32_t w = 1; w < nw; w++) { lo = min(lo, s_red[w]); hi = max(hi, s_red[nw + w]); }
    BucketMap m;
    m.nb = nb;
    m.dmin = __uint_as_float(lo);
    const float span = lo < hi ? __uint_as_float(hi) - __uint_as_float(lo) : 0.f;
    float sc = span > 0.f ? (float)nb / span : 0.f;
    if (!(sc < 1.0e30f)) sc = 0.f;   // a span of a few ulps: one bucket (still correct, the bucket sort orders it)
    m.scale = sc;
    return m;
}


}  // namespace gom_rank
