// The shadow MLP on the bf16 matrix cores at fp32 precision (round 4; models/modules/shadow_module.py:66-117 at its default shape:
// D0 = 39 -> 128 -> 128 -> 128 -> 1, ReLU, sigmoid).  csrc/mlp.hip runs the layers as packed fp32 FMAs: 35 TFLOP/s, 49 us forward and
// 63 us backward per frame of the drop-in Model's iteration -- the VALU peak is 157.  Here every operand is TWO bf16 planes, hi = bf16(v)
// and lo = bf16(v - hi), and a product is three v_mfma_f32_16x16x32_bf16 -- (w_hi, x_hi), (w_hi, x_lo), (w_lo, x_hi), fp32 accumulation;
// the dropped w_lo x_lo term is 2^-16 of a product -- exactly the "bf16x3" scheme of the LPIPS trunk (vgg_bf16.hip).
//
// A workgroup takes 64 rows through all layers.  Activations live in LDS as two bf16 planes [row][k] (row stride 272 bytes: the sixteen rows
// of a fragment read land in sixteen different 16-byte bank groups); the weights are packed once per call into bf16 planes (k_mlp_pack: both
// orientations, 0.3 MB) and read straight from global memory into the A fragments (they stay in L2: every workgroup reads the same 0.3 MB).
// Wave w owns output channels [32 w, 32 w + 32) of a layer for all 64 rows: 4 x 2 accumulator tiles, 24 MFMA per 32-wide k-step.
// D[i = channel][j = row]: a lane ends with 4 consecutive channels of one row -- one float4 store of the fp32 activation (kept for the
// backward and the weight gradients, which stay on csrc/mlp.hip's fp32 kernels: they are streams over the rows, HBM-bound) and one
// 8-byte store per plane into LDS for the next layer.
#include "gom_internal.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef unsigned short bf16_t;

constexpr int kR = 64;            // rows per workgroup
constexpr int kH = 128;           // hidden width
constexpr int kLd = 136;          // LDS row stride in bf16 elements (272 bytes)
constexpr int kK1 = 64;           // layer 1's reduction length, padded (D0 <= 64)

__device__ __forceinline__ bf16_t f2bf(float f) {  // round to nearest even
    uint32_t u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}
__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float((uint32_t)h << 16); }
__device__ __forceinline__ void split_bf(float v, bf16_t &hi, bf16_t &lo) { hi = f2bf(v); lo = f2bf(v - bf2f(hi)); }

// packed weights (bf16 elements): per matrix two planes [hi][lo], rows of K elements
//   A1 [128][64]  = W1[o][k] (k < D0, else 0)          forward layer 1
//   A2, A3 [128][128] = W[o][k]                          forward layers 2, 3
//   T3, T2 [128][128] = W[k][i] transposed (row i, k)   backward through layers 3, 2
//   T1 [64][128]  = W1[k][i] transposed (row i < D0)    backward through layer 1
constexpr size_t kOffA1 = 0, kOffA2 = kOffA1 + 2 * 128 * 64, kOffA3 = kOffA2 + 2 * 128 * 128, kOffT3 = kOffA3 + 2 * 128 * 128, kOffT2 = kOffT3 + 2 * 128 * 128,
                 kOffT1 = kOffT2 + 2 * 128 * 128, kPackElems = kOffT1 + 2 * 64 * 128;

__global__ void __launch_bounds__(256) k_mlp_pack(int D0, const float *__restrict__ W1, const float *__restrict__ W2, const float *__restrict__ W3, bf16_t *__restrict__ pk) {
    const int idx = blockIdx.x * 256 + threadIdx.x;   // over 128 x 128
    if (idx >= 128 * 128) return;
    const int r = idx >> 7, c = idx & 127;
    auto put = [&](size_t off, int rows, int K, int rr, int kk, float v) {
        bf16_t hi, lo;
        split_bf(v, hi, lo);
        pk[off + (size_t)rr * K + kk] = hi;
        pk[off + (size_t)rows * K + (size_t)rr * K + kk] = lo;
    };
    put(kOffA2, 128, 128, r, c, W2[r * 128 + c]);
    put(kOffA3, 128, 128, r, c, W3[r * 128 + c]);
    put(kOffT2, 128, 128, r, c, W2[c * 128 + r]);
    put(kOffT3, 128, 128, r, c, W3[c * 128 + r]);
    if (c < 64) put(kOffA1, 128, 64, r, c, c < D0 ? W1[r * D0 + c] : 0.f);
    if (r < 64) put(kOffT1, 64, 128, r, c, r < D0 ? W1[c * D0 + r] : 0.f);
}

// acc[m][n] += A (rows n0 + 16 n .. of `wa`, K-major) x B (the 64 rows in LDS), three bf16 passes per product.
// wa: plane 0 at wa, plane 1 at wa + rows_total * K.
// The B fragments of a k-step are requested from LDS a whole k-step AHEAD of the MFMAs that consume them (register double buffer, the loop fully unrolled), and the
// first k-step's before the weights' global loads are waited for.  This is not a latency measure: an MFMA that consumes a register an LDS read has only just written
// makes a packed-fp32 FMA of ANOTHER wave on the same SIMD lose a term (LABBOOK R6.8, scripts/ubench/pkfma_beside_mfma.hip: "reads one trip ahead" is the variant
// beside which every result is exact) -- the first version of this loop read, waited and multiplied, and corrupted whatever kernel ran beside it.
template <int NT, int K>
__device__ __forceinline__ void gemm_x3(const bf16_t *__restrict__ wa, int rows_total, int n0, const bf16_t *s_act /* [2][kR][kLd] */, f32x4 (&acc)[4][NT]) {
    const int lane = threadIdx.x & 63, l15 = lane & 15, kg = lane >> 4;
    const bf16_t *w_hi = wa + (size_t)(n0 + l15) * K + kg * 8, *w_lo = w_hi + (size_t)rows_total * K;
    constexpr int KS = K / 32;
    bf16x8 ah[2][NT], al[2][NT], bh[2][4], bl[2][4];
    auto read_b = [&](int ks, bf16x8 (&h)[4], bf16x8 (&l)[4]) {
#pragma unroll
        for (int m = 0; m < 4; m++) {
            h[m] = *reinterpret_cast<const bf16x8 *>(s_act + (size_t)(16 * m + l15) * kLd + ks * 32 + kg * 8);
            l[m] = *reinterpret_cast<const bf16x8 *>(s_act + (size_t)kR * kLd + (size_t)(16 * m + l15) * kLd + ks * 32 + kg * 8);
        }
    };
    auto load_a = [&](int ks, bf16x8 (&h)[NT], bf16x8 (&l)[NT]) {
#pragma unroll
        for (int n = 0; n < NT; n++) { h[n] = *reinterpret_cast<const bf16x8 *>(w_hi + (size_t)n * 16 * K + ks * 32); l[n] = *reinterpret_cast<const bf16x8 *>(w_lo + (size_t)n * 16 * K + ks * 32); }
    };
    read_b(0, bh[0], bl[0]);
    load_a(0, ah[0], al[0]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < KS; ks++) {
        const int cur = ks & 1, nxt = cur ^ 1;
        if (ks + 1 < KS) {   // the next k-step's fragments and weights are in flight during this one's MFMAs
            read_b(ks + 1, bh[nxt], bl[nxt]);
            load_a(ks + 1, ah[nxt], al[nxt]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (ks == 0) {   // the FIRST k-step's fragments cannot be a step ahead (they are the previous layer's output, behind a barrier): let them land, then idle 64 clocks
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_sleep 8" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int m = 0; m < 4; m++)
#pragma unroll
            for (int n = 0; n < NT; n++) {
                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[cur][n], bh[cur][m], acc[m][n], 0, 0, 0);
                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[cur][n], bl[cur][m], acc[m][n], 0, 0, 0);
                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[cur][n], bh[cur][m], acc[m][n], 0, 0, 0);
            }
        __builtin_amdgcn_sched_barrier(0);
    }
}
// four consecutive channels of one row into both LDS planes
__device__ __forceinline__ void act_to_lds(bf16_t *s_act, int row, int ch, const float (&v)[4]) {
    bf16_t h[4], l[4];
#pragma unroll
    for (int r = 0; r < 4; r++) split_bf(v[r], h[r], l[r]);
    *reinterpret_cast<uint2 *>(s_act + (size_t)row * kLd + ch) = make_uint2((uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16));
    *reinterpret_cast<uint2 *>(s_act + (size_t)kR * kLd + (size_t)row * kLd + ch) = make_uint2((uint32_t)l[0] | ((uint32_t)l[1] << 16), (uint32_t)l[2] | ((uint32_t)l[3] << 16));
}
template <int NT>
__device__ __forceinline__ void zero_acc(f32x4 (&acc)[4][NT]) {
#pragma unroll
    for (int m = 0; m < 4; m++)
#pragma unroll
        for (int n = 0; n < NT; n++) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
}

__global__ void __launch_bounds__(256, 2) k_mlp3_fwd_mc(int64_t n, int D0, const float *__restrict__ x, const bf16_t *__restrict__ pk, const float *__restrict__ b1,
                                                         const float *__restrict__ b2, const float *__restrict__ b3, const float *__restrict__ w4,
                                                         const float *__restrict__ b4, float *__restrict__ h1, float *__restrict__ h2, float *__restrict__ h3,
                                                         float *__restrict__ out, const int32_t *__restrict__ n_dev) {
    __shared__ __attribute__((aligned(16))) bf16_t s_act[2 * kR * kLd];
    __shared__ float s_red[4][kR];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l15 = lane & 15, kg = lane >> 4;
    const int64_t r0 = (int64_t)blockIdx.x * kR;
    if (n_dev) n = (int64_t)n_dev[0] + 1;
    if (r0 >= n) return;
    const int rows = (int)min<int64_t>(kR, n - r0);
    {   // the rows' inputs -> LDS planes, zero beyond D0 and beyond the last row: thread = (row, 16 columns)
        const int row = tid >> 2, c0 = (tid & 3) * 16;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; r++) { const int c = c0 + 4 * q + r; v[r] = (row < rows && c < D0) ? x[(r0 + row) * D0 + c] : 0.f; }
            act_to_lds(s_act, row, c0 + 4 * q, v);
        }
    }
    __syncthreads();
    float *hs[3] = {h1, h2, h3};
    const float *bs[3] = {b1, b2, b3};
    const size_t offs[3] = {kOffA1, kOffA2, kOffA3};
    f32x4 acc[4][2];
    float part[4] = {0.f, 0.f, 0.f, 0.f};   // layer 4: this lane's share of w4 . h3 per row tile
#pragma unroll
    for (int l = 0; l < 3; l++) {
        zero_acc<2>(acc);
        if (l == 0) gemm_x3<2, kK1>(pk + offs[l], kH, 32 * wv, s_act, acc);
        else gemm_x3<2, kH>(pk + offs[l], kH, 32 * wv, s_act, acc);
        __syncthreads();   // every wave has read the layer's input
#pragma unroll
        for (int m = 0; m < 4; m++) {
            const int row = 16 * m + l15;
#pragma unroll
            for (int nn = 0; nn < 2; nn++) {
                const int ch = 32 * wv + 16 * nn + 4 * kg;
                const float4 bb = *reinterpret_cast<const float4 *>(bs[l] + ch);
                float v[4] = {fmaxf(acc[m][nn][0] + bb.x, 0.f), fmaxf(acc[m][nn][1] + bb.y, 0.f), fmaxf(acc[m][nn][2] + bb.z, 0.f), fmaxf(acc[m][nn][3] + bb.w, 0.f)};
                if (row < rows) *reinterpret_cast<float4 *>(hs[l] + (size_t)(r0 + row) * kH + ch) = make_float4(v[0], v[1], v[2], v[3]);
                if (l < 2) act_to_lds(s_act, row, ch, v);
                else {
                    const float4 ww = *reinterpret_cast<const float4 *>(w4 + ch);
                    part[m] += v[0] * ww.x + v[1] * ww.y + v[2] * ww.z + v[3] * ww.w;
                }
            }
        }
        __syncthreads();
    }
    // layer 4: sum the lanes' shares over the four k-groups (lanes l15 + 16 kg), then over the waves
#pragma unroll
    for (int m = 0; m < 4; m++) {
        float p = part[m];
        p += __shfl_xor(p, 16, 64);
        p += __shfl_xor(p, 32, 64);
        if (kg == 0) s_red[wv][16 * m + l15] = p;
    }
    __syncthreads();
    if (tid < rows) {
        const float z = ((s_red[0][tid] + s_red[1][tid]) + (s_red[2][tid] + s_red[3][tid])) + b4[0];
        out[r0 + tid] = 1.f / (1.f + __expf(-z));
    }
}

// g [n] = dL/d out  ->  dz4 [n], dz3 / dz2 / dz1 [n][128], dx [n][D0]
__global__ void __launch_bounds__(256, 2) k_mlp3_bwd_mc(int64_t n, int D0, const float *__restrict__ g, const float *__restrict__ out, const float *__restrict__ h1,
                                                         const float *__restrict__ h2, const float *__restrict__ h3, const bf16_t *__restrict__ pk,
                                                         const float *__restrict__ w4, float *__restrict__ dz4, float *__restrict__ dz3, float *__restrict__ dz2,
                                                         float *__restrict__ dz1, float *__restrict__ dx, const int32_t *__restrict__ n_dev) {
    __shared__ __attribute__((aligned(16))) bf16_t s_act[2 * kR * kLd];
    __shared__ float s_d4[kR];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l15 = lane & 15, kg = lane >> 4;
    const int64_t r0 = (int64_t)blockIdx.x * kR;
    if (n_dev) n = (int64_t)n_dev[0] + 1;
    if (r0 >= n) return;
    const int rows = (int)min<int64_t>(kR, n - r0);
    if (tid < kR) {
        float d = 0.f;
        if (tid < rows) { const float o = out[r0 + tid]; d = g[r0 + tid] * o * (1.f - o); dz4[r0 + tid] = d; }
        s_d4[tid] = d;
    }
    __syncthreads();
    {   // dz3 = dz4 w4^T (.) [h3 > 0]: thread = (row, 32 channels)
        const int row = tid >> 2, c0 = (tid & 3) * 32;
        const float d = s_d4[row];
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int ch = c0 + 4 * q;
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            if (row < rows) {
                const float4 hh = *reinterpret_cast<const float4 *>(h3 + (size_t)(r0 + row) * kH + ch), ww = *reinterpret_cast<const float4 *>(w4 + ch);
                v[0] = hh.x > 0.f ? d * ww.x : 0.f; v[1] = hh.y > 0.f ? d * ww.y : 0.f; v[2] = hh.z > 0.f ? d * ww.z : 0.f; v[3] = hh.w > 0.f ? d * ww.w : 0.f;
                *reinterpret_cast<float4 *>(dz3 + (size_t)(r0 + row) * kH + ch) = make_float4(v[0], v[1], v[2], v[3]);
            }
            act_to_lds(s_act, row, ch, v);
        }
    }
    __syncthreads();
    const float *hm[2] = {h2, h1};
    float *dzo[2] = {dz2, dz1};
    const size_t offs[2] = {kOffT3, kOffT2};
#pragma unroll
    for (int l = 0; l < 2; l++) {   // dz2 = (dz3 W3) (.) [h2 > 0];  dz1 = (dz2 W2) (.) [h1 > 0]
        f32x4 acc[4][2];
        zero_acc<2>(acc);
        gemm_x3<2, kH>(pk + offs[l], kH, 32 * wv, s_act, acc);
        __syncthreads();
#pragma unroll
        for (int m = 0; m < 4; m++) {
            const int row = 16 * m + l15;
#pragma unroll
            for (int nn = 0; nn < 2; nn++) {
                const int ch = 32 * wv + 16 * nn + 4 * kg;
                float v[4] = {0.f, 0.f, 0.f, 0.f};
                if (row < rows) {
                    const float4 hh = *reinterpret_cast<const float4 *>(hm[l] + (size_t)(r0 + row) * kH + ch);
                    v[0] = hh.x > 0.f ? acc[m][nn][0] : 0.f; v[1] = hh.y > 0.f ? acc[m][nn][1] : 0.f; v[2] = hh.z > 0.f ? acc[m][nn][2] : 0.f; v[3] = hh.w > 0.f ? acc[m][nn][3] : 0.f;
                    *reinterpret_cast<float4 *>(dzo[l] + (size_t)(r0 + row) * kH + ch) = make_float4(v[0], v[1], v[2], v[3]);
                }
                act_to_lds(s_act, row, ch, v);
            }
        }
        __syncthreads();
    }
    {   // dx = dz1 W1: 64 (padded) input columns, wave w takes columns [16 w, 16 w + 16)
        f32x4 acc[4][1];
        zero_acc<1>(acc);
        gemm_x3<1, kH>(pk + kOffT1, 64, 16 * wv, s_act, acc);
#pragma unroll
        for (int m = 0; m < 4; m++) {
            const int row = 16 * m + l15;
            if (row >= rows) continue;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int c = 16 * wv + 4 * kg + r;
                if (c < D0) dx[(size_t)(r0 + row) * D0 + c] = acc[m][0][r];
            }
        }
    }
}

}  // namespace

// ---- called from csrc/mlp.hip's entry points when the caller hands in a pack buffer -----------------------------------------------
size_t gom_mlp3_mc_pack_elems(void) { return kPackElems; }
bool gom_mlp3_mc_supported(int D0, int H) { return H == kH && D0 >= 1 && D0 <= kK1; }

int gom_mlp3_mc_forward(int64_t n, int D0, const float *x, const float *W1, const float *b1, const float *W2, const float *b2, const float *W3, const float *b3,
                        const float *w4, const float *b4, float *h1, float *h2, float *h3, float *out, const int32_t *n_dev, uint16_t *pack, void *stream) {
    hipLaunchKernelGGL(k_mlp_pack, dim3(64), dim3(256), 0, (hipStream_t)stream, D0, W1, W2, W3, pack);
    GOM_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_mlp3_fwd_mc, dim3((unsigned)((n + kR - 1) / kR)), dim3(256), 0, (hipStream_t)stream, n, D0, x, pack, b1, b2, b3, w4, b4, h1, h2, h3, out, n_dev);
    GOM_LAUNCH_CHECK();
    return 0;
}
int gom_mlp3_mc_backward(int64_t n, int D0, const float *g, const float *out, const float *h1, const float *h2, const float *h3, const float *W1, const float *W2,
                         const float *W3, const float *w4, float *dz4, float *dz3, float *dz2, float *dz1, float *dx, const int32_t *n_dev, uint16_t *pack,
                         void *stream) {
    hipLaunchKernelGGL(k_mlp_pack, dim3(64), dim3(256), 0, (hipStream_t)stream, D0, W1, W2, W3, pack);
    GOM_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_mlp3_bwd_mc, dim3((unsigned)((n + kR - 1) / kR)), dim3(256), 0, (hipStream_t)stream, n, D0, g, out, h1, h2, h3, pack, w4, dz4, dz3, dz2, dz1, dx,
                       n_dev);
    GOM_LAUNCH_CHECK();
    return 0;
}
