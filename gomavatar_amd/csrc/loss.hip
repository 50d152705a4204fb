// Fused photometric L1 losses with their backward, one pass over the image.
//
// Replaces unpack (train.py:53-55) + L1 rgb / L1 mask (train.py:101-111) and the
// autograd backward of those ~8 element-wise/reduction launches.  Reads the
// rasterizer's CHW output directly and writes dL/d(pred) in the same CHW layout
// the raster backward consumes.  HBM-bound: 4+1+3+1 floats read, 4(+1) written
// per pixel.  The loss VALUE is only needed for logging, so each block writes its
// partial sums and nobody waits on a same-address atomic (11-13 ns each on
// MI355X); the gradient does not depend on the value.
#include "gom_internal.h"
#include "l1_pixel.hpp"

namespace {

__device__ __forceinline__ float sgn(float x) { return gom_sgn(x); }

__global__ void __launch_bounds__(256) k_l1_loss(int HW, const float *__restrict__ pred, const float *__restrict__ shade,
                                                 const float *__restrict__ gt_rgb, const float *__restrict__ gt_mask,
                                                 const float *__restrict__ bg, float k_rgb, float k_mask,
                                                 float *__restrict__ dpred, float *__restrict__ dshade, float *__restrict__ partials, GomLossSkip skip, int slots) {
    __shared__ float s_red[2][4];
    {  // blockIdx.y = frame of a batched launch: [B][4][HW] images, [B][HW][3] targets, [B][3] backgrounds
        const size_t fr = blockIdx.y;
        pred += fr * 4 * HW; gt_rgb += fr * 3 * HW; gt_mask += fr * HW; bg += fr * 3; dpred += fr * 4 * HW;
        partials += fr * 2 * slots;
        if (shade) shade += fr * HW;
        if (dshade) dshade += fr * HW;
    }
    const float b0 = bg[0], b1 = bg[1], b2 = bg[2];
    float sum_rgb = 0.f, sum_mask = 0.f;
    // (GomLossSkip) a pixel of an empty tile: the rasterizer wrote its background there -- the same value, not loaded
    float e0 = skip.bg[0], e1 = skip.bg[1], e2 = skip.bg[2], e3 = skip.bg[3];
    const uint32_t *tb = nullptr;
    if (skip.tile_base) {
        tb = skip.tile_base + (size_t)blockIdx.y * skip.gx * skip.gy;
        if (skip.cams) { const float *cb = skip.cams[blockIdx.y].bg; e0 = cb[0]; e1 = cb[1]; e2 = cb[2]; e3 = cb[3]; }
    }
    for (int p = blockIdx.x * 256 + threadIdx.x; p < HW; p += GOM_LOSS_BLOCKS * 256) {
        bool empty = false;
        if (tb) {
            const int y = p / skip.W, x = p - y * skip.W, t = (y >> 4) * skip.gx + (x >> 4);
            empty = tb[t + 1] == tb[t];
        }
        const float m = empty ? e3 : pred[3 * (size_t)HW + p];
        const float s = shade ? shade[p] : 1.f;
        const float a0 = empty ? e0 : pred[p], a1 = empty ? e1 : pred[(size_t)HW + p], a2 = empty ? e2 : pred[2 * (size_t)HW + p];
        const float3 g = *reinterpret_cast<const float3 *>(gt_rgb + 3 * (size_t)p);   // one 12-byte load per lane (three strided 4-byte loads cost the texture path three passes)
        const GomL1Px o = gom_l1_pixel(a0, a1, a2, m, s, b0, b1, b2, g.x, g.y, g.z, gt_mask[p], k_rgb, k_mask);
        sum_rgb += o.abs_rgb;
        sum_mask += o.abs_mask;
        if (empty) {           // (no list entry touches the pixel: the backward does not read its gradient)
            if (skip.zero_empty) { dpred[p] = 0.f; dpred[(size_t)HW + p] = 0.f; dpred[2 * (size_t)HW + p] = 0.f; dpred[3 * (size_t)HW + p] = 0.f; }
            continue;
        }
        dpred[p] = o.d0;
        dpred[(size_t)HW + p] = o.d1;
        dpred[2 * (size_t)HW + p] = o.d2;
        dpred[3 * (size_t)HW + p] = o.d3;
        if (dshade) dshade[p] = o.dshade;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        sum_rgb += __shfl_xor(sum_rgb, d, 64);
        sum_mask += __shfl_xor(sum_mask, d, 64);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { s_red[0][wave] = sum_rgb; s_red[1][wave] = sum_mask; }
    __syncthreads();
    if (threadIdx.x == 0) {
        partials[2 * blockIdx.x] = (s_red[0][0] + s_red[0][1]) + (s_red[0][2] + s_red[0][3]);
        partials[2 * blockIdx.x + 1] = (s_red[1][0] + s_red[1][1]) + (s_red[1][2] + s_red[1][3]);
        for (int j = (int)blockIdx.x + GOM_LOSS_BLOCKS; j < slots; j += GOM_LOSS_BLOCKS) { partials[2 * j] = 0.f; partials[2 * j + 1] = 0.f; }   // (a frame step's row: one slot per tile for the riders)
    }
}

}  // namespace

int gom_l1_loss_batch(int B, int H, int W, const float *pred, const float *shade, const float *gt_rgb, const float *gt_mask,
                      const float *bg, float c_rgb, float c_mask, float grad_scale, float *dL_dpred, float *dL_dshade,
                      float *loss_partials, void *stream, const GomLossSkip *skip, int slots) {
    if (H <= 0 || W <= 0) { gom_set_error("gom_l1_loss: bad image size"); return -1; }
    if (!pred || !gt_rgb || !gt_mask || !bg || !dL_dpred || !loss_partials) { gom_set_error("gom_l1_loss: null pointer"); return -1; }
    const int HW = H * W;
    const float k_rgb = grad_scale * c_rgb / (3.0f * (float)HW);
    const float k_mask = grad_scale * c_mask / (float)HW;
    hipLaunchKernelGGL(k_l1_loss, dim3(GOM_LOSS_BLOCKS, B), dim3(256), 0, (hipStream_t)stream, HW, pred, shade, gt_rgb, gt_mask, bg,
                       k_rgb, k_mask, dL_dpred, dL_dshade, loss_partials, skip ? *skip : GomLossSkip{}, slots);
    GOM_LAUNCH_CHECK();
    return 0;
}

extern "C" int gom_l1_loss(int H, int W, const float *pred, const float *shade, const float *gt_rgb, const float *gt_mask,
                           const float *bg, float c_rgb, float c_mask, float grad_scale, float *dL_dpred, float *dL_dshade,
                           float *loss_partials, void *stream) {
    return gom_l1_loss_batch(1, H, W, pred, shade, gt_rgb, gt_mask, bg, c_rgb, c_mask, grad_scale, dL_dpred, dL_dshade, loss_partials,
                             stream);
}

// ---------------------------------------------------------------------------------------------------------------------
// The three "mean |a - b|" terms of the reference's compute_loss (train.py:101-111 rgb and mask, train.py:141-149 normal mask
// against the k x k dilation of the target mask) on ALREADY UNPACKED images, as `Model` + `train_util.compute_loss` see them:
// one launch for the three sums (+ a one-block fold), one for the three gradient images.  Through torch the same arithmetic is
// 10 launches forward and ~14 backward, each 3-5 us of kernel and ~10 us of host time.
namespace {

struct L1Terms {
    const float *a[3];   // predictions: rgb (HW*3), mask (HW), normal mask (HW) -- a null `a` switches the term off
    const float *b[3];   // targets: rgb (HW*3), mask (HW), mask again (dilated on the fly for term 2)
    float *da[3];        // backward only
};

// max over the k x k window of the target mask clipped to the image (F.max_pool2d, stride 1, padding k/2: pads with -inf)
__device__ __forceinline__ float dilated(const float *__restrict__ m, int H, int W, int p, int k) {
    const int y = p / W, x = p - y * W, r = k / 2;
    float v = -INFINITY;
    for (int yy = max(0, y - r); yy <= min(H - 1, y + r); yy++)
        for (int xx = max(0, x - r); xx <= min(W - 1, x + r); xx++) v = fmaxf(v, m[yy * W + xx]);
    return v;
}

template <bool BWD>
__global__ void __launch_bounds__(256) k_l1_terms(L1Terms t, int H, int W, int dil_k, const float *__restrict__ g, float *__restrict__ partials) {
    __shared__ float s_red[3][4];
    const int HW = H * W;
    float sum[3] = {0.f, 0.f, 0.f};
    float gs[3] = {0.f, 0.f, 0.f};
    if (BWD) { gs[0] = g[0] / (3.f * (float)HW); gs[1] = g[1] / (float)HW; gs[2] = g[2] / (float)HW; }
    for (int p = blockIdx.x * 256 + threadIdx.x; p < HW; p += gridDim.x * 256) {
        if (t.a[0]) {
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const float d = t.a[0][3 * (size_t)p + c] - t.b[0][3 * (size_t)p + c];
                if (BWD) t.da[0][3 * (size_t)p + c] = sgn(d) * gs[0]; else sum[0] += fabsf(d);
            }
        }
        if (t.a[1]) {
            const float d = t.a[1][p] - t.b[1][p];
            if (BWD) t.da[1][p] = sgn(d) * gs[1]; else sum[1] += fabsf(d);
        }
        if (t.a[2]) {
            const float d = t.a[2][p] - (dil_k > 1 ? dilated(t.b[2], H, W, p, dil_k) : t.b[2][p]);
            if (BWD) t.da[2][p] = sgn(d) * gs[2]; else sum[2] += fabsf(d);
        }
    }
    if (BWD) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < 3; q++) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) sum[q] += __shfl_xor(sum[q], d, 64);
        if (lane == 0) s_red[q][wave] = sum[q];
    }
    __syncthreads();
    if (threadIdx.x < 3) partials[3 * blockIdx.x + threadIdx.x] = (s_red[threadIdx.x][0] + s_red[threadIdx.x][1]) + (s_red[threadIdx.x][2] + s_red[threadIdx.x][3]);
}

__global__ void __launch_bounds__(64) k_l1_terms_fold(int HW, int nblocks, const float *__restrict__ partials, float *__restrict__ out) {
    float s[3] = {0.f, 0.f, 0.f};
    for (int b = threadIdx.x; b < nblocks; b += 64) { s[0] += partials[3 * b]; s[1] += partials[3 * b + 1]; s[2] += partials[3 * b + 2]; }
#pragma unroll
    for (int q = 0; q < 3; q++)
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) s[q] += __shfl_xor(s[q], d, 64);
    if (threadIdx.x == 0) { out[0] = s[0] / (3.f * (float)HW); out[1] = s[1] / (float)HW; out[2] = s[2] / (float)HW; }
}

}  // namespace

extern "C" int gom_l1_terms_forward(int H, int W, const float *rgb, const float *rgb_gt, const float *mask, const float *mask_gt,
                                    const float *normal_mask, int dil_k, float *out3, float *partials, void *stream) {
    if (H <= 0 || W <= 0 || dil_k < 0 || (dil_k > 1 && !(dil_k & 1))) { gom_set_error("gom_l1_terms_forward: bad image size or even dilation window"); return -1; }
    if (!out3 || !partials || (rgb && !rgb_gt) || ((mask || normal_mask) && !mask_gt)) { gom_set_error("gom_l1_terms_forward: null pointer"); return -1; }
    L1Terms t = {{rgb, mask, normal_mask}, {rgb_gt, mask_gt, mask_gt}, {nullptr, nullptr, nullptr}};
    // one pixel per thread up to 4 x GOM_LOSS_BLOCKS workgroups (the dilation is a 49-load loop per pixel: latency, not bandwidth)
    const int nblocks = min((H * W + 255) / 256, 4 * GOM_LOSS_BLOCKS);
    hipLaunchKernelGGL(k_l1_terms<false>, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, t, H, W, dil_k, (const float *)nullptr, partials);
    GOM_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_l1_terms_fold, dim3(1), dim3(64), 0, (hipStream_t)stream, H * W, nblocks, partials, out3);
    GOM_LAUNCH_CHECK();
    return 0;
}

extern "C" int gom_l1_terms_backward(int H, int W, const float *rgb, const float *rgb_gt, const float *mask, const float *mask_gt,
                                     const float *normal_mask, int dil_k, const float *g3, float *d_rgb, float *d_mask, float *d_normal_mask,
                                     void *stream) {
    if (H <= 0 || W <= 0 || dil_k < 0 || (dil_k > 1 && !(dil_k & 1))) { gom_set_error("gom_l1_terms_backward: bad image size or even dilation window"); return -1; }
    if (!g3 || (rgb && (!rgb_gt || !d_rgb)) || (mask && (!mask_gt || !d_mask)) || (normal_mask && (!mask_gt || !d_normal_mask))) {
        gom_set_error("gom_l1_terms_backward: null pointer"); return -1;
    }
    L1Terms t = {{rgb, mask, normal_mask}, {rgb_gt, mask_gt, mask_gt}, {d_rgb, d_mask, d_normal_mask}};
    hipLaunchKernelGGL(k_l1_terms<true>, dim3((H * W + 255) / 256), dim3(256), 0, (hipStream_t)stream, t, H, W, dil_k, g3, (float *)nullptr);
    GOM_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// Image composition around the rasterizer's (4,H,W) output, one launch each way instead of the slice / permute / multiply chains:
//   compose (models/model.py:262-287):  albedo = img[:3] as (H,W,3), mask = img[3], rgb = albedo * shade
//   unpack  (train.py:53-55):            out = rgb * mask + bg * (1 - mask)
namespace {

__global__ void __launch_bounds__(256) k_compose_fwd(int HW, const float *__restrict__ img, const float *__restrict__ shade, float *__restrict__ albedo,
                                                     float *__restrict__ mask, float *__restrict__ rgb) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    const float s = shade ? shade[p] : 1.f;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float a = img[(size_t)c * HW + p];
        albedo[3 * (size_t)p + c] = a;
        if (rgb) rgb[3 * (size_t)p + c] = a * s;
    }
    mask[p] = img[3 * (size_t)HW + p];
}

__global__ void __launch_bounds__(256) k_compose_bwd(int HW, const float *__restrict__ img, const float *__restrict__ shade, const float *__restrict__ d_albedo,
                                                     const float *__restrict__ d_mask, const float *__restrict__ d_rgb, float *__restrict__ d_img,
                                                     float *__restrict__ d_shade) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    const float s = shade ? shade[p] : 1.f;
    float ds = 0.f;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float gr = d_rgb ? d_rgb[3 * (size_t)p + c] : 0.f;
        d_img[(size_t)c * HW + p] = (d_albedo ? d_albedo[3 * (size_t)p + c] : 0.f) + gr * s;
        ds += gr * img[(size_t)c * HW + p];
    }
    d_img[3 * (size_t)HW + p] = d_mask ? d_mask[p] : 0.f;
    if (d_shade) d_shade[p] = ds;
}

template <bool BWD>
__global__ void __launch_bounds__(256) k_unpack(int HW, const float *__restrict__ rgb, const float *__restrict__ mask, const float *__restrict__ bg,
                                                const float *__restrict__ g, float *__restrict__ out, float *__restrict__ d_rgb, float *__restrict__ d_mask) {
    const int p = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;   // b: image of the batch
    if (p >= HW) return;
    const size_t q = (size_t)b * HW + p;
    const float m = mask[q];
    float dm = 0.f;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float r = rgb[3 * q + c], k = bg[3 * b + c];
        if (!BWD) out[3 * q + c] = r * m + k * (1.f - m);
        else {
            const float gc = g[3 * q + c];
            d_rgb[3 * q + c] = gc * m;
            dm += gc * (r - k);
        }
    }
    if (BWD) d_mask[q] = dm;
}

}  // namespace

extern "C" int gom_compose_forward(int H, int W, const float *img, const float *shade, float *albedo, float *mask, float *rgb, void *stream) {
    if (H <= 0 || W <= 0 || !img || !albedo || !mask) { gom_set_error("gom_compose_forward: bad arguments"); return -1; }
    hipLaunchKernelGGL(k_compose_fwd, dim3((H * W + 255) / 256), dim3(256), 0, (hipStream_t)stream, H * W, img, shade, albedo, mask, rgb);
    GOM_LAUNCH_CHECK();
    return 0;
}

extern "C" int gom_compose_backward(int H, int W, const float *img, const float *shade, const float *d_albedo, const float *d_mask, const float *d_rgb,
                                    float *d_img, float *d_shade, void *stream) {
    if (H <= 0 || W <= 0 || !img || !d_img) { gom_set_error("gom_compose_backward: bad arguments"); return -1; }
    hipLaunchKernelGGL(k_compose_bwd, dim3((H * W + 255) / 256), dim3(256), 0, (hipStream_t)stream, H * W, img, shade, d_albedo, d_mask, d_rgb, d_img, d_shade);
    GOM_LAUNCH_CHECK();
    return 0;
}

extern "C" int gom_unpack_forward(int B, int H, int W, const float *rgb, const float *mask, const float *bg, float *out, void *stream) {
    if (B <= 0 || H <= 0 || W <= 0 || !rgb || !mask || !bg || !out) { gom_set_error("gom_unpack_forward: bad arguments"); return -1; }
    hipLaunchKernelGGL(k_unpack<false>, dim3((H * W + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, H * W, rgb, mask, bg, (const float *)nullptr, out,
                       (float *)nullptr, (float *)nullptr);
    GOM_LAUNCH_CHECK();
    return 0;
}

extern "C" int gom_unpack_backward(int B, int H, int W, const float *rgb, const float *mask, const float *bg, const float *g, float *d_rgb, float *d_mask,
                                   void *stream) {
    if (B <= 0 || H <= 0 || W <= 0 || !rgb || !mask || !bg || !g || !d_rgb || !d_mask) { gom_set_error("gom_unpack_backward: bad arguments"); return -1; }
    hipLaunchKernelGGL(k_unpack<true>, dim3((H * W + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, H * W, rgb, mask, bg, g, (float *)nullptr, d_rgb, d_mask);
    GOM_LAUNCH_CHECK();
    return 0;
}

// ---- the tail of compute_loss (train.py:98-163): every term's partial sums -> the vector of terms, its scaled copy and the total, ONE launch ----
// The terms leave per-workgroup partial sums (or, the L1 terms, their folded values as rows of one element); torch summed each with a launch of
// its own, concatenated, multiplied by the coefficients and summed again: nine launches of ~4.5 us with ~2.5 us between them on a chain of ~150.
namespace {
#define GOM_TAIL_MAX_INPUTS 8
struct LossTailArgs {
    int n_inputs;
    const float *ptr[GOM_TAIL_MAX_INPUTS];   // input i: rows[i] x cols[i] floats, row-major; its first used[i] rows are terms
    int rows[GOM_TAIL_MAX_INPUTS], cols[GOM_TAIL_MAX_INPUTS], used[GOM_TAIL_MAX_INPUTS];
    float pre[GOM_TAIL_MAX_INPUTS];          // term = pre * sum(row)
};
__global__ void __launch_bounds__(1024) k_loss_tail(LossTailArgs a, const float *__restrict__ coeffs, float *__restrict__ vec, float *__restrict__ scaled, float *__restrict__ total) {
    __shared__ float s_scaled[64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int k = 0;
    for (int i = 0; i < a.n_inputs; i++) {
        for (int r = 0; r < a.used[i]; r++, k++) {
            if ((k & 15) != wave) continue;   // a wave per term, terms dealt round robin
            const float *row = a.ptr[i] + (size_t)r * a.cols[i];
            float acc = 0.f;
            for (int c = lane; c < a.cols[i]; c += 64) acc += row[c];
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
            const float v = acc * a.pre[i];
            if (lane == 0) { vec[k] = v; const float sc = v * coeffs[k]; scaled[k] = sc; s_scaled[k] = sc; }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int j = 0; j < k; j++) t += s_scaled[j];   // in term order
        *total = t;
    }
}
}  // namespace

extern "C" int gom_loss_tail(int n_inputs, const float *const *ptrs, const int32_t *rows, const int32_t *cols, const int32_t *used, const float *pre,
                             const float *coeffs, float *vec, float *scaled, float *total, void *stream) {
    if (n_inputs < 1 || n_inputs > GOM_TAIL_MAX_INPUTS || !ptrs || !rows || !cols || !used || !pre || !coeffs || !vec || !scaled || !total) {
        gom_set_error("gom_loss_tail: 1..%d inputs, no null pointers", GOM_TAIL_MAX_INPUTS);
        return -1;
    }
    LossTailArgs a{};
    a.n_inputs = n_inputs;
    int K = 0;
    for (int i = 0; i < n_inputs; i++) {
        if (!ptrs[i] || rows[i] < 1 || cols[i] < 1 || used[i] < 0 || used[i] > rows[i]) { gom_set_error("gom_loss_tail: bad input %d", i); return -1; }
        a.ptr[i] = ptrs[i]; a.rows[i] = rows[i]; a.cols[i] = cols[i]; a.used[i] = used[i]; a.pre[i] = pre[i];
        K += used[i];
    }
    if (K < 1 || K > 64) { gom_set_error("gom_loss_tail: 1..64 terms"); return -1; }
    hipLaunchKernelGGL(k_loss_tail, dim3(1), dim3(1024), 0, (hipStream_t)stream, a, coeffs, vec, scaled, total);
    GOM_LAUNCH_CHECK();
    return 0;
}
