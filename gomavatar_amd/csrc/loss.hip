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

namespace {

__device__ __forceinline__ float sgn(float x) { return (x > 0.f) ? 1.f : ((x < 0.f) ? -1.f : 0.f); }

__global__ void __launch_bounds__(256) k_l1_loss(int HW, const float *__restrict__ pred, const float *__restrict__ shade,
                                                 const float *__restrict__ gt_rgb, const float *__restrict__ gt_mask,
                                                 const float *__restrict__ bg, float k_rgb, float k_mask,
                                                 float *__restrict__ dpred, float *__restrict__ dshade, float *__restrict__ partials, GomBwdOrderRider rider) {
    __shared__ float s_red[2][4];
    if (blockIdx.x >= GOM_LOSS_BLOCKS) {   // the riders (eight workgroups, frame 0's row of the grid only): see GomBwdOrderRider
        if (blockIdx.y != 0 || rider.status->overflow) return;
        // Rider x orders the tasks of queue shard x (the segments with seg mod 8 = x): a counting sort over 512 cost levels, most
        // expensive first; a pair of sub-ranges above GOM_BWD_SPLIT_COST becomes two single-sub-range tasks.
        __shared__ uint32_t s_lvl[512];
        const uint32_t x = blockIdx.x - GOM_LOSS_BLOCKS, nsegs = rider.status->num_segs;
        const uint32_t npairs = nsegs > x ? 2u * ((nsegs - x + 7u) / 8u) : 0u;       // (segment, pair) units of this shard
        const uint32_t region = 4u * ((nsegs + 7u) / 8u);
        uint32_t *out = rider.bwd_order + GOM_BWD_ORDER_BASE + (size_t)x * region;
        for (int k = threadIdx.x; k < 512; k += 256) s_lvl[k] = 0u;
        __syncthreads();
        auto level = [](uint32_t c) { return 511u - min(c, 511u); };                  // level 0 = the most expensive
        for (int pass = 0; pass < 2; pass++) {
            for (uint32_t j0 = threadIdx.x; j0 < npairs; j0 += 2 * 256) {            // 2 units = 4 independent 16-byte loads in flight per thread
                uint2 c[2];
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    const uint32_t j = j0 + u * 256, seg = (j >> 1) * 8u + x;
                    c[u] = make_uint2(0xffffffffu, 0u);
                    if (j < npairs && seg < nsegs) {
                        // what the pair costs its workgroup: wave w takes quadrant w of the first sub-range, then quadrant 3 - w of the second
                        // (k_seg_bwd_pair), and the task lasts as long as its busiest wave; a sub-range alone: its busiest quadrant
                        const uint4 a = *reinterpret_cast<const uint4 *>(rider.seg_cost + 16 * (size_t)seg + 8 * (j & 1u));
                        const uint4 b = *reinterpret_cast<const uint4 *>(rider.seg_cost + 16 * (size_t)seg + 8 * (j & 1u) + 4);
                        const uint32_t both = max(max(a.x + b.w, a.y + b.z), max(a.z + b.y, a.w + b.x));
                        c[u] = make_uint2(max(max(a.x, a.y), max(a.z, a.w)), max(max(b.x, b.y), max(b.z, b.w)));
                        if (both <= GOM_BWD_SPLIT_COST) c[u] = make_uint2(both, 0xfffffffeu);   // (.y = marker: not split)
                    }
                }
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    if (c[u].x == 0xffffffffu) continue;
                    const uint32_t j = j0 + u * 256, seg = (j >> 1) * 8u + x, pair = j & 1u, tot = c[u].x;
                    if (c[u].y != 0xfffffffeu) {
                        if (pass == 0) { atomicAdd(&s_lvl[level(c[u].x)], 1u); atomicAdd(&s_lvl[level(c[u].y)], 1u); }
                        else {
                            out[atomicAdd(&s_lvl[level(c[u].x)], 1u)] = (seg << 3) | (4u + 2u * pair);
                            out[atomicAdd(&s_lvl[level(c[u].y)], 1u)] = (seg << 3) | (5u + 2u * pair);
                        }
                    } else {
                        if (pass == 0) atomicAdd(&s_lvl[level(tot)], 1u);
                        else out[atomicAdd(&s_lvl[level(tot)], 1u)] = (seg << 3) | pair;    // (order inside a level: any)
                    }
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
        return;
    }
    {  // blockIdx.y = frame of a batched launch: [B][4][HW] images, [B][HW][3] targets, [B][3] backgrounds
        const size_t fr = blockIdx.y;
        pred += fr * 4 * HW; gt_rgb += fr * 3 * HW; gt_mask += fr * HW; bg += fr * 3; dpred += fr * 4 * HW;
        partials += fr * 2 * GOM_LOSS_BLOCKS;   // (the grid's x extent may carry one rider block more)
        if (shade) shade += fr * HW;
        if (dshade) dshade += fr * HW;
    }
    const float b0 = bg[0], b1 = bg[1], b2 = bg[2];
    float sum_rgb = 0.f, sum_mask = 0.f;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < HW; p += GOM_LOSS_BLOCKS * 256) {
        const float m = pred[3 * (size_t)HW + p];
        const float s = shade ? shade[p] : 1.f;
        const float a0 = pred[p], a1 = pred[(size_t)HW + p], a2 = pred[2 * (size_t)HW + p];
        const float3 g = *reinterpret_cast<const float3 *>(gt_rgb + 3 * (size_t)p);   // one 12-byte load per lane (three strided 4-byte loads cost the texture path three passes)
        const float r0 = a0 * s * m + b0 * (1.f - m) - g.x;
        const float r1 = a1 * s * m + b1 * (1.f - m) - g.y;
        const float r2 = a2 * s * m + b2 * (1.f - m) - g.z;
        const float rm = m - gt_mask[p];
        sum_rgb += fabsf(r0) + fabsf(r1) + fabsf(r2);
        sum_mask += fabsf(rm);
        const float s0 = sgn(r0) * k_rgb, s1 = sgn(r1) * k_rgb, s2 = sgn(r2) * k_rgb;
        dpred[p] = s0 * s * m;
        dpred[(size_t)HW + p] = s1 * s * m;
        dpred[2 * (size_t)HW + p] = s2 * s * m;
        dpred[3 * (size_t)HW + p] = s0 * (a0 * s - b0) + s1 * (a1 * s - b1) + s2 * (a2 * s - b2) + sgn(rm) * k_mask;
        if (dshade) dshade[p] = (s0 * a0 + s1 * a1 + s2 * a2) * m;
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
    }
}

}  // namespace

int gom_l1_loss_batch(int B, int H, int W, const float *pred, const float *shade, const float *gt_rgb, const float *gt_mask,
                      const float *bg, float c_rgb, float c_mask, float grad_scale, float *dL_dpred, float *dL_dshade,
                      float *loss_partials, void *stream, const GomBwdOrderRider *rider) {
    if (H <= 0 || W <= 0) { gom_set_error("gom_l1_loss: bad image size"); return -1; }
    if (!pred || !gt_rgb || !gt_mask || !bg || !dL_dpred || !loss_partials) { gom_set_error("gom_l1_loss: null pointer"); return -1; }
    const int HW = H * W;
    const float k_rgb = grad_scale * c_rgb / (3.0f * (float)HW);
    const float k_mask = grad_scale * c_mask / (float)HW;
    hipLaunchKernelGGL(k_l1_loss, dim3(GOM_LOSS_BLOCKS + (rider ? 8 : 0), B), dim3(256), 0, (hipStream_t)stream, HW, pred, shade, gt_rgb, gt_mask, bg,
                       k_rgb, k_mask, dL_dpred, dL_dshade, loss_partials, rider ? *rider : GomBwdOrderRider{});
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
