// LPIPS-VGG value and image gradient as ONE native call on the bf16 matrix-core trunk (vgg_bf16.hip):
// 2 x (prepare + 13 convolutions + 4 pools) forward, 5 head forwards, then 5 head backwards + 4 pool backwards +
// 13 backward-data convolutions + the image gradient -- ~75 launches enqueued back to back on the caller's stream
// (hipGraph-capturable: no allocation after the first call of a given size, no host sync).
// Replaces train.py:113-121 (LPIPS(2*pred-1, 2*gt-1).mean() and its autograd backward).
#include <stdlib.h>
#include <string.h>

#include "gom_internal.h"

namespace {
const int kPoolBefore[13] = {0, 0, 1, 0, 1, 0, 0, 1, 0, 0, 1, 0, 0};
const int kTapIndex[13] = {-1, 0, -1, 1, -1, -1, 2, -1, -1, 3, -1, -1, 4};
}  // namespace

struct GomLpipsVgg {
    const void *w_fwd[13], *w_bwd[13];
    const float *bias[13], *lin[5];
    int cin[13], cout[13];
    const void *w1_fwd = nullptr, *w1_bwd = nullptr;   // first layer as a 1 x 1 convolution over im2col rows (gom_lpips_vgg_set_first_layer; vgg_bf16.hip)
    int x3 = 0;                                  // GOM_LPIPS_PRECISION_BF16X3: two bf16 planes per tensor, three MFMA passes (vgg_bf16.hip)
    size_t lo_x = 0, lo_act[13] = {}, lo_pooled[13] = {}, lo_g = 0;   // element offsets of the lo planes (0 in the plain mode)
    // workspace for (B, H, W)
    int B = 0, H = 0, W = 0;
    void *x[2] = {nullptr, nullptr};            // trunk inputs (B,H,W,32)
    void *act[2][13] = {};                       // post-ReLU activations of both images
    void *pooled[2][13] = {};                    // pool outputs feeding conv i (i in kPoolBefore)
    void *grad[2] = {nullptr, nullptr};          // ping-pong gradient buffers (largest activation)
    void *gtap = nullptr;                        // head gradient of the current tap
    float *splitk = nullptr, *splitk_target = nullptr;   // split-K partial sums; the target-only pass (its own stream) has its own
    float *go = nullptr;                         // [B] d value / d value_b
    float go_scale = -1.f;                       // what `go` holds (a value no caller passes: filled at the first call)
    float *head_sums = nullptr;                  // [5][B][GOM_LPIPS_HEAD_BLOCKS]: per-workgroup value sums of the taps when the backward kernels produce them
    size_t splitk_elems = 0;
    // captured launch sequence (GOM_LPIPS_USE_GRAPH), valid for exactly these arguments
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    const float *g_pred = nullptr, *g_gt = nullptr;
    float *g_partials = nullptr, *g_dpred = nullptr;
    float g_scale = 0.f;
    uint32_t g_flags = 0;
};

static void lp_drop_graph(GomLpipsVgg *h) {
    if (h->exec) (void)hipGraphExecDestroy(h->exec);
    if (h->graph) (void)hipGraphDestroy(h->graph);
    h->exec = nullptr; h->graph = nullptr;
}

static void lp_free(GomLpipsVgg *h) {
    lp_drop_graph(h);
    void *ptrs[] = {h->x[0], h->grad[0], h->grad[1], h->gtap, h->splitk, h->splitk_target, h->go, h->head_sums};   // (x[1], act[1][], pooled[1][] are the second halves of [0])
    for (void *p : ptrs) if (p) (void)hipFree(p);
    for (int i = 0; i < 13; i++) {
        if (h->act[0][i]) (void)hipFree(h->act[0][i]);
        if (h->pooled[0][i]) (void)hipFree(h->pooled[0][i]);
        for (int k = 0; k < 2; k++) h->act[k][i] = h->pooled[k][i] = nullptr;
    }
    h->x[0] = h->x[1] = h->grad[0] = h->grad[1] = h->gtap = nullptr;
    h->splitk = h->splitk_target = nullptr; h->go = h->head_sums = nullptr; h->go_scale = -1.f;
    h->B = h->H = h->W = 0;
}

extern "C" GomLpipsVgg *gom_lpips_vgg_create(const void *const *w_fwd, const void *const *w_bwd, const float *const *bias, const float *const *lin,
                                             const int32_t *cin, const int32_t *cout) {
    if (!w_fwd || !w_bwd || !bias || !lin || !cin || !cout) { gom_set_error("gom_lpips_vgg_create: null argument"); return nullptr; }
    GomLpipsVgg *h = new GomLpipsVgg();
    for (int i = 0; i < 13; i++) { h->w_fwd[i] = w_fwd[i]; h->w_bwd[i] = w_bwd[i]; h->bias[i] = bias[i]; h->cin[i] = cin[i]; h->cout[i] = cout[i]; }
    for (int k = 0; k < 5; k++) h->lin[k] = lin[k];
    return h;
}

extern "C" int gom_lpips_vgg_set_precision(GomLpipsVgg *h, int32_t precision) {
    if (!h || (precision != GOM_LPIPS_PRECISION_BF16 && precision != GOM_LPIPS_PRECISION_BF16X3)) { gom_set_error("gom_lpips_vgg_set_precision: bad argument"); return -1; }
    if (h->x3 != (precision == GOM_LPIPS_PRECISION_BF16X3)) lp_free(h);   // other buffer sizes; the weights the handle points at must match
    h->x3 = precision == GOM_LPIPS_PRECISION_BF16X3;
    return 0;
}

extern "C" int gom_lpips_vgg_set_first_layer(GomLpipsVgg *h, const void *w1x1_fwd, const void *w1x1_bwd) {
    if (!h || (w1x1_fwd == nullptr) != (w1x1_bwd == nullptr)) { gom_set_error("gom_lpips_vgg_set_first_layer: both weight blocks or none"); return -1; }
    if ((h->w1_fwd != nullptr) != (w1x1_fwd != nullptr)) lp_drop_graph(h);   // (a recorded sequence holds the other layer-0 launches)
    h->w1_fwd = w1x1_fwd; h->w1_bwd = w1x1_bwd;
    return 0;
}

extern "C" void gom_lpips_vgg_destroy(GomLpipsVgg *h) {
    if (!h) return;
    lp_free(h);
    delete h;
}

static int lp_ensure(GomLpipsVgg *h, int B, int H, int W) {
    if (h->B == B && h->H == H && h->W == W) return 0;
    lp_free(h);
    size_t maxact = 0, maxsplit = 0;
    int hh = H, ww = W;
    // prediction and target walk the trunk as ONE batch of 2B images (twice the workgroups per launch, half the launches):
    // every forward buffer is [2][B][...], its second half is image set 1
    // (bf16x3: [hi of set 0][hi of set 1][lo of set 0][lo of set 1] -- the lo plane 2n elements behind the hi plane for both sets)
    const size_t planes = h->x3 ? 2 : 1;
    auto alloc2 = [planes](void **p0, void **p1, size_t bytes) -> hipError_t {
        const hipError_t e = hipMalloc(p0, 2 * bytes * planes);
        *p1 = e == hipSuccess ? (void *)((char *)*p0 + bytes) : nullptr;
        return e;
    };
    GOM_HIP_CHECK(alloc2(&h->x[0], &h->x[1], (size_t)B * H * W * 32 * 2));
    h->lo_x = h->x3 ? 2 * (size_t)B * H * W * 32 : 0;
    for (int i = 0; i < 13; i++) {
        if (kPoolBefore[i]) {
            hh /= 2; ww /= 2;
            GOM_HIP_CHECK(alloc2(&h->pooled[0][i], &h->pooled[1][i], (size_t)B * hh * ww * h->cin[i] * 2));
            h->lo_pooled[i] = h->x3 ? 2 * (size_t)B * hh * ww * h->cin[i] : 0;
        }
        const size_t n = (size_t)B * hh * ww * h->cout[i];
        GOM_HIP_CHECK(alloc2(&h->act[0][i], &h->act[1][i], n * 2));
        h->lo_act[i] = h->x3 ? 2 * n : 0;
        maxact = n > maxact ? n : maxact;
        const size_t nin = (size_t)B * hh * ww * (h->cin[i] < 64 ? 64 : h->cin[i]);
        maxact = nin > maxact ? nin : maxact;
        const size_t sf = (size_t)gom_conv3x3_splits(2 * B, hh, ww, h->cin[i], h->cout[i]) * 2 * n;
        const size_t sb = (size_t)gom_conv3x3_splits(B, hh, ww, h->cout[i], h->cin[i] < 64 ? 64 : h->cin[i]) * nin;
        maxsplit = sf > maxsplit ? sf : maxsplit;
        const size_t s1 = (size_t)gom_conv3x3_splits(B, hh, ww, h->cin[i], h->cout[i]) * n;   // (one image set alone: GOM_LPIPS_TARGET_READY / target_features)
        maxsplit = s1 > maxsplit ? s1 : maxsplit;
        maxsplit = sb > maxsplit ? sb : maxsplit;
    }
    for (int k = 0; k < 2; k++) GOM_HIP_CHECK(hipMalloc(&h->grad[k], maxact * 2 * planes));
    GOM_HIP_CHECK(hipMalloc(&h->gtap, maxact * 2 * planes));
    h->lo_g = h->x3 ? maxact : 0;
    GOM_HIP_CHECK(hipMalloc((void **)&h->splitk, maxsplit * sizeof(float)));
    GOM_HIP_CHECK(hipMalloc((void **)&h->splitk_target, maxsplit * sizeof(float)));
    GOM_HIP_CHECK(hipMalloc((void **)&h->go, (size_t)B * sizeof(float)));
    GOM_HIP_CHECK(hipMalloc((void **)&h->head_sums, (size_t)5 * B * GOM_LPIPS_HEAD_BLOCKS * sizeof(float)));
    h->splitk_elems = maxsplit;
    h->B = B; h->H = H; h->W = W;
    return 0;
}

static int lp_conv(float *splitk, int B, int hh, int ww, int cin, int cout, const void *in, const void *wt, const float *bias, const void *mask,
                   void *out, uint32_t flags, size_t in_lo, size_t out_lo, void *stream, void *pooled = nullptr, size_t pooled_lo = 0) {
    const int s = gom_conv3x3_splits(B, hh, ww, cin, cout);
    return gom_conv3x3_planes(B, hh, ww, cin, cout, in, wt, bias, mask, out, flags, s, s > 1 ? splitk : nullptr, in_lo, out_lo, pooled, pooled_lo, stream);
}

__global__ void k_fill(float *p, int n, float v) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = v;
}

static int lp_enqueue(GomLpipsVgg *h, int B, int H, int W, const float *pred, const float *gt, float *value_partials, float grad_scale,
                      float *d_pred, bool target_ready, void *stream);
static int lp_trunk_forward(GomLpipsVgg *h, int k0, int nsets, int B, int H, int W, const float *const *img, float *splitk, void *stream);

extern "C" int gom_lpips_vgg_target_features(GomLpipsVgg *h, int B, int H, int W, const float *gt, void *stream) {
    if (!h || !gt) { gom_set_error("gom_lpips_vgg_target_features: null argument"); return -1; }
    if (B <= 0 || H <= 0 || W <= 0 || H % 16 || W % 16) { gom_set_error("gom_lpips_vgg_target_features: H and W must be multiples of 16"); return -1; }
    int rc;
    if ((rc = lp_ensure(h, B, H, W))) return rc;
    const float *img[2] = {nullptr, gt};
    return lp_trunk_forward(h, 1, 1, B, H, W, img, h->splitk_target, stream);
}

extern "C" int gom_lpips_vgg_value_and_grad(GomLpipsVgg *h, int B, int H, int W, const float *pred, const float *gt, float *value_partials,
                                            float grad_scale, float *d_pred, uint32_t flags, void *stream) {
    if (!h || !pred || !gt || !value_partials) { gom_set_error("gom_lpips_vgg_value_and_grad: null argument"); return -1; }
    if (B <= 0 || H <= 0 || W <= 0 || H % 16 || W % 16) { gom_set_error("gom_lpips_vgg_value_and_grad: H and W must be multiples of 16"); return -1; }
    int rc;
    const bool same_size = h->B == B && h->H == H && h->W == W;
    if ((rc = lp_ensure(h, B, H, W))) return rc;
    const bool target_ready = (flags & GOM_LPIPS_TARGET_READY) != 0;
    if (target_ready && !same_size) { gom_set_error("gom_lpips_vgg_value_and_grad: GOM_LPIPS_TARGET_READY without gom_lpips_vgg_target_features at this size"); return -1; }
    if (!(flags & GOM_LPIPS_USE_GRAPH) || stream == nullptr) return lp_enqueue(h, B, H, W, pred, gt, value_partials, grad_scale, d_pred, target_ready, stream);
    hipStream_t st = (hipStream_t)stream;
    if (h->exec && same_size && h->g_pred == pred && h->g_gt == gt && h->g_partials == value_partials && h->g_dpred == d_pred && h->g_scale == grad_scale && h->g_flags == flags) {
        GOM_HIP_CHECK(hipGraphLaunch(h->exec, st));
        return 0;
    }
    lp_drop_graph(h);
    GOM_HIP_CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    rc = lp_enqueue(h, B, H, W, pred, gt, value_partials, grad_scale, d_pred, target_ready, stream);
    hipError_t ce = hipStreamEndCapture(st, &h->graph);
    if (rc) { lp_drop_graph(h); return rc; }
    if (ce != hipSuccess) { gom_set_error("hipStreamEndCapture failed: %s", hipGetErrorString(ce)); h->graph = nullptr; return -2; }
    GOM_HIP_CHECK(hipGraphInstantiate(&h->exec, h->graph, nullptr, nullptr, 0));
    h->g_pred = pred; h->g_gt = gt; h->g_partials = value_partials; h->g_dpred = d_pred; h->g_scale = grad_scale; h->g_flags = flags;
    GOM_HIP_CHECK(hipGraphLaunch(h->exec, st));
    return 0;
}

// prepare + 13 convolutions + 4 pools of `nsets` image sets starting at set k0 (0 = prediction, 1 = target) as one batch of nsets * B images
static int lp_trunk_forward(GomLpipsVgg *h, int k0, int nsets, int B, int H, int W, const float *const *img, float *splitk, void *stream) {
    int rc;
    const bool im2col = h->w1_fwd != nullptr;   // conv1_1 without its channel padding (vgg_bf16.hip: k_lpips_prepare_im2col)
    // conv1_1 reads the image itself (k_conv1_1_image) unless GOM_LPIPS_FIRST_LAYER_FUSED=0 asks for the two-kernel form (im2col rows, then a 1 x 1
    // convolution: the same bits, 33 MB more traffic per image and plane) -- read per call so that a test can compare the two in one process
    const char *fenv = getenv("GOM_LPIPS_FIRST_LAYER_FUSED");
    const bool fused1 = im2col && W % 32 == 0 && !(fenv && fenv[0] == '0');
    const char *penv = getenv("GOM_LPIPS_FUSED_POOL");                 // (development switch: 0 = the pools as launches of their own)
    const bool fused_pool = !(penv && penv[0] == '0');
    for (int k = k0; k < k0 + nsets && !fused1; k++)
        if ((rc = im2col ? gom_lpips_prepare_im2col_planes(B, H, W, img[k], h->x[k], h->lo_x, stream) : gom_lpips_prepare_planes(B, H, W, img[k], h->x[k], h->lo_x, stream))) return rc;
    const int nb = nsets * B;
    int hh = H, ww = W;
    const void *cur = h->x[k0];
    size_t cur_lo = h->lo_x;
    for (int i = 0; i < 13; i++) {
        if (kPoolBefore[i]) {   // (written by the convolution in front of it: conv_store / k_splitk_epilogue with `pooled`)
            if (!fused_pool && (rc = gom_maxpool2x2_planes(nb, hh, ww, h->cin[i], cur, h->pooled[k0][i], cur_lo, h->lo_pooled[i], stream))) return rc;
            hh /= 2; ww /= 2;
            cur = h->pooled[k0][i]; cur_lo = h->lo_pooled[i];
        }
        const bool pool_next = fused_pool && i + 1 < 13 && kPoolBefore[i + 1];
        if (i == 0 && fused1) {
            for (int k = k0; k < k0 + nsets; k++)   // (one launch per image set: the sets' images are separate tensors)
                if ((rc = gom_conv1_1_image_planes(B, hh, ww, img[k], h->w1_fwd, h->bias[0], h->act[k][0], h->lo_act[0], stream))) return rc;
        } else if (i == 0 && im2col) {
            if ((rc = gom_conv1x1_planes((size_t)nb * hh * ww, 32, h->cout[0], cur, h->w1_fwd, h->bias[0], h->act[k0][0], 1, cur_lo, h->lo_act[0], stream))) return rc;
        } else if ((rc = lp_conv(splitk, nb, hh, ww, h->cin[i], h->cout[i], cur, h->w_fwd[i], h->bias[i], nullptr, h->act[k0][i], GOM_CONV_RELU, cur_lo, h->lo_act[i], stream,
                                 pool_next ? h->pooled[k0][i + 1] : nullptr, pool_next ? h->lo_pooled[i + 1] : 0))) return rc;
        cur = h->act[k0][i]; cur_lo = h->lo_act[i];
    }
    return 0;
}

static int lp_enqueue(GomLpipsVgg *h, int B, int H, int W, const float *pred, const float *gt, float *value_partials, float grad_scale,
                      float *d_pred, bool target_ready, void *stream) {
    int rc;
    const float *img[2] = {pred, gt};
    const bool im2col = h->w1_fwd != nullptr;
    // prediction and target as ONE batch of 2B images -- or, when the caller ran the target's trunk beforehand
    // (gom_lpips_vgg_target_features, typically on another stream under the frame's forward), the prediction alone
    if ((rc = lp_trunk_forward(h, 0, target_ready ? 1 : 2, B, H, W, img, h->splitk, stream))) return rc;
    if (!d_pred) {   // value only: the head forwards (with a gradient wanted, each tap's backward kernel leaves its value too: one read of the feature maps)
        int hh = H, ww = W;
        for (int i = 0; i < 13; i++) {
            if (kPoolBefore[i]) { hh /= 2; ww /= 2; }
            const int t = kTapIndex[i];
            if (t < 0) continue;
            if ((rc = gom_lpips_layer_forward_planes(B, h->cout[i], hh * ww, h->act[0][i], h->act[1][i], h->lin[t],
                                                     value_partials + (size_t)t * B * GOM_LOSS_BLOCKS, h->lo_act[i], stream)))
                return rc;
        }
    }
    if (!d_pred) return 0;
    if (h->go_scale != grad_scale) {   // d value / d value_b, the same number for every image: rewritten only when it changes (not a launch per call)
        hipLaunchKernelGGL(k_fill, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, h->go, B, grad_scale);
        GOM_LAUNCH_CHECK();
        h->go_scale = grad_scale;
    }
    // backward: g = gradient w.r.t. the pre-ReLU output of conv i, walking the trunk of image 0 in reverse
    int hs[13], wsz[13];
    {
        int hh = H, ww = W;
        for (int i = 0; i < 13; i++) { if (kPoolBefore[i]) { hh /= 2; ww /= 2; } hs[i] = hh; wsz[i] = ww; }
    }
    const void *g = nullptr;
    int pp = 0, head_blocks[5] = {0, 0, 0, 0, 0};
    const char *benv = getenv("GOM_LPIPS_FIRST_LAYER_BWD_FUSED");   // (development switch, read per call: 0 = conv1_1's backward as a 1 x 1 convolution + the col2im kernel)
    const bool first_bwd_fused = !(benv && benv[0] == '0');
    for (int i = 12; i >= 0; i--) {
        const int hh = hs[i], ww = wsz[i], t = kTapIndex[i];
        if (t >= 0) {
            void *gh = h->gtap;
            // (the gradient of the pool this activation feeds -- g, from the layers above -- is routed inside the same kernel: k_lpips_head_nhwc_1p POOL)
            if ((rc = gom_lpips_layer_backward_value_planes(B, h->cout[i], hh * ww, h->act[0][i], h->act[1][i], h->lin[t], h->go, gh,
                                                            h->head_sums + (size_t)t * B * GOM_LPIPS_HEAD_BLOCKS, &head_blocks[t], h->lo_act[i], h->lo_g, g, h->lo_g, ww, stream))) return rc;
            g = gh;
        }
        const int co = h->cin[i] < 64 ? 64 : h->cin[i];
        const void *mask = (i > 0 && !kPoolBefore[i]) ? h->act[0][i - 1] : nullptr;
        void *dst = h->grad[pp];
        pp ^= 1;
        if (i == 0 && im2col && first_bwd_fused && h->cout[0] == 64) {   // the same two steps in one kernel, the im2col gradient rows in LDS (k_conv1_1_bwd_image)
            if ((rc = gom_lpips_fold_values(B, h->head_sums, head_blocks, value_partials, stream))) return rc;
            return gom_conv1_1_bwd_image_planes(B, H, W, g, h->w1_bwd, d_pred, h->lo_g, stream);
        }
        if (i == 0 && im2col) {   // d(im2col rows) = W^T dY as a 1 x 1 convolution 64 -> 32; the col2im gather rides in the unprepare kernel
            if ((rc = gom_conv1x1_planes((size_t)B * hh * ww, h->cout[0], 32, g, h->w1_bwd, nullptr, dst, 0, h->lo_g, h->lo_g, stream))) return rc;
            if ((rc = gom_lpips_fold_values(B, h->head_sums, head_blocks, value_partials, stream))) return rc;
            return gom_lpips_unprepare_col2im_planes(B, H, W, dst, d_pred, h->lo_g, stream);
        }
        if ((rc = lp_conv(h->splitk, B, hh, ww, h->cout[i], co, g, h->w_bwd[i], nullptr, mask, dst, 0, h->lo_g, h->lo_g, stream))) return rc;
        g = dst;
    }
    if ((rc = gom_lpips_fold_values(B, h->head_sums, head_blocks, value_partials, stream))) return rc;
    return gom_lpips_unprepare_planes(B, H, W, 64, g, d_pred, h->lo_g, stream);
}
