/*
 * gom_hip.h -- C ABI of libgom_hip.so, the MI355X (gfx950) implementation of
 * the GoMAvatar per-frame hot path.
 *
 * Plain C: device pointers + sizes + a hipStream_t passed as void*.  No torch
 * types cross this boundary.  Every entry point only ENQUEUES work on the given
 * stream (no host synchronisation, no allocation in the steady state), so a
 * whole frame is hipGraph-capturable.  All functions return 0 on success and
 * a negative code on failure; gom_last_error() then returns a description
 * (thread-local).
 *
 * Each entry point names the reference interface it replaces
 * (paths relative to the wenj/GoMAvatar tree):
 *
 *   gom_raster_forward / gom_raster_backward
 *       diff_gaussian_rasterization._C.rasterize_gaussians(_backward), i.e. the
 *       CUDA extension behind `GaussianRasterizer.forward`, called from
 *       models/modules/renderer/gaussian.py:83-91 (settings built at :53-67).
 *   gom_fk_forward / gom_fk_backward
 *       utils/body_util.py:612-638  get_global_RTs  (+ :591-609).
 *   gom_lbs_forward, gom_vertex_backward
 *       utils/body_util.py:641-644  apply_lbs.
 *   gom_face_forward / gom_face_backward
 *       models/model.py:225-234 (centroid, so3_exp_map, Steiner frame :27-41,
 *       covariance) + gaussian.py:71-75 (6-pack).
 *   gom_l1_loss
 *       train.py:53-55 (unpack) + train.py:101-111 (L1 rgb, L1 mask) and their
 *       autograd backward.
 *   gom_conv3x3_bf16, gom_maxpool2x2_*, gom_lpips_prepare_bf16, gom_lpips_layer_*_nhwc_bf16
 *       utils/lpips/pretrained_networks.py:96-134 (VGG16 trunk) + utils/lpips/lpips.py:81-133 on the matrix cores.
 *   gom_mesh_raster_forward / gom_mesh_raster_backward
 *       models/modules/renderer/mesh.py:65-128 (PyTorch3D MeshRasterizer + NormalShader + SoftSilhouetteShader).
 *   gom_mesh_laplacian, gom_mesh_normal_consistency, gom_mesh_color_consistency (+ _backward)
 *       train.py:123-160 (PyTorch3D mesh_laplacian_smoothing / mesh_normal_consistency, network_util.py:795-799).
 *   gom_ssim
 *       eval.py:106-108 (skimage structural_similarity, multichannel) and eval.py:157 (torchmetrics SSIM).
 *   gom_lpips_layer_forward / gom_lpips_layer_backward
 *       utils/lpips/lpips.py:104-115 (normalize_tensor, squared difference, lin layer, spatial average;
 *       utils/lpips/__init__.py:40-42) for one VGG tap, called from train.py:113-121.
 */
#ifndef GOM_HIP_H
#define GOM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GOM_ABI_VERSION 11

/* Camera of one rasterizer call: the 12 fields of GaussianRasterizationSettings
 * that matter on this path (gaussian.py:53-66).  view/proj are the 16 floats of
 * the row-major tensors the reference passes (viewmatrix = E^T,
 * projmatrix = E^T K_ndc^T). bg: up to 4 channels (only the first C are read). */
typedef struct GomCamera {
    int32_t H, W;
    float tanfovx, tanfovy;
    float view[16];
    float proj[16];
    float bg[4];
} GomCamera;

typedef struct GomState GomState; /* opaque: binning/geometry/image scratch of one in-flight frame */

/* flags for gom_raster_forward */
#define GOM_FWD_REUSE_BINNING 1u /* geometry+camera identical to the previous forward on this state:
                                    skip projection/binning/sort, composite new colours only */

/* buffer ids for gom_state_export (bit-exact parity checks) */
enum {
    GOM_BUF_DEPTH = 0,       /* float  [P]            */
    GOM_BUF_XY = 1,          /* float  [P][2]         */
    GOM_BUF_CONIC_OPACITY = 2, /* float [P][4]        */
    GOM_BUF_TILES_TOUCHED = 3, /* uint32 [P]          */
    GOM_BUF_RECT = 4,        /* uint16 [P][4] xmin ymin xmax ymax (tiles) */
    GOM_BUF_TILE_BASE = 5,   /* uint32 [tiles+1] exclusive scan of per-tile counts = range starts */
    GOM_BUF_KEYS = 6,        /* uint64 [D] sorted (depth_bits<<32 | gaussian) per tile range       */
    GOM_BUF_POINT_LIST = 7,  /* uint32 [D] sorted gaussian ids                                      */
    GOM_BUF_FINAL_T = 8,     /* float  [H][W]         */
    GOM_BUF_N_CONTRIB = 9,   /* uint32 [H][W]         */
    GOM_BUF_STATUS = 10      /* uint32 [4]: num_pairs, overflow, num_segments, - */
};

/* options for gom_state_set_option */
enum {
    GOM_OPT_SORT_CAP = 0,   /* tile-list length sorted in one register/LDS chunk (64..8192, rounded down to a power of
                               two; default 8192); longer lists take the chunked merge path -- lowered by tests */
    GOM_OPT_PAIR_CAPACITY = 1, /* capacity (entries) of the (tile, gaussian) pair buffers */
    GOM_OPT_PROFILE = 2,       /* 1: bracket every raster kernel launch with HIP events on the caller's stream */
    GOM_OPT_SEG_SHIFT = 3,     /* log2 of the tile-list segment size: 7 (128 entries), 8 (256) or 0 = auto (8 for a batched launch and for
                                  one frame with >= 32 Gaussians per tile of the image, else 7).  Results for different sizes agree to fp32 round-off, not bitwise. */
    GOM_OPT_TASK_GRID_PCT = 4, /* 10..100 (default 100): share of the chip the persistent task-queue grids of a batched launch
                                  occupy.  100 is fastest when the step has the GPU to itself; with several steps in flight on
                                  separate streams ~50 lets their kernels run side by side (a full grid holds every workgroup
                                  slot until it ends).  Results do not depend on it. */
    GOM_OPT_BWD_MODE = 6,      /* render backward: 0 = a workgroup replays two consecutive sub-ranges between barriers, every wave taking
                                  diagonally opposite 8x8 quadrants in the two (evens out the quadrant imbalance of a tile); 1 = one
                                  sub-range per barrier (round 1); -1 (default) = auto (0 for a batch, 1 for one frame).  0 and 1 give
                                  bitwise the same gradients.  2 (the (sub-range, 4x4 block) kernel of round 3) and 3 (the records backward
                                  of round 4) were measured slower and exist in -DGOM_LAB builds only (gom_hip_lab.h): refused here. */
    GOM_OPT_SORT_MODE = 5,     /* how the tile lists get their (depth, index) order: 0 = auto, 1 = merge sort per tile, 2 = rank the
                                  frame's Gaussians by depth once, then a linear bitmap pass per tile (auto picks it when the
                                  bitmap of one frame fits comfortably in LDS: up to 2^18 Gaussians per frame).  Bit-identical results. */
    GOM_OPT_FUSE_FACE = 7,     /* frame step (gom_frame_forward_backward / gom_batch_forward_backward) only: 1 (default) = the per-face
                                  Gaussian frame and its backward run inside the rasterizer's per-Gaussian kernels and the kinematic chain
                                  inside the skinning launch (three launches and a round trip of the means / covariances / their gradients
                                  through HBM less); 0 = separate kernels.  Bit-identical results. */
    GOM_OPT_FUSE_LOSS = 8      /* frame step only: 1 (default) = the photometric L1 losses and dL/d(image) are computed inside the forward's
                                  own launches (the workgroup that assembles a tile has its pixels in registers; the workgroups that paint
                                  the empty tiles sum their loss), tile t leaving its sums in slot t of the frame's loss_partials row: one launch
                                  and one read of the image less.  0 = the stand-alone loss kernel (gom_l1_loss's).  The same bits for the
                                  image gradient; the loss VALUE is summed in another order (per tile instead of per strided block). */
};

/* kernel ids for gom_state_kernel_times */
enum {
    GOM_K_PREPROCESS = 0, GOM_K_SCAN = 1, GOM_K_EMIT = 2, GOM_K_SORT = 3, GOM_K_SEG_T = 4, GOM_K_SEG_FWD = 5,
    GOM_K_COMBINE = 6, GOM_K_SEG_BWD = 7, GOM_K_PREPROCESS_BWD = 8,
    GOM_K_DEPTH_HIST = 9,  /* depth ranking: bucket histogram                                   */
    GOM_K_DEPTH_RANK = 10  /* depth ranking: bucket scatter + per-bucket sort (two launches)    */
};
#define GOM_NUM_KERNELS 11

const char *gom_last_error(void);
int gom_abi_version(void);

GomState *gom_state_create(void);          /* on the current HIP device */
void gom_state_destroy(GomState *s);
int gom_state_set_option(GomState *s, int option, int64_t value);
/* Synchronises `stream` and reports the last forward's pair count and whether
 * the pair buffers overflowed (outputs of an overflowed frame are NaN). */
int gom_state_poll(GomState *s, int64_t *num_pairs, int32_t *overflow, void *stream);
/* With GOM_OPT_PROFILE on: synchronises and returns the duration (ms) of each raster kernel's most
 * recent launch on this state, measured with HIP events on the stream it was launched on
 * (-1 for kernels that have not run). ms_out has GOM_NUM_KERNELS entries. */
int gom_state_kernel_times(GomState *s, float *ms_out);
/* Asynchronous copy of an internal buffer into caller device memory. */
int gom_state_export(GomState *s, int buffer_id, void *dst_device, int64_t dst_bytes, void *stream);

/* ---- splat rasterizer (C = 3 or 4 channels) -------------------------------
 * means3D [P][3], cov6 [P][6] (xx xy xz yy yz zz), colors [P][C], opacity [P]
 * out_color [C][H][W], radii [P] int32 (may be NULL). */
int gom_raster_forward(GomState *s, const GomCamera *cam, int P, int C,
                       const float *means3D, const float *cov6, const float *colors, const float *opacity,
                       float *out_color, int32_t *radii, uint32_t flags, void *stream);

/* flags for gom_raster_backward */
#define GOM_BWD_RECOMPUTE_FORWARD 1u /* another forward with different COLOURS ran on this state since the forward
                                        being differentiated (same geometry, GOM_FWD_REUSE_BINNING): re-create the
                                        colour-dependent checkpoints first */

/* dL_dcolor [C][H][W] -> dL_dmeans3D [P][3], dL_dcov6 [P][6], dL_dcolors [P][C],
 * dL_dopacity [P], dL_dmeans2D [P][3] (screen space, z = 0; may be NULL).
 * Must follow the gom_raster_forward on the same state with the same inputs. */
int gom_raster_backward(GomState *s, const GomCamera *cam, int P, int C,
                        const float *means3D, const float *cov6, const float *colors, const float *opacity,
                        const float *dL_dcolor,
                        float *dL_dmeans3D, float *dL_dcov6, float *dL_dcolors, float *dL_dopacity, float *dL_dmeans2D,
                        uint32_t flags, void *stream);

/* gom_raster_forward / gom_raster_backward with the camera read from DEVICE memory by the kernels: nothing of
 * tanfov / view / proj / bg is baked into the launches, so the calls can sit inside a captured hipGraph (the caller's,
 * e.g. a whole training iteration) that is replayed for other cameras -- overwrite *cam_device before the replay.
 * H and W are launch geometry and stay host values; cam_device->H / W must hold the same numbers.
 * (The reference reads its camera back to the host with .item() every frame, gaussian.py:30-31.) */
/* The camera block of renderer/gaussian.py:28-51 from the reference's K (3,3) and E (4,4) DEVICE tensors (fp32, row-major) into *cam_device, one
 * launch, no host read: tanfov = tan(atan(size / 2f)) in fp64, view = E^T, proj = E^T K_ndc^T (znear / zfar as in the reference: 0.001 / 100);
 * bg4 (device, 4 floats) or null = leave cam_device->bg as it is.  Bit for bit gomavatar_amd.camera.camera_block's struct. */
int gom_camera_update_device(const float *K, const float *E, int H, int W, double znear, double zfar, const float *bg4, GomCamera *cam_device, void *stream);
int gom_raster_forward_dcam(GomState *s, int H, int W, const GomCamera *cam_device, int P, int C,
                            const float *means3D, const float *cov6, const float *colors, const float *opacity,
                            float *out_color, int32_t *radii, uint32_t flags, void *stream);
int gom_raster_backward_dcam(GomState *s, int H, int W, const GomCamera *cam_device, int P, int C,
                             const float *means3D, const float *cov6, const float *colors, const float *opacity,
                             const float *dL_dcolor,
                             float *dL_dmeans3D, float *dL_dcov6, float *dL_dcolors, float *dL_dopacity, float *dL_dmeans2D,
                             uint32_t flags, void *stream);

/* ---- positional encoding of the shadow MLP's input (models/modules/shadow_module.py:96-97, utils/network_util.py get_embedder) --
 * x [n][3] -> out [n][3 + 6 L] = [x, sin(2^0 x), cos(2^0 x), ..., sin(2^(L-1) x), cos(2^(L-1) x)];  backward: g_out -> dx [n][3]. */
int gom_posenc_forward(int64_t n, int L, const float *x, float *out, void *stream);
int gom_posenc_backward(int64_t n, int L, const float *x, const float *g_out, float *dx, void *stream);

/* ---- weight / bias gradient of a Linear layer with a long batch dimension (the shadow MLP, models/modules/shadow_module.py:66-117:
 * one row per pixel under the mesh): dW [out][in] = dY^T X, db [out] = column sums of dY (may be NULL); X [n][in], dY [n][out],
 * in, out <= 128.  workspace: gom_linear_wgrad_slices() * 129 * 128 floats.  Rows are split over the workgroups, partials summed
 * in a fixed order. */
int gom_linear_wgrad_slices(void);
int gom_linear_wgrad(int64_t n, int in_dim, int out_dim, const float *X, const float *dY, float *dW, float *db, float *workspace, void *stream);

/* ---- image composition around the rasterizer's output, one launch each way ------------------------------------------------
 * compose (models/model.py:262-287): img (4,H,W) = albedo rgb + alpha, shade (H,W) or NULL -> albedo (H,W,3), mask (H,W),
 *   rgb (H,W,3) = albedo * shade (rgb may be NULL).  backward: any of d_albedo / d_mask / d_rgb may be NULL (= zero);
 *   d_img (4,H,W), d_shade (H,W) or NULL.
 * unpack (train.py:53-55): out = rgb * mask + bg * (1 - mask) for B images, rgb (B,H,W,3), mask (B,H,W), bg (B,3).
 *   backward: g (B,H,W,3) -> d_rgb, d_mask (no gradient for the background colour: data). */
int gom_compose_forward(int H, int W, const float *img, const float *shade, float *albedo, float *mask, float *rgb, void *stream);
int gom_compose_backward(int H, int W, const float *img, const float *shade, const float *d_albedo, const float *d_mask, const float *d_rgb,
                         float *d_img, float *d_shade, void *stream);
int gom_unpack_forward(int B, int H, int W, const float *rgb, const float *mask, const float *bg, float *out, void *stream);
int gom_unpack_backward(int B, int H, int W, const float *rgb, const float *mask, const float *bg, const float *g, float *d_rgb, float *d_mask,
                        void *stream);

/* ---- the three "mean |a - b|" terms of compute_loss on unpacked images (train.py:101-111: rgb (H,W,3) and mask (H,W) against
 * their targets; train.py:141-149: normal mask (H,W) against the dil_k x dil_k max-pool dilation of the target mask, dil_k odd or
 * <= 1 for none).  A null prediction switches its term off (its output is 0).  forward: out3 = the three means, partials
 * [4 * GOM_LOSS_BLOCKS][3] scratch.  backward: g3 = dL/d out3 (device), d_* = g * sign(a - b) / count, like torch's abs / mean. */
int gom_l1_terms_forward(int H, int W, const float *rgb, const float *rgb_gt, const float *mask, const float *mask_gt,
                         const float *normal_mask, int dil_k, float *out3, float *partials, void *stream);
int gom_l1_terms_backward(int H, int W, const float *rgb, const float *rgb_gt, const float *mask, const float *mask_gt,
                          const float *normal_mask, int dil_k, const float *g3, float *d_rgb, float *d_mask, float *d_normal_mask, void *stream);

/* The tail of compute_loss (train.py:98-163) in one launch: input i is a rows[i] x cols[i] matrix of partial sums (device, row-major) whose first
 * used[i] rows are loss terms: term = pre[i] * sum(row).  coeffs (device, one per term, in input order) -> vec[k] = the terms, scaled[k] =
 * coeffs[k] * vec[k], total = sum of scaled in term order.  Up to 8 inputs, 64 terms; ptrs / rows / cols / used / pre are host arrays. */
int gom_loss_tail(int n_inputs, const float *const *ptrs, const int32_t *rows, const int32_t *cols, const int32_t *used, const float *pre,
                  const float *coeffs, float *vec, float *scaled, float *total, void *stream);

/* ---- the shadow MLP at its default shape (shadow_module.py:66-117 with mlp_depth 3: D0 -> H -> H -> H -> 1, ReLU x 3, sigmoid),
 * D0, H <= 128, nn.Linear weight layout [out][in].  forward: x [n][D0] -> h1, h2, h3 [n][H] (post-ReLU, kept for the backward) and
 * out [n].  backward: g [n] = dL/d out -> dz4 [n], dz3, dz2, dz1 [n][H] (the dY of every layer: feed them to gom_linear_wgrad with
 * X = h3, h2, h1, x) and dx [n][D0]. */
int gom_mlp3_forward(int64_t n, int D0, int H, const float *x, const float *W1, const float *b1, const float *W2, const float *b2,
                     const float *W3, const float *b3, const float *w4, const float *b4, float *h1, float *h2, float *h3, float *out, void *stream);
int gom_mlp3_backward(int64_t n, int D0, int H, const float *g, const float *out, const float *h1, const float *h2, const float *h3,
                      const float *W1, const float *W2, const float *W3, const float *w4, float *dz4, float *dz3, float *dz2, float *dz1,
                      float *dx, void *stream);
/* the weight / bias gradients of all four layers from what gom_mlp3_forward / _backward left behind (x, h1..h3, dz1..dz4), two
 * launches instead of four gom_linear_wgrad calls; workspace: 4 * gom_linear_wgrad_slices() * 129 * 128 floats. */
int gom_mlp3_wgrad(int64_t n, int D0, int H, const float *x, const float *h1, const float *h2, const float *h3, const float *dz1, const float *dz2,
                   const float *dz3, const float *dz4, float *dW1, float *db1, float *dW2, float *db2, float *dW3, float *db3, float *dW4, float *db4,
                   float *workspace, void *stream);

/* ---- skeleton + skinning ----------------------------------------------------
 * cnl_gtfms [24][4][4], dst_Rs [24][3][3], dst_Ts [24][3] -> RT [24][12]
 * (row-major 3x3 R then T).  fk_save [24][32] keeps the chain for backward. */
int gom_fk_forward(const float *cnl_gtfms, const float *dst_Rs, const float *dst_Ts, float *RT, float *fk_save, void *stream);
int gom_fk_backward(const float *dst_Rs, const float *dst_Ts, const float *fk_save, const float *dRT,
                    float *d_dst_Rs, float *d_dst_Ts, void *stream);
/* xyz [3][N] channel-first, weights [J+1][N] (row J = background, ignored), RT [J][12] -> out [3][N] */
int gom_lbs_forward(int N, int J, const float *xyz, const float *weights, const float *RT, float *out, void *stream);
/* Both of the above for one frame in ONE launch (ABI 10): utils/body_util.py:612-644 as models/model.py:213-216 calls them back to back.
 * Same bits as gom_fk_forward followed by gom_lbs_forward with J = 24; RT and fk_save are written as by gom_fk_forward (for the backward). */
int gom_fk_lbs_forward(int N, const float *cnl_gtfms, const float *dst_Rs, const float *dst_Ts, const float *xyz, const float *weights,
                       float *RT, float *fk_save, float *out, void *stream);

/* ---- per-face Gaussians -----------------------------------------------------
 * verts [3][N], faces [F][3] int32, so3 [3][F], scale [3][F], sigma ->
 * xyz [F][3], cov6 [F][6].  Optional (both or neither): appearance [3][F] ->
 * feat4 [F][4] = (r, g, b, 1), the rasterizer's colour rows (gaussian.py:49). */
int gom_face_forward(int N, int F, const float *verts, const int32_t *faces, const float *so3, const float *scale,
                     float sigma, float *xyz, float *cov6, const float *appearance, float *feat4, void *stream);
/* d_xyz [F][3], d_cov6 [F][6] -> d_corner [F][3][3] (per face corner, xyz), d_so3 [3][F], d_scale [3][F].
 * Optional (both or neither): d_feat4 [F][4] -> d_appearance [3][F]. */
int gom_face_backward(int N, int F, const float *verts, const int32_t *faces, const float *so3, const float *scale,
                      float sigma, const float *d_xyz, const float *d_cov6,
                      float *d_corner, float *d_so3, float *d_scale, const float *d_feat4, float *d_appearance, void *stream);
/* Gathers per-corner gradients onto vertices through the CSR vertex->corner
 * adjacency (csr_off [N+1], csr_idx [3F] = face*3+corner; no atomics), adds
 * d_verts_extra [3][N] (may be NULL), and applies the LBS backward:
 * d_xyz [3][N] = sum_j w_j R_j^T g.  If dRT != NULL also writes
 * dRT [J][12] = sum_n w_jn (g_n x_n^T | g_n) (overwritten; fixed summation order, no atomics). */
int gom_vertex_backward(int N, int J, const float *xyz, const float *weights, const float *RT,
                        const int32_t *csr_off, const int32_t *csr_idx, const float *d_corner, const float *d_verts_extra,
                        float *d_verts_obs, float *d_xyz, float *dRT, void *stream);

/* ---- photometric L1 losses ---------------------------------------------------
 * pred [4][H][W] (albedo rgb + alpha, the rasterizer's CHW output), shade [H][W] or NULL,
 * gt_rgb [H][W][3], gt_mask [H][W], bg [3].
 * rgb' = albedo*shade*alpha + bg*(1-alpha);  L = c_rgb*mean|rgb'-gt| + c_mask*mean|alpha-gt_mask|.
 * Writes dL_dpred [4][H][W] (scaled by grad_scale), dL_dshade [H][W] (may be NULL) and
 * loss_partials [GOM_LOSS_BLOCKS][2] (per-block sums of |rgb'-gt| and |alpha-gt_mask|; the
 * caller reduces them -- the gradient does not depend on the loss value). */
#define GOM_LOSS_BLOCKS 256
int gom_l1_loss(int H, int W, const float *pred, const float *shade, const float *gt_rgb, const float *gt_mask, const float *bg,
                float c_rgb, float c_mask, float grad_scale,
                float *dL_dpred, float *dL_dshade, float *loss_partials, void *stream);

/* ---- LPIPS head, one feature tap per call (utils/lpips/lpips.py:104-115) -----------------------------------------
 * f0 (prediction branch), f1 (target branch): [B][C][HW] fp32 feature maps of the VGG trunk; w [C] = the tap's
 * NetLinLayer weights (1x1 conv, no bias; Dropout is inert in eval mode, lpips.py:78-79).
 *   value_b = mean_hw sum_c w_c (f0_c/m0 - f1_c/m1)^2,  m = sqrt(sum_c f_c^2 + 1e-10) + 1e-10   (normalize_tensor)
 * forward: partials [B][GOM_LOSS_BLOCKS], value_b = sum of row b (no atomics; the caller reduces).
 * backward: d_f0 [B][C][HW] = grad_out[b] * d value_b / d f0 (grad_out [B] in device memory). */
int gom_lpips_layer_forward(int B, int C, int HW, const float *f0, const float *f1, const float *w, float *partials, void *stream);
int gom_lpips_layer_backward(int B, int C, int HW, const float *f0, const float *f1, const float *w, const float *grad_out,
                             float *d_f0, void *stream);

/* ---- bf16 VGG16 trunk of LPIPS on the matrix cores (utils/lpips/pretrained_networks.py:96-134) ---------------------
 * Activations are NHWC bf16.  gom_conv3x3_bf16: 3x3, stride 1, zero padding 1; Cin multiple of 32, Cout multiple of 64;
 * weights packed [Cin/32][9 taps (ky*3+kx)][Cout][32] bf16; bias fp32 [Cout] or NULL; flags GOM_CONV_RELU;
 * mask (same shape as out) or NULL: out = (mask > 0) ? out : 0 -- the ReLU derivative of the layer below, which makes
 * the same kernel the backward-data convolution when given the 180-degree-rotated, transposed weights.
 * gom_maxpool2x2_*: 2x2 / stride 2 pooling and its backward (x = the pool's input, a post-ReLU activation; the routed
 * gradient is also multiplied by [x > 0]; accumulate != 0: dx += ...).
 * gom_lpips_prepare_bf16: (B,H,W,3) fp32 image in [0,1] -> ((2x-1) - shift)/scale (lpips.py:126-133, train.py:113),
 * NHWC bf16 padded to 32 channels; gom_lpips_unprepare_bf16: gradient wrt that image from the gradient wrt the trunk input.
 * gom_lpips_layer_*_nhwc_bf16: the LPIPS head (see gom_lpips_layer_forward) on NHWC bf16 taps; C = 64 or k*128.  The
 * backward returns the gradient w.r.t. the tap's PRE-ReLU value (d/d f0 times [f0 > 0]): what the backward-data
 * convolution of the layer consumes. */
#define GOM_CONV_RELU 1u
int gom_conv3x3_bf16(int B, int H, int W, int Cin, int Cout, const void *in, const void *wt, const float *bias, const void *mask,
                     void *out, uint32_t flags, void *stream);
/* split-K variant for layers with few pixel tiles: `splits` workgroups share the input channels of an output tile, fp32
 * partial sums go through workspace [splits][B][H][W][Cout] and a second launch applies bias / ReLU / mask.
 * gom_conv3x3_splits returns the split count the library would pick (1 = plain kernel). */
int gom_conv3x3_splits(int B, int H, int W, int Cin, int Cout);
int gom_conv3x3_bf16_splitk(int B, int H, int W, int Cin, int Cout, const void *in, const void *wt, const float *bias, const void *mask,
                            void *out, uint32_t flags, int splits, float *workspace, void *stream);
int gom_maxpool2x2_bf16(int B, int H, int W, int C, const void *x, void *y, void *stream);
int gom_maxpool2x2_backward_bf16(int B, int H, int W, int C, const void *x, const void *dy, void *dx, int accumulate, void *stream);
int gom_lpips_prepare_bf16(int B, int H, int W, const float *rgb, void *out32, void *stream);
int gom_lpips_unprepare_bf16(int B, int H, int W, int Cpad, const void *d_in, float *d_rgb, void *stream);
int gom_lpips_layer_forward_nhwc_bf16(int B, int C, int HW, const void *f0, const void *f1, const float *w, float *partials, void *stream);
int gom_lpips_layer_backward_nhwc_bf16(int B, int C, int HW, const void *f0, const void *f1, const float *w, const float *grad_out,
                                       void *d_f0, void *stream);

/* LPIPS-VGG value and image gradient in one native call (train.py:113-121 and its backward) on the kernels above.
 * create: 13 packed forward / backward-data weight tensors, 13 biases (fp32, padded to cout), 5 lin vectors, padded channel
 * counts cin[13] / cout[13]; all DEVICE pointers that must outlive the handle.
 * value_and_grad: pred, gt (B,H,W,3) fp32 in [0,1], H and W multiples of 16.  value_partials [5][B][GOM_LOSS_BLOCKS]:
 * LPIPS of image b = sum over taps and blocks.  d_pred (B,H,W,3) = grad_scale * d LPIPS_b / d pred (NULL: value only). */
typedef struct GomLpipsVgg GomLpipsVgg;
GomLpipsVgg *gom_lpips_vgg_create(const void *const *w_fwd, const void *const *w_bwd, const float *const *bias, const float *const *lin,
                                  const int32_t *cin, const int32_t *cout);
void gom_lpips_vgg_destroy(GomLpipsVgg *h);
/* Arithmetic of the trunk.  BF16 (default): bf16 activations, fp32 accumulation -- 3 % on the LPIPS value, narrower than the
 * reference's fp32 convolutions (utils/lpips/pretrained_networks.py:96-134 through cuDNN).  BF16X3: every activation, gradient and
 * weight as two bf16 planes (hi + lo = 16 mantissa bits), three MFMA passes per product (hi hi + lo hi + hi lo, fp32 accumulation):
 * the reference's precision (<= 1e-5 relative on the value against fp32 library convolutions) at matrix-core speed.  The handle's
 * weight tensors must then carry 3 x Cin/32 chunks per layer (w_hi, w_hi, w_lo per 32 input channels).  Set before the first call. */
#define GOM_LPIPS_PRECISION_BF16 0
#define GOM_LPIPS_PRECISION_BF16X3 1
int gom_lpips_vgg_set_precision(GomLpipsVgg *h, int32_t precision);
/* The trunk's FIRST layer without its channel padding: conv1_1 (3 input channels) as a 1 x 1 convolution over im2col rows (channel
 * 3 (3 ky + kx) + c of a pixel = its neighbour (ky-1, kx-1), zero outside the image) and its backward-data pass as a 1 x 1 convolution
 * 64 -> 32 + a col2im gather -- 9x / 21x fewer MFMA than the padded 3 x 3 kernels, same products and accumulation
 * (pretrained_networks.py:96-101: `slice1`'s first Conv2d(3, 64, 3, padding=1)).  w1x1_fwd: [chunks][64][32] bf16 with
 * w[co][3 (3 ky + kx) + c] = W[co][c][ky][kx]; w1x1_bwd: [chunks][32][32] per 32 output channels of conv1_1, the transpose; chunks = 1
 * (bf16) or 3 per 32 input channels (bf16x3: hi, hi, lo).  NULL, NULL: back to the padded 3 x 3 path. */
int gom_lpips_vgg_set_first_layer(GomLpipsVgg *h, const void *w1x1_fwd, const void *w1x1_bwd);
#define GOM_LPIPS_USE_GRAPH 1u   /* capture the ~75 launches once per (sizes, pointers) and replay them as one hipGraph */
#define GOM_LPIPS_TARGET_READY 2u /* the target's trunk features are already in the handle (gom_lpips_vgg_target_features at this size, ordered
                                    before this call by the caller): walk the trunk with the prediction alone */
/* The target image's half of the trunk forward on its own: it depends on nothing the frame's forward computes, so a caller can
 * enqueue it on a second stream under the geometry / raster launches of the same iteration (train.py:113-121 evaluates both images
 * where the loss is computed; same arithmetic, earlier).  The features stay in the handle until the next call of either function;
 * the caller orders the two streams (event) both ways: target_features after the previous value_and_grad, value_and_grad after it. */
int gom_lpips_vgg_target_features(GomLpipsVgg *h, int B, int H, int W, const float *gt, void *stream);
int gom_lpips_vgg_value_and_grad(GomLpipsVgg *h, int B, int H, int W, const float *pred, const float *gt, float *value_partials,
                                 float grad_scale, float *d_pred, uint32_t flags, void *stream);

/* ---- mesh normal map + soft silhouette (models/modules/renderer/mesh.py:65-128, called at models/model.py:270-273) -----
 * verts_ndc [N][3]: what utils/pc_util.py:30-46 `ndc_T_world` returns (x, y negated NDC with the shorter image side in
 * [-1,1], z = camera depth); faces [F][3]; vnormals [N][3] (already rotated into the camera frame, model.py:271-273).
 * normal_map [H][W][3] = n0 + n1 + n2 of the nearest face under the pixel, 0 where none (NormalShader + hard_rgb_blend,
 * then x alpha: mesh.py:23-30,121-124).  alpha [H][W] (NULL: evaluation mode) = soft silhouette of
 * MeshRenderer(SoftSilhouetteShader) with blur_radius = ln(1/1e-4 - 1) * cfg.sigma and blend sigma 1e-4 (mesh.py:99-112,127).
 * The state keeps binning, pix_to_face and the per-pixel product for the backward; it is a GomState used for nothing else.
 * backward: d_normal_map [H][W][3], d_alpha [H][W] or NULL -> d_verts_ndc [N][3] (z = 0: depth only orders the faces),
 * d_vnormals [N][3]; csr_off [N+1] / csr_idx [3F] = vertex -> (face*3 + corner). */
int gom_mesh_raster_forward(GomState *s, int N, int F, int H, int W, const float *verts_ndc, const int32_t *faces, const float *vnormals,
                            float blur_radius, float sigma, float *normal_map, float *alpha, void *stream);
int gom_mesh_raster_backward(GomState *s, int N, int F, int H, int W, const int32_t *csr_off, const int32_t *csr_idx, const float *d_normal_map,
                             const float *d_alpha, float *d_verts_ndc, float *d_vnormals, void *stream);
int gom_mesh_pix_to_face(GomState *s, int32_t *dst /*[H][W], -1 = background*/, void *stream);
/* world -> the mesh rasterizer's NDC (utils/pc_util.py:30-46 ndc_T_world) for one frame: verts [3][N] world points, K [9] and
 * E [16] row-major in DEVICE memory -> out [N][3] = (negated NDC x, negated NDC y with the shorter image side in [-1, 1], camera z).
 * backward: d_out [N][3] -> d_verts [3][N] (no gradient for K, E: data in the reference). */
int gom_ndc_from_world_forward(int N, int H, int W, const float *verts, const float *K, const float *E, float *out, void *stream);
int gom_ndc_from_world_backward(int N, int H, int W, const float *verts, const float *K, const float *E, const float *d_out, float *d_verts, void *stream);

/* vertex normals of a mesh (PyTorch3D Meshes.verts_normals_padded, models/model.py:271): verts [N][3], faces [F][3];
 * sums [N][3] = un-normalised sums (kept for the backward), normals [N][3] = R (sums / max(|sums|, 1e-6)) with R [9] a row-major
 * 3x3 in DEVICE memory or NULL for none (models/model.py:272 rotates the normals into the camera frame with E[:3,:3]).
 * backward: d_normals [N][3] -> d_verts [N][3]; d_corner_scratch [F][9]; same R as the forward. */
int gom_vertex_normals_forward(int N, int F, const float *verts, const int32_t *faces, const int32_t *csr_off, const int32_t *csr_idx,
                               const float *R, float *sums, float *normals, void *stream);
int gom_vertex_normals_backward(int N, int F, const float *verts, const int32_t *faces, const int32_t *csr_off, const int32_t *csr_idx,
                                const float *R, const float *sums, const float *d_normals, float *d_corner_scratch, float *d_verts,
                                void *stream);

/* ---- mesh regularisers (train.py:123-160) ------------------------------------------------------------------------------
 * verts [N][3]; nbr_off [N+1] / nbr_idx [2E]: vertex -> neighbouring vertices; pairs [P][2]: faces sharing an edge
 * (models/model.py:115-125); fp_off [F+1] / fp_idx: face -> (pair*2 + side); csr_off / csr_idx: vertex -> face*3 + corner.
 * Forward calls write GOM_LOSS_BLOCKS partial sums (loss = their sum) and the per-element data the backward needs;
 * grad_out is a 1-element DEVICE array (the upstream gradient of the scalar loss).
 *   laplacian:           mean_i || mean_{j in N(i)} v_j - v_i ||     (PyTorch3D mesh_laplacian_smoothing, method="uniform")
 *   normal consistency:  mean_pairs 1 - cos(n_a, n_b)                (PyTorch3D mesh_normal_consistency, oriented manifold mesh)
 *   colour consistency:  mean |c_a - c_b|, colours [3][F]            (utils/network_util.py:795-799) */
int gom_mesh_laplacian(int N, const float *verts, const int32_t *nbr_off, const int32_t *nbr_idx, float *dir /*[N][3]*/, float *partials, void *stream);
int gom_mesh_laplacian_backward(int N, const float *dir, const int32_t *nbr_off, const int32_t *nbr_idx, const float *grad_out, float *d_verts, void *stream);
int gom_mesh_normal_consistency(int P, const int32_t *pairs, const float *verts, const int32_t *faces, float *pair_grad /*[P][2][3]*/, float *partials,
                                void *stream);
int gom_mesh_normal_consistency_backward(int N, int F, int P, const int32_t *fp_off, const int32_t *fp_idx, const float *pair_grad, const float *verts,
                                         const int32_t *faces, const int32_t *csr_off, const int32_t *csr_idx, const float *grad_out,
                                         float *d_corner_scratch /*[F][9]*/, float *d_verts, void *stream);
int gom_mesh_color_consistency(int P, int F, const int32_t *pairs, const float *colors, float *pair_sign /*[P][3]*/, float *partials, void *stream);
int gom_mesh_color_consistency_backward(int F, int P, const int32_t *fp_off, const int32_t *fp_idx, const float *pair_sign, const float *grad_out,
                                        float *d_colors /*[3][F]*/, void *stream);

/* ---- SSIM (evaluation metric; eval.py:106-108,157) --------------------------------------------
 * img0, img1 [H][W][C] fp32; weights [win][win] fp64 window (sums to 1; win odd); the SSIM map is evaluated where the
 * window lies inside the image.  partials [GOM_LOSS_BLOCKS] fp64: their sum / ((H-win+1)(W-win+1)C) is the mean SSIM.
 *   skimage 0.18 (eval.py:107): win 7 uniform, cov_norm 49/48, C1 = (0.01*2)^2, C2 = (0.03*2)^2 (float input => data_range 2)
 *   torchmetrics (eval.py:157): win 11 gaussian sigma 1.5, cov_norm 1, C1 = 0.01^2, C2 = 0.03^2 */
int gom_ssim(int H, int W, int C, const float *img0, const float *img1, int win, const double *weights, double cov_norm, double C1,
             double C2, double *partials, void *stream);

/* ---- whole frame ---------------------------------------------------------------
 * The per-frame hot path as ONE call: FK -> LBS -> per-face Gaussians -> splat forward (4 channels) -> fused
 * unpack + L1 losses (forward and backward) -> splat backward -> face backward -> vertex gather + LBS backward
 * (reference models/model.py:213-250, renderer/gaussian.py:22-100, train.py:53-55,101-111 and their autograd
 * backward).  11 kernel launches (12 for a batch: + the frame sum; one more with GOM_OPT_FUSE_LOSS 0, four more with GOM_OPT_FUSE_FACE 0) enqueued back to back from native code (the kinematic chain runs inside the skinning launch, the
 * per-face frame, the depth histogram and the frame's backward inside the rasterizer's per-Gaussian kernels: GOM_OPT_FUSE_FACE); every pointer is caller-owned device
 * memory, `work_*` are scratch tensors of the stated sizes. */
typedef struct GomFrame {
    int32_t N, F, H, W;                 /* vertices, faces (= Gaussians), image size; 24 joints                  */
    float sigma, c_rgb, c_mask;         /* Steiner normal thickness; loss coefficients                          */
    GomCamera cam;
    /* mesh topology (static until subdivide) */
    const int32_t *faces;               /* [F][3]                                                               */
    const int32_t *csr_off, *csr_idx;   /* [N+1], [3F] vertex -> face*3+corner                                  */
    const float *lbs_weights;           /* [25][N]                                                              */
    /* parameters (reference layouts) */
    const float *vertices, *so3, *scale, *appearance;   /* [3][N] [3][F] [3][F] [3][F]                          */
    /* per-frame inputs */
    const float *cnl_gtfms, *dst_Rs, *dst_Ts;           /* [24][4][4] [24][3][3] [24][3]                        */
    const float *gt_rgb, *gt_mask, *bgcolor;            /* [H][W][3] [H][W] [3]                                 */
    /* outputs */
    float *image;                       /* [4][H][W] albedo rgb + alpha                                         */
    float *loss_partials;               /* [gom_frame_loss_slots(H, W)][2]: the caller sums the slots (ABI 9)   */
    float *g_vertices, *g_so3, *g_scale, *g_appearance; /* gradients, same layouts as the parameters            */
    /* scratch */
    float *work_RT, *work_fk;           /* [24][12], [24][32]                                                   */
    float *work_vobs;                   /* [3][N] posed vertices (also an output for callers that need it)      */
    float *work_xyz, *work_cov6, *work_feat, *work_opacity;   /* [F][3] [F][6] [F][4] [F] (opacity preset to 1) */
    float *work_dimage, *work_dxyz, *work_dcov6, *work_dfeat, *work_dopacity, *work_dcorner; /* [4][H][W] [F][3] [F][6] [F][4] [F] [F][9] */
    int32_t *work_radii;                /* [F]                                                                  */
} GomFrame;

#define GOM_FRAME_FORWARD_ONLY 1u
#define GOM_FRAME_BACKWARD_ONLY 4u  /* second half of a split call: the previous call on this state was the same frame with
                                       GOM_FRAME_FORWARD_ONLY (which also leaves d(L1 losses)/d(image) in work_dimage).  In between the
                                       caller may ADD any other image-space gradient to work_dimage -- that is how LPIPS
                                       (gom_lpips_vgg_value_and_grad on the unpacked image, train.py:113-121) joins the native path.
                                       work_dimage is a gradient for the BACKWARD: the pixels of tiles no Gaussian touches are never read
                                       by it and the loss kernel leaves them as they are (allocate the buffer zeroed; adding to it is fine) */
#define GOM_FRAME_USE_GRAPH 2u      /* capture the launch sequence of this exact GomFrame (all pointers/sizes equal) into a
                                       hipGraph on first use and replay it afterwards: one submission instead of 12.  A recording is
                                       dropped (and made again at its next use) when a buffer of the state is re-allocated -- a larger
                                       frame through the same state -- or one of its options changes */
/* Slots (pairs of floats) per frame of GomFrame.loss_partials: max(GOM_LOSS_BLOCKS, 16x16 tiles of the image).  With GOM_OPT_FUSE_LOSS the
 * loss rides in the forward's launches and tile t leaves its sums in slot t; the stand-alone kernel fills the first GOM_LOSS_BLOCKS slots
 * and zeroes the rest.  Either way the loss value is the sum over a frame's slots. */
int gom_frame_loss_slots(int H, int W);
int gom_frame_forward_backward(GomState *s, const GomFrame *f, uint32_t flags, void *stream);

/* B frames in ONE launch sequence (the same 11 kernels, each over all B frames, + one frame sum): the launch-latency- and tail-bound
 * kernels of a single 512x512 frame become B times larger launches, which is what fills 256 CUs.  The reference has
 * batch size 1 (train.py:309-349); a batch here is B frames whose gradients are SUMMED, i.e. one optimizer step on B
 * frames, the same semantics as the frame-parallel all-reduce across GPUs.
 *   - every per-frame input, output and scratch pointer of `f` (cnl_gtfms, dst_Rs, dst_Ts, gt_rgb, gt_mask, bgcolor,
 *     image, loss_partials, work_*) carries a leading dimension B; parameters, topology and the g_* gradient
 *     outputs (sum over the B frames, accumulated in frame order: bitwise reproducible) keep their single-frame shapes;
 *   - `cams_device` is a DEVICE array of B GomCamera (same H, W as f->cam; view/proj/bg/tanfov per frame): a replayed
 *     hipGraph picks up new cameras without re-capture.  f->cam only provides H and W.
 *   - results are bit-identical to B separate gom_frame_forward_backward calls. */
int gom_batch_forward_backward(GomState *s, const GomFrame *f, int32_t B, const GomCamera *cams_device, uint32_t flags, void *stream);

/* ONE step of B = sum(Bs) frames as K CONCURRENT launch sequences (ABI 11): branch k runs frames[k] (a GomFrame as for
 * gom_batch_forward_backward, Bs[k] frames, device cameras cams_device[k]) on ITS OWN state states[k]; branch 0 on `stream`, the others
 * on streams the library owns, between a fork behind everything already enqueued on `stream` and a join; one frame sum over all B
 * frames in frame order (branch 0's frames first) closes the step on `stream`.  Why: the segment kernels of a batched launch are
 * resident grids draining a task queue, and the chip idles behind the last workgroups of each of a step's ~12 launches; side by side, one
 * branch's tails are filled by the other's workgroups (MI355X, 55 104 Gaussians at 512x512, 8 frames: 2 x 4 is the fastest cut).  It stays
 * ONE step -- nothing of the next step starts before this one's gradients are complete, unlike several steps in flight -- i.e. the
 * reference's semantics of one optimizer step per batch (train.py:309-349).
 *   - every branch's g_vertices / g_so3 / g_scale / g_appearance must be the SAME four tensors (the step's gradients: the sum over all B
 *     frames), N and F equal; everything else of a GomFrame is per branch (inputs, image, loss_partials, work_*: leading dimension Bs[k]);
 *   - K <= 4, B <= 16; flags: 0 or GOM_FRAME_USE_GRAPH (fork, branches, join and sum recorded as ONE hipGraph, replayed while the K
 *     descriptors, states and camera arrays stay the same);
 *   - results: images, losses, radii bitwise those of the frames rendered one by one; gradients bitwise those of
 *     gom_batch_forward_backward over the same B frames in the same order. */
int gom_split_forward_backward(GomState *const *states, const GomFrame *frames, int32_t K, const int32_t *Bs, const GomCamera *const *cams_device,
                               uint32_t flags, void *stream);

/* ---- frame-parallel step, behind the gradient -----------------------------------------------------------------
 * Adam on the FLAT parameter buffer: the reference's torch.optim.Adam(param_groups, betas=(0.9, 0.999)) (train.py:263-267,
 * per-group learning rates models/model.py:305-327, decayed by update_lr train.py:166-175) as one launch over the buffer the
 * gradient all-reduce leaves behind.  Segment i = elements [seg_begin[i], seg_begin[i + 1]) with learning rate seg_lr[i] (host
 * arrays, n_segments + 1 bounds); elements outside every segment (padding) are left alone.  `step` counts from 1 (bias correction);
 * the gradient is multiplied by grad_scale first (1 / world size when the collective summed).  No weight decay, no amsgrad.
 * `seg_start` (host array of n_segments, or NULL = all zero): optimizer steps that had been taken when segment i JOINED -- torch.optim.Adam
 * skips a parameter whose .grad is None and counts that parameter's steps from its first gradient, which is how the reference's non-rigid /
 * pose-refinement MLPs behave before their kick_in_iter (models/model.py:193,200; exps/zju-mocap_377.yaml:73,85): segment i's bias
 * corrections use step - seg_start[i]; seg_start[i] < 0 (or >= step) = not joined yet, its elements are left alone like padding. */
#define GOM_ADAM_MAX_SEGMENTS 12
int gom_adam_flat(int64_t n, float *params, const float *grads, float *exp_avg, float *exp_avg_sq, int32_t n_segments,
                  const int64_t *seg_begin, const float *seg_lr, const int64_t *seg_start, int64_t step, float beta1, float beta2, float eps,
                  float grad_scale, void *stream);

/* The same step with the step COUNT in device memory -- step_device[0] = steps taken so far, step_device[1] = 0 (scratch), advanced by the
 * kernel -- and the reference's learning-rate schedule (update_lr: base * 0.1^(iteration / lr_decay_steps); 0 = constant) derived from it
 * on the device: no argument changes from step to step, so the launch can be captured into a hipGraph with the frame step in front of it.
 * step_device == NULL: exactly gom_adam_flat. */
int gom_adam_flat_graphable(int64_t n, float *params, const float *grads, float *exp_avg, float *exp_avg_sq, int32_t n_segments,
                            const int64_t *seg_begin, const float *seg_lr, const int64_t *seg_start, int64_t step, int64_t *step_device,
                            float lr_decay_steps, float beta1, float beta2, float eps, float grad_scale, void *stream);

/* Shading of the pixels under the mesh, fused (models/model.py:279-283: shadow = 2 * shadow_module(normal) for every pixel; the normal map is
 * zero outside the mesh, where the MLP is one constant).  `normal`: (HW, 3); `pos`: (HW) int32 row of a pixel or -1; `pe`: (HW + 1, 3 + 6 L)
 * rows of the positional encoding (shadow_module.py:96-97), row n = the background's; `workspace`: gom_shade_workspace_ints(HW) int32, ZEROED
 * once by the caller, holds the row count n in device memory (no host synchronisation; the *_rows entry points read it and take HW + 1 as the
 * capacity of every row-indexed buffer).  Forward: select -> gom_mlp3_forward_rows -> scatter (shading = scale * out[row | background]);
 * backward: backward_gather (d out rows; the background row sums the pixels outside the mesh in a fixed order) -> gom_mlp3_backward_rows /
 * gom_mlp3_wgrad_rows -> backward_scatter (d normal through the encoding's backward, zero outside the mesh).
 * `pack`: NULL = the fp32 VALU layers of csrc/mlp.hip; a scratch buffer of gom_mlp3_pack_elems() uint16 = the layers on the bf16 matrix cores at
 * fp32 precision (csrc/mlp_mc.hip: hi / lo planes, three MFMA passes per product; D0 <= 64, H = 128 -- other shapes take the VALU kernels). */
int gom_mlp3_pack_elems(void);
int gom_shade_workspace_ints(int64_t HW);
int gom_shade_select(int64_t HW, int L, const float *normal, int32_t *pos, float *pe, int32_t *workspace, void *stream);
int gom_mlp3_forward_rows(int64_t HW, const int32_t *workspace, int D0, int H, const float *x, const float *W1, const float *b1, const float *W2,
                          const float *b2, const float *W3, const float *b3, const float *w4, const float *b4, float *h1, float *h2, float *h3,
                          float *out, uint16_t *pack, void *stream);
int gom_mlp3_backward_rows(int64_t HW, const int32_t *workspace, int D0, int H, const float *g, const float *out, const float *h1, const float *h2,
                           const float *h3, const float *W1, const float *W2, const float *W3, const float *w4, float *dz4, float *dz3, float *dz2,
                           float *dz1, float *dx, uint16_t *pack, void *stream);
int gom_mlp3_wgrad_rows(int64_t HW, const int32_t *workspace, int D0, int H, const float *x, const float *h1, const float *h2, const float *h3,
                        const float *dz1, const float *dz2, const float *dz3, const float *dz4, float *dW1, float *db1, float *dW2, float *db2,
                        float *dW3, float *db3, float *dW4, float *db4, float *wgrad_workspace, void *stream);
int gom_shade_scatter(int64_t HW, const int32_t *pos, const float *out, const int32_t *workspace, float scale, float *shading, void *stream);
int gom_shade_backward_gather(int64_t HW, const int32_t *pos, const float *g, int32_t *workspace, float scale, float *g_rows, void *stream);
int gom_shade_backward_scatter(int64_t HW, int L, const int32_t *pos, const float *normal, const float *dpe, float *d_normal, void *stream);

/* The same Adam step over a LIST of separately allocated tensors -- torch.optim.Adam(Model.get_param_groups()) as the reference builds it
 * (train.py:263-267) in one launch per GOM_ADAM_MULTI_MAX tensors instead of ~35 multi-tensor launches: host arrays of n_tensors device
 * pointers / element counts / learning rates (one per tensor: its group's).  step counts from 1; with step_device != NULL the count lives in
 * device memory (steps taken so far; advanced behind the update) and nothing changes between calls, so the launch can be graph-captured.
 * Arithmetic of torch/optim/adam.py::_single_tensor_adam (no weight decay, no amsgrad, maximize = False); betas and eps arrive as DOUBLES, as
 * torch holds them (1 - beta2 is formed in double and rounded once: 1.f - 0.999f would be 4.7e-5 away from torch's 0.001f). */
#define GOM_ADAM_MULTI_MAX 16
int gom_adam_multi(int32_t n_tensors, float *const *params, const float *const *grads, float *const *exp_avg, float *const *exp_avg_sq, const int64_t *numel,
                   const float *lr, int64_t step, int64_t *step_device, double beta1, double beta2, double eps, void *stream);

/* Direct all-reduce of the flat gradient buffer over peer pointers ("for this latency-bound size use a direct one-/two-shot
 * algorithm, not a ring"): one process per GPU; every rank creates a region, the 64-byte IPC handles are exchanged once (any transport:
 * the process group), and `run` enqueues two kernels that leave scale x (sum over the ranks, in RANK ORDER) in `out` -- the same bits on
 * every rank.  `buffer` is this rank's input (n_floats, device memory inside the region): the backward writes the gradient there.
 * `out` may be that same buffer.  The region is fine-grained device memory (flags are polled inside a running kernel): create fails
 * where that cannot be had -- use the library collective then.
 * Failure is loud, never a hang: a wait gives up after the time limit (30 s unless gom_peer_reduce_set_timeout), the rank that gave up
 * publishes nothing, so its peers give up in turn; gom_peer_reduce_poll (no synchronisation: a pinned host word the kernel writes) is 1
 * from then on and every later run/run_adam/run_zero1 on the handle fails until gom_peer_reduce_reset. */
#define GOM_PEER_MAX_RANKS 16
typedef struct GomPeerReduce GomPeerReduce;
GomPeerReduce *gom_peer_reduce_create(int32_t rank, int32_t world, int64_t n_floats);
int gom_peer_reduce_handle(GomPeerReduce *h, void *handle64);
int gom_peer_reduce_connect(GomPeerReduce *h, const void *handles /* world x 64 bytes, rank order */);
float *gom_peer_reduce_buffer(GomPeerReduce *h);
int gom_peer_reduce_run(GomPeerReduce *h, float *out, float scale, void *stream);
/* The same exchange with the optimizer inside its second kernel: every rank applies gom_adam_flat's step of ITS parameter replica straight from
 * the reduced slices as it reads them (n_floats = the flat parameter count; out may be NULL: the reduced gradient is then not materialised). */
int gom_peer_reduce_run_adam(GomPeerReduce *h, float scale, float *out, float *params, float *exp_avg, float *exp_avg_sq, int32_t n_segments,
                             const int64_t *seg_begin, const float *seg_lr, const int64_t *seg_start, int64_t step, float beta1, float beta2, float eps,
                             void *stream);
/* ZeRO-1: the rank that reduced a slice applies THAT slice's Adam step (moments are touched for the own slice only) and
 * publishes the updated parameters; the second kernel gathers parameters.  Replicas end with the bits gom_peer_reduce_run_adam gives. */
int gom_peer_reduce_run_zero1(GomPeerReduce *h, float scale, float *params, float *exp_avg, float *exp_avg_sq, int32_t n_segments,
                              const int64_t *seg_begin, const float *seg_lr, const int64_t *seg_start, int64_t step, float beta1, float beta2, float eps,
                              void *stream);
int gom_peer_reduce_set_timeout(GomPeerReduce *h, double seconds);
int gom_peer_reduce_poll(GomPeerReduce *h);     /* 0 / 1, non-blocking (one step late at most) */
int gom_peer_reduce_status(GomPeerReduce *h);   /* 0 / 1, synchronises the device */
/* After a timeout, on every rank, between two barriers of the process group; `epoch` = the largest gom_peer_reduce_epoch over the ranks. */
int gom_peer_reduce_reset(GomPeerReduce *h, uint32_t epoch);
uint32_t gom_peer_reduce_epoch(GomPeerReduce *h);
/* Call with the device synchronised and behind a barrier of the process group: no peer may still be reading this rank's region. */
void gom_peer_reduce_destroy(GomPeerReduce *h);

#ifdef __cplusplus
}
#endif
#endif /* GOM_HIP_H */
