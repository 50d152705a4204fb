// Internal definitions shared by the translation units of libgom_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

#include "gom_hip.h"

#define GOM_TILE 16
#define GOM_SORT_CAP_MAX 8192       // tile-list entries sortable in LDS (64 KiB of 64-bit keys)
#define GOM_RANK_WIN 2048           // list positions per work item of k_tile_rank (raster_rank.hip)
#define GOM_SORT_SMALL 2048         // lists up to this length are sorted by the 256-thread instantiation of k_sort
#define GOM_PARTIAL_STRIDE 12       // floats per (tile, gaussian) partial-gradient record (10 used)
#ifndef GOM_SEG
#define GOM_SEG 128                 // smallest tile-list segment (the unit of parallel compositing); GomState::segShift picks 128 or 256
#endif
#define GOM_TPX 256                 // pixels per tile = threads of the per-tile / per-segment workgroups
#ifndef GOM_SEG_GRID
#define GOM_SEG_GRID 1024           // workgroups launched for the segment kernels (grid-stride over segments)
#endif

#ifndef GOM_TQ_SHARDS
#define GOM_TQ_SHARDS 8               // heads per task queue (raster_render.hip: TaskQueue)
#endif
#define GOM_TASK_CTR_WORDS (3 * (32 * GOM_TQ_SHARDS + 32))   // three sharded task queues
// Which segments a queue shard owns.  A workgroup's shard is blockIdx % 8 = the XCD it runs on (workgroups are dealt to the XCDs round-robin),
// and every XCD has its own L2: the segments of ONE tile (consecutive segment numbers) should meet in one of them -- a compositing task reads
// the transmittance rows of every segment in front of it, a backward task the tile's image-gradient / last-contributor rows.  So the shards
// take GROUPS of 2^GOM_TQ_GROUP_LOG2 consecutive segments in turn (0: segment s to shard s % 8, rounds 1-3 until here).  Measured (FETCH_SIZE per
// 8-frame launch, KB): k_seg_bwd_pair 85.7 k at 0, 67.8 k at 3, 61.8 k at 5, 60.5 k at 7; k_seg_fwd 40.5 k -> 35.1 k; durations unchanged up to 6.
#ifndef GOM_TQ_GROUP_LOG2
#define GOM_TQ_GROUP_LOG2 5
#endif
__host__ __device__ inline uint32_t gom_shard_segments(uint32_t nsegs, uint32_t x) {   // how many segments shard x owns
    const uint32_t G = 1u << GOM_TQ_GROUP_LOG2, per = G * GOM_TQ_SHARDS, rem = nsegs % per, lo = x * G;
    return (nsegs / per) * G + (rem > lo ? (rem - lo < G ? rem - lo : G) : 0u);
}
__host__ __device__ inline uint32_t gom_shard_segment(uint32_t x, uint32_t k) {        // its k-th segment
    return (((k >> GOM_TQ_GROUP_LOG2) * GOM_TQ_SHARDS + x) << GOM_TQ_GROUP_LOG2) | (k & ((1u << GOM_TQ_GROUP_LOG2) - 1u));
}
__host__ __device__ inline uint32_t gom_bwd_order_region(uint32_t nsegs) {              // entries of a shard's part of the backward's order table (two per (segment, pair) unit)
    return 4u * ((nsegs + GOM_TQ_SHARDS - 1u) / GOM_TQ_SHARDS + (1u << GOM_TQ_GROUP_LOG2));
}

#define GOM_REC_SHARDS 32            // cursors of the record allocator (one device-scope word takes ~88 atomics per microsecond)
#define GOM_REC_PER_PAIR 8           // record capacity per unit of pair capacity (the metric workload writes ~4.7 records per (tile, entry) pair)
struct GomDevStatus {
    uint32_t num_pairs;
    uint32_t overflow;
    uint32_t num_segs;
    uint32_t pair_cursor;   // allocator for the per-gaussian ranges of pair_pos (mesh rasterizer; reset by the scan kernel)
    // The splat preprocess allocates from 8 cursors, one per eighth of pair_pos, 128 bytes apart: a single device-scope word takes
    // ~88 atomics per microsecond (MI355X_MICROARCH.md), and 1 728 workgroups of a batched launch queued ~20 us on it.
    uint32_t shard_overflow;   // a shard ran past its eighth of the buffer (folded into `overflow` by the scan kernel)
    uint32_t n_work_items;     // length of the work list of k_tile_rank
    uint32_t rec_overflow;     // the forward's per-(pixel, entry) records ran past a shard of their buffer (render backward, records mode): gradients poisoned
    uint32_t pad_[25];
    uint32_t shard_cursor[8][32];   // [shard][0] used
    uint32_t rec_cursor[GOM_REC_SHARDS][32];   // [shard][0] used: allocator of the record regions of the live (segment, sub-range, quadrant) pieces (k_seg_fwd)
};

// The frame step's loss INSIDE the rasterizer's forward (GOM_OPT_FUSE_LOSS, default on): the pixels of a non-empty tile are in the registers of
// the k_combine_fwd workgroup that assembles them -- it writes dL/d(image) next to the image -- and the empty tiles' loss (a function of the
// target and the background alone: 16 bytes per pixel, bandwidth) is summed by rider workgroups of the same launch, dispatched in front of the
// tiles' latency chains and done long before them (in k_emit's painters the same work made that launch 8 us longer at B = 8: its tail is a
// queue for workgroup slots).  Tile t of a frame leaves (sum |rgb|, sum |mask|) in slot t
// of the frame's row of `partials` (gom_frame_loss_slots(H, W) = max(GOM_LOSS_BLOCKS, tiles) slots per frame: the caller sums them, as it sums
// the stand-alone kernel's per-block slots).  One launch (7 us at B = 1, 14 at B = 8) and one read of the image less; no hand-off between
// workgroups (a ticket per workgroup with its device-scope release cost k_combine_fwd 100 us: measured, dropped).
struct GomLossRider {
    const float *gt_rgb;        // [B][HW][3]; null: no rider (the caller launches k_l1_loss)
    const float *gt_mask;       // [B][HW]
    const float *bgcolor;       // [B][3] background of the unpack (train.py:53-55)
    float k_rgb, k_mask;
    float *dpred;               // [B][4][HW]
    float *partials;            // [B][slots][2] (caller)
    int slots;
    int zero_empty;             // GomLossSkip::zero_empty
};

struct GomGraphEntry {
    GomFrame key;
    uint32_t flags;
    int B;
    const GomCamera *cams;
    hipGraph_t graph;
    hipGraphExec_t exec;
    uint64_t last_use;
    uint64_t alloc_gen;   // GomState::allocGen at capture: a recorded launch sequence holds the addresses of the state's buffers
    // host-side state the recorded sequence left behind (a replay skips the host code that derives it; a later stand-alone call on
    // the state -- gom_raster_backward, REUSE_BINNING -- must see what the replayed forward really did)
    int gx, gy, segShift;
    bool rankSort, bwdOrderReady, recCounts, recForward;
};

// gom_split_forward_backward: one recorded graph of K concurrent launch sequences (kept by states[0])
#define GOM_SPLIT_MAX 4
#define GOM_SPLIT_MAX_FRAMES 16
#define GOM_FRAME_NO_SUM 0x40000000u   // (internal) frame_enqueue: per-frame gradient slices even for one frame, no frame sum, no optimizer
struct GomSplitHostState { int gx, gy, segShift; bool rankSort, bwdOrderReady, recCounts, recForward; };
struct GomSplitGraphEntry {
    int K;
    uint32_t flags;
    GomState *states[GOM_SPLIT_MAX];
    uint64_t uids[GOM_SPLIT_MAX];      // GomState::uid of the branch states at capture: a destroyed state's address may be handed out again
    GomFrame keys[GOM_SPLIT_MAX];
    int Bs[GOM_SPLIT_MAX];
    const GomCamera *cams[GOM_SPLIT_MAX];
    uint64_t alloc_gen[GOM_SPLIT_MAX];
    GomSplitHostState host[GOM_SPLIT_MAX];
    hipGraph_t graph;
    hipGraphExec_t exec;
    uint64_t last_use;
};

struct GomState {
    int device = 0;
    uint64_t uid = 0;                 // unique per created state (gom_state_create): what a recording made with OTHER states remembers them by
    // capacities
    int capP = 0, capTiles = 0, capPix = 0;
    int64_t capPairs = 0;
    int64_t wantPairs = 0;           // user-set pair capacity (0 = auto)
    int sortCap = GOM_SORT_CAP_MAX;
    // last forward (P, H, W, gx, gy are PER FRAME; a batched launch stacks B frames: B*P Gaussians on a gx x B*gy tile grid)
    int P = 0, H = 0, W = 0, C = 0, gx = 0, gy = 0;
    int B = 1;
    int wantSegShift = 0;             // GOM_OPT_SEG_SHIFT (0 = auto)
    int taskGridPct = 100;            // GOM_OPT_TASK_GRID_PCT
    bool lossSkip = true;             // the frame step's loss kernel skips loads and stores of empty tiles (GomLossSkip)
    bool sortSplit = false;           // set by the mesh rasterizer around its per-tile sort: keys are face indices alone (gom_launch_sort)
    bool fuseLoss = true;             // GOM_OPT_FUSE_LOSS: the frame step's loss rides in k_emit (empty tiles) and k_combine_fwd (the others): GomLossRider
    GomLossRider lossRider{};         // set by the frame step around its forward
    bool emptyFilled = false;         // this forward's k_emit has painted the empty tiles of the image k_combine_fwd is about to write
    bool bwdOrder = true;             // development switch: cost-ordered backward queue in the frame step
    bool fuseFace = true;             // GOM_OPT_FUSE_FACE: the frame step builds / differentiates the per-face frame inside k_preprocess / k_preprocess_bwd
    int bwdMode = -1;                 // GOM_OPT_BWD_MODE: 3 = lane per (pixel, entry) record the forward left (round 4); 0 = two sub-ranges between barriers
                                      // with opposite quadrants per wave, 1 = one sub-range per barrier (round 1); -1 = auto
    bool recCounts = false;           // the current binning's transmittance pre-pass counted the pieces' records (piece_ub)
    bool recForward = false;          // the checkpoints of the last render forward are those of the records mode (records + inclusive sub_C rows)
    int segShift = 7;                 // log2 of the segment size of the current binning: 7 for one frame, 8 for a batch
    const GomCamera *cams = nullptr;  // device array of B cameras for a batched launch; nullptr: the by-value camera
    bool haveForward = false;
    // mesh normal / silhouette rasterizer (mesh_raster.hip) on this state: per-face geometry + per-face gradients
    float *mesh_face = nullptr;
    size_t capMeshFace = 0;
    float meshBlurRadius = 0.f, meshSigma = 1e-4f;
    bool meshForward = false;
    // per-frame parameter gradients of a batched frame call, summed by k_sum_frames
    float *batch_grads = nullptr;
    size_t capBatchGrads = 0;
    // per gaussian
    float *depth = nullptr;
    float2 *xy = nullptr;
    float4 *conic_opacity = nullptr;
    uint32_t *tiles_touched = nullptr;
    ushort4 *rect = nullptr;
    int32_t *radii = nullptr;
    // per tile
    uint32_t *tile_count = nullptr;
    uint32_t *tile_base = nullptr;    // [tiles+1]
    uint32_t *tile_cursor = nullptr;
    uint32_t *tile_nmax = nullptr;    // max n_contrib over the tile's pixels (entries beyond it are dead for backward)
    uint32_t *tile_qlim = nullptr;    // 1 + packed depth rank of the tile's last contributing entry (0: none): liveness test of the per-Gaussian backward
    uint32_t *work_items = nullptr;   // [tiles + capPairs / GOM_RANK_WIN] (tile | window << 24) items of k_tile_rank, listed by the scan kernel
    int64_t capItems = 0;
    uint32_t *seg_base = nullptr;     // [tiles+1] exclusive scan of ceil(count/GOM_SEG)
    // per gaussian: start of its private range in pair_pos
    uint32_t *pair_off = nullptr;
    // per pair
    uint64_t *keys = nullptr;
    uint32_t *point_list = nullptr;
    uint32_t *pair_pos = nullptr;     // [capPairs] sorted position of (gaussian, k-th tile of its rect)
    float *partial = nullptr;         // [capPairs][GOM_PARTIAL_STRIDE] gradient records.  Splat path: GAUSSIAN-major (record of the k-th tile of
                                      // Gaussian g at pair_off[g] + k: the per-Gaussian backward streams them); mesh path: list order
    uint32_t *ent_slot = nullptr;     // [capPairs] list order: record slot of the entry (= pair_off[g] + k)
    // per segment (x 256 pixels of the tile, quadrant-major)
    int64_t capSegs = 0;
    uint4 *seg_qmax = nullptr;        // [capSegs] max n_contrib over each 8x8 quadrant of the segment's tile (combine pass, for the backward)
    uint32_t *seg_cost = nullptr;     // [capSegs][4 sub-ranges][4 quadrants] entries that survived the cull in the pieces k_seg_fwd found alive = cost estimate of the backward's tasks
    uint32_t *bwd_order = nullptr;    // the backward's tasks per queue shard, most expensive first (riders of the loss kernel): GOM_BWD_ORDER_* below
    bool rideBwdOrder = false;        // set by the frame step around its forward: k_combine_fwd carries the riders
    bool bwdOrderReady = false;       // bwd_order belongs to the forward that has just run (frame step, batched launches)
    const uint32_t *rank_minmax = nullptr;   // where the depth ranking of the current forward takes its depth range from: depth_minmax (per
    int rank_blocks = 0;                     //   block of k_preprocess) or the frame step's vertex ranges (GomFaceArgs::vdepth_minmax)
    uint32_t *vdepth_minmax = nullptr;       // [frames][skinning blocks][2], frame step
    size_t capVdepth = 0;
    uint32_t *big_list = nullptr;     // [frames][GOM_BIG_CAP] the Gaussians of each frame that touch more than GOM_BIG_NT tiles (k_preprocess_bwd gives each a whole wave)
    uint32_t *big_count = nullptr;    // [2][frames] their number per frame: [0] published by the scan kernel, [1] being counted by k_preprocess
    int capBigFrames = 0;
    uint4 *seg_desc = nullptr;        // [capSegs] {tile, first list position, entries, index of the segment inside its tile}
    float2 *ent_geo = nullptr;        // [capPairs][3] list-ordered geometry of the entries: (x,y) (A,B) (Cq,lo) = the conic and the opacity pre-scaled for alpha_eval (entry_record.hpp)
    float *ent_col = nullptr;         // [capPairs][4] list-ordered colours
    float *seg_T = nullptr;           // [capSegs][256]     product of (1-alpha) over the segment
    float *seg_C = nullptr;           // [capSegs][4][256]  colour the segment adds to the pixel
    uint32_t *seg_last = nullptr;     // [capSegs][256]     1+list index of the last contributing entry (0: none)
    float *seg_Tend = nullptr;        // [capSegs][256]     transmittance after the segment (combine pass)
    float *seg_Sbehind = nullptr;     // [capSegs][4][256]  colour still to come behind the segment
    // per 32-entry sub-range of a segment (4 per segment) x 256 pixels
    unsigned long long *cull_masks = nullptr;   // [capSegs][4 sub-ranges][4 quadrants] the entries of a sub-range that can reach the quadrant at all (k_seg_T; read by the passes behind it)
    float *sub_T = nullptr;           // [capSegs][4][256]     product of (1-alpha) over the sub-range
    float *sub_C = nullptr;           // [capSegs][4][4][256]  colour the sub-range really added to the pixel
    float *sub_Tend = nullptr;        // [capSegs][4][256]     transmittance behind the sub-range
    // records mode of the render backward: per (segment, sub-range, quadrant) piece and per blending (pixel, entry) pair
    uint32_t *piece_ub = nullptr;     // [capSegs][4][4]  lanes x surviving entries with alpha > 0 (k_seg_T): upper bound of the piece's records
    uint2 *piece_rec = nullptr;       // [capSegs][4][4]  (first record, number of records) of a piece k_seg_fwd found alive
    uint8_t *piece_cnt = nullptr;     // [capSegs][4][4][64]  records per entry of the piece (entry-major inside the region: a lane walks its entry's run)
    float2 *rec_ti = nullptr;         // [capRec] (T in front of the entry at the pixel, bits: entry of the sub-range << 6 | pixel of the quadrant)
    float4 *rec_acc = nullptr;        // [capRec] colour the piece had added to the pixel in front of the entry
    int64_t capRec = 0, capPieceSegs = 0;   // (allocated for a state in records mode only: -DGOM_LAB builds)
    // per pixel
    // depth ranking of the splat path (raster_rank.hip)
    bool rankSort = false;            // this binning used it (decided per forward: GOM_OPT_SORT_MODE, P small enough for the LDS bitmap)
    int sortMode = 0;                 // GOM_OPT_SORT_MODE: 0 auto, 1 per-tile merge sort, 2 depth ranking
    int nbShift = 8;                  // log2 of the depth buckets per frame
    int capFrames = 0;
    int64_t capBuckets = 0;
    uint32_t *depth_minmax = nullptr; // [capFrames = frames x preprocess blocks][2] bit patterns of the min / max visible depth of a block
    uint32_t *bucket_count = nullptr, *bucket_base = nullptr, *bucket_cursor = nullptr;   // [capBuckets (+1)]
    uint64_t *bkeys = nullptr, *bkeys_scratch = nullptr;   // [capP] (depth_bits << 32 | index in frame), bucket-major
    float4 *rec_g = nullptr;          // [capP][2] 32-byte record of a Gaussian: (x, y, conic a, b) (conic c, opacity, pair_off, rect x0 | width << 10 | y0 << 20)
    uint32_t *order = nullptr;        // [capP] Gaussian at packed depth rank q
    uint32_t *rank_of = nullptr;      // [capP] packed rank of a (visible) Gaussian
    uint32_t *keys32 = nullptr;       // [capPairs] emitted ranks, tile-major
    float *final_T = nullptr;
    uint32_t *n_contrib = nullptr;
    float *scratch_img = nullptr;     // [4][capPix] image sink when the backward has to re-create its checkpoints
    GomDevStatus *status = nullptr;
    uint32_t *task_ctr = nullptr;     // heads of the task queues of k_seg_T / k_seg_fwd / k_seg_bwd
    // optional per-kernel HIP-event timing (GOM_OPT_PROFILE); events bracket each launch on the caller's stream
    bool profile = false;
    hipEvent_t ev[2 * GOM_NUM_KERNELS] = {};
    bool evValid[GOM_NUM_KERNELS] = {};
    // optimizer step appended to the frame step's launch sequence (gom_state_set_frame_optimizer): Adam on a flat parameter buffer with the
    // step count in device memory, so that it is part of the recorded graph
    struct {
        bool on = false;
        int64_t n = 0;
        float *params = nullptr, *exp_avg = nullptr, *exp_avg_sq = nullptr;
        const float *grads = nullptr;
        int32_t n_segments = 0;
        int64_t seg_begin[GOM_ADAM_MAX_SEGMENTS + 1] = {};
        float seg_lr[GOM_ADAM_MAX_SEGMENTS] = {};
        int64_t *step_device = nullptr;
        float lr_decay_steps = 0.f, beta1 = 0.9f, beta2 = 0.999f, eps = 1e-8f, grad_scale = 1.f;
    } adam;
    // captured whole-frame launch sequences (GOM_FRAME_USE_GRAPH), keyed by the exact frame descriptor
    std::vector<GomGraphEntry> graphs;
    std::vector<GomSplitGraphEntry> splitGraphs;          // recorded split steps this state leads (gom_split_forward_backward)
    hipStream_t splitStreams[GOM_SPLIT_MAX] = {};         // side streams of the split step's branches 1 .. K-1 (created on first use)
    hipEvent_t splitFork = nullptr, splitJoin[GOM_SPLIT_MAX] = {};
    uint64_t graphClock = 0;
    uint64_t allocGen = 0;            // bumped whenever a buffer of the state is re-allocated: older recordings are dropped, not replayed
};

struct GomKernelTimer {
    GomState *s; int k; hipStream_t st;
    GomKernelTimer(GomState *s_, int k_, hipStream_t st_) : s(s_), k(k_), st(st_) {
        if (s->profile) { (void)hipEventRecord(s->ev[2 * k], st); }
    }
    ~GomKernelTimer() {
        if (s->profile) { (void)hipEventRecord(s->ev[2 * k + 1], st); s->evValid[k] = true; }
    }
};

void gom_set_error(const char *fmt, ...);

#define GOM_HIP_CHECK(expr)                                                                       \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess) {                                                                   \
            gom_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return -2;                                                                            \
        }                                                                                         \
    } while (0)

#define GOM_LAUNCH_CHECK()                                                                        \
    do {                                                                                          \
        hipError_t _e = hipGetLastError();                                                        \
        if (_e != hipSuccess) {                                                                   \
            gom_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e), __FILE__, __LINE__); \
            return -3;                                                                            \
        }                                                                                         \
    } while (0)

int gom_ensure_capacity(GomState *s, int P_frame, int H, int W, int B);

// The frame step's geometry around the rasterizer's per-Gaussian kernels (frame_enqueue, gom_api.hip): with it k_preprocess builds the
// Gaussian from its posed triangle itself (means3D / cov6 become outputs) and k_preprocess_bwd runs the frame's backward in the same thread.
struct GomFaceArgs {
    int N;                       // vertices per frame
    const float *verts;          // (B, 3, N) posed vertices
    const int32_t *faces;        // (F, 3)
    const float *so3, *scale;    // (3, F)
    float sigma;
    const float *appearance;     // (3, F)
    float *feat4;                // (B, F, 4) rasterizer features [r g b 1], written by the forward
    const uint32_t *vdepth_minmax;   // (B, vdepth_blocks, 2) view-depth range of the posed vertices per skinning block (k_fk_lbs_fwd): the depth
    int vdepth_blocks;               //   ranking's bucket map comes from it and k_preprocess builds the bucket histogram itself (null: k_depth_hist)
    float *d_corner;             // (B, F, 9)      written by the backward
    float *d_so3, *d_scale, *d_appearance;   // (B, 3, F) per-frame slices
};

// ---- launchers (one per kernel family; defined in the .hip files) ----------
int gom_launch_preprocess(GomState *s, const GomCamera &cam, int P, const float *means3D, const float *cov6,
                          const float *opacity, int32_t *radii_out, hipStream_t st, const GomFaceArgs *face = nullptr);
// Background of the tiles no Gaussian touches, painted by spare blocks of k_emit (see there); first_block = blocks that emit.
// A Gaussian close to the camera touches a hundred tiles; its gradient records are summed by one WAVE (rider blocks of k_preprocess_bwd)
// instead of one lane walking them while the kernel waits (13 of its 62 us on the metric workload).
#define GOM_BIG_NT 32u
#define GOM_BIG_CAP 512u    // per FRAME (a frame with more keeps the per-lane walk: the choice never depends on what else is in the batch)
#define GOM_FILL_TILES 4
struct GomEmptyFill {
    int first_block, H, W, C;
    float bg[4];
    const GomCamera *cams;
    const uint32_t *tile_base;
    float *out_color, *final_T;
    uint32_t *n_contrib;
};
// The bucket sorts of the depth ranking as the FIRST blocks of the emit launch (one per (frame, bucket)): the emission writes Gaussian ids,
// not ranks, so the two are independent, and the sorts' latency chains run beside the emission's atomics.
struct GomSortRider {
    int B;                      // frames of the launch (always set: the grid is frame-minor)
    int blocks;                 // = buckets per frame; 0: no riders
    int P;
    uint32_t log_chunk;
    const uint32_t *bucket_base;
    uint64_t *bkeys, *scratch;
    uint32_t *order, *rank_of;
    uint32_t *bucket_count, *bucket_cursor;   // zeroed here for the next frame's histogram and scatter
};
// The bucket scatter of the depth ranking as further workgroups of the scan launch (1 024 Gaussians each): it needs the bucket COUNTS, which
// are complete when the scan starts, not the scan's result -- every scatter workgroup folds the counts in front of its frame and scans its
// frame's itself (a few KB from L2) instead of waiting a launch for one workgroup to do it.
struct GomScatterRider {
    int P, B;                   // P = 0: no riders
    uint32_t nb;
    int nblk;                   // (min, max) pairs per frame behind `minmax`
    const float *depth;
    const int32_t *radii;
    const uint32_t *minmax;
    uint64_t *bkeys;
};
// fill_out: the image this forward writes (C planes per frame) -- the emit kernel then also paints the empty tiles and the compositing
// assembly skips them (GomState::emptyFilled).
int gom_launch_scan_emit(GomState *s, int P, hipStream_t st, bool rank = false, float *fill_out = nullptr, int fill_C = 0, const float *fill_bg = nullptr);
int gom_launch_depth_hist(GomState *s, int P, hipStream_t st);
int gom_launch_tile_rank(GomState *s, hipStream_t st);
int gom_launch_rebuild_keys(GomState *s, hipStream_t st);
// lists of up to this many entries go to the 4-wave instantiations of the per-tile kernels (0: one instantiation takes all)
static inline uint32_t gom_sort_small_max(const GomState *s) { return s->B > 1 ? GOM_SORT_SMALL : 0u; }
int gom_launch_sort(GomState *s, hipStream_t st);
int gom_launch_render_forward(GomState *s, const GomCamera &cam, int C, const float *colors, float *out_color, bool reuse_T,
                              hipStream_t st);
int gom_launch_render_backward(GomState *s, const GomCamera &cam, int C, const float *colors, const float *dL_dcolor,
                               hipStream_t st);
// batched (B frames per launch) geometry / loss launches behind the single-frame C-ABI entry points
int gom_fk_forward_batch(int B, const float *cnl_gtfms, const float *dst_Rs, const float *dst_Ts, float *RT, float *fk_save, void *stream);
int gom_lbs_forward_batch(int B, int N, int J, const float *xyz, const float *weights, const float *RT, float *out, void *stream);
int gom_fk_lbs_forward_batch(int B, int N, const float *cnl_gtfms, const float *dst_Rs, const float *dst_Ts, const float *xyz, const float *weights,
                             float *RT, float *fk_save, float *out, void *stream, const GomCamera *cam1 = nullptr, const GomCamera *cams = nullptr,
                             uint32_t *vdepth_minmax = nullptr);
int gom_face_forward_batch(int B, int N, int F, const float *verts, const int32_t *faces, const float *so3, const float *scale,
                           float sigma, float *xyz, float *cov6, const float *appearance, float *feat4, void *stream);
int gom_face_backward_batch(int B, int N, int F, const float *verts, const int32_t *faces, const float *so3, const float *scale,
                            float sigma, const float *d_xyz, const float *d_cov6, float *d_corner, float *d_so3, float *d_scale,
                            const float *d_feat4, float *d_appearance, void *stream);
int gom_vertex_backward_batch(int B, int F, int N, int J, const float *xyz, const float *weights, const float *RT, const int32_t *csr_off,
                              const int32_t *csr_idx, const float *d_corner, const float *d_verts_extra, float *d_verts_obs,
                              float *d_xyz, float *dRT, void *stream);
// The backward's task word: (segment << 3) | code, code 0 / 1 = the pair of sub-ranges (0,1) / (2,3), code 4 + j = sub-range j alone (the riders
// split a pair whose busiest wave would see more than GOM_BWD_SPLIT_COST surviving entries: the densest pairs are 120 us tasks).
// Order table: [0, 8) tasks of shard x; then shard x's tasks at GOM_BWD_ORDER_BASE + x * region, region = gom_bwd_order_region(nsegs) entries.
#ifndef GOM_BWD_SPLIT_COST
#define GOM_BWD_SPLIT_COST 110u
#endif
#define GOM_BWD_ORDER_BASE 64u
// Riders of the compositing assembly (k_combine_fwd) in the frame step: eight extra workgroups, one per shard of the render backward's task
// queue, order its tasks by the cost k_seg_fwd counted (counting sort, 512 levels, most expensive first; bwd_order.hpp).  Their chain (~17 us)
// runs beside the tiles' -- in the loss kernel, where they first lived, the launch lasted as long as they did.
struct GomBwdOrderRider {
    const GomDevStatus *status;
    const uint32_t *seg_cost;
    uint32_t *bwd_order;
};
// The frame step's loss knows which tiles are EMPTY (five in six on a body): their prediction is the rasterizer's background, which is not
// read, and their image gradient, which the backward never reads (no list entry there), is not written: 16 instead of 48 bytes per pixel.
struct GomLossSkip {
    const uint32_t *tile_base;   // [frames * tiles + 1]; null: every pixel is read and written
    const GomCamera *cams;       // per-frame background, or null: bg
    float bg[4];
    int gx, gy, W;
    int zero_empty;              // split frame call (GOM_FRAME_FORWARD_ONLY: a caller's hook reads / adds to the gradient image between the halves):
                                 // the pixels of empty tiles get an explicit 0 instead of being left as they were
};
int gom_l1_loss_batch(int B, int H, int W, const float *pred, const float *shade, const float *gt_rgb, const float *gt_mask,
                      const float *bg, float c_rgb, float c_mask, float grad_scale, float *dL_dpred, float *dL_dshade,
                      float *loss_partials, void *stream, const GomLossSkip *skip = nullptr, int slots = GOM_LOSS_BLOCKS);
int gom_sum_frames_multi(int B, const size_t n[4], const float *const src[4][GOM_SPLIT_MAX_FRAMES], float *const dst[4], void *stream);
int gom_sum_frames4(int B, size_t n0, const float *s0, float *d0, size_t n1, const float *s1, float *d1, size_t n2, const float *s2,
                    float *d2, size_t n3, const float *s3, float *d3, void *stream);
int gom_launch_preprocess_backward(GomState *s, const GomCamera &cam, int P, int C, const float *means3D,
                                   const float *cov6, float *dL_dmeans3D, float *dL_dcov6, float *dL_dcolors,
                                   float *dL_dopacity, float *dL_dmeans2D, hipStream_t st, const GomFaceArgs *face = nullptr);

// ---- LPIPS trunk with one or two bf16 planes per tensor (vgg_bf16.hip; `*_lo` = element offset of the lo plane, 0 = plain bf16) ----
int gom_conv3x3_planes(int B, int H, int W, int Cin, int Cout, const void *in, const void *wt, const float *bias, const void *mask,
                       void *out, uint32_t flags, int splits, float *workspace, size_t in_lo, size_t out_lo, void *pooled, size_t pooled_lo, void *stream);
int gom_maxpool2x2_planes(int B, int H, int W, int C, const void *x, void *y, size_t x_lo, size_t y_lo, void *stream);
int gom_maxpool2x2_backward_planes(int B, int H, int W, int C, const void *x, const void *dy, void *dx, int accumulate, size_t x_lo, size_t dy_lo, size_t dx_lo, void *stream);
int gom_lpips_prepare_planes(int B, int H, int W, const float *rgb, void *out32, size_t out_lo, void *stream);
int gom_lpips_prepare_im2col_planes(int B, int H, int W, const float *rgb, void *out32, size_t out_lo, void *stream);
int gom_lpips_unprepare_col2im_planes(int B, int H, int W, const void *d_col, float *d_rgb, size_t in_lo, void *stream);
int gom_conv1_1_image_planes(int B, int H, int W, const float *rgb, const void *wt, const float *bias, void *out, size_t out_lo, void *stream);
int gom_conv1_1_bwd_image_planes(int B, int H, int W, const void *g, const void *wt, float *d_rgb, size_t in_lo, void *stream);
int gom_conv1x1_planes(size_t npix, int Cin, int Cout, const void *in, const void *wt, const float *bias, void *out, int relu, size_t in_lo, size_t out_lo, void *stream);
int gom_lpips_unprepare_planes(int B, int H, int W, int Cpad, const void *d_in, float *d_rgb, size_t in_lo, void *stream);
int gom_lpips_layer_forward_planes(int B, int C, int HW, const void *f0, const void *f1, const float *w, float *partials, size_t f_lo, void *stream);
#define GOM_LPIPS_HEAD_BLOCKS 4096   // workgroups per image of a head backward launch = per-block value sums of a tap (lpips_vgg_api.hip: head_sums)
int gom_lpips_layer_backward_value_planes(int B, int C, int HW, const void *f0, const void *f1, const float *w, const float *grad_out, void *d_f0, float *block_sums,
                                          int *n_blocks, size_t f_lo, size_t d_lo, const void *pool_dy, size_t pool_dy_lo, int W, void *stream);
int gom_lpips_fold_values(int B, const float *block_sums, const int *n_blocks5, float *partials, void *stream);
int gom_lpips_layer_backward_planes(int B, int C, int HW, const void *f0, const void *f1, const float *w, const float *grad_out, void *d_f0, size_t f_lo, size_t d_lo,
                                    void *stream);
