// C-ABI entry points of libgom_hip.so (see include/gom_hip.h) and the
// GomState scratch management.  Host code only; kernels live in the other
// translation units.
#include <stdarg.h>
#include <string.h>

#include "gom_internal.h"
#include <atomic>
#include <unistd.h>
#ifdef GOM_LAB
#include "gom_hip_lab.h"
#endif
#include <cstdlib>

static thread_local char g_err[512] = "";

void gom_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *gom_last_error(void) { return g_err; }
extern "C" int gom_abi_version(void) { return GOM_ABI_VERSION; }

template <typename T>
static int grow(T **p, size_t count) {
    if (*p) {
        GOM_HIP_CHECK(hipFree(*p));
        *p = nullptr;
    }
    GOM_HIP_CHECK(hipMalloc((void **)p, (count ? count : 1) * sizeof(T)));
    // GOM_DEBUG_POISON=1 (development): every new scratch buffer starts as 0xA5 bytes instead of whatever the pages held -- a kernel that reads a word
    // nobody wrote yet shows itself at once instead of once in a dozen multi-process runs (scripts/soak_two_ranks.sh, LABBOOK R5.9)
    static const bool addrs = getenv("GOM_DEBUG_ADDRS") && atoi(getenv("GOM_DEBUG_ADDRS")) != 0;   // (development: which buffer does a faulting address belong to)
    static int n_alloc = 0;
    if (addrs) fprintf(stderr, "[gom alloc pid %d] #%d %p .. %p (%zu bytes, elements of %zu)\n", (int)getpid(), n_alloc++, (void *)*p, (void *)((char *)*p + (count ? count : 1) * sizeof(T)),
                       (count ? count : 1) * sizeof(T), sizeof(T));
    static const bool poison = getenv("GOM_DEBUG_POISON") && atoi(getenv("GOM_DEBUG_POISON")) != 0;
    if (poison) GOM_HIP_CHECK(hipMemset(*p, 0xA5, (count ? count : 1) * sizeof(T)));
    return 0;
}

// Zero a freshly allocated counter array NOW.  hipMemset on device memory is enqueued on the legacy NULL stream and may return before it has run;
// the caller's kernels go to the caller's stream, and a stream created non-blocking (every torch.cuda.Stream) is NOT ordered behind the NULL stream:
// the first frame's kernels could meet the counters as the pages were left -- tile counts of garbage, lists of garbage lengths, a wild address.
// Alone on a device the fill always won that race by tens of microseconds; with a second process' persistent kernels on the same device it lost it
// about once in ten start-ups (`bench.py --gpus 2` on one GPU: "Memory access fault", LABBOOK R5.9).  Allocation path only: the wait costs nothing.
static int zero_now(void *p, size_t bytes) {
    GOM_HIP_CHECK(hipMemset(p, 0, bytes));
    GOM_HIP_CHECK(hipDeviceSynchronize());
    return 0;
}

// a buffer OF THE STATE: recorded launch sequences that hold its old address must not be replayed (GomState::allocGen)
template <typename T>
static int grow_s(GomState *s, T **p, size_t count) {
    s->allocGen++;
    return grow(p, count);
}

extern "C" GomState *gom_state_create(void) {
    GomState *s = new GomState();
    // development switches (A / B measurements; both default on)
    if (const char *e = getenv("GOM_BWD_ORDER")) s->bwdOrder = atoi(e) != 0;   // the cost-ordered backward queue of the batched frame step
    if (const char *e = getenv("GOM_LOSS_SKIP")) s->lossSkip = atoi(e) != 0;   // the frame step's loss kernel leaving the pixels of empty tiles alone
    if (const char *e = getenv("GOM_FUSE_LOSS")) s->fuseLoss = atoi(e) != 0;   // initial GOM_OPT_FUSE_LOSS
    if (const char *e = getenv("GOM_BWD_MODE")) s->bwdMode = atoi(e);          // initial GOM_OPT_BWD_MODE (A / B runs of the whole test suite)
#ifndef GOM_LAB
    if (s->bwdMode >= 2) s->bwdMode = -1;                                      // (modes 2 and 3 exist in -DGOM_LAB builds only)
#endif
    if (hipGetDevice(&s->device) != hipSuccess) {
        gom_set_error("hipGetDevice failed (no HIP device?)");
        delete s;
        return nullptr;
    }
    if (hipMalloc((void **)&s->status, sizeof(GomDevStatus)) != hipSuccess ||
        hipMemset(s->status, 0, sizeof(GomDevStatus)) != hipSuccess ||
        hipMalloc((void **)&s->task_ctr, GOM_TASK_CTR_WORDS * sizeof(uint32_t)) != hipSuccess || hipMemset(s->task_ctr, 0, GOM_TASK_CTR_WORDS * sizeof(uint32_t)) != hipSuccess ||
        hipDeviceSynchronize() != hipSuccess) {   // (the fills have run before anybody's stream meets the words: see zero_now)
        gom_set_error("hipMalloc(status) failed");
        delete s;
        return nullptr;
    }
    static std::atomic<uint64_t> next_uid{1};
    s->uid = next_uid.fetch_add(1);
    return s;
}

extern "C" void gom_state_destroy(GomState *s) {
    if (!s) return;
    void *ptrs[] = {s->depth, s->xy, s->conic_opacity, s->tiles_touched, s->rect, s->radii, s->pair_off, s->tile_count, s->tile_base,
                    s->tile_cursor, s->tile_nmax, s->seg_base, s->keys, s->point_list, s->pair_pos, s->ent_slot, s->partial, s->seg_desc, s->seg_qmax, s->ent_geo, s->ent_col, s->seg_T,
                    s->seg_C, s->seg_last, s->seg_Tend, s->seg_Sbehind, s->sub_T, s->sub_C, s->sub_Tend, s->final_T, s->n_contrib, s->scratch_img, s->status, s->task_ctr, s->batch_grads, s->mesh_face,
                    s->depth_minmax, s->bucket_count, s->bucket_base, s->bucket_cursor, s->bkeys, s->bkeys_scratch, s->rec_g, s->order, s->rank_of, s->keys32, s->tile_qlim, s->work_items, s->seg_cost, s->bwd_order, s->big_list, s->big_count, s->vdepth_minmax, s->cull_masks, s->piece_ub, s->piece_rec, s->piece_cnt, s->rec_ti, s->rec_acc};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    for (hipEvent_t e : s->ev)
        if (e) (void)hipEventDestroy(e);
    for (auto &g : s->graphs) {
        (void)hipGraphExecDestroy(g.exec);
        (void)hipGraphDestroy(g.graph);
    }
    for (auto &g : s->splitGraphs) {
        (void)hipGraphExecDestroy(g.exec);
        (void)hipGraphDestroy(g.graph);
    }
    for (hipStream_t t : s->splitStreams)
        if (t) (void)hipStreamDestroy(t);
    if (s->splitFork) (void)hipEventDestroy(s->splitFork);
    for (hipEvent_t e : s->splitJoin)
        if (e) (void)hipEventDestroy(e);
    delete s;
}

extern "C" int gom_state_set_option(GomState *s, int option, int64_t value) {
    if (!s) { gom_set_error("null state"); return -1; }
    switch (option) {
        case GOM_OPT_SORT_CAP:
            if (value < 64 || value > GOM_SORT_CAP_MAX) { gom_set_error("sort cap must be in [64, %d]", GOM_SORT_CAP_MAX); return -1; }
            s->sortCap = (int)value;
            s->allocGen++;   // (a recorded launch sequence was made under the old setting: dropped at its next use)
            return 0;
        case GOM_OPT_PAIR_CAPACITY:
            if (value < 0 || value > 0xffffffffLL) { gom_set_error("pair capacity out of range"); return -1; }
            s->wantPairs = value;
            s->allocGen++;   // (a recorded launch sequence was made under the old setting: dropped at its next use)
            return 0;
        case GOM_OPT_SEG_SHIFT:
            if (value != 0 && value != 7 && value != 8) { gom_set_error("segment shift must be 0 (auto), 7 or 8"); return -1; }
            s->wantSegShift = (int)value;
            s->allocGen++;   // (a recorded launch sequence was made under the old setting: dropped at its next use)
            return 0;
        case GOM_OPT_TASK_GRID_PCT:
            if (value < 10 || value > 100) { gom_set_error("task grid share must be in [10, 100] percent"); return -1; }
            s->taskGridPct = (int)value;
            s->allocGen++;   // (a recorded launch sequence was made under the old setting: dropped at its next use)
            return 0;
        case GOM_OPT_BWD_MODE:
            if (value < -1 || value > 3) { gom_set_error("backward mode must be -1 (auto), 0 (paired sub-ranges), 1 (one sub-range per barrier), 2 (4x4-block items per DPP row, GOM_LAB builds) or 3 (lane per (pixel, entry) record)"); return -1; }
#ifndef GOM_LAB
            if (value == 2) { gom_set_error("backward mode 2 (4x4-block items) exists in -DGOM_LAB builds only (include/gom_hip_lab.h)"); return -1; }
            if (value == 3) { gom_set_error("backward mode 3 (records) exists in -DGOM_LAB builds only (include/gom_hip_lab.h): measured slower than the replay"); return -1; }
#endif
            s->bwdMode = (int)value;
            s->allocGen++;   // (a recorded launch sequence was made under the old setting: dropped at its next use)
            return 0;
        case GOM_OPT_FUSE_FACE:
            if (value != 0 && value != 1) { gom_set_error("GOM_OPT_FUSE_FACE is 0 or 1"); return -1; }
            if ((value != 0) != s->fuseFace) s->allocGen++;   // recorded graphs hold the other launch sequence
            s->fuseFace = value != 0;
            return 0;
        case GOM_OPT_FUSE_LOSS:
            if (value != 0 && value != 1) { gom_set_error("GOM_OPT_FUSE_LOSS is 0 or 1"); return -1; }
            if ((value != 0) != s->fuseLoss) s->allocGen++;   // recorded graphs hold the other launch sequence
            s->fuseLoss = value != 0;
            return 0;
        case GOM_OPT_SORT_MODE:
            if (value < 0 || value > 2) { gom_set_error("sort mode must be 0 (auto), 1 (per-tile merge sort) or 2 (depth ranking)"); return -1; }
            s->sortMode = (int)value;
            s->allocGen++;   // (a recorded launch sequence was made under the old setting: dropped at its next use)
            return 0;
        case GOM_OPT_PROFILE:
            if (value && !s->ev[0]) {
                for (int i = 0; i < 2 * GOM_NUM_KERNELS; i++) GOM_HIP_CHECK(hipEventCreate(&s->ev[i]));
            }
            s->profile = value != 0;
            return 0;
        default:
            gom_set_error("unknown option %d", option);
            return -1;
    }
}

// Make sure the scratch fits (P, H, W).  Reallocation synchronises the device
// (hipFree); it only happens when a dimension grows.
static int ensure_capacity(GomState *s, int P_frame, int H, int W, int B = 1);
int gom_ensure_capacity(GomState *s, int P_frame, int H, int W, int B) { return ensure_capacity(s, P_frame, H, W, B); }

static int ensure_capacity(GomState *s, int P_frame, int H, int W, int B) {
    const int gx = (W + GOM_TILE - 1) / GOM_TILE, gy = (H + GOM_TILE - 1) / GOM_TILE;
    if ((int64_t)P_frame * B > 0x7fffffffLL || (int64_t)gy * B > 65535 || (int64_t)H * W * B > 0x7fffffffLL) {
        gom_set_error("batch of %d frames too large (P=%d, %dx%d)", B, P_frame, H, W);
        return -1;
    }
    const int P = P_frame * B;
    const int tiles = gx * gy * B;
    const int pix = H * W * B;
    if (P > s->capP) {
        const int cap = P + P / 8 + 256;
        if (grow_s(s, &s->depth, cap) || grow_s(s, &s->xy, cap) || grow_s(s, &s->conic_opacity, cap) || grow_s(s, &s->tiles_touched, cap) ||
            grow_s(s, &s->rect, cap) || grow_s(s, &s->radii, cap) || grow_s(s, &s->pair_off, cap) || grow_s(s, &s->bkeys, cap) || grow_s(s, &s->bkeys_scratch, cap) ||
            grow_s(s, &s->rec_g, (size_t)cap * 2) || grow_s(s, &s->order, cap) || grow_s(s, &s->rank_of, cap))
            return -2;
        s->capP = cap;
    }
    // depth ranking: NB = 2^nbShift buckets per frame, ~200 Gaussians each (one 256-thread sort workgroup per bucket)
    {
        int sh = 6;
        while (sh < 11 && (P_frame >> sh) > 256) sh++;
        s->nbShift = sh;
        const int64_t nbuck = (int64_t)B << sh;
        if (nbuck > s->capBuckets) {
            if (grow_s(s, &s->bucket_count, (size_t)nbuck) || grow_s(s, &s->bucket_base, (size_t)nbuck + 1) || grow_s(s, &s->bucket_cursor, (size_t)nbuck)) return -2;
            if (zero_now(s->bucket_count, (size_t)nbuck * sizeof(uint32_t)) || zero_now(s->bucket_cursor, (size_t)nbuck * sizeof(uint32_t))) return -2;
            s->capBuckets = nbuck;
        }
        const int64_t nmm = (int64_t)B * ((P_frame + 255) / 256);      // one (min, max) pair per preprocess block
        if (nmm > s->capFrames) {
            if (grow_s(s, &s->depth_minmax, (size_t)nmm * 2)) return -2;
            s->capFrames = (int)nmm;
        }
    }
    if (B > s->capBigFrames) {
        if (grow_s(s, &s->big_list, (size_t)B * GOM_BIG_CAP) || grow_s(s, &s->big_count, (size_t)2 * B)) return -2;
        if (zero_now(s->big_count, (size_t)2 * B * sizeof(uint32_t))) return -2;
        s->capBigFrames = B;
    }
    if (tiles > s->capTiles) {
        if (grow_s(s, &s->tile_count, tiles) || grow_s(s, &s->tile_base, (size_t)tiles + 1) || grow_s(s, &s->tile_cursor, tiles) ||
            grow_s(s, &s->tile_nmax, tiles) || grow_s(s, &s->seg_base, (size_t)tiles + 1) || grow_s(s, &s->tile_qlim, tiles))
            return -2;
        if (zero_now(s->tile_count, (size_t)tiles * sizeof(uint32_t))) return -2;
        s->capTiles = tiles;
        s->capSegs = 0;  // segment buffers depend on the tile count too
    }
    if (pix > s->capPix) {
        if (grow_s(s, &s->final_T, pix) || grow_s(s, &s->n_contrib, pix) || grow_s(s, &s->scratch_img, (size_t)pix * 4)) return -2;
        s->capPix = pix;
    }
    // Pair buffers: sized for 288 GB of HBM, not for frugality.  Default 16 pairs
    // per Gaussian (the GoMAvatar workload averages ~2.7) and at least 4 M.
    int64_t want = s->wantPairs > 0 ? s->wantPairs : (int64_t)P * 16;
    if (s->wantPairs <= 0 && want < (4 << 20)) want = 4 << 20;
    if (want > 0xffffffffLL) want = 0xffffffffLL;
    if (want != s->capPairs && (want > s->capPairs || s->wantPairs > 0)) {
        if (grow_s(s, &s->keys, (size_t)want) || grow_s(s, &s->point_list, (size_t)want) || grow_s(s, &s->pair_pos, (size_t)want) || grow_s(s, &s->ent_slot, (size_t)want) || grow_s(s, &s->keys32, (size_t)want) ||
            grow_s(s, &s->ent_geo, (size_t)want * 3) || grow_s(s, &s->ent_col, (size_t)want * 4) ||
            grow_s(s, &s->partial, (size_t)want * GOM_PARTIAL_STRIDE))
            return -2;
        s->capPairs = want;
        s->capSegs = 0;
    }
    {
        const int64_t wantItems = s->capTiles + s->capPairs / GOM_RANK_WIN + 1;
        if (wantItems > s->capItems) {
            if (grow_s(s, &s->work_items, (size_t)wantItems)) return -2;
            s->capItems = wantItems;
        }
    }
    // every tile has at most count/GOM_SEG + 1 segments
    const int64_t wantSegs = s->capPairs / GOM_SEG + s->capTiles + 1;
    if (wantSegs > s->capSegs) {
        const size_t n = (size_t)wantSegs;
        if (grow_s(s, &s->seg_desc, n) || grow_s(s, &s->seg_cost, 16 * n) || grow_s(s, &s->bwd_order, GOM_BWD_ORDER_BASE + GOM_TQ_SHARDS * (size_t)gom_bwd_order_region((uint32_t)n)) || grow_s(s, &s->seg_qmax, n) || grow_s(s, &s->seg_T, n * GOM_TPX) || grow_s(s, &s->seg_C, n * 4 * GOM_TPX) || grow_s(s, &s->seg_last, n * GOM_TPX) ||
            grow_s(s, &s->seg_Tend, n * GOM_TPX) || grow_s(s, &s->seg_Sbehind, n * 4 * GOM_TPX) || grow_s(s, &s->sub_T, n * 4 * GOM_TPX) || grow_s(s, &s->cull_masks, n * 16) ||
            grow_s(s, &s->sub_C, n * 16 * GOM_TPX) || grow_s(s, &s->sub_Tend, n * 4 * GOM_TPX))
            return -2;
        s->capSegs = wantSegs;
    }
#ifdef GOM_LAB
    if (s->bwdMode == 3) {   // records of the laboratory render backward (lab/rec_bwd.hpp), only for a state that uses it: one per blending (pixel, entry) pair in
                             // GOM_REC_SHARDS equal regions (24 bytes x 8 per unit of pair capacity), and the per-piece bookkeeping
        int64_t wantRec = s->capPairs * GOM_REC_PER_PAIR / GOM_REC_SHARDS * GOM_REC_SHARDS;
        if (wantRec > 0xf0000000LL) wantRec = 0xf0000000LL / GOM_REC_SHARDS * GOM_REC_SHARDS;
        if (wantRec != s->capRec) {
            if (grow_s(s, &s->rec_ti, (size_t)wantRec) || grow_s(s, &s->rec_acc, (size_t)wantRec)) return -2;
            s->capRec = wantRec;
        }
        if (s->capSegs > s->capPieceSegs) {
            const size_t n = (size_t)s->capSegs;
            if (grow_s(s, &s->piece_ub, n * 16) || grow_s(s, &s->piece_rec, n * 16) || grow_s(s, &s->piece_cnt, n * 16 * 64)) return -2;
            s->capPieceSegs = s->capSegs;
        }
    }
#endif
    s->gx = gx;
    s->gy = gy;
    s->B = B;
    // auto: 256-entry segments for a batch; for one frame too when its tiles are busy -- at >= 32 Gaussians per tile of the image (55 104 at 512^2 or
    // 540^2: 54 / 48; 220 416 at 1024^2: 54) the halved number of segments pays (B = 1: 4.78 -> 5.00 k, 4.69 -> 4.99 k, 2.08 -> 2.40 k frames/s), at
    // 13 per tile (55 104 at 1024^2) a frame is short of independent tasks and 128-entry segments win (3.27 k against 2.31 k).
    s->segShift = s->wantSegShift ? s->wantSegShift : ((B > 1 || (int64_t)P_frame >= 32 * (int64_t)gx * gy) ? 8 : 7);   // (capSegs above is sized for the 128-entry case)
    return 0;
}

static bool valid_dims(int P, int C, const GomCamera *cam) {
    if (!cam) { gom_set_error("null camera"); return false; }
    if (P < 0) { gom_set_error("negative P"); return false; }
    if (C != 3 && C != 4) { gom_set_error("C must be 3 or 4 (got %d)", C); return false; }
    if (cam->H <= 0 || cam->W <= 0 || cam->H > 65535 * 16 || cam->W > 65535 * 16) { gom_set_error("bad image size %dx%d", cam->H, cam->W); return false; }
    return true;
}

// P, H, W are per frame; B frames with device-resident cameras `cams` (nullptr: one frame, camera by value).
// All tensors then carry a leading B dimension.
static int raster_forward_impl(GomState *s, const GomCamera *cam, const GomCamera *cams, int B, int P, int C, const float *means3D,
                               const float *cov6, const float *colors, const float *opacity, float *out_color, int32_t *radii,
                               uint32_t flags, void *stream, const GomFaceArgs *face = nullptr) {
    if (!s) { gom_set_error("null state"); return -1; }
    if (!valid_dims(P, C, cam)) return -1;
    if (!out_color || (P > 0 && (!means3D || !cov6 || !colors || !opacity))) { gom_set_error("null tensor pointer"); return -1; }
    hipStream_t st = (hipStream_t)stream;
    const bool reuse = (flags & GOM_FWD_REUSE_BINNING) != 0;
    if (reuse) {
        if (!s->haveForward || s->P != P || s->H != cam->H || s->W != cam->W || s->B != B || s->cams != cams) {
            gom_set_error("GOM_FWD_REUSE_BINNING without a matching previous forward");
            return -1;
        }
    } else {
        if (int rc = ensure_capacity(s, P, cam->H, cam->W, B)) return rc;
        s->bwdOrderReady = false;   // a new binning: the cost-ordered task table of an earlier forward (frame step, forward-only call) is stale
        s->P = P; s->H = cam->H; s->W = cam->W; s->cams = cams;
        // Tile-list order: rank the frame's Gaussians by depth once + a linear bitmap pass per tile (raster_rank.hip), unless
        // the frame's bitmap would not fit in LDS (P > 2^19) or the caller asked for the per-tile merge sort.
        // (k_tile_rank's work items carry the tile in 24 bits; the Gaussian record packs its rect as 10-bit x0 / width and 12-bit stacked y0)
        const bool rank_fits = (int64_t)s->gx * s->gy * B < (1 << 24) && s->gx < 1024 && (int64_t)s->gy * B < 4096;
        s->rankSort = rank_fits && (s->sortMode == 2 || (s->sortMode == 0 && P <= (1 << 18)));
        if (s->rankSort && P > 393216) { gom_set_error("GOM_OPT_SORT_MODE 2 needs P <= 393216 per frame (the frame's rank bitmap lives in 64 KiB of LDS)"); return -1; }
        if (int rc = gom_launch_preprocess(s, *cam, P, means3D, cov6, opacity, radii, st, face)) return rc;
        if (s->rankSort) {
            // the depth range behind the bucket map: per block of k_preprocess, or the frame step's vertex ranges (then k_preprocess has
            // built the histogram as well)
            const bool vranges = face && face->vdepth_minmax;
            s->rank_minmax = vranges ? face->vdepth_minmax : s->depth_minmax;
            s->rank_blocks = vranges ? face->vdepth_blocks : (P + 255) / 256;
            if (!vranges) { if (int rc = gom_launch_depth_hist(s, P, st)) return rc; }
            if (int rc = gom_launch_scan_emit(s, P, st, true, out_color, C, cam->bg)) return rc;
            if (int rc = gom_launch_tile_rank(s, st)) return rc;
        } else {
            if (int rc = gom_launch_scan_emit(s, P, st, false, out_color, C, cam->bg)) return rc;
            if (int rc = gom_launch_sort(s, st)) return rc;
        }
        s->emptyFilled = P > 0;   // (no Gaussians: no emit launch)
    }
    s->C = C;
    const int rrc = gom_launch_render_forward(s, *cam, C, colors, out_color, reuse, st);
    s->emptyFilled = false;       // (a later pass over this binning -- other colours, other image -- paints its own)
    if (rrc) return rrc;
    if (reuse && radii) GOM_HIP_CHECK(hipMemcpyAsync(radii, s->radii, (size_t)P * B * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
    s->haveForward = true;
    return 0;
}

extern "C" int gom_raster_forward(GomState *s, const GomCamera *cam, int P, int C, const float *means3D, const float *cov6,
                                  const float *colors, const float *opacity, float *out_color, int32_t *radii,
                                  uint32_t flags, void *stream) {
    return raster_forward_impl(s, cam, nullptr, 1, P, C, means3D, cov6, colors, opacity, out_color, radii, flags, stream);
}

static int raster_backward_impl(GomState *s, const GomCamera *cam, const GomCamera *cams, int B, int P, int C, const float *means3D,
                                const float *cov6, const float *colors, const float *opacity, const float *dL_dcolor,
                                float *dL_dmeans3D, float *dL_dcov6, float *dL_dcolors, float *dL_dopacity, float *dL_dmeans2D,
                                uint32_t flags, void *stream, const GomFaceArgs *face = nullptr) {
    (void)opacity;
    if (!s) { gom_set_error("null state"); return -1; }
    if (!valid_dims(P, C, cam)) return -1;
    if (!s->haveForward || s->P != P || s->H != cam->H || s->W != cam->W || s->B != B || s->cams != cams) {
        gom_set_error("gom_raster_backward without a matching forward on this state");
        return -1;
    }
    if (!dL_dcolor || (!face && (!dL_dmeans3D || !dL_dcov6 || !dL_dcolors || !dL_dopacity))) { gom_set_error("null gradient pointer"); return -1; }
    hipStream_t st = (hipStream_t)stream;
    if (flags & GOM_BWD_RECOMPUTE_FORWARD) {
        if (int rc = gom_launch_render_forward(s, *cam, C, colors, s->scratch_img, true, st)) return rc;
    }
    const int brc = gom_launch_render_backward(s, *cam, C, colors, dL_dcolor, st);
    s->bwdOrderReady = false;   // (the order belongs to ONE forward / backward pair)
    if (brc) return brc;
    if (int rc = gom_launch_preprocess_backward(s, *cam, P, C, means3D, cov6, dL_dmeans3D, dL_dcov6, dL_dcolors, dL_dopacity,
                                                dL_dmeans2D, st, face))
        return rc;
    return 0;
}

extern "C" int gom_raster_backward(GomState *s, const GomCamera *cam, int P, int C, const float *means3D, const float *cov6,
                                   const float *colors, const float *opacity, const float *dL_dcolor, float *dL_dmeans3D,
                                   float *dL_dcov6, float *dL_dcolors, float *dL_dopacity, float *dL_dmeans2D, uint32_t flags,
                                   void *stream) {
    return raster_backward_impl(s, cam, nullptr, 1, P, C, means3D, cov6, colors, opacity, dL_dcolor, dL_dmeans3D, dL_dcov6, dL_dcolors,
                                dL_dopacity, dL_dmeans2D, flags, stream);
}

static bool dcam_stub(int H, int W, const GomCamera *cam_device, GomCamera *stub) {
    if (!cam_device) { gom_set_error("null device camera"); return false; }
    *stub = GomCamera{};
    stub->H = H; stub->W = W;
    return true;
}

extern "C" int gom_raster_forward_dcam(GomState *s, int H, int W, const GomCamera *cam_device, int P, int C, const float *means3D,
                                       const float *cov6, const float *colors, const float *opacity, float *out_color, int32_t *radii,
                                       uint32_t flags, void *stream) {
    GomCamera stub;
    if (!dcam_stub(H, W, cam_device, &stub)) return -1;
    return raster_forward_impl(s, &stub, cam_device, 1, P, C, means3D, cov6, colors, opacity, out_color, radii, flags, stream);
}

extern "C" int gom_raster_backward_dcam(GomState *s, int H, int W, const GomCamera *cam_device, int P, int C, const float *means3D,
                                        const float *cov6, const float *colors, const float *opacity, const float *dL_dcolor,
                                        float *dL_dmeans3D, float *dL_dcov6, float *dL_dcolors, float *dL_dopacity, float *dL_dmeans2D,
                                        uint32_t flags, void *stream) {
    GomCamera stub;
    if (!dcam_stub(H, W, cam_device, &stub)) return -1;
    return raster_backward_impl(s, &stub, cam_device, 1, P, C, means3D, cov6, colors, opacity, dL_dcolor, dL_dmeans3D, dL_dcov6,
                                dL_dcolors, dL_dopacity, dL_dmeans2D, flags, stream);
}

extern "C" int gom_state_poll(GomState *s, int64_t *num_pairs, int32_t *overflow, void *stream) {
    if (!s) { gom_set_error("null state"); return -1; }
    GomDevStatus h;
    GOM_HIP_CHECK(hipMemcpyAsync(&h, s->status, sizeof(h), hipMemcpyDeviceToHost, (hipStream_t)stream));
    GOM_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    if (num_pairs) *num_pairs = h.num_pairs;
    if (overflow) *overflow = (int32_t)(h.overflow | (h.rec_overflow ? 2u : 0u));   // bit 0: pair buffers (image poisoned); bit 1: records of the render backward (gradients poisoned)
    return 0;
}

extern "C" int gom_state_kernel_times(GomState *s, float *ms_out) {
    if (!s || !ms_out) { gom_set_error("null argument"); return -1; }
    for (int k = 0; k < GOM_NUM_KERNELS; k++) {
        ms_out[k] = -1.f;
        if (s->ev[0] && s->evValid[k]) {
            GOM_HIP_CHECK(hipEventSynchronize(s->ev[2 * k + 1]));
            GOM_HIP_CHECK(hipEventElapsedTime(&ms_out[k], s->ev[2 * k], s->ev[2 * k + 1]));
        }
    }
    return 0;
}

extern "C" int gom_state_export(GomState *s, int id, void *dst, int64_t dst_bytes, void *stream) {
    if (!s || !s->haveForward) { gom_set_error("export without a forward"); return -1; }
    const void *src = nullptr;
    int64_t bytes = 0;
    const int64_t P = (int64_t)s->P * s->B, tiles = (int64_t)s->gx * s->gy * s->B, pix = (int64_t)s->H * s->W * s->B;
    switch (id) {
        case GOM_BUF_DEPTH: src = s->depth; bytes = P * 4; break;
        case GOM_BUF_XY: src = s->xy; bytes = P * 8; break;
        case GOM_BUF_CONIC_OPACITY: src = s->conic_opacity; bytes = P * 16; break;
        case GOM_BUF_TILES_TOUCHED: src = s->tiles_touched; bytes = P * 4; break;
        case GOM_BUF_RECT: src = s->rect; bytes = P * 8; break;
        case GOM_BUF_TILE_BASE: src = s->tile_base; bytes = (tiles + 1) * 4; break;
        case GOM_BUF_KEYS:
            if (s->rankSort) { if (int rc = gom_launch_rebuild_keys(s, (hipStream_t)stream)) return rc; }   // (the ranking path never materialises them)
            src = s->keys; bytes = dst_bytes < s->capPairs * 8 ? dst_bytes : s->capPairs * 8; break;
        case GOM_BUF_POINT_LIST: src = s->point_list; bytes = dst_bytes < s->capPairs * 4 ? dst_bytes : s->capPairs * 4; break;
        case GOM_BUF_FINAL_T: src = s->final_T; bytes = pix * 4; break;
        case GOM_BUF_N_CONTRIB: src = s->n_contrib; bytes = pix * 4; break;
        case GOM_BUF_STATUS: src = s->status; bytes = 16; break;
        default: gom_set_error("unknown buffer id %d", id); return -1;
    }
    if (dst_bytes < bytes) { gom_set_error("export buffer too small (%lld < %lld)", (long long)dst_bytes, (long long)bytes); return -1; }
    if (bytes > 0) GOM_HIP_CHECK(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return 0;
}

static int frame_enqueue(GomState *s, const GomFrame *f, int B, const GomCamera *cams, uint32_t flags, void *stream);

// per-frame slices of the parameter gradients of a batched call: [so3 | scale | appearance : B x 3F each][vertices : B x 3N]
static int ensure_batch_grads(GomState *s, int B, int N, int F, bool slices_for_one = false) {
    const size_t vneed = (size_t)B * ((N + 255) / 256) * 2;   // vertex depth ranges of the skinning blocks (frame step)
    if (vneed > s->capVdepth) {
        if (grow_s(s, &s->vdepth_minmax, vneed)) return -2;
        s->capVdepth = vneed;
    }
    const size_t need = (B > 1 || slices_for_one) ? (size_t)B * (9 * (size_t)F + 3 * (size_t)N) : 0;
    if (need > s->capBatchGrads) {
        if (grow_s(s, &s->batch_grads, need)) return -2;
        s->capBatchGrads = need;
    }
    return 0;
}

static int frame_call(GomState *s, const GomFrame *f, int B, const GomCamera *cams, uint32_t flags, void *stream) {
    if (!s || !f) { gom_set_error("gom_frame_forward_backward: null argument"); return -1; }
    if (f->cam.H != f->H || f->cam.W != f->W) { gom_set_error("gom_frame_forward_backward: camera/image size mismatch"); return -1; }
    if (B < 1 || (B > 1 && !cams)) { gom_set_error("batched frame call needs B >= 1 and a device camera array"); return -1; }
    // the legacy NULL stream cannot be captured; profiling wants individually timed launches
    if (!(flags & GOM_FRAME_USE_GRAPH) || s->profile || stream == nullptr) {
        if (int rc = ensure_batch_grads(s, B, f->N, f->F)) return rc;
        return frame_enqueue(s, f, B, cams, flags & ~GOM_FRAME_USE_GRAPH, stream);
    }
    hipStream_t st = (hipStream_t)stream;
    const uint32_t kflags = flags & ~GOM_FRAME_USE_GRAPH;
    s->graphClock++;
    for (size_t i = 0; i < s->graphs.size();) {   // recordings made before a buffer of the state moved (another frame size on the same state)
        if (s->graphs[i].alloc_gen != s->allocGen) {
            (void)hipGraphExecDestroy(s->graphs[i].exec);
            (void)hipGraphDestroy(s->graphs[i].graph);
            s->graphs[i] = s->graphs.back();
            s->graphs.pop_back();
        } else {
            i++;
        }
    }
    for (auto &g : s->graphs) {
        if (g.flags == kflags && g.B == B && g.cams == cams && memcmp(&g.key, f, sizeof(GomFrame)) == 0) {
            g.last_use = s->graphClock;
            GOM_HIP_CHECK(hipGraphLaunch(g.exec, st));
            s->P = f->F; s->H = f->H; s->W = f->W; s->C = 4; s->B = B; s->cams = cams; s->haveForward = true;
            s->gx = g.gx; s->gy = g.gy; s->segShift = g.segShift; s->rankSort = g.rankSort; s->bwdOrderReady = g.bwdOrderReady; s->recCounts = g.recCounts; s->recForward = g.recForward;
            s->emptyFilled = false;
            return 0;
        }
    }
    // first use of this frame descriptor: allocate outside the capture, then record the launch sequence
    if (int rc = ensure_capacity(s, f->F, f->H, f->W, B)) return rc;
    if (int rc = ensure_batch_grads(s, B, f->N, f->F)) return rc;
    GomGraphEntry e;
    memcpy(&e.key, f, sizeof(GomFrame));
    e.flags = kflags;
    e.B = B;
    e.cams = cams;
    e.last_use = s->graphClock;
    e.alloc_gen = s->allocGen;   // (after the two ensure_* calls above: nothing is allocated inside the capture)
    GOM_HIP_CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    const int rc = frame_enqueue(s, f, B, cams, kflags, stream);
    hipError_t ce = hipStreamEndCapture(st, &e.graph);
    if (rc) { if (ce == hipSuccess && e.graph) (void)hipGraphDestroy(e.graph); return rc; }
    if (ce != hipSuccess) { gom_set_error("hipStreamEndCapture failed: %s", hipGetErrorString(ce)); return -2; }
    e.gx = s->gx; e.gy = s->gy; e.segShift = s->segShift; e.rankSort = s->rankSort; e.bwdOrderReady = s->bwdOrderReady; e.recCounts = s->recCounts; e.recForward = s->recForward;
    GOM_HIP_CHECK(hipGraphInstantiate(&e.exec, e.graph, nullptr, nullptr, 0));
    if (s->graphs.size() >= 64) {  // evict the least recently used capture
        size_t victim = 0;
        for (size_t i = 1; i < s->graphs.size(); i++)
            if (s->graphs[i].last_use < s->graphs[victim].last_use) victim = i;
        (void)hipGraphExecDestroy(s->graphs[victim].exec);
        (void)hipGraphDestroy(s->graphs[victim].graph);
        s->graphs[victim] = e;
    } else {
        s->graphs.push_back(e);
    }
    GOM_HIP_CHECK(hipGraphLaunch(e.exec, st));
    return 0;
}

#ifdef GOM_LAB   // (include/gom_hip_lab.h)
extern "C" int gom_state_set_frame_optimizer(GomState *s, int64_t n, float *params, const float *grads, float *exp_avg, float *exp_avg_sq, int32_t n_segments,
                                             const int64_t *seg_begin, const float *seg_lr, int64_t *step_device, float lr_decay_steps, float beta1, float beta2,
                                             float eps, float grad_scale) {
    if (!s) { gom_set_error("null state"); return -1; }
    s->allocGen++;   // recordings made with another (or no) optimizer behind them are dropped at their next use
    if (n == 0 || !params) { s->adam.on = false; return 0; }
    if (!grads || !exp_avg || !exp_avg_sq || !step_device || !seg_begin || !seg_lr || n_segments < 1 || n_segments > GOM_ADAM_MAX_SEGMENTS) {
        gom_set_error("gom_state_set_frame_optimizer: null pointer / 1..%d segments / a device step counter is required", GOM_ADAM_MAX_SEGMENTS);
        return -1;
    }
    s->adam.on = true; s->adam.n = n; s->adam.params = params; s->adam.grads = grads; s->adam.exp_avg = exp_avg; s->adam.exp_avg_sq = exp_avg_sq;
    s->adam.n_segments = n_segments;
    for (int i = 0; i <= n_segments; i++) s->adam.seg_begin[i] = seg_begin[i];
    for (int i = 0; i < n_segments; i++) s->adam.seg_lr[i] = seg_lr[i];
    s->adam.step_device = step_device; s->adam.lr_decay_steps = lr_decay_steps; s->adam.beta1 = beta1; s->adam.beta2 = beta2; s->adam.eps = eps; s->adam.grad_scale = grad_scale;
    return 0;
}
#endif

extern "C" int gom_frame_loss_slots(int H, int W) {
    const int tiles = ((W + GOM_TILE - 1) / GOM_TILE) * ((H + GOM_TILE - 1) / GOM_TILE);
    return tiles > GOM_LOSS_BLOCKS ? tiles : GOM_LOSS_BLOCKS;
}

extern "C" int gom_frame_forward_backward(GomState *s, const GomFrame *f, uint32_t flags, void *stream) {
    return frame_call(s, f, 1, nullptr, flags, stream);
}

extern "C" int gom_batch_forward_backward(GomState *s, const GomFrame *f, int32_t B, const GomCamera *cams_device, uint32_t flags,
                                          void *stream) {
    return frame_call(s, f, B, cams_device, flags, stream);
}

// ---- ONE step as K concurrent launch sequences (include/gom_hip.h: gom_split_forward_backward) ------------------------------------------------
// The segment kernels of a batched launch are resident grids that drain a task queue: the chip idles behind the last workgroups of every one
// of the ~12 launches of a step (three steps in flight measure +13 %, but train on stale parameters).  Here the B frames of ONE step are cut
// into K sub-batches, each with its own GomState, enqueued on K streams between a fork and a join -- the tail of one branch's kernel is filled
// by the other branch's workgroups -- and ONE frame sum over all B frames in frame order closes the step: the gradients are bitwise those of
// gom_batch_forward_backward on the B frames (a frame's gradients do not depend on the launch it rides in: tests/test_gpu_batch.py).
static int split_enqueue(GomState *const *states, const GomFrame *frames, int K, const int32_t *Bs, const GomCamera *const *cams, hipStream_t st, bool serial) {
    GomState *lead = states[0];
    if (serial) {   // GOM_OPT_PROFILE: one branch after the other on the caller's stream, so that every bracketed launch owns the chip
        for (int k = 0; k < K; k++)
            if (int rc = frame_enqueue(states[k], &frames[k], Bs[k], cams[k], GOM_FRAME_NO_SUM, (void *)st)) return rc;
    } else {
        GOM_HIP_CHECK(hipEventRecord(lead->splitFork, st));
        for (int k = 1; k < K; k++) GOM_HIP_CHECK(hipStreamWaitEvent(lead->splitStreams[k], lead->splitFork, 0));
        for (int k = K - 1; k >= 0; k--) {   // (the lead's own branch last: the side branches are already running while the host enqueues it)
            hipStream_t bs = k == 0 ? st : lead->splitStreams[k];
            if (int rc = frame_enqueue(states[k], &frames[k], Bs[k], cams[k], GOM_FRAME_NO_SUM, (void *)bs)) return rc;
            if (k) GOM_HIP_CHECK(hipEventRecord(lead->splitJoin[k], bs));
        }
        for (int k = 1; k < K; k++) GOM_HIP_CHECK(hipStreamWaitEvent(st, lead->splitJoin[k], 0));
    }
    // frame-ordered sum over the slices of every branch: frame b of branch k at batch_grads + [tensor offset in ITS B_k-frame layout] + b * n
    const GomFrame &f0 = frames[0];
    const size_t F3 = 3 * (size_t)f0.F, N3 = 3 * (size_t)f0.N;
    const float *src[4][GOM_SPLIT_MAX_FRAMES];
    int nfr = 0;
    for (int k = 0; k < K; k++) {
        const float *bg = states[k]->batch_grads;
        for (int b = 0; b < Bs[k]; b++, nfr++) {
            src[0][nfr] = bg + (size_t)b * F3;
            src[1][nfr] = bg + (size_t)Bs[k] * F3 + (size_t)b * F3;
            src[2][nfr] = bg + 2 * (size_t)Bs[k] * F3 + (size_t)b * F3;
            src[3][nfr] = bg + 3 * (size_t)Bs[k] * F3 + (size_t)b * N3;
        }
    }
    const size_t n[4] = {F3, F3, F3, N3};
    float *dst[4] = {f0.g_so3, f0.g_scale, f0.g_appearance, f0.g_vertices};
    return gom_sum_frames_multi(nfr, n, src, dst, (void *)st);
}

extern "C" int gom_split_forward_backward(GomState *const *states, const GomFrame *frames, int32_t K, const int32_t *Bs, const GomCamera *const *cams_device,
                                          uint32_t flags, void *stream) {
    if (!states || !frames || !Bs || !cams_device || K < 1 || K > GOM_SPLIT_MAX) { gom_set_error("gom_split_forward_backward: null argument or K outside 1..%d", GOM_SPLIT_MAX); return -1; }
    if (flags & ~GOM_FRAME_USE_GRAPH) { gom_set_error("gom_split_forward_backward: only GOM_FRAME_USE_GRAPH is accepted (a split step is a whole step)"); return -1; }
    int total = 0;
    for (int k = 0; k < K; k++) {
        if (!states[k]) { gom_set_error("gom_split_forward_backward: null state %d", k); return -1; }
        for (int j = 0; j < k; j++)
            if (states[j] == states[k]) { gom_set_error("gom_split_forward_backward: every branch needs its own GomState"); return -1; }
        const GomFrame &f = frames[k];
        if (Bs[k] < 1 || !cams_device[k]) { gom_set_error("gom_split_forward_backward: branch %d needs B >= 1 and a device camera array", k); return -1; }
        if (f.cam.H != f.H || f.cam.W != f.W) { gom_set_error("gom_split_forward_backward: camera/image size mismatch"); return -1; }
        if (f.N != frames[0].N || f.F != frames[0].F || f.g_so3 != frames[0].g_so3 || f.g_scale != frames[0].g_scale || f.g_appearance != frames[0].g_appearance ||
            f.g_vertices != frames[0].g_vertices) {
            gom_set_error("gom_split_forward_backward: the branches must share the topology sizes and the four gradient outputs");
            return -1;
        }
        total += Bs[k];
    }
    if (total > GOM_SPLIT_MAX_FRAMES) { gom_set_error("gom_split_forward_backward: at most %d frames per step", GOM_SPLIT_MAX_FRAMES); return -1; }
    GomState *lead = states[0];
    hipStream_t st = (hipStream_t)stream;
    if (!lead->splitFork || (K > 1 && !lead->splitStreams[K - 1])) {   // the branches' streams and events live on the LEAD state's device (a process may drive several devices)
        int cur = lead->device;
        (void)hipGetDevice(&cur);
        if (cur != lead->device) GOM_HIP_CHECK(hipSetDevice(lead->device));
        if (!lead->splitFork) GOM_HIP_CHECK(hipEventCreateWithFlags(&lead->splitFork, hipEventDisableTiming));
        for (int k = 1; k < K; k++) {   // only as many as this call's K needs
            if (lead->splitStreams[k]) continue;
            // (measured, round 6: a side stream created with hipStreamCreateWithPriority -- highest OR lowest -- makes the step 0.79 ms instead of 0.51:
            //  the two sequences no longer overlap.  Default priority on both.)
            GOM_HIP_CHECK(hipStreamCreateWithFlags(&lead->splitStreams[k], hipStreamNonBlocking));
            GOM_HIP_CHECK(hipEventCreateWithFlags(&lead->splitJoin[k], hipEventDisableTiming));
        }
        if (cur != lead->device) GOM_HIP_CHECK(hipSetDevice(cur));
    }
    for (int k = 1; k < K; k++)
        if (states[k]->device != lead->device) { gom_set_error("gom_split_forward_backward: every branch state must live on the lead state's device"); return -1; }
    for (int k = 0; k < K; k++) {   // every allocation happens here, outside any capture
        if (int rc = ensure_capacity(states[k], frames[k].F, frames[k].H, frames[k].W, Bs[k])) return rc;
        if (int rc = ensure_batch_grads(states[k], Bs[k], frames[k].N, frames[k].F, true)) return rc;
    }
    bool profiling = false;
    for (int k = 0; k < K; k++) profiling = profiling || states[k]->profile;
    if (!(flags & GOM_FRAME_USE_GRAPH) || profiling || stream == nullptr) return split_enqueue(states, frames, K, Bs, cams_device, st, profiling);
    lead->graphClock++;
    for (size_t i = 0; i < lead->splitGraphs.size();) {   // recordings made before a buffer of one of their states moved
        GomSplitGraphEntry &g = lead->splitGraphs[i];
        // (only the states of THIS call are known to be alive: a recording that names other states -- possibly destroyed since -- is left alone
        //  here, never matched below, and ages out of the 16 slots)
        bool stale = false;
        for (int k = 0; k < g.K; k++)
            for (int c = 0; c < K; c++)
                if (g.states[k] == states[c]) stale = stale || g.uids[k] != states[c]->uid || g.alloc_gen[k] != states[c]->allocGen;
        if (stale) {
            (void)hipGraphExecDestroy(g.exec);
            (void)hipGraphDestroy(g.graph);
            lead->splitGraphs[i] = lead->splitGraphs.back();
            lead->splitGraphs.pop_back();
        } else {
            i++;
        }
    }
    auto restore = [&](const GomSplitGraphEntry &g) {
        for (int k = 0; k < K; k++) {
            GomState *s = states[k];
            s->P = frames[k].F; s->H = frames[k].H; s->W = frames[k].W; s->C = 4; s->B = Bs[k]; s->cams = cams_device[k]; s->haveForward = true;
            s->gx = g.host[k].gx; s->gy = g.host[k].gy; s->segShift = g.host[k].segShift; s->rankSort = g.host[k].rankSort;
            s->bwdOrderReady = false; s->recCounts = g.host[k].recCounts; s->recForward = g.host[k].recForward; s->emptyFilled = false;
        }
    };
    for (auto &g : lead->splitGraphs) {
        if (g.K != K || g.flags != 0u) continue;
        bool same = true;
        for (int k = 0; k < K && same; k++)
            same = g.states[k] == states[k] && g.uids[k] == states[k]->uid && g.alloc_gen[k] == states[k]->allocGen && g.Bs[k] == Bs[k] && g.cams[k] == cams_device[k] &&
                   memcmp(&g.keys[k], &frames[k], sizeof(GomFrame)) == 0;
        if (!same) continue;
        g.last_use = lead->graphClock;
        GOM_HIP_CHECK(hipGraphLaunch(g.exec, st));
        restore(g);
        return 0;
    }
    GomSplitGraphEntry e{};
    e.K = K; e.flags = 0u; e.last_use = lead->graphClock;
    for (int k = 0; k < K; k++) {
        e.states[k] = states[k]; e.uids[k] = states[k]->uid; e.Bs[k] = Bs[k]; e.cams[k] = cams_device[k]; e.alloc_gen[k] = states[k]->allocGen;
        memcpy(&e.keys[k], &frames[k], sizeof(GomFrame));
    }
    GOM_HIP_CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    const int rc = split_enqueue(states, frames, K, Bs, cams_device, st, false);
    hipError_t ce = hipStreamEndCapture(st, &e.graph);
    if (rc) { if (ce == hipSuccess && e.graph) (void)hipGraphDestroy(e.graph); return rc; }
    if (ce != hipSuccess) { gom_set_error("hipStreamEndCapture failed: %s", hipGetErrorString(ce)); return -2; }
    for (int k = 0; k < K; k++) {
        const GomState *s = states[k];
        e.host[k] = GomSplitHostState{s->gx, s->gy, s->segShift, s->rankSort, false, s->recCounts, s->recForward};
    }
    GOM_HIP_CHECK(hipGraphInstantiate(&e.exec, e.graph, nullptr, nullptr, 0));
    if (lead->splitGraphs.size() >= 16) {   // evict the least recently used recording
        size_t victim = 0;
        for (size_t i = 1; i < lead->splitGraphs.size(); i++)
            if (lead->splitGraphs[i].last_use < lead->splitGraphs[victim].last_use) victim = i;
        (void)hipGraphExecDestroy(lead->splitGraphs[victim].exec);
        (void)hipGraphDestroy(lead->splitGraphs[victim].graph);
        lead->splitGraphs[victim] = e;
    } else {
        lead->splitGraphs.push_back(e);
    }
    GOM_HIP_CHECK(hipGraphLaunch(e.exec, st));
    restore(e);
    return 0;
}

static int frame_enqueue(GomState *s, const GomFrame *f, int B, const GomCamera *cams, uint32_t flags, void *stream) {
    const int N = f->N, F = f->F, H = f->H, W = f->W, J = 24;
    int rc;
    if ((flags & GOM_FRAME_FORWARD_ONLY) && (flags & GOM_FRAME_BACKWARD_ONLY)) { gom_set_error("FORWARD_ONLY and BACKWARD_ONLY together"); return -1; }
    // The per-face frame runs inside the rasterizer's per-Gaussian kernels (GomFaceArgs) unless GOM_OPT_FUSE_FACE is 0.
    const size_t F3 = 3 * (size_t)F, N3 = 3 * (size_t)N;
    const bool no_sum = (flags & GOM_FRAME_NO_SUM) != 0;   // a branch of gom_split_forward_backward: slices even for one frame, the caller sums all branches' frames
    flags &= ~GOM_FRAME_NO_SUM;
    const bool slices = B > 1 || no_sum;
    float *b_so3 = slices ? s->batch_grads : f->g_so3;   // B > 1: every frame writes its own slice, one more launch sums them in frame order (no atomics: reproducible)
    float *b_scale = slices ? s->batch_grads + B * F3 : f->g_scale;
    float *b_app = slices ? s->batch_grads + 2 * B * F3 : f->g_appearance;
    float *b_vert = slices ? s->batch_grads + 3 * B * F3 : f->g_vertices;
    GomFaceArgs fa{};
    fa.N = N; fa.verts = f->work_vobs; fa.faces = f->faces; fa.so3 = f->so3; fa.scale = f->scale; fa.sigma = f->sigma;
    fa.appearance = f->appearance; fa.feat4 = f->work_feat;
    fa.d_corner = f->work_dcorner; fa.d_so3 = b_so3; fa.d_scale = b_scale; fa.d_appearance = b_app;
    fa.vdepth_minmax = s->vdepth_minmax; fa.vdepth_blocks = (N + 255) / 256;
    const GomFaceArgs *face = s->fuseFace ? &fa : nullptr;
    if (!(flags & GOM_FRAME_BACKWARD_ONLY)) {
        if (face) {   // (GOM_OPT_FUSE_FACE also fuses the kinematic chain into the skinning launch)
            if ((rc = gom_fk_lbs_forward_batch(B, N, f->cnl_gtfms, f->dst_Rs, f->dst_Ts, f->vertices, f->lbs_weights, f->work_RT, f->work_fk, f->work_vobs, stream,
                                               &f->cam, cams, s->vdepth_minmax)))
                return rc;
        } else {
            if ((rc = gom_fk_forward_batch(B, f->cnl_gtfms, f->dst_Rs, f->dst_Ts, f->work_RT, f->work_fk, stream))) return rc;
            if ((rc = gom_lbs_forward_batch(B, N, J, f->vertices, f->lbs_weights, f->work_RT, f->work_vobs, stream))) return rc;
        }
        if (!face) {
            if ((rc = gom_face_forward_batch(B, N, F, f->work_vobs, f->faces, f->so3, f->scale, f->sigma, f->work_xyz, f->work_cov6, f->appearance,
                                             f->work_feat, stream)))
                return rc;
        }
        // batched launches: the assembly pass of the forward carries the riders that order the backward's task queue by the cost the forward counted
        const bool ride = B > 1 && s->bwdOrder;   // (and the depth ranking, decided inside the forward: its tile pass zeroes the cost words)
        s->rideBwdOrder = ride;
        // GOM_OPT_FUSE_LOSS: the loss rides in the forward's emit and assembly launches (GomLossRider) -- when there is an emit launch (F > 0)
        const bool fuse_loss = s->fuseLoss && s->lossSkip && F > 0 && f->gt_rgb && f->gt_mask && f->bgcolor && f->work_dimage && f->loss_partials;
        if (fuse_loss) {
            const int HW = H * W;
            s->lossRider = GomLossRider{f->gt_rgb, f->gt_mask, f->bgcolor, 1.0f * f->c_rgb / (3.0f * (float)HW), 1.0f * f->c_mask / (float)HW, f->work_dimage, f->loss_partials,
                                        gom_frame_loss_slots(H, W), (flags & GOM_FRAME_FORWARD_ONLY) ? 1 : 0};
        }
        rc = raster_forward_impl(s, &f->cam, cams, B, F, 4, f->work_xyz, f->work_cov6, f->work_feat, f->work_opacity, f->image, f->work_radii, 0, stream, face);
        s->rideBwdOrder = false;
        s->lossRider = GomLossRider{};
        if (rc) return rc;

        if (!fuse_loss) {
            GomLossSkip skip{s->tile_base, cams, {f->cam.bg[0], f->cam.bg[1], f->cam.bg[2], f->cam.bg[3]}, s->gx, s->gy, W, (flags & GOM_FRAME_FORWARD_ONLY) ? 1 : 0};
            if ((rc = gom_l1_loss_batch(B, H, W, f->image, nullptr, f->gt_rgb, f->gt_mask, f->bgcolor, f->c_rgb, f->c_mask, 1.0f, f->work_dimage,
                                        nullptr, f->loss_partials, stream, s->lossSkip ? &skip : nullptr, gom_frame_loss_slots(H, W))))
                return rc;
        }
        s->bwdOrderReady = ride && s->rankSort;
    }
    if (flags & GOM_FRAME_FORWARD_ONLY) return 0;
    if ((rc = raster_backward_impl(s, &f->cam, cams, B, F, 4, f->work_xyz, f->work_cov6, f->work_feat, f->work_opacity, f->work_dimage,
                                   f->work_dxyz, f->work_dcov6, f->work_dfeat, f->work_dopacity, nullptr, 0, stream, face)))
        return rc;
    if (!face) {
        if ((rc = gom_face_backward_batch(B, N, F, f->work_vobs, f->faces, f->so3, f->scale, f->sigma, f->work_dxyz, f->work_dcov6,
                                          f->work_dcorner, b_so3, b_scale, f->work_dfeat, b_app, stream)))
            return rc;
    }
    if ((rc = gom_vertex_backward_batch(B, F, N, J, f->vertices, f->lbs_weights, f->work_RT, f->csr_off, f->csr_idx, f->work_dcorner,
                                        nullptr, nullptr, b_vert, nullptr, stream)))
        return rc;
    if (no_sum) return 0;
    if (B > 1) {
        if ((rc = gom_sum_frames4(B, F3, b_so3, f->g_so3, F3, b_scale, f->g_scale, F3, b_app, f->g_appearance, N3, b_vert, f->g_vertices,
                                  stream)))
            return rc;
    }
    if (s->adam.on) {   // the optimizer step behind the gradients, inside the same (recordable) launch sequence
        if ((rc = gom_adam_flat_graphable(s->adam.n, s->adam.params, s->adam.grads, s->adam.exp_avg, s->adam.exp_avg_sq, s->adam.n_segments, s->adam.seg_begin,
                                          s->adam.seg_lr, nullptr, 1, s->adam.step_device, s->adam.lr_decay_steps, s->adam.beta1, s->adam.beta2, s->adam.eps, s->adam.grad_scale,
                                          stream)))
            return rc;
    }
    return 0;
}

// ---- the camera block of renderer/gaussian.py:28-51 written into a DEVICE GomCamera by one launch -------------------------------------------------
// K (3,3) and E (4,4) are device tensors in the reference's interface; its forward reads them back with .item() (a stream synchronisation per
// frame), the package's device-side form was ~40 small tensor launches.  One thread does gomavatar_amd.camera.camera_block's arithmetic in its
// order and precision -- fp64 for K's entries, tan(atan(.)) in fp64 rounded once to fp32, fp32 products summed in k order without contraction
// (a row of K_ndc has at most two non-zeros: adding the zero products is exact) -- so the struct is the host path's, bit for bit on the
// reference's golden camera (tests/test_camera.py).
namespace {
__global__ void __launch_bounds__(64) k_camera_update(const float *__restrict__ K, const float *__restrict__ E, int H, int W, double z_a, double z_b,
                                                      const float *__restrict__ bg, GomCamera *__restrict__ cam) {
#pragma clang fp contract(off)   // every product below is rounded on its own (HIP's __fmul_rn / __fadd_rn are inline functions compiled under the command line's contraction mode: fusable)
    if (threadIdx.x != 0) return;
    const double fx = K[0], fy = K[4], px = K[2], py = K[5];
    cam->H = H; cam->W = W;
    cam->tanfovx = (float)tan(atan((double)W / (2.0 * fx)));
    cam->tanfovy = (float)tan(atan((double)H / (2.0 * fy)));
    float Kn[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) Kn[i][j] = 0.f;
    Kn[0][0] = (float)(2.0 * fx / (double)W); Kn[0][2] = (float)((2.0 * px - (double)W) / (double)W);
    Kn[1][1] = (float)(2.0 * fy / (double)H); Kn[1][2] = (float)((2.0 * py - (double)H) / (double)H);
    Kn[2][2] = (float)z_a; Kn[2][3] = (float)z_b;
    Kn[3][2] = 1.f;
    float view[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int k = 0; k < 4; k++) view[i][k] = E[4 * k + i];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            float acc = view[i][0] * Kn[j][0];
#pragma unroll
            for (int k = 1; k < 4; k++) acc = acc + view[i][k] * Kn[j][k];   // (plain operators: the pragma above governs them, not the bodies of __fmul_rn / __fadd_rn)
            cam->view[4 * i + j] = view[i][j];
            cam->proj[4 * i + j] = acc;
        }
    if (bg) {
#pragma unroll
        for (int c = 0; c < 4; c++) cam->bg[c] = bg[c];
    }
}
}  // namespace

extern "C" int gom_camera_update_device(const float *K, const float *E, int H, int W, double znear, double zfar, const float *bg4, GomCamera *cam_device, void *stream) {
    if (!K || !E || !cam_device || H <= 0 || W <= 0 || !(zfar > znear)) { gom_set_error("gom_camera_update_device: bad arguments"); return -1; }
    hipLaunchKernelGGL(k_camera_update, dim3(1), dim3(64), 0, (hipStream_t)stream, K, E, H, W, zfar / (zfar - znear), -zfar * znear / (zfar - znear), bg4, cam_device);
    GOM_LAUNCH_CHECK();
    return 0;
}
