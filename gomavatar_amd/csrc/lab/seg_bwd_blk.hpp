// Render backward with (sub-range, 4 x 4 pixel block) work items, one per 16-lane DPP ROW of a wave -- included by raster_render.hip
// inside its anonymous namespace (it shares load_sub, cull_entry, gauss_power, PairQueue, ld4 with the kernels there).
//
// Why (profiles/r02_valu.json, r02_cull_granularity.txt): k_seg_bwd / k_seg_bwd_pair evaluate every entry that survives the cull of an
// 8 x 8 quadrant on all 64 lanes and reduce its 10 terms over the wave with 34 cross-lane instructions; 13 % of the evaluated lanes do
// useful work, and the kernel is bound by VALU issue on those dead lanes, not by bytes.  Here
//   * the cull is per 4 x 4 BLOCK (16 per tile): half the lane evaluations of an exact quadrant cull;
//   * a wave carries FOUR items, one per row, each row walking the survivors of its own block back to front: the reduction is the
//     transposed in-row tree of row_reduce.hpp, 21 v_add_f32_dpp for the four entries of a trip (5.25 per entry instead of 34);
//   * the 16 blocks of the tile are dealt to the four waves IN ORDER OF THEIR SURVIVOR COUNT (ranks 4g .. 4g + 3 to one wave), so the
//     four rows of a wave finish within a few trips of each other -- rows of one quadrant would run as long as its busiest block
//     (0.83 of the quadrant kernel's trips instead of 0.5, scripts/cull_granularity.py);
//   * a row leaves the totals of its t-th survivor at position t of its own LDS strip (plain stores) and the flush -- thread = entry --
//     collects an entry's totals from the blocks it survived in, in block order: bitwise reproducible, no atomics of any kind.
// Same task words and the same cost-ordered queue as k_seg_bwd_pair (a pair of sub-ranges is walked one after the other).
#include "row_reduce.hpp"

#ifndef GOM_BWDB_WAVES
#define GOM_BWDB_WAVES 5
#endif
#ifdef GOM_BLK_STATS   // development (scripts/exp_build.py blkstats -DGOM_BLK_STATS; scripts/blk_stats.py): what the launch really did
__device__ unsigned long long g_blk_stats[8];
#define GOM_BLK_STAT(I, V) do { if (lane == 0) atomicAdd(&g_blk_stats[I], (unsigned long long)(V)); } while (0)
#else
#define GOM_BLK_STAT(I, V) do { } while (0)
#endif
#define GOM_BWDB_POS 32   // list positions of a row per chunk of the replay (the rows' totals wait in LDS by POSITION until the flush)

// cull_entry split into what depends on the entry alone and a branch-free test per rectangle (four blocks per wave and task): the
// same conservative bound, never culls when unsure.  `always` = keep whatever the rectangle, `never` = cull whatever the rectangle.
struct CullPre { float ra, rcz, lthr; bool always, never; };
__device__ __forceinline__ CullPre cull_pre(float a, float b, float cz, float o) {
    CullPre p;
    const bool posdef = (a > 0.f) && (cz > 0.f) && (a * cz - b * b > 0.f);
    p.never = posdef && o <= 0.f;
    p.always = !posdef || !(o < 3.0e38f);
    p.ra = __builtin_amdgcn_rcpf(a);
    p.rcz = __builtin_amdgcn_rcpf(cz);
    p.lthr = -__logf(255.0f * o);
    return p;
}
__device__ __forceinline__ bool cull_rect(const CullPre &p, float cx, float cy, float a, float b, float cz, float x0, float y0, float x1, float y1) {
    const float X = cx < x0 ? (x0 - cx) : (cx > x1 ? (x1 - cx) : 0.f);
    const float Y = cy < y0 ? (y0 - cy) : (cy > y1 ? (y1 - cy) : 0.f);
    float dy = -b * X * p.rcz;
    dy = fminf(fmaxf(dy, y0 - cy), y1 - cy);
    const float qx = a * X * X + 2.f * b * X * dy + cz * dy * dy;
    float dx = -b * Y * p.ra;
    dx = fminf(fmaxf(dx, x0 - cx), x1 - cx);
    const float qy = a * dx * dx + 2.f * b * dx * Y + cz * Y * Y;
    const float q = fminf(X != 0.f ? qx : 3.0e38f, Y != 0.f ? qy : 3.0e38f);
    const float DX = fmaxf(fabsf(x0 - cx), fabsf(x1 - cx));
    const float DY = fmaxf(fabsf(y0 - cy), fabsf(y1 - cy));
    const float mag = a * DX * DX + 2.f * fabsf(b) * DX * DY + cz * DY * DY;
    const bool outside = X != 0.f || Y != 0.f;
    const bool far = (-0.5f * q + (1e-5f * mag + 1e-2f)) < p.lthr;
    return p.never || (!p.always && outside && far);
}

__device__ __forceinline__ uint32_t row_max_u32(uint32_t v) {   // maximum over the 16 lanes of the row, in every lane
#define GOM_RMAX(CTRL) { const uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true); v = o > v ? o : v; }
    GOM_RMAX(0xB1) GOM_RMAX(0x4E) GOM_RMAX(0x141) GOM_RMAX(0x140)   // quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror
#undef GOM_RMAX
    return v;
}

// number of the 16 lanes of the row whose key is larger than this lane's (keys are unique inside a row)
__device__ __forceinline__ uint32_t row_rank_desc(uint32_t key) {
    uint32_t rank = 0;
#define GOM_RR(S) { const uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)key, 0x120 + S, 0xf, 0xf, true); rank += o > key ? 1u : 0u; }
    GOM_RR(1) GOM_RR(2) GOM_RR(3) GOM_RR(4) GOM_RR(5) GOM_RR(6) GOM_RR(7) GOM_RR(8) GOM_RR(9) GOM_RR(10) GOM_RR(11) GOM_RR(12) GOM_RR(13) GOM_RR(14) GOM_RR(15)
#undef GOM_RR
    return rank;
}

template <int C>
__global__ void __launch_bounds__(256, GOM_BWDB_WAVES) k_seg_bwd_blk(uint32_t seg_shift, int H, int W, int gx, int gy, float bg0, float bg1, float bg2, float bg3,
                                                  const GomCamera *__restrict__ cams,
                                                  const uint4 *__restrict__ seg_desc, const uint4 *__restrict__ seg_qmax,
                                                  const float2 *__restrict__ ent_geo, const float *__restrict__ ent_col,
                                                  const float *__restrict__ final_T, const uint32_t *__restrict__ n_contrib,
                                                  const float *__restrict__ dL_dpix, const float *__restrict__ sub_Tend,
                                                  const float *__restrict__ sub_C, const float *__restrict__ seg_Sbehind,
                                                  const uint32_t *__restrict__ ent_slot, float *__restrict__ partial, const GomDevStatus *__restrict__ status,
                                                  uint32_t *__restrict__ task_ctr, const uint32_t *__restrict__ task_order, const unsigned long long *__restrict__ cull_masks) {
    constexpr int NV = 6 + C;
    // [wave][row][position in the row's list, modulo a chunk][10 values + 2 padding slots]: every (row, position) is written once per chunk
    // with a plain store -- LDS float atomics on per-entry records cost ~140 cycles an instruction (ds_add_f32, measured: the first version)
    __shared__ __attribute__((aligned(16))) float s_acc[4][4][GOM_BWDB_POS][12];
    // the sub-range's entries, staged once per workgroup (every wave a quarter); slot GOM_SUB_MAX = the null entry (opacity 0)
    __shared__ float4 s_e0[GOM_SUB_MAX + 1], s_e2[GOM_SUB_MAX + 1];
    __shared__ float2 s_e1[GOM_SUB_MAX + 1];
    __shared__ __attribute__((aligned(4))) uint8_t s_idx[4][4][GOM_SUB_MAX];   // [wave][row] the row's survivors, LAST first, padded with the null entry
    __shared__ unsigned long long s_bmask[2][16];                               // [task parity] survivors of block (quadrant * 4 + block inside the quadrant)
    __shared__ uint8_t s_brank[2][16];                                          // [task parity] the block's rank by survivor count (-> which wave and row carried it)
    __shared__ uint32_t s_task[2];
    const uint32_t sub_sz = (1u << seg_shift) >> 2;
    if (status->overflow) return;
    const uint32_t nsegs = status->num_segs;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, row = lane >> 4, l16 = lane & 15;
    const size_t HW = (size_t)H * W;
    const int slot = row_sum10_slot(lane);
    if (threadIdx.x == 0) {
        s_e0[GOM_SUB_MAX] = make_float4(0.f, 0.f, 0.f, 0.f); s_e1[GOM_SUB_MAX] = make_float2(0.f, -INFINITY);   /* (Cq, lo = log2 of opacity 0: alpha 0) */ s_e2[GOM_SUB_MAX] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    uint32_t parity = 0;
    PairQueue tq;
    for (tq.init(task_ctr ? task_ctr + 2 * GOM_TQ_WORDS : nullptr, nsegs, s_task, task_order);; tq.advance()) {
        const uint32_t task = tq.current(s_task);
        if (task == 0xffffffffu) break;
        const uint32_t seg = task >> 3, code = task & 7u;
        const int nhalf = code < 4u ? 2 : 1;
        const int sub_a = code < 4u ? (int)code * 2 : (int)code - 4;
        const uint4 d = seg_desc[seg];
        const uint4 qm4 = seg_qmax[seg];
        const uint32_t tile = d.x, start = d.y, cnt = d.z;
        const uint32_t e0 = d.w << seg_shift;
        const uint32_t tmax = max(max(qm4.x, qm4.y), max(qm4.z, qm4.w));
        if ((uint32_t)sub_a * sub_sz >= cnt || e0 + (uint32_t)sub_a * sub_sz >= tmax) {   // nothing alive: nothing is written (see k_seg_bwd)
            tq.request();
            tq.publish(s_task);
            __syncthreads();
            continue;
        }
        const int tx = tile % gx, fr = (tile / gx) / gy, ty = (tile / gx) % gy;
        float bg[4] = {bg0, bg1, bg2, bg3};
        if (cams) {
#pragma unroll
            for (int ch = 0; ch < 4; ch++) bg[ch] = cams[fr].bg[ch];
        }
        // the second sub-range of a pair may be empty or dead for the whole tile (workgroup-uniform)
        const bool live2 = nhalf == 2 && (uint32_t)(sub_a + 1) * sub_sz < cnt && e0 + (uint32_t)(sub_a + 1) * sub_sz < tmax;
        const int last_half = live2 ? 1 : 0;
        bool requested = false;
        for (int half = 0; half <= last_half; half++) {
            const int sub = sub_a + half;
            const uint32_t s0 = e0 + (uint32_t)sub * sub_sz;
            const uint32_t scnt = min(sub_sz, cnt - (uint32_t)sub * sub_sz);
            // ---- phase 0: this wave's survivor lists
            reinterpret_cast<uint32_t *>(&s_idx[wv][0][0])[lane] = 0x01010101u * (uint32_t)GOM_SUB_MAX;
            // ---- phase 1: wave = quadrant wv, row = one of its four blocks: which entries can reach the block at all
            {
                const int bxq = (wv & 1) * 2, byq = (wv >> 1) * 2;
                const int px = tx * 16 + (bxq + (row & 1)) * 4 + (l16 & 3), py = ty * 16 + (byq + (row >> 1)) * 4 + (l16 >> 2);
                const uint32_t nl = (px < W && py < H) ? n_contrib[(size_t)fr * HW + (size_t)py * W + px] : 0u;
                const uint32_t lim = min(cnt, tmax - e0);
                const EntryRegs<C> r = load_sub<C>(ent_geo, ent_col, start, lim, sub, lane, 0.f, 0.f, 0.f, 0.f, sub_sz, cull_masks + ((size_t)seg * GOM_NSUB + sub) * 4 + wv);
                if (!requested) { tq.request(); requested = true; }   // (behind the loads of this piece)
                if (row == wv) {   // this wave's quarter of the shared slab
                    s_e0[lane] = make_float4(r.x, r.y, r.a, r.b);
                    s_e1[lane] = make_float2(r.c, r.o);
                    float4 cl = make_float4(r.col[0], 0.f, 0.f, 0.f);
                    if (C > 1) cl.y = r.col[1 % C];
                    if (C > 2) cl.z = r.col[2 % C];
                    if (C > 3) cl.w = r.col[3 % C];
                    s_e2[lane] = cl;
                }
                const uint32_t bm = row_max_u32(nl);   // last contributor over the block's 16 pixels
                const float4 ur = entry_unrecord(r.a, r.b, r.c, r.o);   // (conic, opacity) back from the list record for the conservative block cull
                const CullPre cp = cull_pre(ur.x, ur.y, ur.z, ur.w);
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    const uint32_t bmax = (uint32_t)__builtin_amdgcn_readlane((int)bm, 16 * b);
                    unsigned long long m = 0ull;
                    if (bmax > s0) {   // (wave-uniform) the block still had a pixel alive when the list reached this sub-range
                        const float x0 = (float)(tx * 16 + (bxq + (b & 1)) * 4), y0 = (float)(ty * 16 + (byq + (b >> 1)) * 4);
                        const bool keep = r.keep && s0 + (uint32_t)lane < bmax && !cull_rect(cp, r.x, r.y, ur.x, ur.y, ur.z, x0, y0, x0 + 3.f, y0 + 3.f);
                        m = __ballot(keep);
                    }
                    if (lane == 0) s_bmask[parity][wv * 4 + b] = m;
                }
            }
            __syncthreads();   // (A) masks and entries of all four quadrants
            // ---- phase 2: deal the 16 blocks to the waves by survivor count; wave wv takes ranks 4g .. 4g + 3
            const unsigned long long mj = s_bmask[parity][l16];
            const uint32_t rank = row_rank_desc(((uint32_t)__popcll(mj) << 4) | (uint32_t)(15 - l16));   // (count, lower block id first) descending
            if (threadIdx.x < 16) s_brank[parity][l16] = (uint8_t)rank;
            const uint32_t nmax = (uint32_t)__builtin_amdgcn_readfirstlane((int)row_max_u32((uint32_t)__popcll(mj)));   // the longest list of the 16 blocks
            const uint32_t g = ((uint32_t)wv + seg + (uint32_t)sub) & 3u;   // rotates the heavy group over the waves (= the SIMDs) from task to task; a function of the task alone
            uint32_t bid_r[4], n_r[4];
            unsigned long long M_r[4];
            uint32_t trips = 0;
#pragma unroll
            for (int rr = 0; rr < 4; rr++) {
                const unsigned long long hit = __ballot(rank == 4u * g + (uint32_t)rr);
                bid_r[rr] = (uint32_t)__builtin_ctzll(hit) & 15u;   // (every row holds the same 16 ranks: the lowest hit is in row 0)
                const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)mj, (int)bid_r[rr]);
                const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(mj >> 32), (int)bid_r[rr]);
                M_r[rr] = ((unsigned long long)hi << 32) | lo;
                n_r[rr] = (uint32_t)__popcll(M_r[rr]);
                trips = max(trips, n_r[rr]);
                if ((M_r[rr] >> lane) & 1ull) {   // lane = entry: its place in the row's list, last survivor first
                    const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(M_r[rr] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)M_r[rr], 0u));
                    s_idx[wv][rr][n_r[rr] - 1u - below] = (uint8_t)lane;
                }
            }
            GOM_BLK_STAT(0, wv == 0 ? 1 : 0); GOM_BLK_STAT(1, trips); GOM_BLK_STAT(3, n_r[0] + n_r[1] + n_r[2] + n_r[3]); GOM_BLK_STAT(5, wv == 0 ? nmax : 0);
            GOM_BLK_STAT(4, wv == 0 && nmax > GOM_BWDB_POS ? 1 : 0);
            // ---- phase 3: lane = pixel of the row's block
            const uint32_t bid = row == 0 ? bid_r[0] : (row == 1 ? bid_r[1] : (row == 2 ? bid_r[2] : bid_r[3]));
            const int q = (int)(bid >> 2), bq = (int)(bid & 3u);
            const int pxt = (q & 1) * 8 + (bq & 1) * 4 + (l16 & 3), pyt = (q >> 1) * 8 + (bq >> 1) * 4 + (l16 >> 2);
            const int pxi = q * 64 + (pyt & 7) * 8 + (pxt & 7);
            const int px = tx * 16 + pxt, py = ty * 16 + pyt;
            const bool inside = px < W && py < H;
            const size_t pix = (size_t)py * W + px;
            const size_t fpix = (size_t)fr * HW + pix;
            const float pfx = (float)px, pfy = (float)py;
            float T = 0.f, R_acc = 0.f, U_last = 0.f, last_alpha = 0.f, T_final = 0.f, bg_dot = 0.f, dpix[C];
            uint32_t my_last = 0u;
#pragma unroll
            for (int ch = 0; ch < C; ch++) dpix[ch] = 0.f;
            if (trips) {   // (wave-uniform)
                my_last = inside ? n_contrib[fpix] : 0u;
                T_final = final_T[inside ? fpix : 0];
#pragma unroll
                for (int ch = 0; ch < C; ch++) {
                    dpix[ch] = inside ? dL_dpix[((size_t)fr * C + ch) * HW + pix] : 0.f;
                    bg_dot += bg[ch] * dpix[ch];
                }
                T = sub_Tend[((size_t)seg * GOM_NSUB + sub) * GOM_TPX + pxi];
                float S[C], cu[C];
                ld4<C>(seg_Sbehind, seg, pxi, S);
                if (sub < GOM_NSUB - 1) {
                    ld4<C>(sub_C, (size_t)seg * GOM_NSUB + sub, pxi, cu);
#pragma unroll
                    for (int ch = 0; ch < C; ch++) S[ch] += cu[ch];
                }
                float sd = 0.f;
#pragma unroll
                for (int ch = 0; ch < C; ch++) sd += S[ch] * dpix[ch];
                R_acc = sd * (T > 0.f ? 1.f / T : 0.f);   // (same operations as k_seg_bwd)
            }
            // ---- the replay, in chunks of GOM_BWDB_POS list positions: trip t = the t-th survivor from the back of each row's list
            const uint8_t *my_idx = &s_idx[wv][row][0];
            const int j3 = l16 & 3;
            float4 fr0 = make_float4(0.f, 0.f, 0.f, 0.f), fr1 = fr0;   // the flush threads' record (entry = thread), summed over the chunks
            float2 fr2 = make_float2(0.f, 0.f);
            const uint32_t nchunks = (nmax + GOM_BWDB_POS - 1u) / GOM_BWDB_POS;   // (workgroup-uniform; at least one list is non-empty or nmax = 0)
            for (uint32_t c = 0; c < max(nchunks, 1u); c++) {
                const uint32_t t_end = min(trips, (c + 1u) * GOM_BWDB_POS);
                for (uint32_t t = c * GOM_BWDB_POS; t < t_end; t++) {
                    const uint32_t k = my_idx[t];
                    const float4 g0 = s_e0[k], c4 = s_e2[k];
                    const float2 g1 = s_e1[k];
                    float *dst = &s_acc[wv][row][t & (GOM_BWDB_POS - 1u)][slot];
                    const AlphaEval<float> ev = alpha_eval<float>(g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, pfx, pfy);   // (the forward's alphas, bit for bit)
                    const float dx = ev.dx, dy = ev.dy;
                    const float a = (s0 + k < my_last) ? ev.al : 0.f;   // beyond this pixel's last contributor (the null entry: opacity 0)
                    if (__ballot(a > 0.f) == 0ull) { *dst = 0.f; GOM_BLK_STAT(2, 1); continue; }   // wave-uniform: four zero records
                    GOM_BLK_STAT(6, __popcll(__ballot(a > 0.f)));
                    const float G0 = (a > 0.f) ? ev.og : 0.f;   // (opacity * G: the records carry the opacity since round 3)
                    const float inv1ma = __builtin_amdgcn_rcpf(1.f - a);
                    T = T * inv1ma;
                    const float w = a * T;
                    float w10[10], U = 0.f;
                    {
                        const float cv[4] = {c4.x, c4.y, c4.z, c4.w};
#pragma unroll
                        for (int ch = 0; ch < 4; ch++) w10[ch] = 0.f;
#pragma unroll
                        for (int ch = 0; ch < C; ch++) {
                            U += cv[ch] * dpix[ch];
                            w10[ch] = w * dpix[ch];
                        }
                    }
                    R_acc = last_alpha * U_last + (1.f - last_alpha) * R_acc;
                    U_last = U;
                    float dL_dalpha = (U - R_acc) * T;
                    last_alpha = a;
                    dL_dalpha += (-T_final * inv1ma) * bg_dot;
                    const float Q = G0 * dL_dalpha;
                    w10[4] = Q;
                    w10[5] = Q * dx;
                    w10[6] = Q * dy;
                    w10[7] = Q * dx * dx;
                    w10[8] = Q * dx * dy;
                    w10[9] = Q * dy * dy;
                    // a lane whose entry was skipped for its pixel (a == 0) contributes exact zeros and leaves T / R_acc as skipping would
                    float t0, t1, t2;
                    row_sum10_t(w10, t0, t1, t2);
                    *dst = j3 == 0 ? t0 : (j3 == 1 ? t1 : t2);   // (slots 10, 11: padding)
                }
                if (half == last_half && c + 1u >= nchunks) {
                    if (!requested) tq.request();
                    tq.publish(s_task);
                }
                __syncthreads();   // (B) the rows' totals of this chunk are in LDS
                if (threadIdx.x < scnt) {   // entry = thread: its totals from the blocks it survived in, in BLOCK order (a function of the tile's geometry alone)
                    const uint32_t e = threadIdx.x;
#pragma unroll 4
                    for (int b = 0; b < 16; b++) {
                        const unsigned long long Mb = s_bmask[parity][b];
                        if ((Mb >> e) & 1ull) {
                            const uint32_t pos = (uint32_t)__popcll(Mb >> e) - 1u;   // survivors at or behind e, minus itself = its place counted from the back
                            if (pos / GOM_BWDB_POS == c) {
                                const uint32_t rk = s_brank[parity][b];
                                const uint32_t w4 = ((rk >> 2) - seg - (uint32_t)sub) & 3u;   // the wave that carried rank group rk / 4
                                const float4 *p = reinterpret_cast<const float4 *>(&s_acc[w4][rk & 3u][pos & (GOM_BWDB_POS - 1u)][0]);
                                const float4 a0 = p[0], a1 = p[1], a2 = p[2];
                                fr0.x += a0.x; fr0.y += a0.y; fr0.z += a0.z; fr0.w += a0.w;
                                fr1.x += a1.x; fr1.y += a1.y; fr1.z += a1.z; fr1.w += a1.w;
                                fr2.x += a2.x; fr2.y += a2.y;
                            }
                        }
                    }
                }
                if (c + 1u < nchunks) __syncthreads();   // the next chunk overwrites the positions
            }
            (void)NV;
            if (threadIdx.x < scnt) {   // one 48-byte record per entry
                const uint32_t rs = ent_slot[start + (uint32_t)sub * sub_sz + threadIdx.x];
                float4 *rec = reinterpret_cast<float4 *>(partial + (size_t)rs * GOM_PARTIAL_STRIDE);
                rec[0] = fr0; rec[1] = fr1; rec[2] = make_float4(fr2.x, fr2.y, 0.f, 0.f);
            }
            parity ^= 1u;
        }
    }
    tq.finish();
}
