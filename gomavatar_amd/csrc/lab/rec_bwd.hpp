// Render backward, records mode (round 4; included inside raster_render.hip's anonymous namespace).
//
// Replaces the back-to-front replay of renderCUDA's backward (the CUDA extension the reference calls at
// models/modules/renderer/gaussian.py:83-91; algorithm SURVEY.md App. A.4) by a walk over the BLENDING (pixel, entry) pairs only.
//
// Why.  On a GoMAvatar frame an entry of a tile list blends ~4.7 of the tile's 256 pixels.  The replay kernels (k_seg_bwd*, lane = pixel,
// scalar loop over the entries of a sub-range) evaluated every surviving entry on 64 lanes of which ~8 were alive, and paid a 64-lane
// reduction tree per entry: 49 M VALU wave-instructions per 8-frame launch, 0.10 of the HBM roofline, three rounds of in-loop tuning.
// The gradient of a pair does not need the replay: with w_j = alpha_j T_j, u_j = c_j . dL/dpix, suffix_i = sum_{j > i} w_j u_j,
//     dL/dalpha_i = T_i u_i - (suffix_i + T_final bg . dL/dpix) / (1 - alpha_i)
// and suffix_i = S_from(piece) - acc_i - w_i u_i, where S_from is the colour from the piece's first entry to the end of the pixel's
// list (k_seg_fwd's fold + k_combine_fwd leave it per piece and pixel) and acc_i the colour the piece had added in front of entry i --
// which, with T_i, the compositing pass holds in registers at the moment the lane blends: it writes them as a record (GomRecArgs),
// entry-major inside the piece's region, with the number of records of every entry beside it (piece_cnt).
// The cancellation in S_from - acc_i is confined to one piece (<= 64 entries): error ~ eps x the piece's colour.
//
// Version 2 (this file): lane = ENTRY.  A wave owns a (segment, sub-range); lane e keeps entry e in registers and walks ITS records of
// the four quadrants one after the other -- alpha (alpha_eval: the forward's, bit for bit), the ten terms of the pair, ten adds into ten
// registers -- and writes the entry's 48-byte record at its Gaussian-major slot, exactly as the replay kernels did.  No reduction across
// lanes, no barrier, no shared accumulator: the sum of an entry has one order (quadrant by quadrant, record by record) by construction.
// A trip costs what its busiest lane needs, so the lanes are ~25 % busy -- but a trip is ~60 instructions where the replay spent ~100 on
// a 13 %-alive wave plus its reduction tree.
// Version 1 (a lane per record, ten ds_add_f32 per record into [value][entry] rows in LDS; git history of this file) was parity-green and
// 3x SLOWER than the replay: LDS float atomics serialise on the lanes that share an address and cost ~4 cycles per lane and instruction on
// gfx950 -- 365 of its 474 us (knock-outs: profiles/r04_records_backward.txt).
#pragma once

#ifndef GOM_RECB_WAVES
#define GOM_RECB_WAVES 8
#endif

__device__ __forceinline__ uint32_t wave_excl_scan_u32(uint32_t v, int lane) {   // exclusive prefix sum over the 64 lanes
    uint32_t s = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(s, d, 64);
        s += lane >= d ? o : 0u;
    }
    return s - v;
}

template <int C>
__global__ void __launch_bounds__(256, GOM_RECB_WAVES) k_rec_bwd(uint32_t seg_shift, int H, int W, int gx, int gy, float bg0, float bg1, float bg2, float bg3,
                                                  const GomCamera *__restrict__ cams,
                                                  const uint4 *__restrict__ seg_desc, const uint4 *__restrict__ seg_qmax,
                                                  const float2 *__restrict__ ent_geo, const float *__restrict__ ent_col,
                                                  const float *__restrict__ final_T, const uint32_t *__restrict__ n_contrib,
                                                  const float *__restrict__ dL_dpix, const float *__restrict__ sub_C, const float *__restrict__ seg_Sbehind,
                                                  const uint32_t *__restrict__ ent_slot, float *__restrict__ partial, const GomDevStatus *__restrict__ status,
                                                  GomRecArgs rec) {
    __shared__ float4 s_pix[4][64][2];   // per WAVE: the current quadrant's pixels -- dL/dpix[0..3]; (S_from . dL/dpix, T_final bg . dL/dpix, n_contrib bits, -)
    const uint32_t sub_sz = (1u << seg_shift) >> 2;
    if (status->overflow) return;
    const bool poisoned = status->rec_overflow != 0u;   // the forward ran out of record space: NaN rows, loudly (gom_state_poll bit 1)
    const uint32_t nsegs = status->num_segs;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const size_t HW = (size_t)H * W;
    // A wave's tasks: (segment, sub-range) items of shard x = blockIdx % 8 (the XCD: raster_render.hip TaskQueue), the four waves of a
    // workgroup taking the four sub-ranges of one segment; plain striding (the tasks are many and short: ~19 000 per 8-frame launch).
    const bool sharded = gridDim.x % GOM_TQ_SHARDS == 0;
    const uint32_t shard = blockIdx.x % GOM_TQ_SHARDS, per_shard = gridDim.x / GOM_TQ_SHARDS;
    for (uint32_t it = 0;; it++) {
        uint32_t seg;
        if (sharded) {
            const uint32_t k = blockIdx.x / GOM_TQ_SHARDS + it * per_shard;
            if (k >= gom_shard_segments(nsegs, shard)) break;
            seg = gom_shard_segment(shard, k);
        } else {
            seg = blockIdx.x + it * gridDim.x;
            if (seg >= nsegs) break;
        }
        const int sub = wv;
        const uint4 d = seg_desc[seg];
        const uint4 qm4 = seg_qmax[seg];
        const uint32_t tile = d.x, start = d.y, cnt = d.z;
        const uint32_t e0 = d.w << seg_shift;
        const uint32_t s0 = e0 + (uint32_t)sub * sub_sz;  // list index of the first entry of the sub-range
        if ((uint32_t)sub * sub_sz >= cnt) continue;      // no entries
        const uint32_t scnt = min(sub_sz, cnt - (uint32_t)sub * sub_sz);
        const uint32_t tmax = max(max(qm4.x, qm4.y), max(qm4.z, qm4.w));
        if (s0 >= tmax) continue;   // every pixel of the tile stopped before this sub-range: nothing is written (the per-Gaussian backward knows)
        const int tx = tile % gx, fr = (tile / gx) / gy, ty = (tile / gx) % gy;  // fr: frame of a batched launch
        // my entry (lane = entry of the sub-range)
        float ex = 0.f, ey = 0.f, eA = 0.f, eB = 0.f, eC = 0.f, elo = -INFINITY, c0 = 0.f, c1 = 0.f, c2 = 0.f, c3 = 0.f;
        uint32_t my_slot = 0;
        const bool have_entry = (uint32_t)lane < scnt;
        if (have_entry) {
            const size_t ei = (size_t)start + (uint32_t)sub * sub_sz + (uint32_t)lane;
            const float2 *g = ent_geo + 3 * ei;
            const float2 g0 = g[0], g1 = g[1], g2 = g[2];
            const float4 cl = *reinterpret_cast<const float4 *>(ent_col + 4 * ei);
            ex = g0.x; ey = g0.y; eA = g1.x; eB = g1.y; eC = g2.x; elo = g2.y;
            c0 = cl.x; c1 = cl.y; c2 = cl.z; c3 = cl.w;
            my_slot = ent_slot[ei];
        }
        float rr[10];
#pragma unroll
        for (int v = 0; v < 10; v++) rr[v] = 0.f;
        float bg[4] = {bg0, bg1, bg2, bg3};
        if (cams) {
#pragma unroll
            for (int ch = 0; ch < 4; ch++) bg[ch] = cams[fr].bg[ch];
        }
#pragma unroll 1
        for (int q = 0; q < 4; q++) {
            const uint32_t wmax = q == 0 ? qm4.x : (q == 1 ? qm4.y : (q == 2 ? qm4.z : qm4.w));  // max n_contrib over the quadrant's 8x8 pixels
            if (!(wmax > s0)) continue;   // (wave-uniform) the quadrant does not reach into the sub-range: its piece may never have run
            const size_t piece = (((size_t)seg * GOM_NSUB + sub) << 2) | (uint32_t)q;
            const uint2 pr = rec.piece_rec[piece];
            if (pr.y == 0u || pr.y == 0xffffffffu) continue;
            const uint32_t my_n = rec.piece_cnt[piece * 64 + lane];
            // the quadrant's pixels (lane = pixel) -> LDS
            {
                const int qx0 = tx * 16 + (q & 1) * 8, qy0 = ty * 16 + (q >> 1) * 8;
                const int px = qx0 + (lane & 7), py = qy0 + (lane >> 3);
                const int pxi = q * 64 + lane;
                const bool inside = px < W && py < H;
                const size_t pix = (size_t)py * W + px;
                const size_t fpix = (size_t)fr * HW + pix;
                const uint32_t my_last = inside ? n_contrib[fpix] : 0u;
                const float T_final = final_T[inside ? fpix : 0];
                float dpix[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ch = 0; ch < C; ch++) dpix[ch] = inside ? dL_dpix[((size_t)fr * C + ch) * HW + pix] : 0.f;
                float Sb[C], Sc[C];
                ld4<C>(seg_Sbehind, seg, pxi, Sb);
                ld4<C>(sub_C, (size_t)seg * GOM_NSUB + sub, pxi, Sc);
                float Sd = 0.f, bg_dot = 0.f;
#pragma unroll
                for (int ch = 0; ch < C; ch++) {
                    Sd += (Sb[ch] + Sc[ch]) * dpix[ch];
                    bg_dot += bg[ch] * dpix[ch];
                }
                s_pix[wv][lane][0] = make_float4(dpix[0], dpix[1], dpix[2], dpix[3]);
                s_pix[wv][lane][1] = make_float4(Sd, T_final * bg_dot, __uint_as_float(my_last), 0.f);
            }
            const uint32_t my_off = pr.x + wave_excl_scan_u32(my_n, lane);
            const uint32_t trips = wave_max_u32(my_n);
            const float qxf = (float)(tx * 16 + (q & 1) * 8), qyf = (float)(ty * 16 + (q >> 1) * 8);
            // One record per trip, the next one in flight meanwhile.  The loop is memory-latency-bound (every lane gathers its own cache
            // lines): measured 298 us per 8-frame launch against 33 us without it.  Four records per trip with their loads issued together
            // -- what should have hidden the latency -- spilled 64 registers at six waves per SIMD and measured 795 us (658 at two per trip,
            // 1 301 at eight): profiles/r04_records_backward.txt.
            float2 n_ti = make_float2(0.f, 0.f);
            float4 n_acc = make_float4(0.f, 0.f, 0.f, 0.f);
            if (my_n > 0u) { n_ti = rec.rec_ti[my_off]; n_acc = rec.rec_acc[my_off]; }
#if defined(GOM_KO_REC) && GOM_KO_REC == 2   // development knock-out: no record loop
            for (uint32_t k = 0; k < 0u; k++) {
#else
            for (uint32_t k = 0; k < trips; k++) {
#endif
                const float2 c_ti = n_ti;
                const float4 c_acc = n_acc;
                const bool act = k < my_n;
                if (k + 1u < my_n) { n_ti = rec.rec_ti[my_off + k + 1u]; n_acc = rec.rec_acc[my_off + k + 1u]; }
                const uint32_t p = act ? (__float_as_uint(c_ti.y) & 63u) : 0u;
                const float4 P0 = s_pix[wv][p][0], P1 = s_pix[wv][p][1];
                const float pfx = qxf + (float)(p & 7u), pfy = qyf + (float)(p >> 3);
                const AlphaEval<float> ev = alpha_eval<float>(ex, ey, eA, eB, eC, elo, pfx, pfy);
                // entries at or beyond the pixel's last contributor (a piece the fold did not count for it) do not exist for the gradient
                const bool valid = act && (s0 + (uint32_t)lane) < __float_as_uint(P1.z) && ev.al > 0.f;
                const float T = c_ti.x;
                const float al = ev.al;
                const float inv = __builtin_amdgcn_rcpf(1.f - al);
                const float w = valid ? al * T : 0.f;
                float u = c0 * P0.x, pd = c_acc.x * P0.x;
                if (C > 1) { u = __fmaf_rn(c1, P0.y, u); pd = __fmaf_rn(c_acc.y, P0.y, pd); }
                if (C > 2) { u = __fmaf_rn(c2, P0.z, u); pd = __fmaf_rn(c_acc.z, P0.z, pd); }
                if (C > 3) { u = __fmaf_rn(c3, P0.w, u); pd = __fmaf_rn(c_acc.w, P0.w, pd); }
                const float sufd = (P1.x - pd) - w * u;                      // colour behind the entry . dL/dpix
                const float dLa = __fmaf_rn(T, u, -((sufd + P1.y) * inv));   // dL/dalpha
                const float Q = valid ? (ev.og * ev.mm) * dLa : 0.f;         // opacity G dL/dalpha (the reference does not mask the 0.99 clamp here)
                const float z5 = Q * ev.dx, z6 = Q * ev.dy;
                rr[0] += w * P0.x;
                if (C > 1) rr[1] += w * P0.y;
                if (C > 2) rr[2] += w * P0.z;
                if (C > 3) rr[3] += w * P0.w;
                rr[4] += Q;
                rr[5] += z5;
                rr[6] += z6;
                rr[7] += z5 * ev.dx;
                rr[8] += z5 * ev.dy;
                rr[9] += z6 * ev.dy;
            }
        }
        if (have_entry) {   // one 48-byte record per entry of the sub-range (also the entries nothing blended: zeros)
            if (poisoned) {
#pragma unroll
                for (int v = 0; v < 10; v++) rr[v] = __uint_as_float(0x7fc00000u);
            }
            float4 *recp = reinterpret_cast<float4 *>(partial + (size_t)my_slot * GOM_PARTIAL_STRIDE);
            recp[0] = make_float4(rr[0], rr[1], rr[2], rr[3]);
            recp[1] = make_float4(rr[4], rr[5], rr[6], rr[7]);
            recp[2] = make_float4(rr[8], rr[9], 0.f, 0.f);
        }
    }
}
