// Render backward, records mode (round 4; included inside raster_render.hip's anonymous namespace).
//
// Replaces the back-to-front replay of renderCUDA's backward (the CUDA extension the reference calls at
// models/modules/renderer/gaussian.py:83-91; algorithm SURVEY.md App. A.4) by a lane per BLENDING (pixel, entry) pair.
//
// Why.  On a GoMAvatar frame an entry of a tile list blends ~4.7 of the tile's 256 pixels.  The replay kernels (k_seg_bwd*, lane = pixel,
// scalar loop over the entries of a sub-range) evaluated every surviving entry on 64 lanes of which ~8 were alive, and paid a 64-lane
// reduction tree per entry: 49 M VALU wave-instructions per 8-frame launch, 0.10 of the HBM roofline, three rounds of in-loop tuning.
// The gradient of a pair does not need the replay: with w_j = alpha_j T_j, u_j = c_j . dL/dpix, suffix_i = sum_{j > i} w_j u_j,
//     dL/dalpha_i = T_i u_i - (suffix_i + T_final bg . dL/dpix) / (1 - alpha_i)
// and suffix_i = S_from(piece) - acc_i - w_i u_i, where S_from is the colour from the piece's first entry to the end of the pixel's
// list (k_seg_fwd's fold + k_combine_fwd leave it per piece and pixel) and acc_i the colour the piece had added in front of entry i --
// which, with T_i, the compositing pass holds in registers at the moment the lane blends: it writes them as a record (GomRecArgs).
// The cancellation in S_from - acc_i is confined to one piece (<= 64 entries): error ~ eps x the piece's colour.
//
// Workgroup = (segment, sub-range), wave = 8x8 quadrant (as k_seg_bwd).  Per task: the sub-range's entries and the quadrant's pixels go
// to LDS once; the wave then walks its piece's records 64 at a time -- alpha (alpha_eval: the forward's, bit for bit), the ten terms of
// the pair, and ten ds_add_f32 into the wave's private [value][entry] accumulator rows.  The records of an entry are consecutive lanes;
// lanes of one instruction that share an address are served in lane order and the wave's instructions execute in order, so the sum of
// an entry has ONE order (bitwise repeatable; tests/test_gpu_raster.py holds it to that) -- no memory shared between waves is ever
// added to atomically.  Behind the barrier wave 0 adds the four quadrants' rows in a fixed order and writes the 48-byte record of
// every entry at its Gaussian-major slot, exactly as the replay kernels did.
#pragma once

#ifndef GOM_RECB_WAVES
#define GOM_RECB_WAVES 7
#endif

template <int C>
__global__ void __launch_bounds__(256, GOM_RECB_WAVES) k_rec_bwd(uint32_t seg_shift, int H, int W, int gx, int gy, float bg0, float bg1, float bg2, float bg3,
                                                  const GomCamera *__restrict__ cams,
                                                  const uint4 *__restrict__ seg_desc, const uint4 *__restrict__ seg_qmax,
                                                  const float2 *__restrict__ ent_geo, const float *__restrict__ ent_col,
                                                  const float *__restrict__ final_T, const uint32_t *__restrict__ n_contrib,
                                                  const float *__restrict__ dL_dpix, const float *__restrict__ sub_C, const float *__restrict__ seg_Sbehind,
                                                  const uint32_t *__restrict__ ent_slot, float *__restrict__ partial, const GomDevStatus *__restrict__ status,
                                                  uint32_t *__restrict__ task_ctr, GomRecArgs rec) {
    __shared__ float4 s_ent[GOM_SUB_MAX][3];      // (x, y, A, B) (Cq, lo, c0, c1) (c2, c3, -, -): the sub-range's entries, shared by the four waves
    __shared__ float4 s_pix[4][64][2];            // per quadrant and pixel: dL/dpix[0..3]; (S_from . dL/dpix, T_final bg . dL/dpix, n_contrib bits, -)
    __shared__ float s_acc[4][10][GOM_SUB_MAX];   // per quadrant: [value][entry] sums; zero between tasks (wave 0 clears what it folds)
    __shared__ uint32_t s_task[2];
    const uint32_t sub_sz = (1u << seg_shift) >> 2;
    if (status->overflow) return;
    const bool poisoned = status->rec_overflow != 0u;   // the forward ran out of record space: NaN rows, loudly (gom_state_poll bit 1)
    const uint32_t nsegs = status->num_segs;
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    const size_t HW = (size_t)H * W;
    for (int i = threadIdx.x; i < 4 * 10 * GOM_SUB_MAX; i += 256) (&s_acc[0][0][0])[i] = 0.f;
    __syncthreads();
    TaskQueue tq;
    for (tq.init(task_ctr ? task_ctr + 2 * GOM_TQ_WORDS : nullptr, nsegs, s_task);; tq.advance()) {
        const uint32_t task = tq.current(s_task);
        if (task == 0xffffffffu) break;
        const uint32_t seg = task >> 2;
        const int sub = (int)(task & 3);
        // everything indexed by the task alone in one round trip
        const uint4 d = seg_desc[seg];
        const uint4 qm4 = seg_qmax[seg];
        const uint2 pr = rec.piece_rec[(((size_t)seg * GOM_NSUB + sub) << 2) | (uint32_t)q];   // (meaningful only where the quadrant reaches into the piece)
        const uint32_t tile = d.x, start = d.y, cnt = d.z;
        const uint32_t e0 = d.w << seg_shift;
        const uint32_t s0 = e0 + (uint32_t)sub * sub_sz;  // list index of the first entry of the sub-range
        const bool empty = (uint32_t)sub * sub_sz >= cnt;
        const uint32_t scnt = empty ? 0u : min(sub_sz, cnt - (uint32_t)sub * sub_sz);
        const uint32_t tmax = max(max(qm4.x, qm4.y), max(qm4.z, qm4.w));
        if (empty || s0 >= tmax) {   // no entries, or every pixel of the tile stopped before this sub-range: nothing is written (the per-Gaussian backward knows)
            tq.request();
            tq.publish(s_task);
            __syncthreads();
            continue;
        }
        const uint32_t wmax = q == 0 ? qm4.x : (q == 1 ? qm4.y : (q == 2 ? qm4.z : qm4.w));  // max n_contrib over this wave's 8x8 pixels
        const bool live = wmax > s0;
        const uint32_t n_rec = live && pr.y != 0xffffffffu ? pr.y : 0u;
        const int tx = tile % gx, fr = (tile / gx) / gy, ty = (tile / gx) % gy;  // fr: frame of a batched launch
        const int qx0 = tx * 16 + (q & 1) * 8, qy0 = ty * 16 + (q >> 1) * 8;
        // ---- every load of the task, back to back: entry `lane` of the sub-range (wave 0), pixel `lane` of the quadrant, the first 64 records
        float2 g0 = make_float2(0.f, 0.f), g1 = g0, g2 = make_float2(0.f, -INFINITY);
        float4 ecl = make_float4(0.f, 0.f, 0.f, 0.f);
        uint32_t my_slot = 0;
        if (q == 0 && (uint32_t)lane < scnt) {
            const size_t ei = (size_t)start + (uint32_t)sub * sub_sz + (uint32_t)lane;
            const float2 *g = ent_geo + 3 * ei;
            g0 = g[0]; g1 = g[1]; g2 = g[2];
            ecl = *reinterpret_cast<const float4 *>(ent_col + 4 * ei);
            my_slot = ent_slot[ei];
        }
        float2 r_ti = make_float2(0.f, 0.f);
        float4 r_acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if ((uint32_t)lane < n_rec) {
            r_ti = rec.rec_ti[pr.x + (uint32_t)lane];
            r_acc = rec.rec_acc[pr.x + (uint32_t)lane];
        }
        if (live) {
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
            float bg[4] = {bg0, bg1, bg2, bg3};
            if (cams) {
#pragma unroll
                for (int ch = 0; ch < 4; ch++) bg[ch] = cams[fr].bg[ch];
            }
            float Sd = 0.f, bg_dot = 0.f;
#pragma unroll
            for (int ch = 0; ch < C; ch++) {
                Sd += (Sb[ch] + Sc[ch]) * dpix[ch];
                bg_dot += bg[ch] * dpix[ch];
            }
            s_pix[q][lane][0] = make_float4(dpix[0], dpix[1], dpix[2], dpix[3]);
            s_pix[q][lane][1] = make_float4(Sd, T_final * bg_dot, __uint_as_float(my_last), 0.f);
        }
        tq.request();  // (behind every load of this task)
        if (q == 0) {
            s_ent[lane][0] = make_float4(g0.x, g0.y, g1.x, g1.y);
            s_ent[lane][1] = make_float4(g2.x, g2.y, ecl.x, ecl.y);
            s_ent[lane][2] = make_float4(ecl.z, ecl.w, 0.f, 0.f);
        }
        __syncthreads();   // the entries (wave 0) are staged; wave 0 has cleared the rows it folded in the previous task
#if defined(GOM_KO_REC) && GOM_KO_REC == 2   // development knock-outs (scripts/exp_build.py NAME -DGOM_KO_REC=1|2): no sums / no record loop
        for (uint32_t r0 = 0; r0 < 0u; r0 += 64) {
#else
        for (uint32_t r0 = 0; r0 < n_rec; r0 += 64) {
#endif
            const float2 c_ti = r_ti;
            const float4 c_acc = r_acc;
            const bool have = r0 + (uint32_t)lane < n_rec;
            if (r0 + 64 + (uint32_t)lane < n_rec) {   // the next 64 records, in flight during this trip
                r_ti = rec.rec_ti[pr.x + r0 + 64 + (uint32_t)lane];
                r_acc = rec.rec_acc[pr.x + r0 + 64 + (uint32_t)lane];
            }
            const uint32_t id = __float_as_uint(c_ti.y);
            const uint32_t e = have ? (id >> 6) & 63u : 0u, p = have ? id & 63u : 0u;
            const float4 E0 = s_ent[e][0], E1 = s_ent[e][1], E2 = s_ent[e][2];
            const float4 P0 = s_pix[q][p][0], P1 = s_pix[q][p][1];
            const float pfx = (float)(qx0 + (int)(p & 7u)), pfy = (float)(qy0 + (int)(p >> 3));
            const AlphaEval<float> ev = alpha_eval<float>(E0.x, E0.y, E0.z, E0.w, E1.x, E1.y, pfx, pfy);
            // entries at or beyond the pixel's last contributor (a piece the fold did not count for it) do not exist for the gradient
            const bool valid = have && (s0 + e) < __float_as_uint(P1.z) && ev.al > 0.f;
            if (valid) {
                const float T = c_ti.x;
                const float al = ev.al;
                const float inv = __builtin_amdgcn_rcpf(1.f - al);
                const float w = al * T;
                float u = E1.z * P0.x, pd = c_acc.x * P0.x;
                if (C > 1) { u = __fmaf_rn(E1.w, P0.y, u); pd = __fmaf_rn(c_acc.y, P0.y, pd); }
                if (C > 2) { u = __fmaf_rn(E2.x, P0.z, u); pd = __fmaf_rn(c_acc.z, P0.z, pd); }
                if (C > 3) { u = __fmaf_rn(E2.y, P0.w, u); pd = __fmaf_rn(c_acc.w, P0.w, pd); }
                const float sufd = (P1.x - pd) - w * u;                      // colour behind the entry . dL/dpix
                const float dLa = __fmaf_rn(T, u, -((sufd + P1.y) * inv));   // dL/dalpha
                const float Q = (ev.og * ev.mm) * dLa;                       // opacity G dL/dalpha (the reference does not mask the 0.99 clamp here)
                const float z5 = Q * ev.dx, z6 = Q * ev.dy;
                float *a = &s_acc[q][0][e];
#if defined(GOM_KO_REC) && GOM_KO_REC == 1
#define GOM_RB_ADD(V, X) do { if ((X) == 123.456f) a[(V) * GOM_SUB_MAX] = (X); } while (0)
#else
#define GOM_RB_ADD(V, X) (void)__hip_atomic_fetch_add(a + (V) * GOM_SUB_MAX, (X), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT)
#endif
                GOM_RB_ADD(0, w * P0.x);
                if (C > 1) GOM_RB_ADD(1, w * P0.y);
                if (C > 2) GOM_RB_ADD(2, w * P0.z);
                if (C > 3) GOM_RB_ADD(3, w * P0.w);
                GOM_RB_ADD(4, Q);
                GOM_RB_ADD(5, z5);
                GOM_RB_ADD(6, z6);
                GOM_RB_ADD(7, z5 * ev.dx);
                GOM_RB_ADD(8, z5 * ev.dy);
                GOM_RB_ADD(9, z6 * ev.dy);
#undef GOM_RB_ADD
            }
        }
        tq.publish(s_task);
        __syncthreads();   // the four quadrants' sums are complete
        if (q == 0 && (uint32_t)lane < scnt) {   // one 48-byte record per entry, quadrants summed in a fixed order; the rows are left zero for the next task
            float rr[10];
#pragma unroll
            for (int v = 0; v < 10; v++) {
                rr[v] = ((s_acc[0][v][lane] + s_acc[1][v][lane]) + s_acc[2][v][lane]) + s_acc[3][v][lane];
                s_acc[0][v][lane] = 0.f; s_acc[1][v][lane] = 0.f; s_acc[2][v][lane] = 0.f; s_acc[3][v][lane] = 0.f;
            }
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
    tq.finish();
}
