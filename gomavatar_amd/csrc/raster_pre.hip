// Per-Gaussian kernels of the splat rasterizer: projection + EWA footprint,
// tile counting, tile-offset scan, pair emission, and the per-Gaussian backward.
//
// Replaces preprocessCUDA / InclusiveSum / duplicateWithKeys /
// computeCov2DCUDA+preprocessCUDA(backward) of the CUDA extension the
// reference calls at models/modules/renderer/gaussian.py:83-91
// (algorithm: SURVEY.md App. A.1, A.2, A.5).
//
// THIS FILE IS COMPILED WITH -ffp-contract=off: the floating-point operation
// order below is the bit-exact binning contract shared with the CPU oracle
// (radii, tile rects, depth key bits must be identical on both sides).
//
// MI355X design notes
//  * binning is a per-tile counting sort, not a global 64-bit radix sort:
//    count (LDS histogram per 256-Gaussian block, one global atomic per touched
//    tile) -> single-block scan over tiles -> emit into exact per-tile ranges
//    (LDS-aggregated slot reservation) -> per-tile LDS sort in the render kernel.
//    4 launches instead of CUB's ~15, no device->host read of the pair count.
//  * the backward gathers per-(tile,gaussian) partial sums written by the
//    render backward through a position table (pair_pos) the sort fills in
//    (no float atomics anywhere -> bitwise reproducible, no searching).
#include "gom_internal.h"
#include "geom_face.hpp"
#include "entry_record.hpp"
#include "rank_map.hpp"
#include "rank_sort.hpp"

namespace {

__device__ __forceinline__ int f2i_sat(float x) {
    if (!(x >= -1073741824.0f)) return -1073741824;
    if (x >= 1073741824.0f) return 1073741824;
    return (int)x;
}

struct ProjJac {
    float t[3];
    float M0[3], M1[3];
    float xmul, ymul;
};

__device__ __forceinline__ void xform4x3(const float *m, const float *p, float *o) {
    o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
    o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
    o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
}
__device__ __forceinline__ void xform4x4(const float *m, const float *p, float *o) {
    o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
    o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
    o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
    o[3] = m[3] * p[0] + m[7] * p[1] + m[11] * p[2] + m[15];
}

__device__ __forceinline__ void proj_jacobian(const GomCamera &cam, const float *mean, float fx, float fy, ProjJac &o) {
    const float *v = cam.view;
    float t[3];
    xform4x3(v, mean, t);
    const float limx = 1.3f * cam.tanfovx;
    const float limy = 1.3f * cam.tanfovy;
    const float txtz = t[0] / t[2];
    const float tytz = t[1] / t[2];
    o.xmul = (txtz < -limx || txtz > limx) ? 0.0f : 1.0f;
    o.ymul = (tytz < -limy || tytz > limy) ? 0.0f : 1.0f;
    {
        const float cx = -limx > txtz ? -limx : txtz;
        const float cy = -limy > tytz ? -limy : tytz;
        t[0] = (limx < cx ? limx : cx) * t[2];
        t[1] = (limy < cy ? limy : cy) * t[2];
    }
    const float J00 = fx / t[2];
    const float J02 = -(fx * t[0]) / (t[2] * t[2]);
    const float J11 = fy / t[2];
    const float J12 = -(fy * t[1]) / (t[2] * t[2]);
#pragma unroll
    for (int k = 0; k < 3; k++) {
        o.M0[k] = J00 * v[4 * k + 0] + J02 * v[4 * k + 2];
        o.M1[k] = J11 * v[4 * k + 1] + J12 * v[4 * k + 2];
    }
    o.t[0] = t[0];
    o.t[1] = t[1];
    o.t[2] = t[2];
}

__device__ __forceinline__ void cov2d_from(const float *c6, const ProjJac &pj, float &a, float &b, float &c, float *SM0, float *SM1) {
    const float S[3][3] = {{c6[0], c6[1], c6[2]}, {c6[1], c6[3], c6[4]}, {c6[2], c6[4], c6[5]}};
#pragma unroll
    for (int k = 0; k < 3; k++) {
        SM0[k] = S[k][0] * pj.M0[0] + S[k][1] * pj.M0[1] + S[k][2] * pj.M0[2];
        SM1[k] = S[k][0] * pj.M1[0] + S[k][1] * pj.M1[1] + S[k][2] * pj.M1[2];
    }
    a = (pj.M0[0] * SM0[0] + pj.M0[1] * SM0[1] + pj.M0[2] * SM0[2]) + 0.3f;
    b = pj.M0[0] * SM1[0] + pj.M0[1] * SM1[1] + pj.M0[2] * SM1[2];
    c = (pj.M1[0] * SM1[0] + pj.M1[1] * SM1[1] + pj.M1[2] * SM1[2]) + 0.3f;
}

// ---------------------------------------------------------------- A.1 ------
// One thread per Gaussian.  LDS_HIST: per-block tile histogram in LDS, flushed
// with one global atomic per touched tile; otherwise direct global atomics
// (tile grids too large for LDS).
// Batched launches (B frames, see gom_batch_forward_backward): blockIdx.y = frame.  The frames are laid out as ONE
// tall problem -- Gaussian i of frame b is global Gaussian b*P + i, tile (x, y) of frame b is tile (x, b*gy + y)
// of a gx x (B*gy) grid -- so that scan, emit, sort and the segment kernels see a single, larger frame.  Only
// the camera, the tile-row offset and (in the pixel kernels) the image addressing know about b.
__device__ __forceinline__ GomCamera pick_camera(const GomCamera &cam, const GomCamera *__restrict__ cams, int b) {
    GomCamera c = cam;
    if (cams) c = cams[b];
    return c;
}

// FACE: the Gaussian is made here too, from its posed triangle (geom_face.hpp; the frame step, gom_api.hip): `means` / `cov6` are
// then OUTPUTS of this kernel (the backward and the exports read them) -- one launch, and one trip of nine floats per Gaussian
// through HBM, less than the face kernel followed by this one.
template <bool LDS_HIST, bool FACE>
__global__ void __launch_bounds__(256) k_preprocess(GomCamera cam1, const GomCamera *__restrict__ cams, int P, float *__restrict__ means,
                                                    float *__restrict__ cov6, const float *__restrict__ opacity, GomFaceArgs face,
                                                    float *__restrict__ depth, float2 *__restrict__ xy,
                                                    float4 *__restrict__ conic_opacity, uint32_t *__restrict__ tiles_touched,
                                                    ushort4 *__restrict__ rect, int32_t *__restrict__ radii,
                                                    int32_t *__restrict__ radii_user, uint32_t *__restrict__ tile_count,
                                                    uint32_t *__restrict__ pair_off, GomDevStatus *__restrict__ status,
                                                    int gx, int gy, uint32_t cap_pairs, uint32_t *__restrict__ depth_minmax,
                                                    float4 *__restrict__ rec_g, uint32_t *__restrict__ big_list, uint32_t *__restrict__ big_count, int big_frames,
                                                    uint32_t *__restrict__ bucket_count, uint32_t nb) {
    extern __shared__ uint32_t s_hist[];
    __shared__ uint32_t s_red[8];
    __shared__ uint32_t s_wsum[4];
    __shared__ uint32_t s_dmin[4], s_dmax[4];
    __shared__ uint32_t s_blockbase;
    const int n_tiles = gx * gy;  // per frame
    const int fr = blockIdx.y;
    const GomCamera cam = pick_camera(cam1, cams, fr);
    {  // this frame's slice of every per-Gaussian / per-tile array
        const size_t go = (size_t)fr * P;
        means += 3 * go; cov6 += 6 * go; opacity += go; depth += go; xy += go; conic_opacity += go; tiles_touched += go;
        rect += go; radii += go; pair_off += go;
        if (radii_user) radii_user += go;
        if (rec_g) rec_g += 2 * go;
        tile_count += (size_t)fr * n_tiles;
    }
    const int ty_off = fr * gy;  // tile rows of this frame in the stacked grid
    uint32_t my_tiles = 0, my_rlo = 0, my_rhi = 0;
    float my_depth = 0.f;
    float4 my_r0 = make_float4(0.f, 0.f, 0.f, 0.f);
    float2 my_r1 = make_float2(0.f, 0.f);
    if (LDS_HIST) {
        for (int i = threadIdx.x; i < n_tiles; i += 256) s_hist[i] = 0;
        __syncthreads();
    }
    // Frame step with the depth ranking: the bucket histogram is built here (the depth range comes from the posed vertices, which the
    // skinning kernel already had in registers) -- k_depth_hist was a launch and a pass over the Gaussians for nothing else.
    const bool hist_here = FACE && face.vdepth_minmax && bucket_count;
    uint32_t *s_bcnt = s_hist + (LDS_HIST ? n_tiles : 0);
    gom_rank::BucketMap bm{};
    if (hist_here) {
        for (uint32_t b = threadIdx.x; b < nb; b += 256) s_bcnt[b] = 0;
        bm = gom_rank::bucket_map(face.vdepth_minmax, fr, face.vdepth_blocks, nb, s_red);   // (ends with a barrier)
    }
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < P) {
        float o_depth = 0.f, o_x = 0.f, o_y = 0.f, o_cx = 0.f, o_cy = 0.f, o_cz = 0.f, o_op = 0.f;
        int o_rad = 0, x0 = 0, y0 = 0, x1 = 0, y1 = 0;
        uint32_t o_tiles = 0;
        const float fx = (float)cam.W / (2.0f * cam.tanfovx);
        const float fy = (float)cam.H / (2.0f * cam.tanfovy);
        float p[3], c6[6];
        if (FACE) {
            gom_face::FaceFwd o;
            float s3[3];
            gom_face::face_forward(face.verts + (size_t)fr * 3 * face.N, face.N, face.faces, face.so3, face.scale, P, i, face.sigma, o, p, s3);
            gom_face::face_cov6(o, c6);
            means[3 * i] = p[0]; means[3 * i + 1] = p[1]; means[3 * i + 2] = p[2];
#pragma unroll
            for (int k = 0; k < 6; k++) cov6[6 * i + k] = c6[k];
            if (face.feat4)   // (3,F) colour parameter -> (F,4) rasterizer features [r g b 1] (gaussian.py:49)
                *reinterpret_cast<float4 *>(face.feat4 + 4 * ((size_t)fr * P + i)) =
                    make_float4(face.appearance[i], face.appearance[(size_t)P + i], face.appearance[2 * (size_t)P + i], 1.0f);
        } else {
            p[0] = means[3 * i]; p[1] = means[3 * i + 1]; p[2] = means[3 * i + 2];
        }
        float pv[3];
        xform4x3(cam.view, p, pv);
        if (pv[2] > 0.2f) {
            float ph[4];
            xform4x4(cam.proj, p, ph);
            const float pw = 1.0f / (ph[3] + 0.0000001f);
            const float ndcx = ph[0] * pw, ndcy = ph[1] * pw;
            ProjJac pj;
            proj_jacobian(cam, p, fx, fy, pj);
            if (!FACE) {
#pragma unroll
                for (int k = 0; k < 6; k++) c6[k] = cov6[6 * i + k];
            }
            float a, b, c, SM0[3], SM1[3];
            cov2d_from(c6, pj, a, b, c, SM0, SM1);
            const float det = a * c - b * b;
            if (det != 0.0f) {
                const float det_inv = 1.0f / det;
                const float mid = 0.5f * (a + c);
                const float dd = mid * mid - det;
                const float disc = sqrtf(0.1f > dd ? 0.1f : dd);
                const float lam1 = mid + disc, lam2 = mid - disc;
                const float rad_f = ceilf(3.0f * sqrtf(lam1 > lam2 ? lam1 : lam2));
                const int rad = f2i_sat(rad_f);
                const float px = ((ndcx + 1.0f) * (float)cam.W - 1.0f) * 0.5f;
                const float py = ((ndcy + 1.0f) * (float)cam.H - 1.0f) * 0.5f;
                const int rx0 = min(gx, max(0, f2i_sat((px - (float)rad) / 16.0f)));
                const int ry0 = min(gy, max(0, f2i_sat((py - (float)rad) / 16.0f)));
                const int rx1 = min(gx, max(0, f2i_sat((px + (float)rad + 15.0f) / 16.0f)));
                const int ry1 = min(gy, max(0, f2i_sat((py + (float)rad + 15.0f) / 16.0f)));
                if ((rx1 - rx0) * (ry1 - ry0) != 0) {
                    o_depth = pv[2];
                    o_rad = rad;
                    o_x = px;
                    o_y = py;
                    o_cx = c * det_inv;
                    o_cy = -b * det_inv;
                    o_cz = a * det_inv;
                    o_op = opacity[i];
                    o_tiles = (uint32_t)((rx1 - rx0) * (ry1 - ry0));
                    x0 = rx0; y0 = ry0; x1 = rx1; y1 = ry1;
                }
            }
        }
        depth[i] = o_depth;
        radii[i] = o_rad;
        if (radii_user) radii_user[i] = o_rad;
        xy[i] = make_float2(o_x, o_y);
        conic_opacity[i] = make_float4(o_cx, o_cy, o_cz, o_op);
        tiles_touched[i] = o_tiles;
        if (hist_here && o_rad > 0) atomicAdd(&s_bcnt[bm(o_depth)], 1u);
        if (big_list && o_tiles > GOM_BIG_NT) {   // (a few hundred per frame at most: one counter per frame is enough)
            const uint32_t bi = atomicAdd(&big_count[big_frames + fr], 1u);   // ([0, big_frames): last forward's counts, published by the scan kernel)
            if (bi < GOM_BIG_CAP) big_list[(size_t)fr * GOM_BIG_CAP + bi] = (uint32_t)i;
        }
        my_tiles = o_tiles;
        my_depth = o_depth;
        rect[i] = make_ushort4((unsigned short)x0, (unsigned short)(y0 + ty_off), (unsigned short)x1, (unsigned short)(y1 + ty_off));
        my_r0 = make_float4(o_x, o_y, o_cx, o_cy);
        my_r1 = make_float2(o_cz, o_op);
        my_rlo = (uint32_t)x0 | ((uint32_t)(y0 + ty_off) << 16);
        my_rhi = (uint32_t)x1 | ((uint32_t)(y1 + ty_off) << 16);
        for (int y = y0; y < y1; y++)
            for (int x = x0; x < x1; x++) {
                if (LDS_HIST) atomicAdd(&s_hist[y * gx + x], 1u);
                else atomicAdd(&tile_count[y * gx + x], 1u);
            }
    }
    // Depth range of the block's visible Gaussians (bit patterns: depths are > 0.2, unsigned order = float order) for the
    // bucket map of the depth ranking (raster_rank.hip): wave reduction, one (min, max) pair per block.
    if (depth_minmax) {
        const bool vis = my_tiles != 0u;
        uint32_t lo = vis ? __float_as_uint(my_depth) : 0xffffffffu, hi = vis ? lo : 0u;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            lo = min(lo, (uint32_t)__shfl_xor((int)lo, d, 64));
            hi = max(hi, (uint32_t)__shfl_xor((int)hi, d, 64));
        }
        if ((threadIdx.x & 63) == 0) { s_dmin[threadIdx.x >> 6] = lo; s_dmax[threadIdx.x >> 6] = hi; }
    }
    // Private range of this gaussian in pair_pos: block-level exclusive scan + ONE atomic per block
    // (the order of the ranges is irrelevant, they only index a scratch table).
    {
        const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
        uint32_t x = my_tiles;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t y = __shfl_up(x, d, 64);
            if (lane >= d) x += y;
        }
        if (lane == 63) s_wsum[wid] = x;
        __syncthreads();
        if (threadIdx.x == 0) {
            const uint32_t tot = s_wsum[0] + s_wsum[1] + s_wsum[2] + s_wsum[3];
            const uint32_t shard = (blockIdx.x + blockIdx.y) & 7u, shard_cap = cap_pairs >> 3;
            uint32_t base = 0;
            if (tot) {
                base = atomicAdd(&status->shard_cursor[shard][0], tot);
                if (base + tot > shard_cap) { atomicOr(&status->shard_overflow, 1u); base = 0; }   // poisoned frame: stay in bounds
            }
            s_blockbase = shard * shard_cap + base;
            if (depth_minmax) {   // per-block partial (no atomics: 1 728 blocks on 16 words cost the kernel 30 us); the rank kernels fold them
                uint2 *mm = reinterpret_cast<uint2 *>(depth_minmax) + (size_t)fr * gridDim.x + blockIdx.x;
                *mm = make_uint2(min(min(s_dmin[0], s_dmin[1]), min(s_dmin[2], s_dmin[3])), max(max(s_dmax[0], s_dmax[1]), max(s_dmax[2], s_dmax[3])));
            }
        }
        __syncthreads();
        uint32_t woff = 0;
        for (int w = 0; w < wid; w++) woff += s_wsum[w];
        if (i < P) {
            const uint32_t po = s_blockbase + woff + (x - my_tiles);
            pair_off[i] = po;
            if (rec_g) {   // everything the tile pass of the depth ranking needs of this Gaussian, in one 32-byte record (one sector per gather):
                // (x, y, A, B) (Cq, lo, first slot, rect: x0 | width << 10 | y0 << 20 in tiles, y0 in the stacked grid)
                float4 *d = rec_g + 2 * (size_t)i;
                const uint32_t rx0 = my_rlo & 0xffffu, ry0 = my_rlo >> 16, rw = (my_rhi & 0xffffu) - rx0;
                const float4 er = gom_entry::entry_record(my_r0.z, my_r0.w, my_r1.x, my_r1.y);   // the list record's (A, B, Cq, lo): entry_record.hpp
                d[0] = make_float4(my_r0.x, my_r0.y, er.x, er.y);
                d[1] = make_float4(er.z, er.w, __uint_as_float(po), __uint_as_float(rx0 | (rw << 10) | (ry0 << 20)));
            }
        }
    }
    if (LDS_HIST) {
        for (int t = threadIdx.x; t < n_tiles; t += 256) {
            const uint32_t c = s_hist[t];
            if (c) atomicAdd(&tile_count[t], c);
        }
    }
    if (hist_here) {   // (the barriers of the pair-range scan above lie between the LDS atomics and these reads)
        for (uint32_t b = threadIdx.x; b < nb; b += 256) {
            const uint32_t c = s_bcnt[b];
            if (c) atomicAdd(&bucket_count[(size_t)fr * nb + b], c);
        }
    }
}

// ---------------------------------------------------------------- A.2a -----
// Exclusive scans of the per-tile counts and of the per-tile segment counts
// (single 1024-thread block); resets the counters for the next frame and
// publishes D, the segment total and the overflow flag.
__device__ __forceinline__ uint32_t block_excl_scan_1024(uint32_t v, uint32_t *s_wave, uint32_t &total) {
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    uint32_t x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t y = __shfl_up(x, d, 64);
        if (lane >= d) x += y;
    }
    __syncthreads();
    if (lane == 63) s_wave[wid] = x;
    __syncthreads();
    uint32_t wave_off = 0, tot = 0;
    for (int w = 0; w < 16; w++) {
        const uint32_t sw = s_wave[w];
        if (w < wid) wave_off += sw;
        tot += sw;
    }
    total = tot;
    return wave_off + (x - v);
}

// One scatter workgroup of the depth ranking (GomScatterRider): 1 024 Gaussians of frame `fr`.  s_mem: 3 * nb words.
__device__ __forceinline__ void scatter_keys_block(const GomScatterRider &sr, int fr, int chunk, const uint32_t *__restrict__ bucket_count,
                                                   uint32_t *__restrict__ bucket_cursor, uint32_t *s_mem, uint32_t *s_red, uint32_t *s_wave) {
    const uint32_t nb = sr.nb, tid = threadIdx.x;
    uint32_t *s_cnt = s_mem, *s_base = s_mem + nb, *s_gbase = s_mem + 2 * nb;
    for (uint32_t b = tid; b < nb; b += 1024) s_cnt[b] = 0;
    const gom_rank::BucketMap bm = gom_rank::bucket_map(sr.minmax, fr, sr.nblk, nb, s_red);   // (ends with a barrier)
    const int il = chunk * 1024 + (int)tid;
    const size_t i = (size_t)fr * sr.P + il;
    const bool vis = il < sr.P && sr.radii[i] > 0;
    uint32_t b = 0, dbits = 0;
    if (vis) {
        const float d = sr.depth[i];
        dbits = __float_as_uint(d);
        b = bm(d);
        atomicAdd(&s_cnt[b], 1u);
    }
    // packed rank at which each bucket of this frame starts: the visible Gaussians of the frames in front + a scan of this frame's counts
    uint32_t part = 0;
    for (uint32_t k = tid; k < (uint32_t)fr * nb; k += 1024) part += bucket_count[k];
    uint32_t before, tot;
    (void)block_excl_scan_1024(part, s_wave, before);
    const uint32_t per = (nb + 1023u) / 1024u;
    uint32_t mine = 0;
    for (uint32_t k = 0; k < per; k++) {
        const uint32_t idx = tid * per + k;
        if (idx < nb) mine += bucket_count[(size_t)fr * nb + idx];
    }
    uint32_t run = before + block_excl_scan_1024(mine, s_wave, tot);
    for (uint32_t k = 0; k < per; k++) {
        const uint32_t idx = tid * per + k;
        if (idx < nb) { s_gbase[idx] = run; run += bucket_count[(size_t)fr * nb + idx]; }
    }
    __syncthreads();
    for (uint32_t k = tid; k < nb; k += 1024) {
        const uint32_t c = s_cnt[k];
        if (c) {
            s_base[k] = s_gbase[k] + atomicAdd(&bucket_cursor[(size_t)fr * nb + k], c);   // (cursor: keys already in the bucket; zeroed by the emit launch)
            s_cnt[k] = 0;
        }
    }
    __syncthreads();
    if (vis) sr.bkeys[s_base[b] + atomicAdd(&s_cnt[b], 1u)] = ((uint64_t)dbits << 32) | (uint32_t)il;   // index INSIDE the frame: ties keep Gaussian order
}

__global__ void __launch_bounds__(1024) k_scan_tiles(uint32_t *__restrict__ tile_count, uint32_t *__restrict__ tile_base,
                                                     uint32_t *__restrict__ tile_cursor, uint32_t *__restrict__ seg_base,
                                                     uint32_t *__restrict__ tile_nmax, int n_tiles,
                                                     GomDevStatus *__restrict__ status, uint32_t cap_pairs, uint32_t seg_shift,
                                                     uint32_t *__restrict__ bucket_count, uint32_t *__restrict__ bucket_base,
                                                     uint32_t *__restrict__ bucket_cursor, int n_buckets,
                                                     uint32_t *__restrict__ work_items, uint32_t cap_items, uint32_t *__restrict__ big_count, int n_frames, int big_frames,
                                                     GomScatterRider sr) {
    __shared__ uint32_t s_wave[16];
    const int tid = threadIdx.x;
    if (blockIdx.x >= 2) {   // the bucket scatter of the depth ranking: independent of the two scans, in their shadow
        extern __shared__ uint32_t s_scatter[];
        __shared__ uint32_t s_red[32];
        const int r = (int)blockIdx.x - 2;
        scatter_keys_block(sr, r % sr.B, r / sr.B, bucket_count, bucket_cursor, s_scatter, s_red, s_wave);
        return;
    }
    // two workgroups: block 0 scans the tiles (and lists the work items), block 1 -- launched with the depth ranking -- the buckets: the two
    // chains (load, block scans, stores) are independent and this kernel is nothing but their latency
    const bool do_tiles = blockIdx.x == 0, do_buckets = bucket_count != nullptr && blockIdx.x == 1;
    uint32_t carry = 0, seg_carry = 0, wi_carry = 0;
    // 8 consecutive tiles per thread and trip: a batched launch (8 192 tiles at 8 x 512x512) is ONE trip = one load latency and
    // two block scans, where one tile per thread took eight dependent trips (20 us of a single workgroup's latency chain).
    constexpr int kPer = 8;
    for (int base = 0; do_tiles && base < n_tiles; base += 1024 * kPer) {
        const int i0 = base + tid * kPer;
        const bool full = i0 + kPer <= n_tiles;   // (arrays come from hipMalloc and i0 is a multiple of 8: 16-byte accesses are aligned)
        uint32_t v[kPer];
        if (full) {
            const uint4 a = *reinterpret_cast<const uint4 *>(tile_count + i0), b = *reinterpret_cast<const uint4 *>(tile_count + i0 + 4);
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        } else {
#pragma unroll
            for (int k = 0; k < kPer; k++) v[k] = i0 + k < n_tiles ? tile_count[i0 + k] : 0u;
        }
        uint32_t tsum = 0, ssum = 0, sv[kPer];
#pragma unroll
        for (int k = 0; k < kPer; k++) {
            sv[k] = (v[k] + (1u << seg_shift) - 1) >> seg_shift;
            tsum += v[k];
            ssum += sv[k];
        }
        uint32_t tot, stot;
        uint32_t excl = carry + block_excl_scan_1024(tsum, s_wave, tot);
        uint32_t sexcl = seg_carry + block_excl_scan_1024(ssum, s_wave, stot);
        uint32_t eb[kPer], sb2[kPer];
#pragma unroll
        for (int k = 0; k < kPer; k++) {
            eb[k] = excl; sb2[k] = sexcl;
            excl += v[k];
            sexcl += sv[k];
        }
        if (full) {
            const uint4 z = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const uint4 e4 = make_uint4(eb[4 * h], eb[4 * h + 1], eb[4 * h + 2], eb[4 * h + 3]);
                *reinterpret_cast<uint4 *>(tile_count + i0 + 4 * h) = z;
                *reinterpret_cast<uint4 *>(tile_nmax + i0 + 4 * h) = z;
                *reinterpret_cast<uint4 *>(tile_base + i0 + 4 * h) = e4;
                *reinterpret_cast<uint4 *>(tile_cursor + i0 + 4 * h) = e4;
                *reinterpret_cast<uint4 *>(seg_base + i0 + 4 * h) = make_uint4(sb2[4 * h], sb2[4 * h + 1], sb2[4 * h + 2], sb2[4 * h + 3]);
            }
        } else {
#pragma unroll
            for (int k = 0; k < kPer; k++)
                if (i0 + k < n_tiles) {
                    tile_count[i0 + k] = 0;
                    tile_nmax[i0 + k] = 0;
                    tile_base[i0 + k] = eb[k];
                    tile_cursor[i0 + k] = eb[k];
                    seg_base[i0 + k] = sb2[k];
                }
        }
        carry += tot;
        seg_carry += stot;
        if (work_items) {   // work items of k_tile_rank: (tile | window << 24) for every window of GOM_RANK_WIN list positions of a non-empty tile
            uint32_t ci = 0;
#pragma unroll
            for (int k = 0; k < kPer; k++) ci += (v[k] + GOM_RANK_WIN - 1u) / GOM_RANK_WIN;
            uint32_t ti;
            uint32_t pi = wi_carry + block_excl_scan_1024(ci, s_wave, ti);
#pragma unroll
            for (int k = 0; k < kPer; k++)
                for (uint32_t w = 0; w * GOM_RANK_WIN < v[k]; w++, pi++)
                    if (pi < cap_items) work_items[pi] = (uint32_t)(i0 + k) | (w << 24);   // (tile_count keeps counting after the pair buffer has overflowed: the list is then longer than its buffer, and unused)
            wi_carry += ti;
        }
    }
    // depth ranking (raster_rank.hip): exclusive scan of the (frame, bucket) counts = packed ranks at which the buckets start
    if (do_buckets) {
        uint32_t bcarry = 0;
        for (int base = 0; base < n_buckets; base += 1024 * kPer) {
            const int i0 = base + tid * kPer;
            uint32_t v[kPer], tsum = 0;
#pragma unroll
            for (int k = 0; k < kPer; k++) { v[k] = i0 + k < n_buckets ? bucket_count[i0 + k] : 0u; tsum += v[k]; }
            uint32_t tot;
            uint32_t excl = bcarry + block_excl_scan_1024(tsum, s_wave, tot);
#pragma unroll
            for (int k = 0; k < kPer; k++) {
                if (i0 + k < n_buckets) bucket_base[i0 + k] = excl;   // (the counts are read by the scatter workgroups beside this one: the emit launch zeroes them)
                excl += v[k];
            }
            bcarry += tot;
        }
        if (tid == 0) bucket_base[n_buckets] = bcarry;
    }
    if (do_tiles && big_count) {   // publish the per-frame counts of the many-tile Gaussians (more than GOM_BIG_CAP: that frame's list is incomplete and unused)
        for (int f = tid; f < n_frames; f += 1024) { big_count[f] = big_count[big_frames + f]; big_count[big_frames + f] = 0u; }
    }
    if (tid == 0 && do_tiles) {
        tile_base[n_tiles] = carry;
        seg_base[n_tiles] = seg_carry;
        const bool over = carry > cap_pairs || status->shard_overflow != 0u;
        status->num_pairs = carry;
        status->overflow = over ? 1u : 0u;
        status->num_segs = over ? 0u : seg_carry;
        status->pair_cursor = 0;
        status->shard_overflow = 0;
        status->n_work_items = (over || wi_carry > cap_items) ? 0u : wi_carry;
        for (int x = 0; x < 8; x++) status->shard_cursor[x][0] = 0;
    }
}

// ---------------------------------------------------------------- A.2b -----
// Emit (depth_bits << 32 | gaussian) into the tile's exact range.  The order
// inside a range is arbitrary here; the per-tile sort on the unique 64-bit key
// makes the final list identical to a stable sort on (tile, depth bits).
// RANK: the depth ranking has run (raster_rank.hip): the entry is the Gaussian's 32-bit packed rank, not a 64-bit key.
template <bool LDS_AGG, bool RANK>
__global__ void __launch_bounds__(256) k_emit(int P, const float *__restrict__ depth, const int32_t *__restrict__ radii,
                                              const ushort4 *__restrict__ rect, uint32_t *__restrict__ tile_cursor,
                                              uint64_t *__restrict__ keys, int gx, int gy,
                                              const GomDevStatus *__restrict__ status, uint32_t *__restrict__ keys32, GomEmptyFill fill,
                                              GomSortRider sorts) {
    extern __shared__ __attribute__((aligned(16))) uint32_t s_mem[];
    if ((int)(blockIdx.x / (uint32_t)sorts.B) < sorts.blocks && threadIdx.x == 0) {   // sort rider of (frame, bucket): its counters are spent (also when the frame overflowed)
        const size_t cb = (size_t)(blockIdx.x % (uint32_t)sorts.B) * sorts.blocks + blockIdx.x / (uint32_t)sorts.B;
        sorts.bucket_count[cb] = 0u;
        sorts.bucket_cursor[cb] = 0u;
    }
    if (status->overflow) return;
    const int n_tiles = gx * gy;  // per frame
    // frame-minor block order: the sort riders of EVERY frame are dispatched first (as the y index of a 2-D grid the last frame's would
    // start behind everything else and the launch would end with their chains)
    const int fr = (int)(blockIdx.x % (uint32_t)sorts.B), bi = (int)(blockIdx.x / (uint32_t)sorts.B);
    // The first blocks sort the buckets of the depth ranking (one each): a latency chain of ~30 active lanes that runs beside the
    // emission's atomics instead of in a launch of its own in front of it (18 us).
    const bool is_sort = bi < sorts.blocks;
    const int sort_b = bi;
    if (is_sort) {
        gom_rank::bucket_sort_block<256>(fr, sort_b, sorts.P, (uint32_t)sorts.blocks, sorts.bucket_base, sorts.bkeys, sorts.scratch, sorts.order,
                                         sorts.rank_of, sorts.log_chunk, reinterpret_cast<uint64_t *>(s_mem));
        return;
    }
    const int bx = bi - sorts.blocks;
    // Blocks beyond the Gaussians paint the tiles NOTHING touches (five in six on a body: background colour, T = 1, no contributor):
    // pure stores that run in the shadow of this kernel's atomics instead of holding 6 800 workgroup slots of the compositing
    // assembly (k_combine_fwd: 54 -> 36 us without them).
    if (bx >= fill.first_block) {
        const int t0 = (bx - fill.first_block) * GOM_FILL_TILES;
        const size_t HW = (size_t)fill.H * fill.W;
        float bg[4] = {fill.bg[0], fill.bg[1], fill.bg[2], fill.bg[3]};
        if (fill.cams) {
#pragma unroll
            for (int ch = 0; ch < 4; ch++) bg[ch] = fill.cams[fr].bg[ch];
        }
        for (int t = t0; t < min(t0 + GOM_FILL_TILES, n_tiles); t++) {
            const uint32_t *tb = fill.tile_base + (size_t)fr * n_tiles + t;
            if (tb[1] != tb[0]) continue;
            const int px = (t % gx) * 16 + (threadIdx.x & 15), py = (t / gx) * 16 + (threadIdx.x >> 4);
            if (px >= fill.W || py >= fill.H) continue;
            const size_t pix = (size_t)py * fill.W + px;
            for (int ch = 0; ch < fill.C; ch++) fill.out_color[((size_t)fr * fill.C + ch) * HW + pix] = bg[ch];
            fill.final_T[(size_t)fr * HW + pix] = 1.f;
            fill.n_contrib[(size_t)fr * HW + pix] = 0u;
        }
        return;
    }
    uint32_t *s_cnt = s_mem;
    uint32_t *s_base = s_mem + n_tiles;
    const int il = bx * 256 + threadIdx.x;
    const size_t i = (size_t)fr * P + il;  // global Gaussian id: the low key bits stay unique across the batch
    const bool vis = (il < P) && (radii[i] > 0);
    ushort4 r = make_ushort4(0, 0, 0, 0);
    uint64_t key = 0;
    if (vis) {
        r = rect[i];
        key = RANK ? (uint64_t)(uint32_t)i : ((uint64_t)__float_as_uint(depth[i]) << 32) | (uint32_t)i;   // (RANK: the tile pass looks the rank up)
    }
    tile_cursor += (size_t)fr * n_tiles;  // rect rows are stacked: bring them back to this frame's tile block
    r.y -= (unsigned short)(vis ? fr * gy : 0);
    r.w -= (unsigned short)(vis ? fr * gy : 0);
    if (LDS_AGG) {
        for (int t = threadIdx.x; t < n_tiles; t += 256) s_cnt[t] = 0;
        __syncthreads();
        for (int y = r.y; y < r.w; y++)
            for (int x = r.x; x < r.z; x++) atomicAdd(&s_cnt[y * gx + x], 1u);
        __syncthreads();
        for (int t0 = threadIdx.x; t0 < n_tiles; t0 += 4 * 256) {   // four returning atomics in flight per thread (a 512 x 512 frame: one trip), not four round trips in a row
            uint32_t c[4], base[4];
#pragma unroll
            for (int u = 0; u < 4; u++) c[u] = t0 + u * 256 < n_tiles ? s_cnt[t0 + u * 256] : 0u;
#pragma unroll
            for (int u = 0; u < 4; u++) base[u] = c[u] ? atomicAdd(&tile_cursor[t0 + u * 256], c[u]) : 0u;
#pragma unroll
            for (int u = 0; u < 4; u++)
                if (c[u]) { s_base[t0 + u * 256] = base[u]; s_cnt[t0 + u * 256] = 0; }
        }
        __syncthreads();
        for (int y = r.y; y < r.w; y++)
            for (int x = r.x; x < r.z; x++) {
                const int t = y * gx + x;
                const uint32_t slot = s_base[t] + atomicAdd(&s_cnt[t], 1u);
                if (RANK) keys32[slot] = (uint32_t)key; else keys[slot] = key;
            }
    } else {
        for (int y = r.y; y < r.w; y++)
            for (int x = r.x; x < r.z; x++) {
                const uint32_t slot = atomicAdd(&tile_cursor[y * gx + x], 1u);
                if (RANK) keys32[slot] = (uint32_t)key; else keys[slot] = key;
            }
    }
}

// ---------------------------------------------------------------- A.5 ------
// One thread per Gaussian: gather the per-(tile, gaussian) partial sums the
// render backward left in `partial` (position found by binary search on the
// unique sort key), then conic -> cov2D -> (cov3D, mean3D) and the projection
// term of the screen-space mean gradient.
#ifndef GOM_PB_W
#define GOM_PB_W 8   // live records in flight per trip of k_preprocess_bwd (4: 54 us, 8: 49 us on the metric workload)
#endif
#ifndef GOM_PB_MB
#define GOM_PB_MB 16  // liveness lookups in flight per trip (4, 8, 16 measure the same: nine Gaussians in ten need one trip)
#endif
// FACE: the gradients of mean and covariance go straight through the backward of the per-face frame (geom_face.hpp) in the same thread:
// corner / so3 / scale / appearance gradients out, nothing per Gaussian written in between.
template <int C, bool RANK, bool FACE>
__global__ void __launch_bounds__(256) k_preprocess_bwd(GomCamera cam1, const GomCamera *__restrict__ cams, GomFaceArgs face, int P, const float *__restrict__ means,
                                                        const float *__restrict__ cov6, const int32_t *__restrict__ radii,
                                                        const uint32_t *__restrict__ tiles_touched,
                                                        const float4 *__restrict__ conic_opacity,
                                                        const uint32_t *__restrict__ pair_off, const uint32_t *__restrict__ pair_pos,
                                                        const ushort4 *__restrict__ rect, const uint32_t *__restrict__ tile_base,
                                                        const uint32_t *__restrict__ tile_nmax, const uint32_t *__restrict__ rank_of,
                                                        const uint32_t *__restrict__ tile_qlim, int gx, const float *__restrict__ partial,
                                                        const GomDevStatus *__restrict__ status,
                                                        float *__restrict__ dL_dmeans, float *__restrict__ dL_dcov6,
                                                        float *__restrict__ dL_dcolors, float *__restrict__ dL_dopacity,
                                                        float *__restrict__ dL_dmeans2D, const uint32_t *__restrict__ big_list, const uint32_t *__restrict__ big_count) {
    // Blocks behind the per-Gaussian grid are RIDERS: each of their waves takes one Gaussian of its frame's big_list -- more than
    // GOM_BIG_NT tiles, i.e. close to the camera -- sums its records with all 64 lanes and lets lane 0 finish it; the lane that owns such a
    // Gaussian in the per-Gaussian grid leaves it alone.  (A lane walking 121 records in dependent trips was the kernel's tail.)
    const int main_blocks = (P + 255) / 256;
    const bool rider = (int)blockIdx.x >= main_blocks;
    const int fr = blockIdx.y;
    const uint32_t n_big = big_list ? big_count[fr] : GOM_BIG_CAP + 1u;
    const bool use_big = n_big <= GOM_BIG_CAP;   // per frame: a frame is treated the same alone and in a batch
    int i;
    if (rider) {
        const uint32_t w = ((uint32_t)blockIdx.x - (uint32_t)main_blocks) * 4u + (threadIdx.x >> 6);
        if (!use_big || w >= n_big) return;
        i = (int)big_list[(size_t)fr * GOM_BIG_CAP + w];
    } else {
        i = blockIdx.x * 256 + threadIdx.x;
        if (i >= P) return;
    }
    // the inputs of the face arithmetic at the end (an index load and the gathers behind it): issued here, in front of the record gather
    gom_face::FaceIn fin;
    if (FACE) gom_face::face_load(face.verts + (size_t)fr * 3 * face.N, face.N, face.faces, face.so3, face.scale, P, i, fin);
    const GomCamera cam = pick_camera(cam1, cams, fr);
    {
        const size_t go = (size_t)fr * P;
        means += 3 * go; cov6 += 6 * go; radii += go; tiles_touched += go; conic_opacity += go; pair_off += go; rect += go;
        if (!FACE) { dL_dmeans += 3 * go; dL_dcov6 += 6 * go; dL_dcolors += C * go; dL_dopacity += go; }
        if (dL_dmeans2D) dL_dmeans2D += 3 * go;
    }
    float gm[3] = {0.f, 0.f, 0.f}, gc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float acc[GOM_PARTIAL_STRIDE];
#pragma unroll
    for (int k = 0; k < GOM_PARTIAL_STRIDE; k++) acc[k] = 0.f;
    float g2x = 0.f, g2y = 0.f, gop = 0.f;
    const bool bad = status->overflow != 0;
    if (!bad && radii[i] > 0) {
        // The records of this Gaussian (one per tile of its rect, written by the render backward) are ONE contiguous run at
        // pair_off[i]: 48 nt bytes streamed, where the list-ordered layout of round 1 cost a 128-byte line per record (4.7x
        // the algorithmic traffic).  The k-th tile's entry is dead -- never written, not read -- when its list index is at or
        // beyond the tile's last contributor: about half of all pairs on a body (back-facing surface).  With the depth ranking that
        // is a comparison of the Gaussian's rank with the rank of the tile's last contributor (tile_qlim, from the forward's combine
        // pass); with the per-tile merge sort, of its list index (pair_pos) with tile_nmax.
        // Two passes per window of 64 tiles: a few Gaussians close to the camera own a hundred records (the metric workload: up to
        // 121, fifty beyond 64 in one frame) and every dependent trip of such a lane is a memory latency the kernel's tail waits for
        // (with liveness and record of four tiles per trip: 31 trips of two dependent loads, 62 us; without the gather 18 us):
        // (1) the liveness bits, GOM_PB_MB tiles per trip (independent loads; lanes past their last tile re-read their first one);
        // (2) the LIVE records only, GOM_PB_W per trip, in ascending k (fixed summation order).
        const uint32_t nt = tiles_touched[i];
        if (!rider && use_big && nt > GOM_BIG_NT) return;   // a rider wave has it
        // (what the arithmetic behind the gather reads: in flight while the records arrive)
        const float4 co = conic_opacity[i];
        const float p[3] = {means[3 * i], means[3 * i + 1], means[3 * i + 2]};
        float c6[6];
#pragma unroll
        for (int k = 0; k < 6; k++) c6[k] = cov6[6 * i + k];
        const uint32_t po = pair_off[i];
        const ushort4 rc = rect[i];
        const uint32_t rw = (uint32_t)(rc.z - rc.x);
        const uint32_t myq = RANK ? rank_of[(size_t)fr * P + i] : 0u;
        const uint32_t tile0 = (uint32_t)rc.y * (uint32_t)gx + (uint32_t)rc.x;   // (stacked tile rows: rect carries the frame offset)
        if (rider) {   // lane l sums the records l, l + 64, ... (liveness as below), then the wave adds the 64 partial sums
            const uint32_t lane = threadIdx.x & 63u;
            for (uint32_t k = lane; k < nt; k += 64) {
                const uint32_t tile = tile0 + (k / rw) * (uint32_t)gx + k % rw;
                const bool lv = RANK ? myq < tile_qlim[tile] : pair_pos[po + k] - tile_base[tile] < tile_nmax[tile];
                if (lv) {
                    const float4 *rec = reinterpret_cast<const float4 *>(partial + (size_t)(po + k) * GOM_PARTIAL_STRIDE);
                    const float4 q0 = rec[0], q1 = rec[1], q2 = rec[2];
                    acc[0] += q0.x; acc[1] += q0.y; acc[2] += q0.z; acc[3] += q0.w;
                    acc[4] += q1.x; acc[5] += q1.y; acc[6] += q1.z; acc[7] += q1.w;
                    acc[8] += q2.x; acc[9] += q2.y;
                }
            }
#pragma unroll
            for (int v = 0; v < 10; v++) {
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) acc[v] += __shfl_xor(acc[v], d, 64);
            }
            if (lane != 0) return;   // lane 0 finishes the Gaussian
        }
        uint32_t kx = 0, trow = tile0;   // column inside the rect and first tile of the row of tile k
        for (uint32_t w0 = 0; w0 < (rider ? 0u : nt); w0 += 64) {
            const uint32_t wn = min(64u, nt - w0);
            unsigned long long m = 0ull;
            for (uint32_t b0 = 0; b0 < wn; b0 += GOM_PB_MB) {
                uint32_t lim[GOM_PB_MB], pos[GOM_PB_MB];
#pragma unroll
                for (int u = 0; u < GOM_PB_MB; u++) {
                    const bool in = b0 + u < wn;
                    const uint32_t tile = in ? trow + kx : tile0;
                    if (RANK) { lim[u] = tile_qlim[tile]; pos[u] = myq; }
                    else { lim[u] = tile_nmax[tile]; pos[u] = pair_pos[po + (in ? w0 + b0 + u : 0u)] - tile_base[tile]; }
                    if (!in) lim[u] = 0u;
                    if (++kx == rw) { kx = 0; trow += (uint32_t)gx; }
                }
#pragma unroll
                for (int u = 0; u < GOM_PB_MB; u++)
                    if (pos[u] < lim[u]) m |= 1ull << (b0 + u);
            }
            const float *run = partial + (size_t)(po + w0) * GOM_PARTIAL_STRIDE;
            while (m) {
                bool live[GOM_PB_W];
                float4 q0[GOM_PB_W], q1[GOM_PB_W], q2[GOM_PB_W];
#pragma unroll
                for (int u = 0; u < GOM_PB_W; u++) {
                    live[u] = m != 0ull;
                    const uint32_t k = live[u] ? (uint32_t)__builtin_ctzll(m) : 0u;   // (spent lanes re-read the window's first slot, masked below)
                    m &= m - 1ull;
                    const float4 *rec = reinterpret_cast<const float4 *>(run + (size_t)k * GOM_PARTIAL_STRIDE);
                    q0[u] = rec[0]; q1[u] = rec[1]; q2[u] = rec[2];
                }
#pragma unroll
                for (int u = 0; u < GOM_PB_W; u++) {
                    if (live[u]) {
                        acc[0] += q0[u].x; acc[1] += q0[u].y; acc[2] += q0[u].z; acc[3] += q0[u].w;
                        acc[4] += q1[u].x; acc[5] += q1[u].y; acc[6] += q1[u].z; acc[7] += q1[u].w;
                        acc[8] += q2[u].x; acc[9] += q2[u].y;
                    }
                }
            }
        }
        // record layout: [0..3] colours, [4] sum Q, [5] sum Q dx, [6] sum Q dy, [7] sum Q dx dx, [8] sum Q dx dy, [9] sum Q dy dy
        // with Q = opacity * G * dL/dalpha and d = centre - pixel (App. A.4 regrouped; the render backward's exp2 delivers opacity * G, so
        // the opacity that multiplied the five geometry sums here now divides the opacity gradient instead).
        const float o = co.w;
        g2x = -(0.5f * (float)cam.W) * (co.x * acc[5] + co.y * acc[6]);
        g2y = -(0.5f * (float)cam.H) * (co.z * acc[6] + co.y * acc[5]);
        const float gxx = -0.5f * acc[7];
        const float gxy = -0.5f * acc[8];
        const float gyy = -0.5f * acc[9];
        gop = o > 0.f ? acc[4] / o : 0.f;

        const float fx = (float)cam.W / (2.0f * cam.tanfovx);
        const float fy = (float)cam.H / (2.0f * cam.tanfovy);
        const float *v = cam.view, *pr = cam.proj;
        ProjJac pj;
        proj_jacobian(cam, p, fx, fy, pj);
        float a, b, c, SM0[3], SM1[3];
        cov2d_from(c6, pj, a, b, c, SM0, SM1);
        const float denom = a * c - b * b;
        const float d2 = 1.0f / (denom * denom + 0.0000001f);
        float dL_da = 0.f, dL_db = 0.f, dL_dc = 0.f;
        if (d2 != 0.0f) {
            dL_da = d2 * (-c * c * gxx + 2.0f * b * c * gxy + (denom - a * c) * gyy);
            dL_dc = d2 * (-a * a * gyy + 2.0f * a * b * gxy + (denom - a * c) * gxx);
            dL_db = d2 * 2.0f * (b * c * gxx - (denom + 2.0f * b * b) * gxy + a * b * gyy);
            const float *M0 = pj.M0, *M1 = pj.M1;
            gc[0] = M0[0] * M0[0] * dL_da + M0[0] * M1[0] * dL_db + M1[0] * M1[0] * dL_dc;
            gc[3] = M0[1] * M0[1] * dL_da + M0[1] * M1[1] * dL_db + M1[1] * M1[1] * dL_dc;
            gc[5] = M0[2] * M0[2] * dL_da + M0[2] * M1[2] * dL_db + M1[2] * M1[2] * dL_dc;
            gc[1] = 2.0f * M0[0] * M0[1] * dL_da + (M0[0] * M1[1] + M0[1] * M1[0]) * dL_db + 2.0f * M1[0] * M1[1] * dL_dc;
            gc[2] = 2.0f * M0[0] * M0[2] * dL_da + (M0[0] * M1[2] + M0[2] * M1[0]) * dL_db + 2.0f * M1[0] * M1[2] * dL_dc;
            gc[4] = 2.0f * M0[2] * M0[1] * dL_da + (M0[1] * M1[2] + M0[2] * M1[1]) * dL_db + 2.0f * M1[1] * M1[2] * dL_dc;
        }
        float dM0[3], dM1[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            dM0[k] = 2.0f * SM0[k] * dL_da + SM1[k] * dL_db;
            dM1[k] = 2.0f * SM1[k] * dL_dc + SM0[k] * dL_db;
        }
        const float dJ00 = v[0] * dM0[0] + v[4] * dM0[1] + v[8] * dM0[2];
        const float dJ02 = v[2] * dM0[0] + v[6] * dM0[1] + v[10] * dM0[2];
        const float dJ11 = v[1] * dM1[0] + v[5] * dM1[1] + v[9] * dM1[2];
        const float dJ12 = v[2] * dM1[0] + v[6] * dM1[1] + v[10] * dM1[2];
        const float tz = 1.0f / pj.t[2], tz2 = tz * tz, tz3 = tz2 * tz;
        const float dtx = pj.xmul * -fx * tz2 * dJ02;
        const float dty = pj.ymul * -fy * tz2 * dJ12;
        const float dtz = -fx * tz2 * dJ00 - fy * tz2 * dJ11 + (2.0f * fx * pj.t[0]) * tz3 * dJ02 + (2.0f * fy * pj.t[1]) * tz3 * dJ12;
        gm[0] = v[0] * dtx + v[1] * dty + v[2] * dtz;
        gm[1] = v[4] * dtx + v[5] * dty + v[6] * dtz;
        gm[2] = v[8] * dtx + v[9] * dty + v[10] * dtz;
        float ph[4];
        xform4x4(pr, p, ph);
        const float mw = 1.0f / (ph[3] + 0.0000001f);
        const float mul1 = ph[0] * mw * mw, mul2 = ph[1] * mw * mw;
        gm[0] += (pr[0] * mw - pr[3] * mul1) * g2x + (pr[1] * mw - pr[3] * mul2) * g2y;
        gm[1] += (pr[4] * mw - pr[7] * mul1) * g2x + (pr[5] * mw - pr[7] * mul2) * g2y;
        gm[2] += (pr[8] * mw - pr[11] * mul1) * g2x + (pr[9] * mw - pr[11] * mul2) * g2y;
    }
    if (bad) {
        const float nanv = __uint_as_float(0x7fc00000u);
        gm[0] = gm[1] = gm[2] = nanv;
#pragma unroll
        for (int k = 0; k < 6; k++) gc[k] = nanv;
#pragma unroll
        for (int k = 0; k < 4; k++) acc[k] = nanv;
        gop = g2x = g2y = nanv;
    }
    if (FACE) {   // blockIdx.y = frame: the parameter gradients go to per-frame slices (summed by k_sum_frames)
        gom_face::FaceFwd o;
        float cdummy[3], s3[3], dcr[9], dw[3], dS[3];
        gom_face::face_forward_in(fin, face.sigma, o, cdummy, s3);
        gom_face::face_backward(o, s3, face.sigma, gm, gc, dcr, dw, dS);
        const size_t P3 = 3 * (size_t)P;
        float *dc = face.d_corner + (size_t)fr * 9 * P + 9 * (size_t)i;
#pragma unroll
        for (int k = 0; k < 9; k++) dc[k] = dcr[k];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            face.d_so3[fr * P3 + (size_t)k * P + i] = dw[k];
            face.d_scale[fr * P3 + (size_t)k * P + i] = dS[k];
            face.d_appearance[fr * P3 + (size_t)k * P + i] = acc[k];
        }
        return;
    }
    dL_dmeans[3 * i] = gm[0];
    dL_dmeans[3 * i + 1] = gm[1];
    dL_dmeans[3 * i + 2] = gm[2];
#pragma unroll
    for (int k = 0; k < 6; k++) dL_dcov6[6 * i + k] = gc[k];
#pragma unroll
    for (int k = 0; k < C; k++) dL_dcolors[C * i + k] = acc[k];
    dL_dopacity[i] = gop;
    if (dL_dmeans2D) {
        dL_dmeans2D[3 * i] = g2x;
        dL_dmeans2D[3 * i + 1] = g2y;
        dL_dmeans2D[3 * i + 2] = 0.f;
    }
}

}  // namespace

#define GOM_LDS_TILE_LIMIT 8192

int gom_launch_preprocess(GomState *s, const GomCamera &cam, int P, const float *means3D, const float *cov6,
                          const float *opacity, int32_t *radii_out, hipStream_t st, const GomFaceArgs *face) {
    const int n_tiles = s->gx * s->gy;
    const int blocks = (P + 255) / 256;
    if (blocks == 0) return 0;
    GomKernelTimer timer(s, GOM_K_PREPROCESS, st);
    const dim3 grid(blocks, s->B);
    const uint32_t cap = (uint32_t)(s->capPairs > 0xffffffffLL ? 0xffffffffLL : s->capPairs);
    const bool lds = n_tiles <= GOM_LDS_TILE_LIMIT;
    const GomFaceArgs fa = face ? *face : GomFaceArgs{};
    // (with `face` the two arrays are this kernel's outputs: the frame step owns them)
    float *m = const_cast<float *>(means3D), *c = const_cast<float *>(cov6);
    const bool hist_here = face && face->vdepth_minmax && s->rankSort;
    const uint32_t nb = 1u << s->nbShift;
#define GOM_PP(LDSH, FACEV) hipLaunchKernelGGL((k_preprocess<LDSH, FACEV>), grid, dim3(256), ((LDSH ? n_tiles : 0) + (hist_here ? nb : 0)) * sizeof(uint32_t), st, cam, s->cams, P, m, c, opacity, fa, \
                                               s->depth, s->xy, s->conic_opacity, s->tiles_touched, s->rect, s->radii, radii_out, s->tile_count, s->pair_off,  \
                                               s->status, s->gx, s->gy, cap, s->rankSort ? s->depth_minmax : nullptr, s->rankSort ? s->rec_g : nullptr, s->big_list, s->big_count, s->capBigFrames, \
                                               hist_here ? s->bucket_count : nullptr, nb)
    if (lds) { if (face) GOM_PP(true, true); else GOM_PP(true, false); }
    else { if (face) GOM_PP(false, true); else GOM_PP(false, false); }
#undef GOM_PP
    GOM_LAUNCH_CHECK();
    return 0;
}

// rank: the splat path with depth ranking (the caller has run gom_launch_depth_hist): the scan kernel also turns the bucket counts
// into ranges, the Gaussians are ranked (gom_launch_depth_rank) and the emission writes 32-bit ranks.
int gom_launch_scan_emit(GomState *s, int P, hipStream_t st, bool rank, float *fill_out, int fill_C, const float *fill_bg) {
    const int n_tiles = s->gx * s->gy;
    const uint32_t cap = (uint32_t)(s->capPairs > 0xffffffffLL ? 0xffffffffLL : s->capPairs);
    {
        GomKernelTimer timer(s, GOM_K_SCAN, st);
        GomScatterRider sr{};
        unsigned scatter_blocks = 0;
        if (rank && P > 0) {
            sr.P = P; sr.B = s->B; sr.nb = 1u << s->nbShift; sr.nblk = s->rank_blocks; sr.depth = s->depth; sr.radii = s->radii; sr.minmax = s->rank_minmax;
            sr.bkeys = s->bkeys;
            scatter_blocks = (unsigned)((P + 1023) / 1024) * (unsigned)s->B;
        }
        hipLaunchKernelGGL(k_scan_tiles, dim3((rank ? 2 : 1) + scatter_blocks), dim3(1024), scatter_blocks ? 3 * sr.nb * sizeof(uint32_t) : 0, st, s->tile_count,
                           s->tile_base, s->tile_cursor, s->seg_base,
                           s->tile_nmax, n_tiles * s->B, s->status, cap, (uint32_t)s->segShift, rank ? s->bucket_count : nullptr, s->bucket_base,
                           s->bucket_cursor, rank ? (s->B << s->nbShift) : 0, rank ? s->work_items : nullptr, (uint32_t)s->capItems, s->big_count, s->B, s->capBigFrames, sr);
    }
    GOM_LAUNCH_CHECK();
    const int blocks = (P + 255) / 256;
    if (blocks == 0) return 0;
    GomKernelTimer timer(s, GOM_K_EMIT, st);
    GomEmptyFill fill{};
    fill.first_block = blocks;
    if (fill_out) {
        fill.out_color = fill_out; fill.final_T = s->final_T; fill.n_contrib = s->n_contrib; fill.tile_base = s->tile_base; fill.cams = s->cams;
        fill.H = s->H; fill.W = s->W; fill.C = fill_C;
        for (int ch = 0; ch < 4; ch++) fill.bg[ch] = fill_bg[ch];
    }
    GomSortRider sorts{};
    if (rank) {
        const int cap = 8 * 256;
        sorts.blocks = 1 << s->nbShift;
        sorts.P = P;
        sorts.log_chunk = (uint32_t)(31 - __builtin_clz((unsigned)(s->sortCap < cap ? s->sortCap : cap)));
        sorts.bucket_base = s->bucket_base; sorts.bkeys = s->bkeys; sorts.scratch = s->bkeys_scratch; sorts.order = s->order; sorts.rank_of = s->rank_of;
        sorts.bucket_count = s->bucket_count; sorts.bucket_cursor = s->bucket_cursor;
    }
    sorts.B = s->B;
    const dim3 grid((unsigned)((size_t)(sorts.blocks + blocks + (fill_out ? (n_tiles + GOM_FILL_TILES - 1) / GOM_FILL_TILES : 0)) * s->B));
    const size_t sort_lds = rank ? 8 * 256 * sizeof(uint64_t) : 0;   // the riders' key slab shares the dynamic LDS
#define GOM_EMIT(AGG, RK, LDS) hipLaunchKernelGGL((k_emit<AGG, RK>), grid, dim3(256), (LDS) > sort_lds ? (LDS) : sort_lds, st, P, s->depth, s->radii, s->rect, s->tile_cursor, s->keys, s->gx, s->gy, \
                                                  s->status, s->keys32, fill, sorts)
    if (n_tiles <= GOM_LDS_TILE_LIMIT) {
        if (rank) GOM_EMIT(true, true, 2 * n_tiles * sizeof(uint32_t)); else GOM_EMIT(true, false, 2 * n_tiles * sizeof(uint32_t));
    } else {
        if (rank) GOM_EMIT(false, true, (size_t)0); else GOM_EMIT(false, false, (size_t)0);
    }
#undef GOM_EMIT
    GOM_LAUNCH_CHECK();
    return 0;
}

int gom_launch_preprocess_backward(GomState *s, const GomCamera &cam, int P, int C, const float *means3D,
                                   const float *cov6, float *dL_dmeans3D, float *dL_dcov6, float *dL_dcolors,
                                   float *dL_dopacity, float *dL_dmeans2D, hipStream_t st, const GomFaceArgs *face) {
    const int blocks = (P + 255) / 256;
    if (blocks == 0) return 0;
    if (face && C != 4) { gom_set_error("the fused face backward carries [r g b 1] features (C = 4)"); return -1; }
    GomKernelTimer timer(s, GOM_K_PREPROCESS_BWD, st);
    const dim3 grid(blocks + GOM_BIG_CAP / 4, s->B);   // + the rider blocks (one wave per many-tile Gaussian of the frame)
    const GomFaceArgs fa = face ? *face : GomFaceArgs{};
#define GOM_PB(CC, RK, FC) hipLaunchKernelGGL((k_preprocess_bwd<CC, RK, FC>), grid, dim3(256), 0, st, cam, s->cams, fa, P, means3D, cov6, s->radii, s->tiles_touched,    \
                                              s->conic_opacity, s->pair_off, s->pair_pos, s->rect, s->tile_base, s->tile_nmax, s->rank_of, s->tile_qlim, s->gx, s->partial, \
                                              s->status, dL_dmeans3D, dL_dcov6, dL_dcolors, dL_dopacity, dL_dmeans2D, s->big_list, s->big_count)
    if (face) { if (s->rankSort) GOM_PB(4, true, true); else GOM_PB(4, false, true); }
    else if (C == 3) { if (s->rankSort) GOM_PB(3, true, false); else GOM_PB(3, false, false); }
    else { if (s->rankSort) GOM_PB(4, true, false); else GOM_PB(4, false, false); }
#undef GOM_PB
    GOM_LAUNCH_CHECK();
    return 0;
}
