// Mesh normal map + soft silhouette of the posed mesh: replaces the PyTorch3D MeshRasterizer
// (naive O(pixels x faces) path, bin_size=0) + NormalShader / hard_rgb_blend and MeshRenderer(SoftSilhouetteShader)
// that the reference runs every frame at models/modules/renderer/mesh.py:65-128 (called from models/model.py:270-273).
//
//   normal(p)  = n0 + n1 + n2 of the nearest (smallest interpolated z) face covering pixel p, 0 where none
//                (phong_normal_shading with barycentrics = 1, hard_rgb_blend, x alpha: mesh.py:23-30,57-61,121-124)
//   alpha(p)   = 1 - prod_k (1 - sigmoid(-sdist_k / 1e-4)) over the faces whose blurred footprint reaches p
//                (blur_radius = ln(1/1e-4 - 1) * sigma_cfg, faces_per_pixel = 50: mesh.py:99-112,127)
//
// MI355X design: faces are tile-binned exactly like the Gaussians (count -> scan -> emit -> per-tile sort by face index
// = the GomState machinery of raster_pre.hip / raster_render.hip, which also yields the (face, tile) -> list position
// table for the backward), but on 8x8-pixel tiles: a face plus its blur band covers ~7x7 pixels, so 16x16 bins would make
// every pixel wade through ~5x more candidates.  One wave per tile walks its LDS-staged face list once for both outputs;
// a face none of the wave's 64 pixels can reach is skipped wave-uniformly before any per-pixel math.
// faces_per_pixel: every qualifying face has sigmoid(-sdist/1e-4) >= 0.285 (sdist < blur_radius = 9.2e-5), so with 50
// or more of them the product is < 6e-8 whichever 50 are kept: alpha is computed over ALL qualifying faces and
// differs from the K = 50 truncation by less than one fp32 ulp of 1.0.
// Backward: one thread per (tile, face) list entry re-walks the few pixels of the face's footprint in that tile and
// writes one record; a per-face gather and a per-vertex CSR gather follow.  No float atomics: bitwise reproducible.
#include "gom_internal.h"

namespace {

constexpr float kEpsArea = 1e-8f;
constexpr int kFaceStride = 12;  // x0 y0 z0 x1 y1 z1 x2 y2 z2 area . .
constexpr int kMeshTile = 8;     // pixels per side of a mesh-raster bin (the GomState tile grid is built for 2H x 2W / 16)

struct MeshGrid {
    int H, W, gx, gy;
    float rngx, offx, rngy, offy;  // pix_to_non_square_ndc
};

// a / b as a * v_rcp_f32(b) (1 ulp): an IEEE division is a ~10-instruction sequence on gfx950 and the per-(face, pixel) evaluation
// below had twelve of them; the forward, the backward and pix_to_face all go through the same functions, so they stay consistent
// with each other, and against the fp64 oracle 1 ulp is far inside the tolerances (tests/test_gpu_mesh.py).  The pixel centres
// keep their IEEE divisions: they decide on which side of an edge a pixel lies, and with approximate centres the silhouette of
// 2 of 14 random scenes of the soak differed from the oracle's by 4e-3 at single pixels.
__device__ __forceinline__ float qdiv(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
__device__ __forceinline__ float pix_x(const MeshGrid &g, int xi) { return -g.offx + (g.rngx * (float)(g.W - 1 - xi) + g.offx) / (float)g.W; }
__device__ __forceinline__ float pix_y(const MeshGrid &g, int yi) { return -g.offy + (g.rngy * (float)(g.H - 1 - yi) + g.offy) / (float)g.H; }
// continuous pixel coordinate of an NDC value (inverse of the above)
__device__ __forceinline__ float ndc_to_px(const MeshGrid &g, float x) { return (float)(g.W - 1) - ((x + g.offx) * (float)g.W - g.offx) / g.rngx; }
__device__ __forceinline__ float ndc_to_py(const MeshGrid &g, float y) { return (float)(g.H - 1) - ((y + g.offy) * (float)g.H - g.offy) / g.rngy; }

__device__ __forceinline__ float edge_fn(float px, float py, float ax, float ay, float bx, float by) {
    return (px - ax) * (by - ay) - (py - ay) * (bx - ax);
}
// squared distance from p to segment ab; t_out: clamped parameter, degenerate: the edge is (nearly) a point
__device__ __forceinline__ float seg_dist2(float px, float py, float ax, float ay, float bx, float by, float &t_out, bool &degenerate) {
    const float dx = bx - ax, dy = by - ay;
    const float l2 = dx * dx + dy * dy;
    degenerate = !(l2 > kEpsArea);
    if (degenerate) { t_out = 1.f; return (px - bx) * (px - bx) + (py - by) * (py - by); }
    float t = qdiv((px - ax) * dx + (py - ay) * dy, l2);
    t = fminf(fmaxf(t, 0.f), 1.f);
    t_out = t;
    const float qx = ax + t * dx, qy = ay + t * dy;
    return (px - qx) * (px - qx) + (py - qy) * (py - qy);
}

struct FaceEval {
    bool hard, soft, inside;
    float z_hard, prob;
    int edge;       // nearest edge 0: v0v1, 1: v1v2, 2: v2v0
    float t;        // clamped parameter on it
    bool degenerate;
};

__device__ __forceinline__ FaceEval eval_face(const float *f, float px, float py, float blur, float blur_radius, float inv_sigma) {
    FaceEval r;
    r.hard = r.soft = r.inside = false;
    r.z_hard = 0.f; r.prob = 0.f; r.edge = 0; r.t = 0.f; r.degenerate = false;
    const float x0 = f[0], y0 = f[1], z0 = f[2], x1 = f[3], y1 = f[4], z1 = f[5], x2 = f[6], y2 = f[7], z2 = f[8], area = f[9];
    const float xmin = fminf(fminf(x0, x1), x2), xmax = fmaxf(fmaxf(x0, x1), x2);
    const float ymin = fminf(fminf(y0, y1), y2), ymax = fmaxf(fmaxf(y0, y1), y2);
    if (px > xmax + blur || px < xmin - blur || py > ymax + blur || py < ymin - blur) return r;
    const float iden = __builtin_amdgcn_rcpf(area + kEpsArea);
    const float w0 = edge_fn(px, py, x1, y1, x2, y2) * iden, w1 = edge_fn(px, py, x2, y2, x0, y0) * iden, w2 = edge_fn(px, py, x0, y0, x1, y1) * iden;
    r.inside = w0 > 0.f && w1 > 0.f && w2 > 0.f;
    if (r.inside) {
        const float pz = w0 * z0 + w1 * z1 + w2 * z2;
        r.hard = pz >= 0.f;
        r.z_hard = pz;
    }
    // soft pass: clipped barycentrics for the depth test
    const float c0 = fminf(fmaxf(w0, 0.f), 1.f), c1 = fminf(fmaxf(w1, 0.f), 1.f), c2 = fminf(fmaxf(w2, 0.f), 1.f);
    const float is = __builtin_amdgcn_rcpf(fmaxf(c0 + c1 + c2, 1e-5f));
    const float pzs = (c0 * is) * z0 + (c1 * is) * z1 + (c2 * is) * z2;
    if (!(pzs >= 0.f)) return r;
    float t0, t1, t2;
    bool g0, g1, g2;
    const float d0 = seg_dist2(px, py, x0, y0, x1, y1, t0, g0), d1 = seg_dist2(px, py, x1, y1, x2, y2, t1, g1), d2 = seg_dist2(px, py, x2, y2, x0, y0, t2, g2);
    // (selects on values: written as assignments under `if`, the compiler kept t0..t2 / g0..g2 in a scratch-memory array and indexed it with the edge --
    //  12 scratch stores and 2 loads per evaluation in k_mesh_backward_entries, the only consumer of t and degenerate: 110 -> 103 us)
    const bool u1 = d1 < d0;
    const float da = u1 ? d1 : d0, ta = u1 ? t1 : t0;
    const int ga = u1 ? (int)g1 : (int)g0;
    const bool u2 = d2 < da;
    const float d = u2 ? d2 : da;
    r.edge = u2 ? 2 : (u1 ? 1 : 0);
    r.t = u2 ? t2 : ta;
    r.degenerate = (u2 ? (int)g2 : ga) != 0;
    if (!r.inside && !(d < blur_radius)) return r;
    r.soft = true;
    const float sd = r.inside ? -d : d;
    r.prob = __builtin_amdgcn_rcpf(1.f + __expf(sd * inv_sigma));   // sigmoid(-sd / sigma)
    return r;
}

// ---- per face: projected geometry, conservative tile rect, tile counts (feeds k_scan_tiles / k_emit / k_sort) --------
__global__ void __launch_bounds__(256) k_mesh_preprocess(MeshGrid g, int F, const float *__restrict__ verts, const int32_t *__restrict__ faces, float blur,
                                                         float *__restrict__ face_geo, float *__restrict__ depth, float2 *__restrict__ xy,
                                                         float4 *__restrict__ conic_opacity, uint32_t *__restrict__ tiles_touched,
                                                         ushort4 *__restrict__ rect, int32_t *__restrict__ radii, uint32_t *__restrict__ tile_count,
                                                         uint32_t *__restrict__ pair_off, GomDevStatus *__restrict__ status, int lds_hist) {
    extern __shared__ uint32_t s_hist[];   // per-block tile histogram (gx * gy entries; 0 bytes = count straight in global memory)
    __shared__ uint32_t s_wsum[4];
    __shared__ uint32_t s_blockbase;
    const int n_tiles = g.gx * g.gy;
    if (lds_hist) {
        for (int i = threadIdx.x; i < n_tiles; i += 256) s_hist[i] = 0;
        __syncthreads();
    }
    const int f = blockIdx.x * 256 + threadIdx.x;
    uint32_t my_tiles = 0;
    if (f < F) {
        const int i0 = faces[3 * f], i1 = faces[3 * f + 1], i2 = faces[3 * f + 2];
        float v[9];
#pragma unroll
        for (int k = 0; k < 3; k++) { v[k] = verts[3 * (size_t)i0 + k]; v[3 + k] = verts[3 * (size_t)i1 + k]; v[6 + k] = verts[3 * (size_t)i2 + k]; }
        const float area = edge_fn(v[6], v[7], v[0], v[1], v[3], v[4]);
        float *dst = face_geo + (size_t)f * kFaceStride;
#pragma unroll
        for (int k = 0; k < 9; k++) dst[k] = v[k];
        dst[9] = area;
        int x0 = 0, y0 = 0, x1 = 0, y1 = 0;
        const bool finite = isfinite(v[0]) && isfinite(v[1]) && isfinite(v[3]) && isfinite(v[4]) && isfinite(v[6]) && isfinite(v[7]);
        if (finite && fabsf(area) > kEpsArea && fmaxf(fmaxf(v[2], v[5]), v[8]) >= 0.f) {
            const float xmin = fminf(fminf(v[0], v[3]), v[6]) - blur, xmax = fmaxf(fmaxf(v[0], v[3]), v[6]) + blur;
            const float ymin = fminf(fminf(v[1], v[4]), v[7]) - blur, ymax = fmaxf(fmaxf(v[1], v[4]), v[7]) + blur;
            // +X is left / +Y is up: the largest NDC value maps to the smallest pixel index; one pixel of safety margin
            const float pxa = fminf(fmaxf(ndc_to_px(g, xmax) - 1.f, -1.f), (float)g.W), pxb = fminf(fmaxf(ndc_to_px(g, xmin) + 1.f, -1.f), (float)g.W);
            const float pya = fminf(fmaxf(ndc_to_py(g, ymax) - 1.f, -1.f), (float)g.H), pyb = fminf(fmaxf(ndc_to_py(g, ymin) + 1.f, -1.f), (float)g.H);
            const int ixa = max(0, (int)floorf(pxa)), ixb = min(g.W - 1, (int)ceilf(pxb));
            const int iya = max(0, (int)floorf(pya)), iyb = min(g.H - 1, (int)ceilf(pyb));
            if (ixa <= ixb && iya <= iyb) {
                x0 = ixa / kMeshTile; x1 = ixb / kMeshTile + 1; y0 = iya / kMeshTile; y1 = iyb / kMeshTile + 1;
            }
        }
        my_tiles = (uint32_t)((x1 - x0) * (y1 - y0));
        depth[f] = 0.f;                               // the sort key is the face index alone
        xy[f] = make_float2(0.f, 0.f);
        conic_opacity[f] = make_float4(0.f, 0.f, 0.f, 0.f);
        radii[f] = my_tiles ? 1 : 0;
        tiles_touched[f] = my_tiles;
        rect[f] = make_ushort4((unsigned short)x0, (unsigned short)y0, (unsigned short)x1, (unsigned short)y1);
        for (int y = y0; y < y1; y++)
            for (int x = x0; x < x1; x++) atomicAdd(lds_hist ? &s_hist[y * g.gx + x] : &tile_count[y * g.gx + x], 1u);
    }
    // private range of this face in pair_pos (same scheme as k_preprocess)
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
        s_blockbase = tot ? atomicAdd(&status->pair_cursor, tot) : 0u;
    }
    __syncthreads();
    uint32_t woff = 0;
    for (int w = 0; w < wid; w++) woff += s_wsum[w];
    if (f < F) pair_off[f] = s_blockbase + woff + (x - my_tiles);
    if (lds_hist) {
        for (int t = threadIdx.x; t < n_tiles; t += 256) {
            const uint32_t c = s_hist[t];
            if (c) atomicAdd(&tile_count[t], c);
        }
    }
}

// ---- forward, pass 1: one wave per (8x8 tile, 128-face segment of its list), lane = pixel ----------------------------------
// A UV-sphere pole or a crumpled region puts thousands of faces into one tile; min-z and the product over faces are
// associative, so the lists are cut into the same 128-entry segments as the splat lists (seg_desc of k_sort) and every
// segment is an independent wave.  Partial results: product of (1 - p), nearest depth and its face per pixel.
// MESH_FW waves per segment: wave w takes the faces [w n/MESH_FW, (w+1) n/MESH_FW) of the segment for all 64 pixels (a lone wave per 128 faces
// left ~2.7 waves per SIMD, each a ~9 000-instruction chain), the four partial (product, nearest z, face) triples are folded in
// list order in LDS.
#ifndef MESH_FW
#define MESH_FW 8   // waves per segment (2: 115 us, 4: 89 us, 8: 83 us at 55 104 faces, 512x512)
#endif
constexpr int kChunk = 64;
__global__ void __launch_bounds__(64 * MESH_FW) k_mesh_forward_seg(MeshGrid g, const uint4 *__restrict__ seg_desc, const uint32_t *__restrict__ point_list,
                                                          const float *__restrict__ face_geo, float blur, float blur_radius, float inv_sigma,
                                                          float *__restrict__ seg_Q, float *__restrict__ seg_z, uint32_t *__restrict__ seg_face,
                                                          const GomDevStatus *__restrict__ status) {
    __shared__ float s_f[MESH_FW][kChunk][10];
    __shared__ uint32_t s_id[MESH_FW][kChunk];
    __shared__ float s_pQ[MESH_FW][64], s_pz[MESH_FW][64];
    __shared__ uint32_t s_pf[MESH_FW][64];
    if (status->overflow) return;
    const uint32_t nsegs = status->num_segs;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (uint32_t seg = blockIdx.x; seg < nsegs; seg += gridDim.x) {
        const uint4 d = seg_desc[seg];
        const int tile = (int)d.x;
        const uint32_t n_all = d.z, per = (n_all + MESH_FW - 1) / MESH_FW;
        const uint32_t first = min((uint32_t)w * per, n_all), n = min(per, n_all - first), start = d.y + first;
        const int tx0 = (tile % g.gx) * kMeshTile, ty0 = (tile / g.gx) * kMeshTile;
        const float px = pix_x(g, tx0 + (lane & 7)), py = pix_y(g, ty0 + (lane >> 3));
        // the wave's pixel rectangle in NDC (+X left / +Y up: the first pixel has the largest coordinate)
        const float wx_hi = pix_x(g, tx0), wx_lo = pix_x(g, min(tx0 + kMeshTile, g.W) - 1), wy_hi = pix_y(g, ty0), wy_lo = pix_y(g, min(ty0 + kMeshTile, g.H) - 1);
        float best_z = 3.0e38f, Q = 1.f;
        uint32_t best = 0xffffffffu;
        for (uint32_t e0 = 0; e0 < n; e0 += kChunk) {   // (wave-private LDS slices: no workgroup barrier inside)
            const uint32_t cn = min((uint32_t)kChunk, n - e0);
            bool reach = false;
            if ((uint32_t)lane < cn) {   // lane = face: stage it and test its blurred box against the wave's rectangle
                const uint32_t f = point_list[start + e0 + lane];
                const float *src = face_geo + (size_t)f * kFaceStride;
                float v[10];
#pragma unroll
                for (int k = 0; k < 10; k++) { v[k] = src[k]; s_f[w][lane][k] = v[k]; }
                s_id[w][lane] = f;
                const float xmin = fminf(fminf(v[0], v[3]), v[6]) - blur, xmax = fmaxf(fmaxf(v[0], v[3]), v[6]) + blur;
                const float ymin = fminf(fminf(v[1], v[4]), v[7]) - blur, ymax = fmaxf(fmaxf(v[1], v[4]), v[7]) + blur;
                reach = !(wx_lo > xmax || wx_hi < xmin || wy_lo > ymax || wy_hi < ymin);
            }
            unsigned long long todo = __ballot(reach);
            while (todo) {   // wave-uniform loop over the faces that can touch this tile at all
                const int j = __builtin_ctzll(todo);
                todo &= todo - 1;
                const FaceEval r = eval_face(s_f[w][j], px, py, blur, blur_radius, inv_sigma);
                if (r.hard && r.z_hard < best_z) { best_z = r.z_hard; best = s_id[w][j]; }   // ascending face index: first wins ties
                if (r.soft) Q *= (1.f - r.prob);
            }
        }
        __syncthreads();   // the previous segment's fold is done
        s_pQ[w][lane] = Q; s_pz[w][lane] = best_z; s_pf[w][lane] = best;
        __syncthreads();
        if (w == 0) {
#pragma unroll
            for (int k = 1; k < MESH_FW; k++) {
                Q *= s_pQ[k][lane];
                if (s_pz[k][lane] < best_z) { best_z = s_pz[k][lane]; best = s_pf[k][lane]; }
            }
            seg_Q[(size_t)seg * 64 + lane] = Q;
            seg_z[(size_t)seg * 64 + lane] = best_z;
            seg_face[(size_t)seg * 64 + lane] = best;
        }
    }
}

// ---- forward, pass 2: per tile, fold its segments in list order ---------------------------------------------------------------
__global__ void __launch_bounds__(256) k_mesh_combine(MeshGrid g, const uint32_t *__restrict__ seg_base, const float *__restrict__ seg_Q,
                                                     const float *__restrict__ seg_z, const uint32_t *__restrict__ seg_face,
                                                     const int32_t *__restrict__ faces, const float *__restrict__ vnormals,
                                                     float *__restrict__ normal_map, float *__restrict__ alpha, uint32_t *__restrict__ pix_to_face,
                                                     float *__restrict__ prodQ, const GomDevStatus *__restrict__ status) {
    const int tile = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;   // a wave per tile, four tiles per workgroup (4 096 single-wave workgroups were dispatch-bound)
    if (tile >= g.gx * g.gy) return;
    const int xi = (tile % g.gx) * kMeshTile + (lane & 7), yi = (tile / g.gx) * kMeshTile + (lane >> 3);
    if (xi >= g.W || yi >= g.H) return;
    const uint32_t sb = seg_base[tile], ns = status->overflow ? 0u : seg_base[tile + 1] - sb;
    float best_z = 3.0e38f, Q = 1.f;
    uint32_t best = 0xffffffffu;
    for (uint32_t k0 = 0; k0 < ns; k0 += 8) {   // 8 segments per trip, their loads issued together: a tile of thousands of faces (dozens of segments) was a chain of one
        float q8[8], z8[8];                      // memory latency per segment, and the launch lasted as long as the longest tile (20 us)
        uint32_t f8[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const size_t o = (size_t)(sb + min(k0 + u, ns - 1)) * 64 + lane;
            q8[u] = seg_Q[o]; z8[u] = seg_z[o]; f8[u] = seg_face[o];
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (k0 + u < ns) {
                Q *= q8[u];
                if (z8[u] < best_z) { best_z = z8[u]; best = f8[u]; }
            }
        }
    }
    const size_t p = (size_t)yi * g.W + xi;
    float nx = 0.f, ny = 0.f, nz = 0.f;
    if (best != 0xffffffffu) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const int v = faces[3 * (size_t)best + c];
            nx += vnormals[3 * (size_t)v]; ny += vnormals[3 * (size_t)v + 1]; nz += vnormals[3 * (size_t)v + 2];
        }
    }
    normal_map[3 * p] = nx; normal_map[3 * p + 1] = ny; normal_map[3 * p + 2] = nz;
    pix_to_face[p] = best;
    prodQ[p] = Q;
    if (alpha) alpha[p] = 1.f - Q;
}

// ---- backward, step 1: one thread per (tile, face) entry -> 9-float record at the entry's list position --------------
//   [0..5] d/d(x0,y0,x1,y1,x2,y2) from the silhouette, [6..8] d/d(n0+n1+n2) from the normal map
__global__ void __launch_bounds__(256) k_mesh_backward_entries(MeshGrid g, const uint4 *__restrict__ seg_desc, const uint32_t *__restrict__ point_list,
                                                               const float *__restrict__ face_geo, float blur, float blur_radius, float inv_sigma,
                                                               const uint32_t *__restrict__ pix_to_face, const float *__restrict__ prodQ,
                                                               const float *__restrict__ d_normal, const float *__restrict__ d_alpha,
                                                               float *__restrict__ partial, const GomDevStatus *__restrict__ status) {
    // One workgroup per (tile, <=128-face segment).  The 64 pixels of the tile are staged in LDS once (every entry walks a part of
    // the same 8x8 pixels; four dependent global loads per visited pixel made the pixel loop a latency chain), and the pixel
    // rows are split over the four waves: thread = (entry, pair of rows), <= 16 pixels each instead of <= 64, four times the
    // waves in flight.  The four partial records of an entry are summed in row order (no atomics: reproducible).
    __shared__ uint32_t s_p2f[64];
    __shared__ float s_Q[64], s_da[64], s_dn[64][3];
    __shared__ float s_rec[4][256][9];   // (a segment holds 128 faces, 256 if the state was switched to large segments)
    if (status->overflow) return;
    const uint32_t nsegs = status->num_segs;
    const int lane = threadIdx.x & 63, part = threadIdx.x >> 6;
    for (uint32_t seg = blockIdx.x; seg < nsegs; seg += gridDim.x) {
        const uint4 sd = seg_desc[seg];
        const int tile = (int)sd.x;
        const int tx0 = (tile % g.gx) * kMeshTile, ty0 = (tile / g.gx) * kMeshTile;
        const uint32_t base = sd.y, n = sd.z;   // n <= 256
        __syncthreads();   // the previous segment's readers are done
        float da_mine = 0.f;
        if (threadIdx.x < 64) {
            const int xi = tx0 + (threadIdx.x & 7), yi = ty0 + (threadIdx.x >> 3);
            const bool in = xi < g.W && yi < g.H;
            const size_t p = in ? (size_t)yi * g.W + xi : 0;
            s_p2f[threadIdx.x] = in ? pix_to_face[p] : 0xffffffffu;
            s_Q[threadIdx.x] = in ? prodQ[p] : 1.f;
            da_mine = (in && d_alpha) ? d_alpha[p] : 0.f;
            s_da[threadIdx.x] = da_mine;
#pragma unroll
            for (int c = 0; c < 3; c++) s_dn[threadIdx.x][c] = in ? d_normal[3 * p + c] : 0.f;
        }
        // A pixel whose dL/d(alpha) is exactly 0 adds +-0 to every silhouette term below (its factors are finite), and that is most of them: under the
        // body alpha = 1 - Q rounds to 1.0f (Q < 2^-25 with dozens of faces in reach) and |alpha - target|'s gradient is sign(0) = 0 there (train.py:142) --
        // only the band around the outline carries a silhouette gradient.  Such pixels skip the per-face evaluation (the same bits: x + (+-0) = x, and the
        // sums start at +0); a tile without any such pixel does the normal-map bookkeeping alone.  (NaN != 0: a poisoned gradient is still evaluated.)
        const bool sil = __syncthreads_or(da_mine != 0.f) != 0;   // (also the barrier behind the staging)
        for (uint32_t e = lane; e < n; e += 64) {
            const uint32_t f = point_list[base + e];
            float fg[10];
#pragma unroll
            for (int k = 0; k < 10; k++) fg[k] = face_geo[(size_t)f * kFaceStride + k];
            const float xmin = fminf(fminf(fg[0], fg[3]), fg[6]) - blur, xmax = fmaxf(fmaxf(fg[0], fg[3]), fg[6]) + blur;
            const float ymin = fminf(fminf(fg[1], fg[4]), fg[7]) - blur, ymax = fmaxf(fmaxf(fg[1], fg[4]), fg[7]) + blur;
            const int ixa = max(tx0, (int)floorf(ndc_to_px(g, xmax) - 1.f)), ixb = min(min(tx0 + kMeshTile, g.W) - 1, (int)ceilf(ndc_to_px(g, xmin) + 1.f));
            int iya = max(ty0, (int)floorf(ndc_to_py(g, ymax) - 1.f)), iyb = min(min(ty0 + kMeshTile, g.H) - 1, (int)ceilf(ndc_to_py(g, ymin) + 1.f));
            iya = max(iya, ty0 + 2 * part);          // this wave's two pixel rows
            iyb = min(iyb, ty0 + 2 * part + 1);
            float gv[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, gn[3] = {0.f, 0.f, 0.f};
            for (int yi = iya; yi <= iyb; yi++)
                for (int xi = ixa; xi <= ixb; xi++) {
                    const int lp = (yi - ty0) * kMeshTile + (xi - tx0);
                    if (s_p2f[lp] == f) { gn[0] += s_dn[lp][0]; gn[1] += s_dn[lp][1]; gn[2] += s_dn[lp][2]; }
                    if (!sil || s_da[lp] == 0.f) continue;
                    const float px = pix_x(g, xi), py = pix_y(g, yi);
                    const FaceEval r = eval_face(fg, px, py, blur, blur_radius, inv_sigma);
                    if (!r.soft) continue;
                    // alpha = 1 - prod(1 - p_j), p = sigmoid(-sd/sigma):  d alpha / d sd_k = -(Q / (1 - p_k)) p_k (1 - p_k) / sigma
                    const float others = qdiv(s_Q[lp], fmaxf(1.f - r.prob, 1e-30f));
                    float gd = -s_da[lp] * others * r.prob * (1.f - r.prob) * inv_sigma;   // d L / d sd
                    if (r.inside) gd = -gd;                                                    // sd = -dist inside
                    // (corner a = r.edge, b = the next one, spelled as selects and a static loop: as fg[3 * ia] / gv[2 * ia] -- register arrays indexed by a value known
                    //  only at run time -- this block cost a third of the kernel: 103 -> 70 us.  Same arithmetic; the compiler contracts other multiply-adds now, so the
                    //  four terms move in their last bit: LABBOOK R5.10)
                    const int ia = r.edge, ib = ia == 2 ? 0 : ia + 1;
                    const float ax = ia == 0 ? fg[0] : (ia == 1 ? fg[3] : fg[6]), ay = ia == 0 ? fg[1] : (ia == 1 ? fg[4] : fg[7]);
                    const float bx = ib == 0 ? fg[0] : (ib == 1 ? fg[3] : fg[6]), by = ib == 0 ? fg[1] : (ib == 1 ? fg[4] : fg[7]);
                    float cax = 0.f, cay = 0.f, cbx, cby;
                    if (r.degenerate) {
                        cbx = gd * -2.f * (px - bx); cby = gd * -2.f * (py - by);
                    } else {
                        const float qx = ax + r.t * (bx - ax), qy = ay + r.t * (by - ay);
                        const float rx = px - qx, ry = py - qy;
                        // dist = |p - q|^2, q = a + t (b - a): for 0 < t < 1 the residual is normal to the edge, so only q's
                        // explicit dependence on a, b counts; at the clamps q is the end point itself
                        cax = gd * -2.f * rx * (1.f - r.t); cay = gd * -2.f * ry * (1.f - r.t);
                        cbx = gd * -2.f * rx * r.t;         cby = gd * -2.f * ry * r.t;
                    }
#pragma unroll
                    for (int k = 0; k < 3; k++) {   // corner a first, then corner b (a != b): the order of the indexed form
                        if (!r.degenerate && k == ia) { gv[2 * k] += cax; gv[2 * k + 1] += cay; }
                        if (k == ib) { gv[2 * k] += cbx; gv[2 * k + 1] += cby; }
                    }
                }
#pragma unroll
            for (int k = 0; k < 6; k++) s_rec[part][e][k] = gv[k];
#pragma unroll
            for (int k = 0; k < 3; k++) s_rec[part][e][6 + k] = gn[k];
        }
        __syncthreads();
        if (threadIdx.x < n) {
            float r9[9];
#pragma unroll
            for (int k = 0; k < 9; k++) r9[k] = ((s_rec[0][threadIdx.x][k] + s_rec[1][threadIdx.x][k]) + s_rec[2][threadIdx.x][k]) + s_rec[3][threadIdx.x][k];
            float4 *rec = reinterpret_cast<float4 *>(partial + (size_t)(base + threadIdx.x) * GOM_PARTIAL_STRIDE);
            rec[0] = make_float4(r9[0], r9[1], r9[2], r9[3]);
            rec[1] = make_float4(r9[4], r9[5], r9[6], r9[7]);
            rec[2] = make_float4(r9[8], 0.f, 0.f, 0.f);
        }
    }
}

// ---- backward, step 2: per face, sum the records of its tiles (fixed order) ------------------------------------------------
__global__ void __launch_bounds__(256) k_mesh_face_gather(int F, const uint32_t *__restrict__ tiles_touched, const uint32_t *__restrict__ pair_off,
                                                          const uint32_t *__restrict__ pair_pos, const float *__restrict__ partial,
                                                          float *__restrict__ d_face, const GomDevStatus *__restrict__ status) {
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f >= F) return;
    float acc[9];
#pragma unroll
    for (int k = 0; k < 9; k++) acc[k] = status->overflow ? __uint_as_float(0x7fc00000u) : 0.f;
    const uint32_t nt = status->overflow ? 0u : tiles_touched[f];
    const uint32_t *pp = pair_pos + pair_off[f];
    for (uint32_t k0 = 0; k0 < nt; k0 += 4) {   // 4 records in flight per trip (independent index -> record chains), summed in list order
        float4 a[4], b[4], c[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const float4 *rec = reinterpret_cast<const float4 *>(partial + (size_t)pp[min(k0 + u, nt - 1)] * GOM_PARTIAL_STRIDE);
            a[u] = rec[0]; b[u] = rec[1]; c[u] = rec[2];
        }
#pragma unroll
        for (int u = 0; u < 4; u++)
            if (k0 + u < nt) {
                acc[0] += a[u].x; acc[1] += a[u].y; acc[2] += a[u].z; acc[3] += a[u].w; acc[4] += b[u].x; acc[5] += b[u].y; acc[6] += b[u].z; acc[7] += b[u].w;
                acc[8] += c[u].x;
            }
    }
#pragma unroll
    for (int k = 0; k < 9; k++) d_face[(size_t)f * 9 + k] = acc[k];
}

// ---- backward, step 3: per vertex, CSR gather over incident (face, corner) pairs -------------------------------------------
__global__ void __launch_bounds__(256) k_mesh_vertex_gather(int N, const int32_t *__restrict__ csr_off, const int32_t *__restrict__ csr_idx,
                                                            const float *__restrict__ d_face, float *__restrict__ d_verts, float *__restrict__ d_vnormals) {
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= N) return;
    float gx = 0.f, gy = 0.f, n0 = 0.f, n1 = 0.f, n2 = 0.f;
    const int kb = csr_off[v], ke = csr_off[v + 1];
    for (int k0 = kb; k0 < ke; k0 += 8) {   // 8 incident corners in flight per trip, summed in list order
        float r[8][5];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int fc = csr_idx[min(k0 + u, ke - 1)], f = fc / 3, c = fc % 3;
            const float *q = d_face + (size_t)f * 9;
            r[u][0] = q[2 * c]; r[u][1] = q[2 * c + 1]; r[u][2] = q[6]; r[u][3] = q[7]; r[u][4] = q[8];
        }
#pragma unroll
        for (int u = 0; u < 8; u++)
            if (k0 + u < ke) { gx += r[u][0]; gy += r[u][1]; n0 += r[u][2]; n1 += r[u][3]; n2 += r[u][4]; }
    }
    d_verts[3 * (size_t)v] = gx; d_verts[3 * (size_t)v + 1] = gy; d_verts[3 * (size_t)v + 2] = 0.f;   // z only orders the faces
    d_vnormals[3 * (size_t)v] = n0; d_vnormals[3 * (size_t)v + 1] = n1; d_vnormals[3 * (size_t)v + 2] = n2;
}

// ---- vertex normals (PyTorch3D Meshes.verts_normals_packed: area-weighted face normals summed on the corners, then
// normalize(eps = 1e-6)); CSR gather instead of three index_add scatters: no atomics, fixed summation order ---------------
__device__ __forceinline__ void cross3(const float *a, const float *b, float *o) {
    o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}
__global__ void __launch_bounds__(256) k_vnormal_fwd(int N, const float *__restrict__ verts, const int32_t *__restrict__ faces,
                                                     const int32_t *__restrict__ csr_off, const int32_t *__restrict__ csr_idx,
                                                     const float *__restrict__ R, float *__restrict__ sums, float *__restrict__ normals) {
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= N) return;
    float acc[3] = {0.f, 0.f, 0.f};
    const int kb = csr_off[v], ke = csr_off[v + 1];
    for (int k0 = kb; k0 < ke; k0 += 8) {   // 8 incident corners in flight per trip (index -> face -> 3 vertices is a 3-hop chain each)
        int iv[8][3];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int fc = csr_idx[min(k0 + u, ke - 1)], f = fc / 3, c = fc % 3;
            // corner c: cross(v[c+1] - v[c], v[c+2] - v[c]) -- the three per-corner expressions of the reference framework
            iv[u][0] = faces[3 * f + c]; iv[u][1] = faces[3 * f + (c + 1) % 3]; iv[u][2] = faces[3 * f + (c + 2) % 3];
        }
        float p[8][3][3];
#pragma unroll
        for (int u = 0; u < 8; u++)
#pragma unroll
            for (int w = 0; w < 3; w++)
#pragma unroll
                for (int d = 0; d < 3; d++) p[u][w][d] = verts[3 * (size_t)iv[u][w] + d];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            float e1[3], e2[3], n[3];
#pragma unroll
            for (int d = 0; d < 3; d++) { e1[d] = p[u][1][d] - p[u][0][d]; e2[d] = p[u][2][d] - p[u][0][d]; }
            cross3(e1, e2, n);
            if (k0 + u < ke) { acc[0] += n[0]; acc[1] += n[1]; acc[2] += n[2]; }
        }
    }
    const float len = fmaxf(sqrtf(acc[0] * acc[0] + acc[1] * acc[1] + acc[2] * acc[2]), 1e-6f);
#pragma unroll
    for (int d = 0; d < 3; d++) sums[3 * (size_t)v + d] = acc[d];
    const float n[3] = {acc[0] / len, acc[1] / len, acc[2] / len};
#pragma unroll
    for (int d = 0; d < 3; d++)   // optional rotation of the unit normal (into the camera frame: models/model.py:272)
        normals[3 * (size_t)v + d] = R ? (R[3 * d] * n[0] + R[3 * d + 1] * n[1]) + R[3 * d + 2] * n[2] : n[d];
}
// per face: g = sum over its corners of d L / d sums[corner vertex]; n = (v1 - v0) x (v2 - v0):
// dL/dv1 = (v2 - v0) x g, dL/dv2 = g x (v1 - v0), dL/dv0 = -(dL/dv1 + dL/dv2)   -> d_corner [F][3][3]
__global__ void __launch_bounds__(256) k_vnormal_bwd_face(int F, const float *__restrict__ verts, const int32_t *__restrict__ faces,
                                                          const float *__restrict__ sums, const float *__restrict__ d_normals,
                                                          const float *__restrict__ R, float *__restrict__ d_corner) {
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f >= F) return;
    int idx[3] = {faces[3 * f], faces[3 * f + 1], faces[3 * f + 2]};
    float g[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float *sv = sums + 3 * (size_t)idx[c], *dr = d_normals + 3 * (size_t)idx[c];
        float dn[3] = {dr[0], dr[1], dr[2]};
        if (R) {   // the forward rotated the unit normal: dL/dn = R^T dL/d(R n)
#pragma unroll
            for (int d = 0; d < 3; d++) dn[d] = (R[d] * dr[0] + R[3 + d] * dr[1]) + R[6 + d] * dr[2];
        }
        const float l = sqrtf(sv[0] * sv[0] + sv[1] * sv[1] + sv[2] * sv[2]);
        if (l > 1e-6f) {   // y = s / |s|: dL/ds = (dn - y (y . dn)) / |s|
            const float y[3] = {sv[0] / l, sv[1] / l, sv[2] / l};
            const float dot = y[0] * dn[0] + y[1] * dn[1] + y[2] * dn[2];
#pragma unroll
            for (int d = 0; d < 3; d++) g[d] += (dn[d] - y[d] * dot) / l;
        } else {           // y = s / 1e-6
#pragma unroll
            for (int d = 0; d < 3; d++) g[d] += dn[d] / 1e-6f;
        }
    }
    float e1[3], e2[3], d1[3], d2[3];
#pragma unroll
    for (int d = 0; d < 3; d++) { e1[d] = verts[3 * (size_t)idx[1] + d] - verts[3 * (size_t)idx[0] + d]; e2[d] = verts[3 * (size_t)idx[2] + d] - verts[3 * (size_t)idx[0] + d]; }
    cross3(e2, g, d1);
    cross3(g, e1, d2);
#pragma unroll
    for (int d = 0; d < 3; d++) {
        d_corner[9 * (size_t)f + d] = -(d1[d] + d2[d]);
        d_corner[9 * (size_t)f + 3 + d] = d1[d];
        d_corner[9 * (size_t)f + 6 + d] = d2[d];
    }
}
__global__ void __launch_bounds__(256) k_corner_gather(int N, const int32_t *__restrict__ csr_off, const int32_t *__restrict__ csr_idx,
                                                       const float *__restrict__ d_corner, float *__restrict__ d_verts) {
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= N) return;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    const int kb = csr_off[v], ke = csr_off[v + 1];
    for (int k0 = kb; k0 < ke; k0 += 8) {   // 8 corners in flight per trip, summed in list order
        float r[8][3];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const float *q = d_corner + 3 * (size_t)csr_idx[min(k0 + u, ke - 1)];
            r[u][0] = q[0]; r[u][1] = q[1]; r[u][2] = q[2];
        }
#pragma unroll
        for (int u = 0; u < 8; u++)
            if (k0 + u < ke) { a0 += r[u][0]; a1 += r[u][1]; a2 += r[u][2]; }
    }
    d_verts[3 * (size_t)v] = a0; d_verts[3 * (size_t)v + 1] = a1; d_verts[3 * (size_t)v + 2] = a2;
}

MeshGrid make_grid(int H, int W) {
    MeshGrid g;
    g.H = H; g.W = W;
    g.gx = (W + kMeshTile - 1) / kMeshTile; g.gy = (H + kMeshTile - 1) / kMeshTile;
    g.rngx = W > H ? 2.f * (float)W / (float)H : 2.f; g.offx = g.rngx / 2.f;
    g.rngy = H > W ? 2.f * (float)H / (float)W : 2.f; g.offy = g.rngy / 2.f;
    return g;
}

}  // namespace

extern "C" int gom_mesh_raster_forward(GomState *s, int N, int F, int H, int W, const float *verts_ndc, const int32_t *faces, const float *vnormals,
                                       float blur_radius, float sigma, float *normal_map, float *alpha, void *stream) {
    if (!s) { gom_set_error("null state"); return -1; }
    if (N <= 0 || F <= 0 || H <= 0 || W <= 0 || !(sigma > 0.f) || blur_radius < 0.f) { gom_set_error("gom_mesh_raster_forward: bad sizes"); return -1; }
    if (!verts_ndc || !faces || !vnormals || !normal_map) { gom_set_error("gom_mesh_raster_forward: null pointer"); return -1; }
    hipStream_t st = (hipStream_t)stream;
    // the shared binning machinery thinks in 16-pixel tiles: give it the doubled image so that its grid is ours (8-pixel tiles)
    if (int rc = gom_ensure_capacity(s, F, 2 * H, 2 * W, 1)) return rc;
    if ((size_t)F * kFaceStride > s->capMeshFace) {
        if (s->mesh_face) GOM_HIP_CHECK(hipFree(s->mesh_face));
        s->mesh_face = nullptr;
        GOM_HIP_CHECK(hipMalloc((void **)&s->mesh_face, (size_t)F * kFaceStride * sizeof(float) * 2));   // geometry + per-face gradient
        s->capMeshFace = (size_t)F * kFaceStride;
    }
    s->P = F; s->H = H; s->W = W; s->C = 3; s->cams = nullptr; s->haveForward = false;
    const MeshGrid g = make_grid(H, W);
    const float blur = sqrtf(blur_radius);
    const int lds_hist = g.gx * g.gy <= 8192 ? 1 : 0;
    hipLaunchKernelGGL(k_mesh_preprocess, dim3((F + 255) / 256), dim3(256), lds_hist ? g.gx * g.gy * sizeof(uint32_t) : 0, st, g, F, verts_ndc, faces, blur,
                       s->mesh_face, s->depth, s->xy, s->conic_opacity, s->tiles_touched, s->rect, s->radii, s->tile_count, s->pair_off, s->status, lds_hist);
    GOM_LAUNCH_CHECK();
    if (int rc = gom_launch_scan_emit(s, F, st)) return rc;
    s->sortSplit = true;      // (keys = face indices: the 4-wave sort for every tile, an index bitmap for the few long lists)
    const int src = gom_launch_sort(s, st);
    s->sortSplit = false;
    if (src) return src;
    hipLaunchKernelGGL(k_mesh_forward_seg, dim3(8192), dim3(64 * MESH_FW), 0, st, g, s->seg_desc, s->point_list, s->mesh_face, blur, blur_radius, 1.0f / sigma,
                       s->seg_T, s->seg_Tend, s->seg_last, s->status);
    GOM_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_mesh_combine, dim3((g.gx * g.gy + 3) / 4), dim3(256), 0, st, g, s->seg_base, s->seg_T, s->seg_Tend, s->seg_last, faces, vnormals,
                       normal_map, alpha, s->n_contrib, s->final_T, s->status);
    GOM_LAUNCH_CHECK();
    s->meshBlurRadius = blur_radius; s->meshSigma = sigma; s->meshForward = true;
    return 0;
}

extern "C" int gom_mesh_raster_backward(GomState *s, int N, int F, int H, int W, const int32_t *csr_off, const int32_t *csr_idx,
                                        const float *d_normal_map, const float *d_alpha, float *d_verts_ndc, float *d_vnormals, void *stream) {
    if (!s || !s->meshForward || s->P != F || s->H != H || s->W != W) { gom_set_error("gom_mesh_raster_backward without a matching forward on this state"); return -1; }
    if (!csr_off || !csr_idx || !d_normal_map || !d_verts_ndc || !d_vnormals) { gom_set_error("gom_mesh_raster_backward: null pointer"); return -1; }
    hipStream_t st = (hipStream_t)stream;
    const MeshGrid g = make_grid(H, W);
    float *d_face = s->mesh_face + s->capMeshFace;
    hipLaunchKernelGGL(k_mesh_backward_entries, dim3(8192), dim3(256), 0, st, g, s->seg_desc, s->point_list, s->mesh_face, sqrtf(s->meshBlurRadius),
                       s->meshBlurRadius, 1.0f / s->meshSigma, s->n_contrib, s->final_T, d_normal_map, d_alpha, s->partial, s->status);
    GOM_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_mesh_face_gather, dim3((F + 255) / 256), dim3(256), 0, st, F, s->tiles_touched, s->pair_off, s->pair_pos, s->partial, d_face,
                       s->status);
    GOM_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_mesh_vertex_gather, dim3((N + 255) / 256), dim3(256), 0, st, N, csr_off, csr_idx, d_face, d_verts_ndc, d_vnormals);
    GOM_LAUNCH_CHECK();
    return 0;
}

extern "C" int gom_mesh_pix_to_face(GomState *s, int32_t *dst, void *stream) {
    if (!s || !s->meshForward || !dst) { gom_set_error("gom_mesh_pix_to_face without a forward"); return -1; }
    GOM_HIP_CHECK(hipMemcpyAsync(dst, s->n_contrib, (size_t)s->H * s->W * sizeof(int32_t), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return 0;
}

extern "C" int gom_vertex_normals_forward(int N, int F, const float *verts, const int32_t *faces, const int32_t *csr_off, const int32_t *csr_idx,
                                          const float *R, float *sums, float *normals, void *stream) {
    (void)F;
    if (N <= 0 || !verts || !faces || !csr_off || !csr_idx || !sums || !normals) { gom_set_error("gom_vertex_normals_forward: bad arguments"); return -1; }
    hipLaunchKernelGGL(k_vnormal_fwd, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, N, verts, faces, csr_off, csr_idx, R, sums, normals);
    GOM_LAUNCH_CHECK();
    return 0;
}

extern "C" int gom_vertex_normals_backward(int N, int F, const float *verts, const int32_t *faces, const int32_t *csr_off, const int32_t *csr_idx,
                                           const float *R, const float *sums, const float *d_normals, float *d_corner_scratch, float *d_verts,
                                           void *stream) {
    if (N <= 0 || F <= 0 || !verts || !faces || !csr_off || !csr_idx || !sums || !d_normals || !d_corner_scratch || !d_verts) {
        gom_set_error("gom_vertex_normals_backward: bad arguments");
        return -1;
    }
    hipLaunchKernelGGL(k_vnormal_bwd_face, dim3((F + 255) / 256), dim3(256), 0, (hipStream_t)stream, F, verts, faces, sums, d_normals, R, d_corner_scratch);
    GOM_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_corner_gather, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, N, csr_off, csr_idx, d_corner_scratch, d_verts);
    GOM_LAUNCH_CHECK();
    return 0;
}

// ---- world -> the mesh rasterizer's NDC (utils/pc_util.py:30-46 ndc_T_world): c = E [v; 1], cam = c.xyz / c.w, p = K cam,
// xy = p.xy / p.z, then x, y -> -(2 xy / S - a) with S the shorter image side; output (x_ndc, y_ndc, cam.z) per vertex.
// Through torch this is 14 launches forward and ~25 backward on a 27 000-vertex array; here one each (the camera matrices are read
// from device memory: nothing for a graph capture to freeze).
namespace {

struct NdcCam { float E[16], K[9]; };

__device__ __forceinline__ void ndc_forward_point(const float *__restrict__ E, const float *__restrict__ K, float x, float y, float z, float (&c)[4], float (&p)[3]) {
#pragma unroll
    for (int r = 0; r < 4; r++) c[r] = E[4 * r] * x + E[4 * r + 1] * y + E[4 * r + 2] * z + E[4 * r + 3];
    const float cam[3] = {c[0] / c[3], c[1] / c[3], c[2] / c[3]};
#pragma unroll
    for (int r = 0; r < 3; r++) p[r] = K[3 * r] * cam[0] + K[3 * r + 1] * cam[1] + K[3 * r + 2] * cam[2];
}

__global__ void __launch_bounds__(256) k_ndc_fwd(int N, float S, float ax, float ay, const float *__restrict__ verts, const float *__restrict__ K,
                                                 const float *__restrict__ E, float *__restrict__ out) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    float c[4], p[3];
    ndc_forward_point(E, K, verts[n], verts[(size_t)N + n], verts[2 * (size_t)N + n], c, p);
    out[3 * (size_t)n] = -((p[0] / p[2] / S) * 2.f - ax);
    out[3 * (size_t)n + 1] = -((p[1] / p[2] / S) * 2.f - ay);
    out[3 * (size_t)n + 2] = c[2] / c[3];
}

__global__ void __launch_bounds__(256) k_ndc_bwd(int N, float S, const float *__restrict__ verts, const float *__restrict__ K, const float *__restrict__ E,
                                                 const float *__restrict__ d_out, float *__restrict__ d_verts) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    float c[4], p[3];
    ndc_forward_point(E, K, verts[n], verts[(size_t)N + n], verts[2 * (size_t)N + n], c, p);
    const float dxy0 = -2.f / S * d_out[3 * (size_t)n], dxy1 = -2.f / S * d_out[3 * (size_t)n + 1];
    const float ip2 = 1.f / p[2];
    const float dp[3] = {dxy0 * ip2, dxy1 * ip2, -(dxy0 * p[0] + dxy1 * p[1]) * ip2 * ip2};
    float dcam[3];
#pragma unroll
    for (int j = 0; j < 3; j++) dcam[j] = K[j] * dp[0] + K[3 + j] * dp[1] + K[6 + j] * dp[2];
    dcam[2] += d_out[3 * (size_t)n + 2];
    const float ic3 = 1.f / c[3];
    const float dc[4] = {dcam[0] * ic3, dcam[1] * ic3, dcam[2] * ic3, -(dcam[0] * c[0] + dcam[1] * c[1] + dcam[2] * c[2]) * ic3 * ic3};
#pragma unroll
    for (int j = 0; j < 3; j++) d_verts[(size_t)j * N + n] = E[j] * dc[0] + E[4 + j] * dc[1] + E[8 + j] * dc[2] + E[12 + j] * dc[3];
}

}  // namespace

extern "C" int gom_ndc_from_world_forward(int N, int H, int W, const float *verts, const float *K, const float *E, float *out, void *stream) {
    if (N <= 0 || H <= 0 || W <= 0 || !verts || !K || !E || !out) { gom_set_error("gom_ndc_from_world_forward: bad arguments"); return -1; }
    const float S = (float)(H < W ? H : W), ax = H < W ? (float)W / (float)H : 1.f, ay = H < W ? 1.f : (float)H / (float)W;
    hipLaunchKernelGGL(k_ndc_fwd, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, N, S, ax, ay, verts, K, E, out);
    GOM_LAUNCH_CHECK();
    return 0;
}

extern "C" int gom_ndc_from_world_backward(int N, int H, int W, const float *verts, const float *K, const float *E, const float *d_out, float *d_verts,
                                        void *stream) {
    if (N <= 0 || H <= 0 || W <= 0 || !verts || !K || !E || !d_out || !d_verts) { gom_set_error("gom_ndc_from_world_backward: bad arguments"); return -1; }
    hipLaunchKernelGGL(k_ndc_bwd, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, N, (float)(H < W ? H : W), verts, K, E, d_out, d_verts);
    GOM_LAUNCH_CHECK();
    return 0;
}
