// Geometry half of the hot path: skeleton forward kinematics, linear-blend
// skinning and the per-face Gaussian frame, forward and backward.
//
// Replaces ~150 eager PyTorch launches per frame in the reference
// (utils/body_util.py:612-644 get_global_RTs/apply_lbs; models/model.py:225-234
// + :27-41 Steiner frame; PyTorch3D so3_exp_map) with
// 3 forward + 2(3) backward kernels.  Backward scatter to vertices goes through
// a CSR vertex->corner adjacency: no atomics, bitwise reproducible.
//
// All of this is HBM/L2-bound streaming work (a few hundred flops per 100-200
// bytes); layouts follow the reference's channel-first parameters ((3,N), (3,F),
// (25,N)) so every per-thread read is a coalesced row read.
#include "gom_internal.h"
#include "geom_face.hpp"

namespace {

__constant__ int c_parent[24] = {-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21};
__constant__ int c_level_start[9] = {1, 4, 7, 10, 15, 18, 20, 22, 24};   // first joint of every depth of that tree below the root (tests/test_abi.py checks this table against c_parent)

__device__ void invert4x4(const float *m, float *inv) {
    // cofactor expansion (row-major in, row-major out)
    float t[16];
    t[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
    t[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
    t[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
    t[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
    t[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
    t[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
    t[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
    t[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
    t[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
    t[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
    t[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
    t[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
    t[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
    t[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
    t[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
    t[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
    const float det = m[0] * t[0] + m[1] * t[4] + m[2] * t[8] + m[3] * t[12];
    const float id = 1.0f / det;
    for (int i = 0; i < 16; i++) inv[i] = t[i] * id;
}

// ---- forward kinematics: one wave ------------------------------------------
// RT[j] = top 3 rows of (prod_{chain} local[k]) * inverse(cnl_gtfms[j]).
// The chain itself, for one frame, by the first 24 / 16 threads of a workgroup of NT threads (every thread takes the barriers): local
// [R | T] matrices into L, inverse canonical transforms into Ci, global transforms G, and the skinning rows RT (24 x 12) into LDS.
template <int NT>
__device__ __forceinline__ void fk_forward_block(int t, const float *__restrict__ cnl, const float *__restrict__ Rs, const float *__restrict__ Ts,
                                                 float (*L)[16], float (*G)[16], float (*Ci)[16], float *rt) {
#pragma clang fp contract(on)
    if (t < 24) {
        for (int r = 0; r < 3; r++) {
            for (int c = 0; c < 3; c++) L[t][4 * r + c] = Rs[9 * t + 3 * r + c];
            L[t][4 * r + 3] = Ts[3 * t + r];
        }
        L[t][12] = 0.f; L[t][13] = 0.f; L[t][14] = 0.f; L[t][15] = 1.f;
        float m[16], inv[16];
        for (int i = 0; i < 16; i++) m[i] = cnl[16 * t + i];
        invert4x4(m, inv);
        for (int i = 0; i < 16; i++) Ci[t][i] = inv[i];
    }
    __syncthreads();
    if (t < 16) G[0][t] = L[0][t];
    __syncthreads();
    // the tree level by level (the joints of a level are a contiguous range of the numbering and depend on the level above only):
    // 8 barriers instead of 23, the same sixteen products per joint
    for (int l = 0; l < 8; l++) {
        const int j0 = c_level_start[l], cnt = c_level_start[l + 1] - j0;
        for (int u = t; u < 16 * cnt; u += NT) {
            const int i = j0 + (u >> 4), e = u & 15, p = c_parent[i], r = e >> 2, c = e & 3;
            G[i][e] = G[p][4 * r + 0] * L[i][c] + G[p][4 * r + 1] * L[i][4 + c] + G[p][4 * r + 2] * L[i][8 + c] + G[p][4 * r + 3] * L[i][12 + c];
        }
        __syncthreads();
    }
    for (int o = t; o < 24 * 12; o += NT) {
        const int j = o / 12, e = o % 12;
        const int r = e < 9 ? e / 3 : e - 9, c = e < 9 ? e % 3 : 3;
        rt[o] = G[j][4 * r + 0] * Ci[j][c] + G[j][4 * r + 1] * Ci[j][4 + c] + G[j][4 * r + 2] * Ci[j][8 + c] + G[j][4 * r + 3] * Ci[j][12 + c];
    }
    __syncthreads();
}

__global__ void __launch_bounds__(64) k_fk_fwd(const float *__restrict__ cnl, const float *__restrict__ Rs, const float *__restrict__ Ts,
                                               float *__restrict__ RT, float *__restrict__ save) {
    __shared__ float L[24][16], G[24][16], Ci[24][16], s_rt[24 * 12];
    const int t = threadIdx.x;
    {  // blockIdx.x = frame of a batched launch
        const size_t fr = blockIdx.x;
        cnl += fr * 24 * 16; Rs += fr * 24 * 9; Ts += fr * 24 * 3; RT += fr * 24 * 12; save += fr * 24 * 32;
    }
    fk_forward_block<64>(t, cnl, Rs, Ts, L, G, Ci, s_rt);
    for (int o = t; o < 24 * 12; o += 64) RT[o] = s_rt[o];
    for (int o = t; o < 24 * 16; o += 64) {
        save[(o / 16) * 32 + (o % 16)] = G[o / 16][o % 16];
        save[(o / 16) * 32 + 16 + (o % 16)] = Ci[o / 16][o % 16];
    }
}

__global__ void __launch_bounds__(64) k_fk_bwd(const float *__restrict__ Rs, const float *__restrict__ Ts, const float *__restrict__ save,
                                               const float *__restrict__ dRT, float *__restrict__ dRs, float *__restrict__ dTs) {
    __shared__ float L[24][16], G[24][16], Ci[24][16], dG[24][16], dL[24][16];
    const int t = threadIdx.x;
    if (t < 24) {
        for (int r = 0; r < 3; r++) {
            for (int c = 0; c < 3; c++) L[t][4 * r + c] = Rs[9 * t + 3 * r + c];
            L[t][4 * r + 3] = Ts[3 * t + r];
        }
        L[t][12] = 0.f; L[t][13] = 0.f; L[t][14] = 0.f; L[t][15] = 1.f;
    }
    for (int o = t; o < 24 * 16; o += 64) {
        G[o / 16][o % 16] = save[(o / 16) * 32 + (o % 16)];
        Ci[o / 16][o % 16] = save[(o / 16) * 32 + 16 + (o % 16)];
    }
    __syncthreads();
    // dG[j] = dF[j] * Ci[j]^T, dF rows 0..2 = [dR | dT], row 3 = 0
    for (int o = t; o < 24 * 16; o += 64) {
        const int j = o / 16, r = (o % 16) >> 2, k = o & 3;
        float acc = 0.f;
        if (r < 3) {
            for (int c = 0; c < 4; c++) {
                const float df = c < 3 ? dRT[12 * j + 3 * r + c] : dRT[12 * j + 9 + r];
                acc += df * Ci[j][4 * k + c];
            }
        }
        dG[j][o % 16] = acc;
    }
    __syncthreads();
    for (int i = 23; i >= 1; i--) {
        const int p = c_parent[i];
        float add = 0.f;
        if (t < 16) {
            const int r = t >> 2, c = t & 3;
            // dL[i] = G[p]^T dG[i]
            dL[i][t] = G[p][0 + r] * dG[i][c] + G[p][4 + r] * dG[i][4 + c] + G[p][8 + r] * dG[i][8 + c] + G[p][12 + r] * dG[i][12 + c];
            // dG[p] += dG[i] L[i]^T
            add = dG[i][4 * r + 0] * L[i][4 * c + 0] + dG[i][4 * r + 1] * L[i][4 * c + 1] + dG[i][4 * r + 2] * L[i][4 * c + 2] + dG[i][4 * r + 3] * L[i][4 * c + 3];
        }
        __syncthreads();
        if (t < 16) dG[p][t] += add;
        __syncthreads();
    }
    if (t < 16) dL[0][t] = dG[0][t];
    __syncthreads();
    for (int o = t; o < 24 * 9; o += 64) dRs[o] = dL[o / 9][4 * ((o % 9) / 3) + (o % 3)];
    for (int o = t; o < 24 * 3; o += 64) dTs[o] = dL[o / 3][4 * (o % 3) + 3];
}

// ---- LBS: one thread per vertex ---------------------------------------------
// v' = sum_j w_j (R_j v + T_j) for vertex n, from the skinning rows in LDS (contraction pinned: k_lbs_fwd and k_fk_lbs_fwd give the same bits)
__device__ __forceinline__ void lbs_blend(int N, int J, int n, const float *__restrict__ xyz, const float *__restrict__ w, const float *s_rt,
                                          float *__restrict__ out, float (&pos)[3]) {
#pragma clang fp contract(on)
    const float x = xyz[n], y = xyz[(size_t)N + n], z = xyz[2 * (size_t)N + n];
    float ox = 0.f, oy = 0.f, oz = 0.f;
    for (int j = 0; j < J; j++) {
        const float wj = w[(size_t)j * N + n];
        if (wj != 0.f) {
            const float *m = s_rt + 12 * j;
            ox += (m[0] * x + m[1] * y + m[2] * z + m[9]) * wj;
            oy += (m[3] * x + m[4] * y + m[5] * z + m[10]) * wj;
            oz += (m[6] * x + m[7] * y + m[8] * z + m[11]) * wj;
        }
    }
    out[n] = ox;
    out[(size_t)N + n] = oy;
    out[2 * (size_t)N + n] = oz;
    pos[0] = ox; pos[1] = oy; pos[2] = oz;
}

__global__ void __launch_bounds__(256) k_lbs_fwd(int N, int J, const float *__restrict__ xyz, const float *__restrict__ w,
                                                 const float *__restrict__ RT, float *__restrict__ out) {
    extern __shared__ float s_rt[];
    RT += (size_t)blockIdx.y * J * 12;  // blockIdx.y = frame of a batched launch (shared canonical vertices and weights)
    out += (size_t)blockIdx.y * 3 * N;
    for (int i = threadIdx.x; i < J * 12; i += 256) s_rt[i] = RT[i];
    __syncthreads();
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    float pos[3];
    lbs_blend(N, J, n, xyz, w, s_rt, out, pos);
}

// FK + LBS in one launch (the frame step): every workgroup runs the frame's 24-joint chain itself (two microseconds of barriers next to a
// launch and a dependent kernel's latency) and skins its 256 vertices from the rows it holds in LDS; the first workgroup of a frame
// also leaves RT and the FK state for the backward.  Same arithmetic as k_fk_fwd + k_lbs_fwd.
__global__ void __launch_bounds__(256) k_fk_lbs_fwd(int N, const float *__restrict__ cnl, const float *__restrict__ Rs, const float *__restrict__ Ts,
                                                    const float *__restrict__ xyz, const float *__restrict__ w, float *__restrict__ RT,
                                                    float *__restrict__ save, float *__restrict__ out, GomCamera cam1, const GomCamera *__restrict__ cams,
                                                    uint32_t *__restrict__ vdepth_minmax) {
    __shared__ float L[24][16], G[24][16], Ci[24][16], s_rt[24 * 12];
    __shared__ float s_zmin[4], s_zmax[4];
    const int t = threadIdx.x;
    {
        const size_t fr = blockIdx.y;
        cnl += fr * 24 * 16; Rs += fr * 24 * 9; Ts += fr * 24 * 3; RT += fr * 24 * 12; save += fr * 24 * 32; out += fr * 3 * N;
    }
    fk_forward_block<256>(t, cnl, Rs, Ts, L, G, Ci, s_rt);
    if (blockIdx.x == 0) {
        for (int o = t; o < 24 * 12; o += 256) RT[o] = s_rt[o];
        for (int o = t; o < 24 * 16; o += 256) {
            save[(o / 16) * 32 + (o % 16)] = G[o / 16][o % 16];
            save[(o / 16) * 32 + 16 + (o % 16)] = Ci[o / 16][o % 16];
        }
    }
    const int n = blockIdx.x * 256 + threadIdx.x;
    float pos[3] = {0.f, 0.f, 0.f};
    if (n < N) lbs_blend(N, 24, n, xyz, w, s_rt, out, pos);
    if (vdepth_minmax) {
        // view depth range of this block's posed vertices (bit patterns of floats >= 0.2: what is visible lies beyond the near cut, and a
        // Gaussian's mean is the centroid of three of these vertices): the depth ranking's bucket map without a pass over the Gaussians
        const float *v = cams ? cams[blockIdx.y].view : cam1.view;
        const float z = v[2] * pos[0] + v[6] * pos[1] + v[10] * pos[2] + v[14];
        float lo = n < N ? z : 3.0e38f, hi = n < N ? z : -3.0e38f;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { lo = fminf(lo, __shfl_xor(lo, d, 64)); hi = fmaxf(hi, __shfl_xor(hi, d, 64)); }
        if ((t & 63) == 0) { s_zmin[t >> 6] = lo; s_zmax[t >> 6] = hi; }
        __syncthreads();
        if (t == 0) {
            lo = fmaxf(fminf(fminf(s_zmin[0], s_zmin[1]), fminf(s_zmin[2], s_zmin[3])), 0.2f);
            hi = fmaxf(fmaxf(fmaxf(s_zmax[0], s_zmax[1]), fmaxf(s_zmax[2], s_zmax[3])), 0.2f);
            reinterpret_cast<uint2 *>(vdepth_minmax)[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = make_uint2(__float_as_uint(lo), __float_as_uint(hi));
        }
    }
}

// ---- per-face Gaussian frame: geom_face.hpp ------------------------------------
using namespace gom_face;

__global__ void __launch_bounds__(256) k_face_fwd(int N, int F, const float *__restrict__ verts, const int32_t *__restrict__ faces,
                                                  const float *__restrict__ so3, const float *__restrict__ scale, float sigma,
                                                  float *__restrict__ xyz, float *__restrict__ cov6,
                                                  const float *__restrict__ appearance, float *__restrict__ feat4) {
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f >= F) return;
    {  // blockIdx.y = frame of a batched launch: per-frame posed vertices in, per-frame Gaussians out
        const size_t fr = blockIdx.y;
        verts += fr * 3 * N; xyz += fr * 3 * F; cov6 += fr * 6 * F;
        if (feat4) feat4 += fr * 4 * F;
    }
    if (feat4) {  // (3,F) colour parameter -> (F,4) rasterizer features [r g b 1] (gaussian.py:49), fused here
        *reinterpret_cast<float4 *>(feat4 + 4 * (size_t)f) =
            make_float4(appearance[f], appearance[(size_t)F + f], appearance[2 * (size_t)F + f], 1.0f);
    }
    FaceFwd o;
    float c[3], s3[3], c6[6];
    face_forward(verts, N, faces, so3, scale, F, f, sigma, o, c, s3);
    face_cov6(o, c6);
    xyz[3 * f] = c[0]; xyz[3 * f + 1] = c[1]; xyz[3 * f + 2] = c[2];
#pragma unroll
    for (int k = 0; k < 6; k++) cov6[6 * f + k] = c6[k];
}

__global__ void __launch_bounds__(256) k_face_bwd(int N, int F, const float *__restrict__ verts, const int32_t *__restrict__ faces,
                                                  const float *__restrict__ so3, const float *__restrict__ scale, float sigma,
                                                  const float *__restrict__ d_xyz, const float *__restrict__ d_cov6,
                                                  float *__restrict__ d_corner, float *__restrict__ d_so3, float *__restrict__ d_scale,
                                                  const float *__restrict__ d_feat4, float *__restrict__ d_appearance) {
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f >= F) return;
    {  // blockIdx.y = frame of a batched launch; the parameter gradients go to per-frame slices (summed by k_sum_frames)
        const size_t fr = blockIdx.y;
        verts += fr * 3 * N; d_xyz += fr * 3 * F; d_cov6 += fr * 6 * F; d_corner += fr * 9 * F;
        d_so3 += fr * 3 * F; d_scale += fr * 3 * F;
        if (d_appearance) { d_feat4 += fr * 4 * F; d_appearance += fr * 3 * F; }
    }
    if (d_appearance) {  // (F,4) feature gradient -> (3,F) colour-parameter gradient
        const float4 g = *reinterpret_cast<const float4 *>(d_feat4 + 4 * (size_t)f);
        d_appearance[f] = g.x; d_appearance[(size_t)F + f] = g.y; d_appearance[2 * (size_t)F + f] = g.z;
    }
    FaceFwd o;
    float cdummy[3], s3[3];
    face_forward(verts, N, faces, so3, scale, F, f, sigma, o, cdummy, s3);
    float gx3[3], gc6[6], dcr[9], dw[3], dS[3];
#pragma unroll
    for (int k = 0; k < 3; k++) gx3[k] = d_xyz[3 * f + k];
#pragma unroll
    for (int k = 0; k < 6; k++) gc6[k] = d_cov6[6 * f + k];
    face_backward(o, s3, sigma, gx3, gc6, dcr, dw, dS);
    d_so3[f] = dw[0]; d_so3[(size_t)F + f] = dw[1]; d_so3[2 * (size_t)F + f] = dw[2];
    d_scale[f] = dS[0]; d_scale[(size_t)F + f] = dS[1]; d_scale[2 * (size_t)F + f] = dS[2];
#pragma unroll
    for (int k = 0; k < 9; k++) d_corner[9 * f + k] = dcr[k];
}

// ---- vertex gather + LBS backward ----------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

__global__ void __launch_bounds__(256) k_vertex_bwd(int N, int J, const float *__restrict__ xyz, const float *__restrict__ w,
                                                    const float *__restrict__ RT, const int32_t *__restrict__ csr_off,
                                                    const int32_t *__restrict__ csr_idx, const float *__restrict__ d_corner,
                                                    const float *__restrict__ d_extra, float *__restrict__ d_obs,
                                                    float *__restrict__ d_xyz, int F) {
    extern __shared__ float s_rt[];
    {  // blockIdx.y = frame of a batched launch (F = faces, the stride of d_corner)
        const size_t fr = blockIdx.y;
        RT += fr * J * 12; d_xyz += fr * 3 * N;
        if (d_corner) d_corner += fr * 9 * F;
        if (d_extra) d_extra += fr * 3 * N;
        if (d_obs) d_obs += fr * 3 * N;
    }
    for (int i = threadIdx.x; i < J * 12; i += 256) s_rt[i] = RT[i];
    __syncthreads();
    const int n = blockIdx.x * 256 + threadIdx.x;
    const bool ok = n < N;
    // the skinning weights first (J <= 24: registers): their loads do not depend on the corner gather and used to follow it one by one
    float wr[24];
    const bool w_regs = J <= 24;
    if (w_regs) {
#pragma unroll
        for (int j = 0; j < 24; j++) wr[j] = (ok && j < J) ? w[(size_t)j * N + n] : 0.f;
    }
    float g[3] = {0.f, 0.f, 0.f};
    if (ok) {
        if (csr_off) {
            // 8 corners in flight per trip: a pole-like vertex with ~100 incident faces would otherwise
            // serialise 100 dependent gathers.  Fixed order -> reproducible sums.
            const int b = csr_off[n], e = csr_off[n + 1];
            for (int k0 = b; k0 < e; k0 += 8) {
                float c[8][3];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int ci = csr_idx[min(k0 + u, e - 1)];
                    c[u][0] = d_corner[3 * (size_t)ci];
                    c[u][1] = d_corner[3 * (size_t)ci + 1];
                    c[u][2] = d_corner[3 * (size_t)ci + 2];
                }
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    if (k0 + u < e) { g[0] += c[u][0]; g[1] += c[u][1]; g[2] += c[u][2]; }
                }
            }
        }
        if (d_extra) {
            g[0] += d_extra[n];
            g[1] += d_extra[(size_t)N + n];
            g[2] += d_extra[2 * (size_t)N + n];
        }
        if (d_obs) {
            d_obs[n] = g[0];
            d_obs[(size_t)N + n] = g[1];
            d_obs[2 * (size_t)N + n] = g[2];
        }
    }
    float ox = 0.f, oy = 0.f, oz = 0.f;
    if (w_regs) {
#pragma unroll
        for (int j = 0; j < 24; j++) {
            const float wj = wr[j];
            if (wj != 0.f) {
                const float *m = s_rt + 12 * j;
                ox += (m[0] * g[0] + m[3] * g[1] + m[6] * g[2]) * wj;
                oy += (m[1] * g[0] + m[4] * g[1] + m[7] * g[2]) * wj;
                oz += (m[2] * g[0] + m[5] * g[1] + m[8] * g[2]) * wj;
            }
        }
    } else {
        for (int j = 0; j < J; j++) {
            const float wj = ok ? w[(size_t)j * N + n] : 0.f;
            if (wj != 0.f) {
                const float *m = s_rt + 12 * j;
                ox += (m[0] * g[0] + m[3] * g[1] + m[6] * g[2]) * wj;
                oy += (m[1] * g[0] + m[4] * g[1] + m[7] * g[2]) * wj;
                oz += (m[2] * g[0] + m[5] * g[1] + m[8] * g[2]) * wj;
            }
        }
    }
    if (ok) {
        d_xyz[n] = ox;
        d_xyz[(size_t)N + n] = oy;
        d_xyz[2 * (size_t)N + n] = oz;
    }
}

// Pose gradient dRT [J][12] = sum_n w_jn (g_n x_n^T | g_n), g_n = gathered gradient of the posed vertex.  One workgroup per
// (joint, frame): thread t owns the vertices t, t + 256, ... (fixed), the 256 partial sums are folded by a shuffle butterfly and
// the four waves in wave order.  No float atomics: bitwise reproducible like the rest of the backward (the first version
// added per-wave sums with atomicAdd in arrival order).  Most joints weigh a few hundred vertices (top-4 skinning weights):
// the row of w is streamed, only its non-zeros gather.
__global__ void __launch_bounds__(256) k_pose_grad(int N, int J, const float *__restrict__ xyz, const float *__restrict__ w,
                                                   const int32_t *__restrict__ csr_off, const int32_t *__restrict__ csr_idx,
                                                   const float *__restrict__ d_corner, const float *__restrict__ d_extra,
                                                   float *__restrict__ dRT, int F) {
    __shared__ float s_part[4][12];
    const int j = blockIdx.x;
    {
        const size_t fr = blockIdx.y;
        if (d_corner) d_corner += fr * 9 * F;
        if (d_extra) d_extra += fr * 3 * N;
        dRT += fr * J * 12;
    }
    float acc[12];
#pragma unroll
    for (int q = 0; q < 12; q++) acc[q] = 0.f;
    for (int n = threadIdx.x; n < N; n += 256) {
        const float wj = w[(size_t)j * N + n];
        if (wj == 0.f) continue;
        float g[3] = {0.f, 0.f, 0.f};
        if (csr_off) {   // same corner order as k_vertex_bwd
            const int b = csr_off[n], e = csr_off[n + 1];
            for (int k = b; k < e; k++) {
                const int ci = csr_idx[k];
                g[0] += d_corner[3 * (size_t)ci]; g[1] += d_corner[3 * (size_t)ci + 1]; g[2] += d_corner[3 * (size_t)ci + 2];
            }
        }
        if (d_extra) { g[0] += d_extra[n]; g[1] += d_extra[(size_t)N + n]; g[2] += d_extra[2 * (size_t)N + n]; }
        const float x = xyz[n], y = xyz[(size_t)N + n], z = xyz[2 * (size_t)N + n];
#pragma unroll
        for (int r = 0; r < 3; r++) {
            const float wg = wj * g[r];
            acc[3 * r] += wg * x; acc[3 * r + 1] += wg * y; acc[3 * r + 2] += wg * z;
            acc[9 + r] += wg;
        }
    }
#pragma unroll
    for (int q = 0; q < 12; q++) acc[q] = wave_sum(acc[q]);
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int q = 0; q < 12; q++) s_part[threadIdx.x >> 6][q] = acc[q];
    }
    __syncthreads();
    if (threadIdx.x < 12) dRT[12 * j + threadIdx.x] = ((s_part[0][threadIdx.x] + s_part[1][threadIdx.x]) + s_part[2][threadIdx.x]) + s_part[3][threadIdx.x];
}

// dst_k[i] = sum_b src_k[b * n_k + i] for up to four tensors in one launch, frames in a fixed order
// (bitwise reproducible)
struct SumFramesArgs {
    size_t n[4];
    const float *src[4];
    float *dst[4];
};
__global__ void __launch_bounds__(256) k_sum_frames(int B, SumFramesArgs a) {
    const size_t total = a.n[0] + a.n[1] + a.n[2] + a.n[3];
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        size_t j = i;
        int k = 0;
        while (j >= a.n[k]) { j -= a.n[k]; k++; }
        const float *src = a.src[k];
        float acc = 0.f;
        for (int b0 = 0; b0 < B; b0 += 8) {   // eight frames' loads in flight, added in frame order
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = b0 + u < B ? src[(size_t)(b0 + u) * a.n[k] + j] : 0.f;
#pragma unroll
            for (int u = 0; u < 8; u++)
                if (b0 + u < B) acc += v[u];
        }
        a.dst[k][j] = acc;
    }
}

// ... and over the frames of SEVERAL launch sequences (gom_split_forward_backward): frame b of tensor k at src[k][b].  Same order, same adds.
struct SumMultiArgs {
    size_t n[4];
    const float *src[4][GOM_SPLIT_MAX_FRAMES];
    float *dst[4];
};
__global__ void __launch_bounds__(256) k_sum_frames_multi(int B, SumMultiArgs a) {
    const size_t total = a.n[0] + a.n[1] + a.n[2] + a.n[3];
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        size_t j = i;
        int k = 0;
        while (j >= a.n[k]) { j -= a.n[k]; k++; }
        float acc = 0.f;
        for (int b0 = 0; b0 < B; b0 += 8) {   // eight frames' loads in flight, added in frame order
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = b0 + u < B ? a.src[k][b0 + u][j] : 0.f;
#pragma unroll
            for (int u = 0; u < 8; u++)
                if (b0 + u < B) acc += v[u];
        }
        a.dst[k][j] = acc;
    }
}

}  // namespace

int gom_sum_frames_multi(int B, const size_t n[4], const float *const src[4][GOM_SPLIT_MAX_FRAMES], float *const dst[4], void *stream) {
    if (B < 1 || B > GOM_SPLIT_MAX_FRAMES) { gom_set_error("gom_sum_frames_multi: 1..%d frames", GOM_SPLIT_MAX_FRAMES); return -1; }
    SumMultiArgs a;
    size_t total = 0;
    for (int k = 0; k < 4; k++) {
        a.n[k] = n[k]; a.dst[k] = dst[k]; total += n[k];
        for (int b = 0; b < GOM_SPLIT_MAX_FRAMES; b++) a.src[k][b] = b < B ? src[k][b] : nullptr;
    }
    if (total == 0) return 0;
    const size_t blocks = (total + 255) / 256;
    hipLaunchKernelGGL(k_sum_frames_multi, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, (hipStream_t)stream, B, a);
    GOM_LAUNCH_CHECK();
    return 0;
}

int gom_sum_frames4(int B, size_t n0, const float *s0, float *d0, size_t n1, const float *s1, float *d1, size_t n2, const float *s2,
                    float *d2, size_t n3, const float *s3, float *d3, void *stream) {
    SumFramesArgs a = {{n0, n1, n2, n3}, {s0, s1, s2, s3}, {d0, d1, d2, d3}};
    const size_t total = n0 + n1 + n2 + n3;
    if (total == 0) return 0;
    const size_t blocks = (total + 255) / 256;
    hipLaunchKernelGGL(k_sum_frames, dim3((unsigned)(blocks < 2048 ? blocks : 2048))   /* one resident round of workgroups; the kernel strides */, dim3(256), 0, (hipStream_t)stream, B, a);
    GOM_LAUNCH_CHECK();
    return 0;
}

int gom_fk_forward_batch(int B, const float *cnl_gtfms, const float *dst_Rs, const float *dst_Ts, float *RT, float *fk_save, void *stream) {
    if (!cnl_gtfms || !dst_Rs || !dst_Ts || !RT || !fk_save) { gom_set_error("gom_fk_forward: null pointer"); return -1; }
    hipLaunchKernelGGL(k_fk_fwd, dim3(B), dim3(64), 0, (hipStream_t)stream, cnl_gtfms, dst_Rs, dst_Ts, RT, fk_save);
    GOM_LAUNCH_CHECK();
    return 0;
}

extern "C" int gom_fk_forward(const float *cnl_gtfms, const float *dst_Rs, const float *dst_Ts, float *RT, float *fk_save, void *stream) {
    return gom_fk_forward_batch(1, cnl_gtfms, dst_Rs, dst_Ts, RT, fk_save, stream);
}

extern "C" int gom_fk_backward(const float *dst_Rs, const float *dst_Ts, const float *fk_save, const float *dRT, float *d_dst_Rs,
                               float *d_dst_Ts, void *stream) {
    if (!dst_Rs || !dst_Ts || !fk_save || !dRT || !d_dst_Rs || !d_dst_Ts) { gom_set_error("gom_fk_backward: null pointer"); return -1; }
    hipLaunchKernelGGL(k_fk_bwd, dim3(1), dim3(64), 0, (hipStream_t)stream, dst_Rs, dst_Ts, fk_save, dRT, d_dst_Rs, d_dst_Ts);
    GOM_LAUNCH_CHECK();
    return 0;
}

int gom_fk_lbs_forward_batch(int B, int N, const float *cnl_gtfms, const float *dst_Rs, const float *dst_Ts, const float *xyz, const float *weights,
                             float *RT, float *fk_save, float *out, void *stream, const GomCamera *cam1, const GomCamera *cams, uint32_t *vdepth_minmax) {
    if (!cnl_gtfms || !dst_Rs || !dst_Ts || !RT || !fk_save || !xyz || !weights || !out) { gom_set_error("gom_fk_lbs_forward: null pointer"); return -1; }
    if (N <= 0) return gom_fk_forward_batch(B, cnl_gtfms, dst_Rs, dst_Ts, RT, fk_save, stream);
    hipLaunchKernelGGL(k_fk_lbs_fwd, dim3((N + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, N, cnl_gtfms, dst_Rs, dst_Ts, xyz, weights, RT, fk_save, out,
                       cam1 ? *cam1 : GomCamera{}, cams, cam1 ? vdepth_minmax : nullptr);
    GOM_LAUNCH_CHECK();
    return 0;
}

int gom_lbs_forward_batch(int B, int N, int J, const float *xyz, const float *weights, const float *RT, float *out, void *stream) {
    if (N < 0 || J <= 0 || J > 64) { gom_set_error("gom_lbs_forward: bad sizes N=%d J=%d", N, J); return -1; }
    if (N == 0) return 0;
    if (!xyz || !weights || !RT || !out) { gom_set_error("gom_lbs_forward: null pointer"); return -1; }
    hipLaunchKernelGGL(k_lbs_fwd, dim3((N + 255) / 256, B), dim3(256), J * 12 * sizeof(float), (hipStream_t)stream, N, J, xyz, weights, RT, out);
    GOM_LAUNCH_CHECK();
    return 0;
}

extern "C" int gom_lbs_forward(int N, int J, const float *xyz, const float *weights, const float *RT, float *out, void *stream) {
    return gom_lbs_forward_batch(1, N, J, xyz, weights, RT, out, stream);
}

// get_global_RTs + apply_lbs in ONE launch for a frame (k_fk_lbs_fwd: every workgroup runs the 24-joint chain itself; the bits of gom_fk_forward + gom_lbs_forward)
extern "C" int gom_fk_lbs_forward(int N, const float *cnl_gtfms, const float *dst_Rs, const float *dst_Ts, const float *xyz, const float *weights, float *RT,
                                  float *fk_save, float *out, void *stream) {
    if (N <= 0) { gom_set_error("gom_fk_lbs_forward: bad size N=%d", N); return -1; }
    return gom_fk_lbs_forward_batch(1, N, cnl_gtfms, dst_Rs, dst_Ts, xyz, weights, RT, fk_save, out, stream);
}

int gom_face_forward_batch(int B, int N, int F, const float *verts, const int32_t *faces, const float *so3, const float *scale,
                           float sigma, float *xyz, float *cov6, const float *appearance, float *feat4, void *stream) {
    if (N < 0 || F < 0) { gom_set_error("gom_face_forward: bad sizes"); return -1; }
    if (F == 0) return 0;
    if (!verts || !faces || !so3 || !scale || !xyz || !cov6) { gom_set_error("gom_face_forward: null pointer"); return -1; }
    if ((appearance == nullptr) != (feat4 == nullptr)) { gom_set_error("gom_face_forward: appearance and feat4 go together"); return -1; }
    hipLaunchKernelGGL(k_face_fwd, dim3((F + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, N, F, verts, faces, so3, scale, sigma, xyz, cov6,
                       appearance, feat4);
    GOM_LAUNCH_CHECK();
    return 0;
}

extern "C" int gom_face_forward(int N, int F, const float *verts, const int32_t *faces, const float *so3, const float *scale,
                                float sigma, float *xyz, float *cov6, const float *appearance, float *feat4, void *stream) {
    return gom_face_forward_batch(1, N, F, verts, faces, so3, scale, sigma, xyz, cov6, appearance, feat4, stream);
}

int gom_face_backward_batch(int B, int N, int F, const float *verts, const int32_t *faces, const float *so3, const float *scale,
                            float sigma, const float *d_xyz, const float *d_cov6, float *d_corner, float *d_so3,
                            float *d_scale, const float *d_feat4, float *d_appearance, void *stream) {
    if (N < 0 || F < 0) { gom_set_error("gom_face_backward: bad sizes"); return -1; }
    if (F == 0) return 0;
    if (!verts || !faces || !so3 || !scale || !d_xyz || !d_cov6 || !d_corner || !d_so3 || !d_scale) { gom_set_error("gom_face_backward: null pointer"); return -1; }
    if ((d_feat4 == nullptr) != (d_appearance == nullptr)) { gom_set_error("gom_face_backward: d_feat4 and d_appearance go together"); return -1; }
    hipLaunchKernelGGL(k_face_bwd, dim3((F + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, N, F, verts, faces, so3, scale, sigma,
                       d_xyz, d_cov6, d_corner, d_so3, d_scale, d_feat4, d_appearance);
    GOM_LAUNCH_CHECK();
    return 0;
}

extern "C" int gom_face_backward(int N, int F, const float *verts, const int32_t *faces, const float *so3, const float *scale,
                                 float sigma, const float *d_xyz, const float *d_cov6, float *d_corner, float *d_so3,
                                 float *d_scale, const float *d_feat4, float *d_appearance, void *stream) {
    return gom_face_backward_batch(1, N, F, verts, faces, so3, scale, sigma, d_xyz, d_cov6, d_corner, d_so3, d_scale, d_feat4,
                                   d_appearance, stream);
}

int gom_vertex_backward_batch(int B, int F, int N, int J, const float *xyz, const float *weights, const float *RT, const int32_t *csr_off,
                              const int32_t *csr_idx, const float *d_corner, const float *d_verts_extra, float *d_verts_obs,
                              float *d_xyz, float *dRT, void *stream) {
    if (N < 0 || J <= 0 || J > 64) { gom_set_error("gom_vertex_backward: bad sizes"); return -1; }
    if (N == 0) return 0;
    if (!xyz || !weights || !RT || !d_xyz) { gom_set_error("gom_vertex_backward: null pointer"); return -1; }
    if ((csr_off == nullptr) != (csr_idx == nullptr) || (csr_off && !d_corner)) { gom_set_error("gom_vertex_backward: inconsistent CSR arguments"); return -1; }
    hipLaunchKernelGGL(k_vertex_bwd, dim3((N + 255) / 256, B), dim3(256), J * 12 * sizeof(float), (hipStream_t)stream, N, J, xyz, weights,
                       RT, csr_off, csr_idx, d_corner, d_verts_extra, d_verts_obs, d_xyz, F);
    GOM_LAUNCH_CHECK();
    if (dRT) {   // pose gradient (pose refinement / test-time pose optimisation only): deterministic per-joint reduction
        hipLaunchKernelGGL(k_pose_grad, dim3(J, B), dim3(256), 0, (hipStream_t)stream, N, J, xyz, weights, csr_off, csr_idx, d_corner,
                           d_verts_extra, dRT, F);
        GOM_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int gom_vertex_backward(int N, int J, const float *xyz, const float *weights, const float *RT, const int32_t *csr_off,
                                   const int32_t *csr_idx, const float *d_corner, const float *d_verts_extra, float *d_verts_obs,
                                   float *d_xyz, float *dRT, void *stream) {
    return gom_vertex_backward_batch(1, 0, N, J, xyz, weights, RT, csr_off, csr_idx, d_corner, d_verts_extra, d_verts_obs, d_xyz, dRT,
                                     stream);
}
