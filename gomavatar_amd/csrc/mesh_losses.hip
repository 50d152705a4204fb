// Mesh regularisers of the training loss (train.py:123-160): uniform Laplacian smoothing and normal consistency
// (PyTorch3D mesh_laplacian_smoothing(method="uniform") / mesh_normal_consistency on a closed manifold mesh) and the
// colour consistency of edge-adjacent faces (utils/network_util.py:795-799).  Each is a value + gradient pair of small
// per-vertex / per-edge kernels over precomputed CSR adjacency: no index_add scatters, no atomics, fixed summation order.
#include "gom_internal.h"

namespace {

__device__ __forceinline__ float block_sum(float v, float *s_red) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
}
__device__ __forceinline__ void cross3(const float *a, const float *b, float *o) {
    o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}

// ---- uniform Laplacian: loss = (1/N) sum_i || mean_{j in N(i)} v_j - v_i ||^2  (the reference's own copy of the PyTorch3D loss,
// utils/network_util.py:782,789,792: L.mm(V), norm(dim=1) ** 2, un-weighted mean) -----------------------------------------------------
__global__ void __launch_bounds__(256) k_lap_fwd(int N, const float *__restrict__ verts, const int32_t *__restrict__ nbr_off,
                                                 const int32_t *__restrict__ nbr_idx, float *__restrict__ dir, float *__restrict__ partials) {
    __shared__ float s_red[4];
    float acc = 0.f;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < N; i += gridDim.x * 256) {
        const int b = nbr_off[i], e = nbr_off[i + 1];
        float s[3] = {0.f, 0.f, 0.f};
        for (int k0 = b; k0 < e; k0 += 8) {   // 8 neighbours in flight per trip, summed in list order
            float r[8][3];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const float *q = verts + 3 * (size_t)nbr_idx[min(k0 + u, e - 1)];
                r[u][0] = q[0]; r[u][1] = q[1]; r[u][2] = q[2];
            }
#pragma unroll
            for (int u = 0; u < 8; u++)
                if (k0 + u < e) { s[0] += r[u][0]; s[1] += r[u][1]; s[2] += r[u][2]; }
        }
        const float inv = e > b ? 1.f / (float)(e - b) : 0.f;
        const float l[3] = {s[0] * inv - verts[3 * (size_t)i], s[1] * inv - verts[3 * (size_t)i + 1], s[2] * inv - verts[3 * (size_t)i + 2]};
        acc += l[0] * l[0] + l[1] * l[1] + l[2] * l[2];
        // d||l||^2 / dl = 2 l
        dir[3 * (size_t)i] = 2.f * l[0]; dir[3 * (size_t)i + 1] = 2.f * l[1]; dir[3 * (size_t)i + 2] = 2.f * l[2];
    }
    const float tot = block_sum(acc, s_red);
    if (threadIdx.x == 0) partials[blockIdx.x] = tot / (float)N;
}
// d loss / d v_i = (1/N) ( -dir_i + sum_{j in N(i)} dir_j / deg_j )       (the Laplacian matrix is a constant of the topology)
__global__ void __launch_bounds__(256) k_lap_bwd(int N, const float *__restrict__ dir, const int32_t *__restrict__ nbr_off,
                                                 const int32_t *__restrict__ nbr_idx, const float *__restrict__ grad_out, float *__restrict__ d_verts) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    float g[3] = {-dir[3 * (size_t)i], -dir[3 * (size_t)i + 1], -dir[3 * (size_t)i + 2]};
    const int kb = nbr_off[i], ke = nbr_off[i + 1];
    for (int k0 = kb; k0 < ke; k0 += 8) {   // 8 neighbours in flight per trip, summed in list order
        float r[8][3], w[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int j = nbr_idx[min(k0 + u, ke - 1)];
            w[u] = 1.f / (float)(nbr_off[j + 1] - nbr_off[j]);
            r[u][0] = dir[3 * (size_t)j]; r[u][1] = dir[3 * (size_t)j + 1]; r[u][2] = dir[3 * (size_t)j + 2];
        }
#pragma unroll
        for (int u = 0; u < 8; u++)
            if (k0 + u < ke) { g[0] += r[u][0] * w[u]; g[1] += r[u][1] * w[u]; g[2] += r[u][2] * w[u]; }
    }
    const float sc = grad_out[0] / (float)N;
    d_verts[3 * (size_t)i] = g[0] * sc; d_verts[3 * (size_t)i + 1] = g[1] * sc; d_verts[3 * (size_t)i + 2] = g[2] * sc;
}

// ---- normal consistency: loss = mean over edge-adjacent face pairs of 1 - cos(n_a, n_b) -------------------------------------------
__device__ __forceinline__ void face_cross(const float *verts, const int32_t *faces, int f, float *c) {
    const int i0 = faces[3 * f], i1 = faces[3 * f + 1], i2 = faces[3 * f + 2];
    float e1[3], e2[3];
#pragma unroll
    for (int d = 0; d < 3; d++) { e1[d] = verts[3 * (size_t)i1 + d] - verts[3 * (size_t)i0 + d]; e2[d] = verts[3 * (size_t)i2 + d] - verts[3 * (size_t)i0 + d]; }
    cross3(e1, e2, c);
}
// value + per-(pair, side) gradient w.r.t. the two un-normalised face normals: pair_grad [P][2][3]
__global__ void __launch_bounds__(256) k_ncons_fwd(int P, const int32_t *__restrict__ pairs, const float *__restrict__ verts,
                                                   const int32_t *__restrict__ faces, float *__restrict__ pair_grad, float *__restrict__ partials) {
    __shared__ float s_red[4];
    float acc = 0.f;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < P; p += gridDim.x * 256) {
        float a[3], b[3];
        face_cross(verts, faces, pairs[2 * p], a);
        face_cross(verts, faces, pairs[2 * p + 1], b);
        // cosine_similarity of the two (un-normalised) normals with torch 1.13's eps rule: a.b / sqrt(max(|a|^2 |b|^2, eps^2)), eps = 1e-8.
        // On a consistently oriented mesh the face normals are PyTorch3D's (n0, -n1) of the shared edge.
        const float aa = a[0] * a[0] + a[1] * a[1] + a[2] * a[2], bb = b[0] * b[0] + b[1] * b[1] + b[2] * b[2];
        const float ab = a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
        const float w = aa * bb;
        const bool clamped = !(w > 1e-16f);
        const float inv = 1.f / sqrtf(clamped ? 1e-16f : w);
        const float c = ab * inv;
        acc += 1.f - c;
        // d(1 - cos)/da = -(b inv - a (a.b) bb inv^3)   (the second term vanishes where the denominator is the clamped constant)
        const float ka = clamped ? 0.f : ab * bb * inv * inv * inv, kb = clamped ? 0.f : ab * aa * inv * inv * inv;
#pragma unroll
        for (int d = 0; d < 3; d++) {
            pair_grad[6 * (size_t)p + d] = -(b[d] * inv - a[d] * ka);
            pair_grad[6 * (size_t)p + 3 + d] = -(a[d] * inv - b[d] * kb);
        }
    }
    const float tot = block_sum(acc, s_red);
    if (threadIdx.x == 0) partials[blockIdx.x] = tot / (float)(P > 0 ? P : 1);
}
// per face: sum the gradients of its pairs (fp_off / fp_idx: face -> pair*2 + side), then through n = (v1-v0) x (v2-v0)
__global__ void __launch_bounds__(256) k_ncons_bwd_face(int F, int P, const int32_t *__restrict__ fp_off, const int32_t *__restrict__ fp_idx,
                                                        const float *__restrict__ pair_grad, const float *__restrict__ verts,
                                                        const int32_t *__restrict__ faces, const float *__restrict__ grad_out, float *__restrict__ d_corner) {
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f >= F) return;
    float g[3] = {0.f, 0.f, 0.f};
    for (int k = fp_off[f]; k < fp_off[f + 1]; k++) {
        const float *r = pair_grad + 3 * (size_t)fp_idx[k];
        g[0] += r[0]; g[1] += r[1]; g[2] += r[2];
    }
    const float sc = grad_out[0] / (float)(P > 0 ? P : 1);
    g[0] *= sc; g[1] *= sc; g[2] *= sc;
    const int i0 = faces[3 * f], i1 = faces[3 * f + 1], i2 = faces[3 * f + 2];
    float e1[3], e2[3], d1[3], d2[3];
#pragma unroll
    for (int d = 0; d < 3; d++) { e1[d] = verts[3 * (size_t)i1 + d] - verts[3 * (size_t)i0 + d]; e2[d] = verts[3 * (size_t)i2 + d] - verts[3 * (size_t)i0 + d]; }
    cross3(e2, g, d1);
    cross3(g, e1, d2);
#pragma unroll
    for (int d = 0; d < 3; d++) {
        d_corner[9 * (size_t)f + d] = -(d1[d] + d2[d]);
        d_corner[9 * (size_t)f + 3 + d] = d1[d];
        d_corner[9 * (size_t)f + 6 + d] = d2[d];
    }
}
__global__ void __launch_bounds__(256) k_corner_gather2(int N, const int32_t *__restrict__ csr_off, const int32_t *__restrict__ csr_idx,
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

// ---- colour consistency: loss = mean |c_a - c_b| over pairs and channels; colours in the (3, F) parameter layout ------------------
__global__ void __launch_bounds__(256) k_ccons_fwd(int P, int F, const int32_t *__restrict__ pairs, const float *__restrict__ colors,
                                                   float *__restrict__ pair_sign, float *__restrict__ partials) {
    __shared__ float s_red[4];
    float acc = 0.f;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < P; p += gridDim.x * 256) {
        const int a = pairs[2 * p], b = pairs[2 * p + 1];
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
            const float d = colors[(size_t)ch * F + a] - colors[(size_t)ch * F + b];
            acc += fabsf(d);
            pair_sign[3 * (size_t)p + ch] = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
        }
    }
    const float tot = block_sum(acc, s_red);
    if (threadIdx.x == 0) partials[blockIdx.x] = tot / (3.f * (float)(P > 0 ? P : 1));
}
__global__ void __launch_bounds__(256) k_ccons_bwd(int F, int P, const int32_t *__restrict__ fp_off, const int32_t *__restrict__ fp_idx,
                                                   const float *__restrict__ pair_sign, const float *__restrict__ grad_out, float *__restrict__ d_colors) {
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f >= F) return;
    float g[3] = {0.f, 0.f, 0.f};
    for (int k = fp_off[f]; k < fp_off[f + 1]; k++) {
        const int ps = fp_idx[k], p = ps >> 1;
        const float sgn = (ps & 1) ? -1.f : 1.f;   // side 1 enters as -c_b
#pragma unroll
        for (int ch = 0; ch < 3; ch++) g[ch] += sgn * pair_sign[3 * (size_t)p + ch];
    }
    const float sc = grad_out[0] / (3.f * (float)(P > 0 ? P : 1));
#pragma unroll
    for (int ch = 0; ch < 3; ch++) d_colors[(size_t)ch * F + f] = g[ch] * sc;
}

}  // namespace

extern "C" int gom_mesh_laplacian(int N, const float *verts, const int32_t *nbr_off, const int32_t *nbr_idx, float *dir, float *partials, void *stream) {
    if (N <= 0 || !verts || !nbr_off || !nbr_idx || !dir || !partials) { gom_set_error("gom_mesh_laplacian: bad arguments"); return -1; }
    hipLaunchKernelGGL(k_lap_fwd, dim3(GOM_LOSS_BLOCKS), dim3(256), 0, (hipStream_t)stream, N, verts, nbr_off, nbr_idx, dir, partials);
    GOM_LAUNCH_CHECK();
    return 0;
}
extern "C" int gom_mesh_laplacian_backward(int N, const float *dir, const int32_t *nbr_off, const int32_t *nbr_idx, const float *grad_out, float *d_verts,
                                           void *stream) {
    if (N <= 0 || !dir || !nbr_off || !nbr_idx || !grad_out || !d_verts) { gom_set_error("gom_mesh_laplacian_backward: bad arguments"); return -1; }
    hipLaunchKernelGGL(k_lap_bwd, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, N, dir, nbr_off, nbr_idx, grad_out, d_verts);
    GOM_LAUNCH_CHECK();
    return 0;
}
extern "C" int gom_mesh_normal_consistency(int P, const int32_t *pairs, const float *verts, const int32_t *faces, float *pair_grad, float *partials,
                                           void *stream) {
    if (P < 0 || !pairs || !verts || !faces || !pair_grad || !partials) { gom_set_error("gom_mesh_normal_consistency: bad arguments"); return -1; }
    hipLaunchKernelGGL(k_ncons_fwd, dim3(GOM_LOSS_BLOCKS), dim3(256), 0, (hipStream_t)stream, P, pairs, verts, faces, pair_grad, partials);
    GOM_LAUNCH_CHECK();
    return 0;
}
extern "C" int gom_mesh_normal_consistency_backward(int N, int F, int P, const int32_t *fp_off, const int32_t *fp_idx, const float *pair_grad,
                                                    const float *verts, const int32_t *faces, const int32_t *csr_off, const int32_t *csr_idx,
                                                    const float *grad_out, float *d_corner_scratch, float *d_verts, void *stream) {
    if (N <= 0 || F <= 0 || !fp_off || !fp_idx || !pair_grad || !verts || !faces || !csr_off || !csr_idx || !grad_out || !d_corner_scratch || !d_verts) {
        gom_set_error("gom_mesh_normal_consistency_backward: bad arguments");
        return -1;
    }
    hipLaunchKernelGGL(k_ncons_bwd_face, dim3((F + 255) / 256), dim3(256), 0, (hipStream_t)stream, F, P, fp_off, fp_idx, pair_grad, verts, faces, grad_out,
                       d_corner_scratch);
    GOM_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_corner_gather2, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, N, csr_off, csr_idx, d_corner_scratch, d_verts);
    GOM_LAUNCH_CHECK();
    return 0;
}
extern "C" int gom_mesh_color_consistency(int P, int F, const int32_t *pairs, const float *colors, float *pair_sign, float *partials, void *stream) {
    if (P < 0 || F <= 0 || !pairs || !colors || !pair_sign || !partials) { gom_set_error("gom_mesh_color_consistency: bad arguments"); return -1; }
    hipLaunchKernelGGL(k_ccons_fwd, dim3(GOM_LOSS_BLOCKS), dim3(256), 0, (hipStream_t)stream, P, F, pairs, colors, pair_sign, partials);
    GOM_LAUNCH_CHECK();
    return 0;
}
extern "C" int gom_mesh_color_consistency_backward(int F, int P, const int32_t *fp_off, const int32_t *fp_idx, const float *pair_sign, const float *grad_out,
                                                   float *d_colors, void *stream) {
    if (F <= 0 || !fp_off || !fp_idx || !pair_sign || !grad_out || !d_colors) { gom_set_error("gom_mesh_color_consistency_backward: bad arguments"); return -1; }
    hipLaunchKernelGGL(k_ccons_bwd, dim3((F + 255) / 256), dim3(256), 0, (hipStream_t)stream, F, P, fp_off, fp_idx, pair_sign, grad_out, d_colors);
    GOM_LAUNCH_CHECK();
    return 0;
}
