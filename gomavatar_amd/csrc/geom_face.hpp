// Per-face Gaussian frame (Steiner ellipse of the posed triangle -> mean, covariance factor M) and its backward, as device
// functions: used by the stand-alone geometry kernels (geom.hip) and by the rasterizer's fused per-Gaussian kernels
// (raster_pre.hip: k_preprocess / k_preprocess_bwd with FACE = true), which run them in the same thread as the projection.
// Reference: models/model.py:232-262 (triangle -> canonical Gaussian frame), gaussian.py:24-47 (covariance).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gom_face {

struct FaceFwd {
    float v0[3], v1[3], v2[3];
    float f1[3], f2[3], cs, sn, p, q;
    float a0[3], a1[3], n[3], nn;
    float A[3][3];    // columns 2a0, 2a1, sigma*n/|n|
    float R[3][3], K[3][3], K2[3][3], th, th2, fac1, fac2, sin_th, cos_th;
    bool clamped;
    float B[3][3], M[3][3];
};

// What face_forward reads from memory (15 floats behind one dependent index load): a kernel with a long gather of its own in front of
// the face arithmetic (k_preprocess_bwd) issues these loads first and computes later.
struct FaceIn {
    float v0[3], v1[3], v2[3], w[3], s[3];
};

__device__ __forceinline__ void face_load(const float *verts, int N, const int32_t *faces, const float *so3, const float *scale, int F, int f, FaceIn &in) {
    const int i0 = faces[3 * f], i1 = faces[3 * f + 1], i2 = faces[3 * f + 2];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        in.v0[k] = verts[(size_t)k * N + i0];
        in.v1[k] = verts[(size_t)k * N + i1];
        in.v2[k] = verts[(size_t)k * N + i2];
        in.w[k] = so3[(size_t)k * F + f];
        in.s[k] = scale[(size_t)k * F + f];
    }
}

__device__ __forceinline__ void face_forward_in(const FaceIn &in, float sigma, FaceFwd &o, float *xyz3, float *s3) {
#pragma clang fp contract(on)   // fused multiply-adds as the expressions spell them, whichever kernel this is inlined into: same bits in all of them
#pragma unroll
    for (int k = 0; k < 3; k++) {
        o.v0[k] = in.v0[k];
        o.v1[k] = in.v1[k];
        o.v2[k] = in.v2[k];
    }
    const float K2C = 0.28867513459481287f;  // 1 / (2 sqrt 3)
    float c[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        c[k] = ((o.v0[k] + o.v1[k]) + o.v2[k]) / 3.0f;
        xyz3[k] = c[k];
        o.f1[k] = 0.5f * (o.v2[k] - c[k]);
        o.f2[k] = K2C * (o.v1[k] - o.v0[k]);
    }
    o.p = 2.f * o.f1[0] * o.f2[0] + 2.f * o.f1[1] * o.f2[1] + 2.f * o.f1[2] * o.f2[2];
    o.q = (o.f1[0] * o.f1[0] + o.f1[1] * o.f1[1] + o.f1[2] * o.f1[2]) - (o.f2[0] * o.f2[0] + o.f2[1] * o.f2[1] + o.f2[2] * o.f2[2]);
    const float t0 = atan2f(o.p, o.q) * 0.5f;
    o.cs = cosf(t0);
    o.sn = sinf(t0);
#pragma unroll
    for (int k = 0; k < 3; k++) {
        o.a0[k] = o.f1[k] * o.cs + o.f2[k] * o.sn;
        o.a1[k] = o.f2[k] * o.cs - o.f1[k] * o.sn;
    }
    o.n[0] = o.a0[1] * o.a1[2] - o.a0[2] * o.a1[1];
    o.n[1] = o.a0[2] * o.a1[0] - o.a0[0] * o.a1[2];
    o.n[2] = o.a0[0] * o.a1[1] - o.a0[1] * o.a1[0];
    o.nn = fmaxf(sqrtf(o.n[0] * o.n[0] + o.n[1] * o.n[1] + o.n[2] * o.n[2]), 1e-12f);
#pragma unroll
    for (int k = 0; k < 3; k++) {
        o.A[k][0] = 2.f * o.a0[k];
        o.A[k][1] = 2.f * o.a1[k];
        o.A[k][2] = o.n[k] / o.nn * sigma;
    }
    // so3 exponential (PyTorch3D: theta^2 clamped at 1e-4)
    const float wx = in.w[0], wy = in.w[1], wz = in.w[2];
    const float n2 = wx * wx + wy * wy + wz * wz;
    o.clamped = !(n2 > 1e-4f);
    o.th2 = o.clamped ? 1e-4f : n2;
    o.th = sqrtf(o.th2);
    const float inv = 1.0f / o.th;
    o.sin_th = sinf(o.th);
    o.cos_th = cosf(o.th);
    o.fac1 = inv * o.sin_th;
    o.fac2 = inv * inv * (1.0f - o.cos_th);
    const float K[3][3] = {{0.f, -wz, wy}, {wz, 0.f, -wx}, {-wy, wx, 0.f}};
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            o.K[i][j] = K[i][j];
            o.K2[i][j] = K[i][0] * K[0][j] + K[i][1] * K[1][j] + K[i][2] * K[2][j];
        }
    s3[0] = in.s[0]; s3[1] = in.s[1]; s3[2] = in.s[2];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            o.R[i][j] = o.fac1 * o.K[i][j] + o.fac2 * o.K2[i][j] + (i == j ? 1.f : 0.f);
            o.B[i][j] = o.R[i][j] * s3[j];
        }
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) o.M[i][j] = o.A[i][0] * o.B[0][j] + o.A[i][1] * o.B[1][j] + o.A[i][2] * o.B[2][j];
}

__device__ __forceinline__ void face_forward(const float *verts, int N, const int32_t *faces, const float *so3, const float *scale,
                                             int F, int f, float sigma, FaceFwd &o, float *xyz3, float *s3) {
    FaceIn in;
    face_load(verts, N, faces, so3, scale, F, f, in);
    face_forward_in(in, sigma, o, xyz3, s3);
}


// cov = M M^T, upper triangle
__device__ __forceinline__ void face_cov6(const FaceFwd &o, float (&cov6)[6]) {
#pragma clang fp contract(on)
    const float (*M)[3] = o.M;
    cov6[0] = M[0][0] * M[0][0] + M[0][1] * M[0][1] + M[0][2] * M[0][2];
    cov6[1] = M[0][0] * M[1][0] + M[0][1] * M[1][1] + M[0][2] * M[1][2];
    cov6[2] = M[0][0] * M[2][0] + M[0][1] * M[2][1] + M[0][2] * M[2][2];
    cov6[3] = M[1][0] * M[1][0] + M[1][1] * M[1][1] + M[1][2] * M[1][2];
    cov6[4] = M[1][0] * M[2][0] + M[1][1] * M[2][1] + M[1][2] * M[2][2];
    cov6[5] = M[2][0] * M[2][0] + M[2][1] * M[2][1] + M[2][2] * M[2][2];
}

// Backward of the frame: gradients of the mean (d_xyz[3]) and of the covariance's upper triangle (d_cov6[6]) ->
// gradients of the three corners (d_corner[9]: v0 | v1 | v2), of the so3 vector and of the scale.  o, s3: from face_forward.
__device__ __forceinline__ void face_backward(const FaceFwd &o, const float (&s3)[3], float sigma, const float (&d_xyz)[3], const float (&d_cov6)[6],
                                              float (&d_corner)[9], float (&d_so3)[3], float (&d_scale)[3]) {
#pragma clang fp contract(on)
    // dL/dM = 2 Gs M, Gs = symmetric gradient with halved off-diagonals
    const float g0 = d_cov6[0], g1 = d_cov6[1], g2 = d_cov6[2], g3 = d_cov6[3], g4 = d_cov6[4], g5 = d_cov6[5];
    const float Gs2[3][3] = {{2.f * g0, g1, g2}, {g1, 2.f * g3, g4}, {g2, g4, 2.f * g5}};  // = 2*Gs
    float dM[3][3], dA[3][3], dB[3][3];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) dM[i][j] = Gs2[i][0] * o.M[0][j] + Gs2[i][1] * o.M[1][j] + Gs2[i][2] * o.M[2][j];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            dA[i][j] = dM[i][0] * o.B[j][0] + dM[i][1] * o.B[j][1] + dM[i][2] * o.B[j][2];   // dM B^T
            dB[i][j] = o.A[0][i] * dM[0][j] + o.A[1][i] * dM[1][j] + o.A[2][i] * dM[2][j];   // A^T dM
        }
    // scale and rotation
    float dS[3], dR[3][3];
#pragma unroll
    for (int j = 0; j < 3; j++) dS[j] = dB[0][j] * o.R[0][j] + dB[1][j] * o.R[1][j] + dB[2][j] * o.R[2][j];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) dR[i][j] = dB[i][j] * s3[j];
    // so3_exp backward
    float dK[3][3];
    float dfac1 = 0.f, dfac2 = 0.f;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            dfac1 += dR[i][j] * o.K[i][j];
            dfac2 += dR[i][j] * o.K2[i][j];
            // d(K K): dR K^T + K^T dR
            const float t1 = dR[i][0] * o.K[j][0] + dR[i][1] * o.K[j][1] + dR[i][2] * o.K[j][2];
            const float t2 = o.K[0][i] * dR[0][j] + o.K[1][i] * dR[1][j] + o.K[2][i] * dR[2][j];
            dK[i][j] = o.fac1 * dR[i][j] + o.fac2 * (t1 + t2);
        }
    float dw[3] = {dK[2][1] - dK[1][2], dK[0][2] - dK[2][0], dK[1][0] - dK[0][1]};
    if (!o.clamped) {
        const float th = o.th, sn = o.sin_th, cs = o.cos_th;
        const float df1 = (th * cs - sn) / (th * th);
        const float df2 = (th * sn - 2.f * (1.f - cs)) / (th * th * th);
        const float dth = dfac1 * df1 + dfac2 * df2;
        const float wx = o.K[2][1], wy = o.K[0][2], wz = o.K[1][0];
        dw[0] += dth * wx / th;
        dw[1] += dth * wy / th;
        dw[2] += dth * wz / th;
    }
#pragma unroll
    for (int k = 0; k < 3; k++) { d_so3[k] = dw[k]; d_scale[k] = dS[k]; }
    // Steiner frame backward
    float da0[3], da1[3], dnh[3];
#pragma unroll
    for (int k = 0; k < 3; k++) { da0[k] = 2.f * dA[k][0]; da1[k] = 2.f * dA[k][1]; dnh[k] = sigma * dA[k][2]; }
    {
        float nh[3] = {o.n[0] / o.nn, o.n[1] / o.nn, o.n[2] / o.nn};
        const float dot = nh[0] * dnh[0] + nh[1] * dnh[1] + nh[2] * dnh[2];
        float dn[3];
        const bool degenerate = !(o.nn > 1e-12f);
#pragma unroll
        for (int k = 0; k < 3; k++) dn[k] = degenerate ? dnh[k] / o.nn : (dnh[k] - nh[k] * dot) / o.nn;
        // n = a0 x a1: da0 += a1 x dn ; da1 += dn x a0
        da0[0] += o.a1[1] * dn[2] - o.a1[2] * dn[1];
        da0[1] += o.a1[2] * dn[0] - o.a1[0] * dn[2];
        da0[2] += o.a1[0] * dn[1] - o.a1[1] * dn[0];
        da1[0] += dn[1] * o.a0[2] - dn[2] * o.a0[1];
        da1[1] += dn[2] * o.a0[0] - dn[0] * o.a0[2];
        da1[2] += dn[0] * o.a0[1] - dn[1] * o.a0[0];
    }
    float df1[3], df2[3];
    float dt0 = 0.f;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        df1[k] = o.cs * da0[k] - o.sn * da1[k];
        df2[k] = o.sn * da0[k] + o.cs * da1[k];
        dt0 += da0[k] * o.a1[k] - da1[k] * o.a0[k];
    }
    {
        const float den = o.p * o.p + o.q * o.q;
        const float dp = den > 0.f ? 0.5f * dt0 * o.q / den : 0.f;
        const float dq = den > 0.f ? -0.5f * dt0 * o.p / den : 0.f;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            df1[k] += 2.f * o.f2[k] * dp + 2.f * o.f1[k] * dq;
            df2[k] += 2.f * o.f1[k] * dp - 2.f * o.f2[k] * dq;
        }
    }
    const float K2C = 0.28867513459481287f;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float dc = (d_xyz[k] - 0.5f * df1[k]) / 3.0f;
        d_corner[0 + k] = dc - K2C * df2[k];
        d_corner[3 + k] = dc + K2C * df2[k];
        d_corner[6 + k] = dc + 0.5f * df1[k];
    }
}

}  // namespace gom_face
