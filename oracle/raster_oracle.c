/*
 * raster_oracle.c -- CPU restatement of the tile-based differentiable Gaussian
 * splat rasterizer that the reference calls through
 * `diff_gaussian_rasterization.GaussianRasterizer`
 * (call site: /root/reference/models/modules/renderer/gaussian.py:9,20,53-67,83-91).
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under gomavatar_amd/ may import, link or
 * execute this file; it is the checker for the HIP path (tests/, smoke(),
 * bench.py's cpu_baseline leg).
 *
 * PARITY UNPINNED: the algorithm lives in a third-party dependency that is not
 * vendored in /root/reference (graphdeco-inria/diff-gaussian-rasterization,
 * installed from git HEAD, no pin: reference README.md:36) and the reference
 * ships no tests / golden vectors for it.  This file restates the published
 * algorithm as transcribed in SURVEY.md Appendix A (A.1 preprocess, A.2
 * binning, A.3 render forward, A.4 render backward, A.5 per-Gaussian
 * backward); each function cites the appendix item it follows.  What *is*
 * pinned: the backward is checked against fp64 finite differences of this
 * file's own forward (tests/test_oracle_raster.py), and the camera
 * conventions are checked against the reference's call site.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off; REAL=float|double).
 * The floating-point operation ORDER in the per-Gaussian functions is part of
 * the contract: the HIP kernels repeat it (compiled with -ffp-contract=off) so
 * that every integer output (radii, tile rects, tiles_touched, sort keys,
 * sorted lists, tile ranges) is bit-identical.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef REAL
#define REAL float
#endif

#define TILE 16

/* typed libm so that REAL=float never detours through double */
#define OR_CAT_(a, b) a##b
#define OR_CAT(a, b) OR_CAT_(a, b)
#define OR_IS_float 1
#if OR_CAT(OR_IS_, REAL)
#define RSQRT sqrtf
#define RCEIL ceilf
#define REXP expf
#define REXP2 exp2f
#define RLOG2 log2f
#define RFMA fmaf
#define RABS fabsf
#else
#define RSQRT sqrt
#define RCEIL ceil
#define REXP exp
#define REXP2 exp2
#define RLOG2 log2
#define RFMA fma
#define RABS fabs
#endif

typedef struct {
    int H, W;
    REAL tanfovx, tanfovy;
    REAL view[16]; /* row-major of E^T: view[4*r+c] = E[c][r] (gaussian.py:60) */
    REAL proj[16]; /* row-major of (K_ndc E)^T (gaussian.py:61)               */
    REAL bg[4];
} OrCamera;

static int g_threads = 1;
void or_set_threads(int n) { g_threads = n > 0 ? n : 1; }
int or_real_size(void) { return (int)sizeof(REAL); }

static inline REAL rmin(REAL a, REAL b) { return a < b ? a : b; }
static inline REAL rmax(REAL a, REAL b) { return a > b ? a : b; }
static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

/* float -> int with defined behaviour out of range (x86 and gfx950 differ on
 * overflow/NaN); identical helper in the HIP kernels. */
static inline int f2i(REAL x) {
    if (!(x >= (REAL)-1073741824.0)) return -1073741824;
    if (x >= (REAL)1073741824.0) return 1073741824;
    return (int)x;
}

/* App. A preamble: p_view = E [p;1] */
static inline void xform4x3(const REAL *m, const REAL *p, REAL *o) {
    o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
    o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
    o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
}
static inline void xform4x4(const REAL *m, const REAL *p, REAL *o) {
    o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
    o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
    o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
    o[3] = m[3] * p[0] + m[7] * p[1] + m[11] * p[2] + m[15];
}

/* Shared by forward (A.1) and backward (A.5): the two non-zero rows of
 * M = J W (W = rotation of the view matrix), with the 1.3*tanfov clamp. */
typedef struct {
    REAL t[3];           /* clamped view-space point            */
    REAL M0[3], M1[3];   /* rows 0,1 of J W                      */
    REAL xmul, ymul;     /* 0 when the clamp was active, else 1  */
} ProjJac;

static inline void proj_jacobian(const OrCamera *cam, const REAL *mean, REAL fx, REAL fy, ProjJac *o) {
    const REAL *v = cam->view;
    REAL t[3];
    xform4x3(v, mean, t);
    const REAL limx = (REAL)1.3 * cam->tanfovx;
    const REAL limy = (REAL)1.3 * cam->tanfovy;
    const REAL txtz = t[0] / t[2];
    const REAL tytz = t[1] / t[2];
    o->xmul = (txtz < -limx || txtz > limx) ? (REAL)0 : (REAL)1;
    o->ymul = (tytz < -limy || tytz > limy) ? (REAL)0 : (REAL)1;
    t[0] = rmin(limx, rmax(-limx, txtz)) * t[2];
    t[1] = rmin(limy, rmax(-limy, tytz)) * t[2];
    const REAL J00 = fx / t[2];
    const REAL J02 = -(fx * t[0]) / (t[2] * t[2]);
    const REAL J11 = fy / t[2];
    const REAL J12 = -(fy * t[1]) / (t[2] * t[2]);
    /* W[i][k] = E[i][k] = view[4*k + i] */
    for (int k = 0; k < 3; k++) {
        o->M0[k] = J00 * v[4 * k + 0] + J02 * v[4 * k + 2];
        o->M1[k] = J11 * v[4 * k + 1] + J12 * v[4 * k + 2];
    }
    o->t[0] = t[0]; o->t[1] = t[1]; o->t[2] = t[2];
}

/* cov2D = M Sigma M^T (+0.3 on the diagonal): returns a,b,c and Sigma*M0, Sigma*M1 */
static inline void cov2d_from(const REAL *c6, const ProjJac *pj, REAL *a, REAL *b, REAL *c, REAL *SM0, REAL *SM1) {
    const REAL S[3][3] = {{c6[0], c6[1], c6[2]}, {c6[1], c6[3], c6[4]}, {c6[2], c6[4], c6[5]}};
    for (int k = 0; k < 3; k++) {
        SM0[k] = S[k][0] * pj->M0[0] + S[k][1] * pj->M0[1] + S[k][2] * pj->M0[2];
        SM1[k] = S[k][0] * pj->M1[0] + S[k][1] * pj->M1[1] + S[k][2] * pj->M1[2];
    }
    *a = (pj->M0[0] * SM0[0] + pj->M0[1] * SM0[1] + pj->M0[2] * SM0[2]) + (REAL)0.3;
    *b = pj->M0[0] * SM1[0] + pj->M0[1] * SM1[1] + pj->M0[2] * SM1[2];
    *c = (pj->M1[0] * SM1[0] + pj->M1[1] * SM1[1] + pj->M1[2] * SM1[2]) + (REAL)0.3;
}

/* ------------------------------------------------------------------ A.1 */
/* Per-Gaussian projection, EWA footprint, tile rectangle.
 * Outputs (all length P unless noted): depth, radii (int32), xy (P*2),
 * conic_opacity (P*4), tiles_touched (uint32), rect (P*4 int32: xmin ymin xmax ymax). */
void or_preprocess(const OrCamera *cam, int P, const REAL *means, const REAL *cov6, const REAL *opacity,
                   REAL *depth, int32_t *radii, REAL *xy, REAL *conic_opacity, uint32_t *tiles_touched, int32_t *rect) {
    const int gx = (cam->W + TILE - 1) / TILE, gy = (cam->H + TILE - 1) / TILE;
    const REAL fx = (REAL)cam->W / ((REAL)2 * cam->tanfovx);
    const REAL fy = (REAL)cam->H / ((REAL)2 * cam->tanfovy);
#pragma omp parallel for schedule(static) num_threads(g_threads)   /* (every Gaussian on its own: the thread count changes no bit) */
    for (int i = 0; i < P; i++) {
        radii[i] = 0; tiles_touched[i] = 0; depth[i] = 0;
        xy[2 * i] = xy[2 * i + 1] = 0;
        conic_opacity[4 * i] = conic_opacity[4 * i + 1] = conic_opacity[4 * i + 2] = conic_opacity[4 * i + 3] = 0;
        rect[4 * i] = rect[4 * i + 1] = rect[4 * i + 2] = rect[4 * i + 3] = 0;
        const REAL *p = means + 3 * i;
        REAL pv[3];
        xform4x3(cam->view, p, pv);
        if (!(pv[2] > (REAL)0.2)) continue; /* near cull: p_view.z <= 0.2 */
        REAL ph[4];
        xform4x4(cam->proj, p, ph);
        const REAL pw = (REAL)1 / (ph[3] + (REAL)0.0000001);
        const REAL ndcx = ph[0] * pw, ndcy = ph[1] * pw;
        ProjJac pj;
        proj_jacobian(cam, p, fx, fy, &pj);
        REAL a, b, c, SM0[3], SM1[3];
        cov2d_from(cov6 + 6 * i, &pj, &a, &b, &c, SM0, SM1);
        const REAL det = a * c - b * b;
        if (det == (REAL)0) continue;
        const REAL det_inv = (REAL)1 / det;
        const REAL cx = c * det_inv, cy = -b * det_inv, cz = a * det_inv;
        const REAL mid = (REAL)0.5 * (a + c);
        const REAL disc = RSQRT(rmax((REAL)0.1, mid * mid - det));
        const REAL lam1 = mid + disc, lam2 = mid - disc;
        const REAL rad_f = RCEIL((REAL)3 * RSQRT(rmax(lam1, lam2)));
        const int rad = f2i(rad_f);
        const REAL px = ((ndcx + (REAL)1) * (REAL)cam->W - (REAL)1) * (REAL)0.5;
        const REAL py = ((ndcy + (REAL)1) * (REAL)cam->H - (REAL)1) * (REAL)0.5;
        const int x0 = imin(gx, imax(0, f2i((px - (REAL)rad) / (REAL)TILE)));
        const int y0 = imin(gy, imax(0, f2i((py - (REAL)rad) / (REAL)TILE)));
        const int x1 = imin(gx, imax(0, f2i((px + (REAL)rad + (REAL)(TILE - 1)) / (REAL)TILE)));
        const int y1 = imin(gy, imax(0, f2i((py + (REAL)rad + (REAL)(TILE - 1)) / (REAL)TILE)));
        if ((x1 - x0) * (y1 - y0) == 0) continue;
        depth[i] = pv[2];
        radii[i] = rad;
        xy[2 * i] = px; xy[2 * i + 1] = py;
        conic_opacity[4 * i] = cx; conic_opacity[4 * i + 1] = cy; conic_opacity[4 * i + 2] = cz;
        conic_opacity[4 * i + 3] = opacity[i];
        tiles_touched[i] = (uint32_t)((x1 - x0) * (y1 - y0));
        rect[4 * i] = x0; rect[4 * i + 1] = y0; rect[4 * i + 2] = x1; rect[4 * i + 3] = y1;
    }
}

/* ------------------------------------------------------------------ A.2 */
/* Stable LSD radix sort, 8 passes of 8 bits.  Threads take CONTIGUOUS chunks of the input; a pass counts per (thread, digit), offsets are
 * the prefix over (digit, then thread) and every thread scatters its chunk in order -- the output is the serial algorithm's, bit for bit,
 * whatever the thread count (stability is the contract: cub::DeviceRadixSort's, which the reference's tile lists inherit). */
static void radix_sort_pairs(uint64_t *keys, uint32_t *vals, int64_t n) {
    uint64_t *k2 = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(n > 0 ? n : 1));
    uint32_t *v2 = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(n > 0 ? n : 1));
    int nt = g_threads;
    if (nt > 64) nt = 64;
    if ((int64_t)nt * 4096 > n) nt = (int)(n / 4096) > 0 ? (int)(n / 4096) : 1;
    int64_t *cnt = (int64_t *)malloc(sizeof(int64_t) * 256 * (size_t)nt);
    for (int pass = 0; pass < 8; pass++) {
        const int sh = pass * 8;
        memset(cnt, 0, sizeof(int64_t) * 256 * (size_t)nt);
#pragma omp parallel num_threads(nt)
        {
            const int t = omp_get_thread_num();
            const int64_t lo = n * t / nt, hi = n * (t + 1) / nt;
            int64_t *c = cnt + 256 * (size_t)t;
            for (int64_t i = lo; i < hi; i++) c[(keys[i] >> sh) & 0xff]++;
#pragma omp barrier
#pragma omp single
            {
                int64_t run = 0;
                for (int b = 0; b < 256; b++)
                    for (int u = 0; u < nt; u++) { const int64_t x = cnt[256 * (size_t)u + b]; cnt[256 * (size_t)u + b] = run; run += x; }
            }   /* (implicit barrier) */
            for (int64_t i = lo; i < hi; i++) {
                const int64_t d = c[(keys[i] >> sh) & 0xff]++;
                k2[d] = keys[i]; v2[d] = vals[i];
            }
        }
        uint64_t *tk = keys; keys = k2; k2 = tk;
        uint32_t *tv = vals; vals = v2; v2 = tv;
    }
    /* 8 passes: data is back in the caller's arrays */
    free(k2); free(v2); free(cnt);
}

/* Inclusive prefix sum of tiles_touched -> offsets; returns D. */
int64_t or_scan(int P, const uint32_t *tiles_touched, uint32_t *offsets) {
    uint64_t s = 0;
    for (int i = 0; i < P; i++) { s += tiles_touched[i]; offsets[i] = (uint32_t)s; }
    return (int64_t)s;
}

/* Duplicate with keys (y outer, x inner), stable sort by key, tile ranges.
 * keys/vals: length D; ranges: tiles*2 (first, one-past-last), (0,0) if empty. */
void or_bin(const OrCamera *cam, int P, const REAL *depth, const int32_t *radii, const int32_t *rect,
            const uint32_t *offsets, int64_t D, uint64_t *keys, uint32_t *vals, uint32_t *ranges) {
    const int gx = (cam->W + TILE - 1) / TILE, gy = (cam->H + TILE - 1) / TILE;
#pragma omp parallel for schedule(static) num_threads(g_threads)   /* (a Gaussian writes its own range of the pair arrays) */
    for (int i = 0; i < P; i++) {
        if (radii[i] <= 0) continue;
        uint32_t off = (i == 0) ? 0 : offsets[i - 1];
        const float df = (float)depth[i];
        uint32_t dbits;
        memcpy(&dbits, &df, 4);
        for (int y = rect[4 * i + 1]; y < rect[4 * i + 3]; y++)
            for (int x = rect[4 * i]; x < rect[4 * i + 2]; x++) {
                keys[off] = ((uint64_t)(uint32_t)(y * gx + x) << 32) | dbits;
                vals[off] = (uint32_t)i;
                off++;
            }
    }
    radix_sort_pairs(keys, vals, D);
    memset(ranges, 0, sizeof(uint32_t) * 2 * (size_t)(gx * gy));
#pragma omp parallel for schedule(static) num_threads(g_threads)   /* (a tile's first / last position is written by exactly one i) */
    for (int64_t i = 0; i < D; i++) {
        const uint32_t t = (uint32_t)(keys[i] >> 32);
        if (i == 0 || (uint32_t)(keys[i - 1] >> 32) != t) ranges[2 * t] = (uint32_t)i;
        if (i == D - 1 || (uint32_t)(keys[i + 1] >> 32) != t) ranges[2 * t + 1] = (uint32_t)(i + 1);
    }
}

/* ------------------------------------------------------------------ A.3 */
/* Two FORMULATIONS of the same alpha (round 6; `form` argument of the *_ex entry points):
 *   form 0  the reference's expression (App. A.3): power = -0.5 (a dx^2 + c dy^2) - b dx dy, alpha = min(0.99, opacity * exp(power)).
 *   form 1  the HIP path's (gomavatar_amd/csrc/raster_render.hip alpha_eval, entry_record.hpp): conic and opacity pre-scaled once per
 *           Gaussian (A = -0.5 log2e a, B = -log2e b, Cq = -0.5 log2e c, lo = log2(opacity)), pw = fma(dx, fma(A, dx, B dy), (Cq dy) dy),
 *           opacity * G = exp2(pw + lo); in the backward T / (1 - alpha) as T * (1 / (1 - alpha)).  Same function, other rounding.  What this
 *           build CANNOT reproduce is the last bit of v_exp_f32 / v_log_f32 / v_rcp_f32 themselves (1-ulp hardware approximations against
 *           glibc's correctly rounded ones) and the association of the segment-parallel transmittance product.
 * MARGIN (forward, optional): per pixel, the smallest RELATIVE distance of any quantity that decides a branch for that pixel to its
 * threshold -- |opacity G - 1/255| / (1/255) for every entry evaluated, |T (1 - alpha) - 1e-4| / 1e-4 for every entry that passed the alpha
 * test, and |power| relative to the magnitude of its terms (the `power > 0` skip).  A pixel whose margin is far above fp32 round-off takes
 * the same branches in ANY faithful fp32 implementation; a pixel below it is one where implementations may legitimately differ (a "threshold
 * flip").  ROUNDOFF (optional): per pixel, an estimate of what one fp32 rounding of each exponent does to the composited value --
 * 1.2e-7 * sum_i w_i max(8, mag_i), w_i = alpha_i T_i the blend weight, mag_i the sum of the magnitudes of the exponent's three products:
 * needle-shaped conics far from their centre (mag ~ 1e4 with power ~ -2) make a pixel ILL-CONDITIONED in fp32 in any formulation, the
 * reference's included (form 0 against the float64 build shows it).  tests/ assert: no pixel with a comfortable margin and a small round-off
 * estimate deviates -- every deviation is a threshold flip or bounded by its own conditioning, as a test instead of an argument. */
typedef struct { REAL A, B, Cq, lo; } OrHipRec;
static inline OrHipRec or_hip_record(const REAL *co) {
    const REAL log2e = (REAL)1.44269504088896340736;
    OrHipRec r;
    r.A = ((REAL)-0.5 * log2e) * co[0];
    r.B = (-log2e) * co[1];
    r.Cq = ((REAL)-0.5 * log2e) * co[2];
    r.lo = co[3] > (REAL)0 ? RLOG2(co[3]) : -(REAL)INFINITY;
    return r;
}
/* -> opacity * G (unclamped) and log2e * power; *skip = the reference's `power > 0` rule */
static inline REAL or_hip_og(const OrHipRec *r, REAL dx, REAL dy, int *skip) {
    const REAL t1 = RFMA(r->A, dx, r->B * dy);
    const REAL pw = RFMA(dx, t1, (r->Cq * dy) * dy);
    *skip = pw > (REAL)0;
    return REXP2(pw + r->lo);
}
static inline REAL or_rel(REAL v, REAL thr) { return RABS(v - thr) / thr; }

/* colors: P*C (C <= 4). out_color: C*H*W (CHW). final_T, n_contrib: H*W.  margin: H*W or NULL. */
void or_render_fwd_ex(const OrCamera *cam, int C, const uint32_t *ranges, const uint32_t *vals,
                      const REAL *xy, const REAL *conic_opacity, const REAL *colors,
                      REAL *out_color, REAL *final_T, uint32_t *n_contrib, int form, REAL *margin, REAL *roundoff) {
    const int H = cam->H, W = cam->W;
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
#pragma omp parallel for schedule(dynamic, 1) num_threads(g_threads)
    for (int tile = 0; tile < gx * gy; tile++) {
        const int tx = tile % gx, ty = tile / gx;
        const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
        for (int ly = 0; ly < TILE; ly++)
            for (int lx = 0; lx < TILE; lx++) {
                const int px = tx * TILE + lx, py = ty * TILE + ly;
                if (px >= W || py >= H) continue;
                const REAL pfx = (REAL)px, pfy = (REAL)py;
                REAL T = 1, Cc[4] = {0, 0, 0, 0};
                REAL mg = (REAL)1e30, ro = 0, tu = 0; /* tu: estimate of T's accumulated relative uncertainty in units of 1e-6 (below) */
                uint32_t contributor = 0, last = 0;
                for (uint32_t e = r0; e < r1; e++) {
                    contributor++;
                    const uint32_t g = vals[e];
                    const REAL dx = xy[2 * g] - pfx, dy = xy[2 * g + 1] - pfy;
                    const REAL *co = conic_opacity + 4 * g;
                    REAL og;
                    int skip;
                    if (form == 1) {
                        const OrHipRec rec = or_hip_record(co);
                        og = or_hip_og(&rec, dx, dy, &skip);
                    } else {
                        const REAL power = (REAL)-0.5 * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                        skip = power > (REAL)0;
                        og = skip ? (REAL)0 : co[3] * REXP(power);
                    }
                    /* conditioning of the exponent: power is a sum of three products that cancel for elongated, correlated conics far from the
                     * centre; its fp32 error is ~eps * mag, mag = the sum of their magnitudes, whatever the association (the reference's, the
                     * HIP path's fma chain): distances to the alpha threshold are measured in units of max(1, mag / 8) (at the threshold
                     * |power| = ln(255 opacity) <= mag, and 8 is where the exponent's own rounding takes over) */
                    REAL cond = 1, magc = 8;
                    if (margin) {
                        const REAL q0 = (REAL)0.5 * co[0] * dx * dx, q1 = (REAL)0.5 * co[2] * dy * dy, q2 = co[1] * dx * dy;
                        const REAL mag = RABS(q0) + RABS(q1) + RABS(q2);
                        if (mag > (REAL)8) { cond = mag / (REAL)8; magc = mag; }
                        /* the sign of the exponent: only an indefinite (or numerically singular) conic brings it near zero away from the centre */
                        if (mag > (REAL)0) mg = rmin(mg, RABS(q0 + q1 + q2) / mag);
                    }
                    if (skip) continue;
                    if (margin) mg = rmin(mg, or_rel(og, (REAL)1 / (REAL)255) / cond);
                    const REAL alpha = rmin((REAL)0.99, og);
                    if (alpha < (REAL)1 / (REAL)255) continue;
                    const REAL test_T = T * ((REAL)1 - alpha);
                    /* T is a product of (1 - alpha_i): a relative error d in alpha_i (a few ulp of the exponent: ~5e-7 * cond) moves it by
                     * d alpha_i / (1 - alpha_i) (nothing when alpha_i sits at the 0.99 cap), a rounding per product adds 6e-8; the distance to
                     * the stop threshold is measured in units of that accumulated uncertainty once it exceeds 1e-6 */
                    tu += (REAL)0.06 + (og < (REAL)0.99 ? (REAL)0.5 * cond * alpha / ((REAL)1 - alpha) : (REAL)0);
                    if (margin) mg = rmin(mg, or_rel(test_T, (REAL)0.0001) / (tu > (REAL)1 ? tu : (REAL)1));
                    if (test_T < (REAL)0.0001) break;
                    for (int ch = 0; ch < C; ch++) Cc[ch] += colors[(size_t)g * C + ch] * alpha * T;
                    /* ROUND-OFF ESTIMATE of the pixel: one fp32 rounding of the exponent's terms (1.2e-7 * mag, any formulation) is a relative
                     * error of alpha and moves the pixel by that much of the entry's blend weight (and, through T, by as much again behind it) */
                    if (og < (REAL)0.99) ro += (REAL)1.2e-7 * magc * alpha * T;
                    T = test_T;
                    last = contributor;
                }
                const size_t pix = (size_t)py * W + px;
                final_T[pix] = T;
                n_contrib[pix] = last;
                if (margin) margin[pix] = mg;
                if (roundoff) roundoff[pix] = ro;
                for (int ch = 0; ch < C; ch++) out_color[(size_t)ch * H * W + pix] = Cc[ch] + T * cam->bg[ch];
            }
    }
}
void or_render_fwd(const OrCamera *cam, int C, const uint32_t *ranges, const uint32_t *vals,
                   const REAL *xy, const REAL *conic_opacity, const REAL *colors,
                   REAL *out_color, REAL *final_T, uint32_t *n_contrib) {
    or_render_fwd_ex(cam, C, ranges, vals, xy, conic_opacity, colors, out_color, final_T, n_contrib, 0, NULL, NULL);
}

/* ------------------------------------------------------------------ A.4 */
static inline void atomic_add(REAL *p, REAL v) {
#pragma omp atomic
    *p += v;
}

/* dL_dpix: C*H*W.  Accumulates (+=) into dL_dcolors (P*C), dL_dmean2D (P*2),
 * dL_dconic (P*3), dL_dopacity (P); caller zero-initialises. */
void or_render_bwd_ex(const OrCamera *cam, int C, const uint32_t *ranges, const uint32_t *vals,
                      const REAL *xy, const REAL *conic_opacity, const REAL *colors,
                      const REAL *final_T, const uint32_t *n_contrib, const REAL *dL_dpix,
                      REAL *dL_dcolors, REAL *dL_dmean2D, REAL *dL_dconic, REAL *dL_dopacity, int form) {
    const int H = cam->H, W = cam->W;
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    const REAL ddelx_dx = (REAL)0.5 * (REAL)W, ddely_dy = (REAL)0.5 * (REAL)H;
#pragma omp parallel for schedule(dynamic, 1) num_threads(g_threads)
    for (int tile = 0; tile < gx * gy; tile++) {
        const int tx = tile % gx, ty = tile / gx;
        const uint32_t r0 = ranges[2 * tile];
        for (int ly = 0; ly < TILE; ly++)
            for (int lx = 0; lx < TILE; lx++) {
                const int px = tx * TILE + lx, py = ty * TILE + ly;
                if (px >= W || py >= H) continue;
                const size_t pix = (size_t)py * W + px;
                const REAL pfx = (REAL)px, pfy = (REAL)py;
                const REAL T_final = final_T[pix];
                REAL T = T_final;
                const uint32_t last = n_contrib[pix];
                REAL accum_rec[4] = {0, 0, 0, 0}, last_color[4] = {0, 0, 0, 0}, dpix[4] = {0, 0, 0, 0};
                REAL last_alpha = 0;
                for (int ch = 0; ch < C; ch++) dpix[ch] = dL_dpix[(size_t)ch * H * W + pix];
                REAL bg_dot = 0;
                for (int ch = 0; ch < C; ch++) bg_dot += cam->bg[ch] * dpix[ch];
                for (uint32_t k = last; k-- > 0;) { /* entries last-1 .. 0, back to front */
                    const uint32_t g = vals[r0 + k];
                    const REAL dx = xy[2 * g] - pfx, dy = xy[2 * g + 1] - pfy;
                    const REAL *co = conic_opacity + 4 * g;
                    REAL G, alpha;
                    if (form == 1) { /* the HIP path's formulation of the same alpha (see A.3 above) */
                        const OrHipRec rec = or_hip_record(co);
                        int skip;
                        const REAL og = or_hip_og(&rec, dx, dy, &skip);
                        if (skip) continue;
                        alpha = rmin((REAL)0.99, og);
                        if (alpha < (REAL)1 / (REAL)255) continue;
                        G = og / co[3];
                        T = T * ((REAL)1 / ((REAL)1 - alpha));
                    } else {
                        const REAL power = (REAL)-0.5 * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                        if (power > (REAL)0) continue;
                        G = REXP(power);
                        alpha = rmin((REAL)0.99, co[3] * G);
                        if (alpha < (REAL)1 / (REAL)255) continue;
                        T = T / ((REAL)1 - alpha);
                    }
                    const REAL dch_dcol = alpha * T;
                    REAL dL_dalpha = 0;
                    for (int ch = 0; ch < C; ch++) {
                        const REAL c = colors[(size_t)g * C + ch];
                        accum_rec[ch] = last_alpha * last_color[ch] + ((REAL)1 - last_alpha) * accum_rec[ch];
                        last_color[ch] = c;
                        dL_dalpha += (c - accum_rec[ch]) * dpix[ch];
                        atomic_add(&dL_dcolors[(size_t)g * C + ch], dch_dcol * dpix[ch]);
                    }
                    dL_dalpha *= T;
                    last_alpha = alpha;
                    dL_dalpha += (-T_final / ((REAL)1 - alpha)) * bg_dot;
                    const REAL dL_dG = co[3] * dL_dalpha;
                    const REAL gdx = G * dx, gdy = G * dy;
                    const REAL dG_ddelx = -gdx * co[0] - gdy * co[1];
                    const REAL dG_ddely = -gdy * co[2] - gdx * co[1];
                    atomic_add(&dL_dmean2D[2 * g], dL_dG * dG_ddelx * ddelx_dx);
                    atomic_add(&dL_dmean2D[2 * g + 1], dL_dG * dG_ddely * ddely_dy);
                    atomic_add(&dL_dconic[3 * g], (REAL)-0.5 * gdx * dx * dL_dG);
                    atomic_add(&dL_dconic[3 * g + 1], (REAL)-0.5 * gdx * dy * dL_dG);
                    atomic_add(&dL_dconic[3 * g + 2], (REAL)-0.5 * gdy * dy * dL_dG);
                    atomic_add(&dL_dopacity[g], G * dL_dalpha);
                }
            }
    }
}

void or_render_bwd(const OrCamera *cam, int C, const uint32_t *ranges, const uint32_t *vals,
                   const REAL *xy, const REAL *conic_opacity, const REAL *colors,
                   const REAL *final_T, const uint32_t *n_contrib, const REAL *dL_dpix,
                   REAL *dL_dcolors, REAL *dL_dmean2D, REAL *dL_dconic, REAL *dL_dopacity) {
    or_render_bwd_ex(cam, C, ranges, vals, xy, conic_opacity, colors, final_T, n_contrib, dL_dpix, dL_dcolors, dL_dmean2D, dL_dconic, dL_dopacity, 0);
}

/* ------------------------------------------------------------------ A.5 */
/* Per-Gaussian backward: conic -> cov2D -> (cov3D, mean3D), plus the
 * projection term of dL_dmean2D.  Writes dL_dmeans (P*3), dL_dcov6 (P*6);
 * both zero for Gaussians with radii <= 0. */
void or_preprocess_bwd(const OrCamera *cam, int P, const REAL *means, const REAL *cov6, const int32_t *radii,
                       const REAL *dL_dconic, const REAL *dL_dmean2D, REAL *dL_dmeans, REAL *dL_dcov6) {
    const REAL fx = (REAL)cam->W / ((REAL)2 * cam->tanfovx);
    const REAL fy = (REAL)cam->H / ((REAL)2 * cam->tanfovy);
    const REAL *v = cam->view, *pr = cam->proj;
#pragma omp parallel for schedule(static) num_threads(g_threads)   /* (every Gaussian on its own) */
    for (int i = 0; i < P; i++) {
        REAL *gm = dL_dmeans + 3 * i, *gc = dL_dcov6 + 6 * i;
        gm[0] = gm[1] = gm[2] = 0;
        for (int k = 0; k < 6; k++) gc[k] = 0;
        if (!(radii[i] > 0)) continue;
        const REAL *p = means + 3 * i;
        ProjJac pj;
        proj_jacobian(cam, p, fx, fy, &pj);
        REAL a, b, c, SM0[3], SM1[3];
        cov2d_from(cov6 + 6 * i, &pj, &a, &b, &c, SM0, SM1);
        const REAL gxx = dL_dconic[3 * i], gxy = dL_dconic[3 * i + 1], gyy = dL_dconic[3 * i + 2];
        const REAL denom = a * c - b * b;
        const REAL d2 = (REAL)1 / (denom * denom + (REAL)0.0000001);
        REAL dL_da = 0, dL_db = 0, dL_dc = 0;
        if (d2 != (REAL)0) {
            dL_da = d2 * (-c * c * gxx + (REAL)2 * b * c * gxy + (denom - a * c) * gyy);
            dL_dc = d2 * (-a * a * gyy + (REAL)2 * a * b * gxy + (denom - a * c) * gxx);
            dL_db = d2 * (REAL)2 * (b * c * gxx - (denom + (REAL)2 * b * b) * gxy + a * b * gyy);
            const REAL *M0 = pj.M0, *M1 = pj.M1;
            gc[0] = M0[0] * M0[0] * dL_da + M0[0] * M1[0] * dL_db + M1[0] * M1[0] * dL_dc;
            gc[3] = M0[1] * M0[1] * dL_da + M0[1] * M1[1] * dL_db + M1[1] * M1[1] * dL_dc;
            gc[5] = M0[2] * M0[2] * dL_da + M0[2] * M1[2] * dL_db + M1[2] * M1[2] * dL_dc;
            gc[1] = (REAL)2 * M0[0] * M0[1] * dL_da + (M0[0] * M1[1] + M0[1] * M1[0]) * dL_db + (REAL)2 * M1[0] * M1[1] * dL_dc;
            gc[2] = (REAL)2 * M0[0] * M0[2] * dL_da + (M0[0] * M1[2] + M0[2] * M1[0]) * dL_db + (REAL)2 * M1[0] * M1[2] * dL_dc;
            gc[4] = (REAL)2 * M0[2] * M0[1] * dL_da + (M0[1] * M1[2] + M0[2] * M1[1]) * dL_db + (REAL)2 * M1[1] * M1[2] * dL_dc;
        }
        /* dL/dM rows: dM0 = 2 Sigma M0 dL_da + Sigma M1 dL_db ; dM1 = 2 Sigma M1 dL_dc + Sigma M0 dL_db */
        REAL dM0[3], dM1[3];
        for (int k = 0; k < 3; k++) {
            dM0[k] = (REAL)2 * SM0[k] * dL_da + SM1[k] * dL_db;
            dM1[k] = (REAL)2 * SM1[k] * dL_dc + SM0[k] * dL_db;
        }
        /* W[i][k] = view[4k+i] */
        const REAL dJ00 = v[0] * dM0[0] + v[4] * dM0[1] + v[8] * dM0[2];
        const REAL dJ02 = v[2] * dM0[0] + v[6] * dM0[1] + v[10] * dM0[2];
        const REAL dJ11 = v[1] * dM1[0] + v[5] * dM1[1] + v[9] * dM1[2];
        const REAL dJ12 = v[2] * dM1[0] + v[6] * dM1[1] + v[10] * dM1[2];
        const REAL tz = (REAL)1 / pj.t[2], tz2 = tz * tz, tz3 = tz2 * tz;
        const REAL dtx = pj.xmul * -fx * tz2 * dJ02;
        const REAL dty = pj.ymul * -fy * tz2 * dJ12;
        const REAL dtz = -fx * tz2 * dJ00 - fy * tz2 * dJ11 + ((REAL)2 * fx * pj.t[0]) * tz3 * dJ02 + ((REAL)2 * fy * pj.t[1]) * tz3 * dJ12;
        /* W^T dt */
        gm[0] = v[0] * dtx + v[1] * dty + v[2] * dtz;
        gm[1] = v[4] * dtx + v[5] * dty + v[6] * dtz;
        gm[2] = v[8] * dtx + v[9] * dty + v[10] * dtz;
        /* projection term of the screen-space mean gradient */
        REAL ph[4];
        xform4x4(pr, p, ph);
        const REAL mw = (REAL)1 / (ph[3] + (REAL)0.0000001);
        const REAL mul1 = ph[0] * mw * mw, mul2 = ph[1] * mw * mw;
        const REAL g2x = dL_dmean2D[2 * i], g2y = dL_dmean2D[2 * i + 1];
        gm[0] += (pr[0] * mw - pr[3] * mul1) * g2x + (pr[1] * mw - pr[3] * mul2) * g2y;
        gm[1] += (pr[4] * mw - pr[7] * mul1) * g2x + (pr[5] * mw - pr[7] * mul2) * g2y;
        gm[2] += (pr[8] * mw - pr[11] * mul1) * g2x + (pr[9] * mw - pr[11] * mul2) * g2y;
    }
}
