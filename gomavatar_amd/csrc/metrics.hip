// SSIM (reference eval.py:106-108 skimage `structural_similarity(multichannel=True)` and eval.py:157 torchmetrics
// `StructuralSimilarityIndexMeasure(data_range=1)`; definitions: SURVEY.md App. C).  Both are: windowed first and second
// moments of the two images, the SSIM map, its mean over the pixels whose window lies inside the image and over the
// channels.  They differ in the window (7x7 uniform vs 11x11 gaussian sigma 1.5), the covariance normalisation
// (sample N/(N-1) vs population) and data_range (2.0 for float input in skimage 0.18 vs 1.0): all parameters here.
// Evaluation-only, so accuracy first: fp64 accumulation (the reference computes in float64), one thread per
// (interior pixel, channel), direct win x win window from L1/L2 -- 0.24 G multiply-adds for a 512x512x3 image.
#include "gom_internal.h"

namespace {

__global__ void __launch_bounds__(256) k_ssim(int H, int W, int C, const float *__restrict__ x, const float *__restrict__ y, int win,
                                              const double *__restrict__ wts, double cov_norm, double C1, double C2,
                                              double *__restrict__ partials) {
    __shared__ double s_red[4];
    const int pad = (win - 1) / 2;
    const int ih = H - 2 * pad, iw = W - 2 * pad;
    const long long total = (long long)ih * iw * C;
    double acc = 0.0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % C);
        const int px = (int)((i / C) % iw), py = (int)(i / ((long long)C * iw));
        double ux = 0, uy = 0, uxx = 0, uyy = 0, uxy = 0;
        for (int dy = 0; dy < win; dy++) {
            const float *rx = x + ((size_t)(py + dy) * W + px) * C + c;
            const float *ry = y + ((size_t)(py + dy) * W + px) * C + c;
            for (int dx = 0; dx < win; dx++) {
                const double w = wts[dy * win + dx];
                const double a = (double)rx[(size_t)dx * C], b = (double)ry[(size_t)dx * C];
                ux += w * a; uy += w * b; uxx += w * a * a; uyy += w * b * b; uxy += w * a * b;
            }
        }
        const double vx = cov_norm * (uxx - ux * ux), vy = cov_norm * (uyy - uy * uy), vxy = cov_norm * (uxy - ux * uy);
        acc += ((2.0 * ux * uy + C1) * (2.0 * vxy + C2)) / ((ux * ux + uy * uy + C1) * (vx + vy + C2));
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
}

}  // namespace

extern "C" int gom_ssim(int H, int W, int C, const float *img0, const float *img1, int win, const double *weights, double cov_norm,
                        double C1, double C2, double *partials, void *stream) {
    if (H <= 0 || W <= 0 || C <= 0 || win < 1 || (win & 1) == 0 || win > H || win > W) { gom_set_error("gom_ssim: bad sizes"); return -1; }
    if (!img0 || !img1 || !weights || !partials) { gom_set_error("gom_ssim: null pointer"); return -1; }
    hipLaunchKernelGGL(k_ssim, dim3(GOM_LOSS_BLOCKS), dim3(256), 0, (hipStream_t)stream, H, W, C, img0, img1, win, weights, cov_norm, C1, C2,
                       partials);
    GOM_LAUNCH_CHECK();
    return 0;
}
