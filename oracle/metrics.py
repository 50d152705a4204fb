"""TEST INFRASTRUCTURE ONLY: numpy/scipy restatement of the evaluation metrics
(eval.py:101-108,157).  skimage 0.18 and torchmetrics are absent
from this image and from /root/reference -> parity unpinned; what is checked is
the published definition (uniform / gaussian window moments, sample vs population
covariance, cropped mean) written with scipy.ndimage filters in float64."""
import numpy as np
from scipy import ndimage


def to_8b(img):
    return (255.0 * np.clip(img, 0.0, 1.0)).astype(np.uint8)


def psnr(p, g):
    return -10.0 * np.log(np.mean((p.astype(np.float64) - g.astype(np.float64)) ** 2)) / np.log(10.0)


def _ssim_channel(x, y, filt, cov_norm, data_range, pad):
    x, y = x.astype(np.float64), y.astype(np.float64)
    ux, uy = filt(x), filt(y)
    uxx, uyy, uxy = filt(x * x), filt(y * y), filt(x * y)
    vx, vy, vxy = cov_norm * (uxx - ux * ux), cov_norm * (uyy - uy * uy), cov_norm * (uxy - ux * uy)
    c1, c2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    s = ((2 * ux * uy + c1) * (2 * vxy + c2)) / ((ux ** 2 + uy ** 2 + c1) * (vx + vy + c2))
    return s[pad:-pad, pad:-pad].mean()


def ssim_skimage(p, g):
    f = lambda a: ndimage.uniform_filter(a, size=7)
    return float(np.mean([_ssim_channel(p[..., c], g[..., c], f, 49.0 / 48.0, 2.0, 3) for c in range(p.shape[-1])]))


def ssim_torchmetrics(p, g):
    k = np.arange(11, dtype=np.float64) - 5.0
    w = np.exp(-(k / 1.5) ** 2 / 2.0)
    w /= w.sum()
    f = lambda a: ndimage.correlate1d(ndimage.correlate1d(a, w, axis=0, mode="reflect"), w, axis=1, mode="reflect")
    return float(np.mean([_ssim_channel(p[..., c], g[..., c], f, 1.0, 1.0, 5) for c in range(p.shape[-1])]))
