"""Host-side image operations of the reference's dataset reader (dataset/train.py:138-200): lens undistortion, resizing, the
random crop around the mask.  The reference calls OpenCV (`cv2.undistort`, `cv2.resize` with INTER_LANCZOS4 / INTER_LINEAR) in its
DataLoader worker; OpenCV is not installed here, so these are numpy restatements of the published algorithms of OpenCV 4.x imgproc
(`initUndistortRectifyMap` + `remap`, `resize`): **parity unpinned** -- no OpenCV output to compare with in this image.  They run
on the host like the reference's; nothing here is on the GPU hot path.

What is restated, with the details that decide the last bits:
  * undistort (8-bit images): destination pixel (u, v) -> normalised (x, y) through the SAME matrix as new camera matrix -> radial
    (k1 k2 k3 [k4 k5 k6]) and tangential (p1 p2) [thin prism s1..s4] distortion -> source position; bilinear `remap` in OpenCV's
    fixed point: positions rounded to 1/32 pixel (INTER_BITS = 5), weights from the 32 x 32 table of 15-bit integers
    (INTER_REMAP_COEF_BITS; exact for bilinear weights on that grid), result (sum + 16384) >> 15; BORDER_CONSTANT 0.
  * resize: source coordinate (d + 0.5) * scale - 0.5 (scale = src / dst); INTER_LANCZOS4 = 8 taps, float coefficients from
    OpenCV's rotation-recurrence form of sin(pi x / 4)-windowed sinc, normalised to sum 1, BORDER_REPLICATE, horizontal pass then
    vertical pass in the image's own floating type; INTER_LINEAR = 2 taps with the coordinate clamped at the borders, and -- as in
    cv::resize -- an exact 2 x 2 decimation takes the INTER_AREA path (the mean of the four pixels)."""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import numpy as np

INTER_LINEAR, INTER_LANCZOS4 = "linear", "lanczos4"


# ---------------------------------------------------------------------------------------------------------------- undistort
def undistort_map(K: np.ndarray, D: Sequence[float], H: int, W: int) -> Tuple[np.ndarray, np.ndarray]:
    """initUndistortRectifyMap(K, D, R = I, newCameraMatrix = K): source position (map_x, map_y) of every destination pixel."""
    K = np.asarray(K, dtype=np.float64)
    d = np.zeros(14, dtype=np.float64)
    D = np.asarray(D, dtype=np.float64).reshape(-1)
    d[: D.size] = D
    k1, k2, p1, p2, k3, k4, k5, k6, s1, s2, s3, s4 = d[:12]
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    u, v = np.meshgrid(np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64))
    x, y = (u - cx) / fx, (v - cy) / fy
    x2, y2 = x * x, y * y
    r2 = x2 + y2
    _2xy = 2.0 * x * y
    kr = (1.0 + ((k3 * r2 + k2) * r2 + k1) * r2) / (1.0 + ((k6 * r2 + k5) * r2 + k4) * r2)
    xd = x * kr + p1 * _2xy + p2 * (r2 + 2.0 * x2) + s1 * r2 + s2 * r2 * r2
    yd = y * kr + p1 * (r2 + 2.0 * y2) + p2 * _2xy + s3 * r2 + s4 * r2 * r2
    return (fx * xd + cx).astype(np.float32), (fy * yd + cy).astype(np.float32)


def _remap_linear_table() -> np.ndarray:
    """The 32 x 32 x (2 x 2) table of 15-bit bilinear weights of cv::remap (initInterTab2D, fixed-point branch).  With positions on
    the 1/32 grid the four products (1 - a or a)(1 - b or b) * 32768 are integers ((32 - i or i)(32 - j or j) * 32) and sum to
    32768 exactly: OpenCV's correction of a rounded table's sum never fires for INTER_LINEAR."""
    t = np.arange(32, dtype=np.int64)
    w1 = np.stack([32 - t, t], -1)                                          # (32, 2)
    return (w1[:, None, :, None] * w1[None, :, None, :] * 32).astype(np.int64)   # [fy][fx][y tap][x tap]


_TAB = None


def remap_linear_u8(img: np.ndarray, map_x: np.ndarray, map_y: np.ndarray) -> np.ndarray:
    """cv::remap(img, map_x, map_y, INTER_LINEAR, BORDER_CONSTANT, 0) for 8-bit images, in OpenCV's fixed point."""
    global _TAB
    if _TAB is None:
        _TAB = _remap_linear_table()
    assert img.dtype == np.uint8
    src = img if img.ndim == 3 else img[..., None]
    H, W, C = src.shape
    sx = np.rint(map_x.astype(np.float64) * 32.0).astype(np.int64)       # saturate_cast<int>(x * INTER_TAB_SIZE): round half to even
    sy = np.rint(map_y.astype(np.float64) * 32.0).astype(np.int64)
    ix, iy, fx, fy = sx >> 5, sy >> 5, sx & 31, sy & 31
    w = _TAB[fy, fx]                                                       # (h, w, 2, 2)
    acc = np.zeros(map_x.shape + (C,), dtype=np.int64)
    for dy in (0, 1):
        for dx in (0, 1):
            yy, xx = iy + dy, ix + dx
            inside = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
            px = src[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)].astype(np.int64)
            px[~inside] = 0                                                # BORDER_CONSTANT, value 0
            acc += px * w[..., dy, dx][..., None]
    out = np.clip((acc + (1 << 14)) >> 15, 0, 255).astype(np.uint8)
    return out if img.ndim == 3 else out[..., 0]


def undistort(img: np.ndarray, K: np.ndarray, D: Sequence[float]) -> np.ndarray:
    """cv2.undistort(img, K, D) (dataset/train.py:151-155)."""
    mx, my = undistort_map(K, D, img.shape[0], img.shape[1])
    return remap_linear_u8(img, mx, my)


# ------------------------------------------------------------------------------------------------------------------- resize
def _lanczos4_coeffs(fx: np.ndarray) -> np.ndarray:
    """interpolateLanczos4 of OpenCV: eight float coefficients per fractional position (taps -3 .. +4)."""
    s45 = 0.70710678118654752440084436210485
    cs = np.array([[1, 0], [-s45, -s45], [0, 1], [s45, -s45], [-1, 0], [s45, s45], [0, -1], [-s45, s45]], dtype=np.float64)
    x = fx.astype(np.float64)
    out = np.zeros(x.shape + (8,), dtype=np.float32)
    tiny = x < np.finfo(np.float32).eps
    y0 = -(x + 3.0) * np.pi * 0.25
    s0, c0 = np.sin(y0), np.cos(y0)
    for i in range(8):
        y = -(x + 3.0 - i) * np.pi * 0.25
        with np.errstate(divide="ignore", invalid="ignore"):
            out[..., i] = ((cs[i, 0] * s0 + cs[i, 1] * c0) / (y * y)).astype(np.float32)
    ssum = np.zeros(x.shape, dtype=np.float32)
    for i in range(8):
        ssum = (ssum + out[..., i]).astype(np.float32)                    # float accumulation, tap order
    with np.errstate(divide="ignore", invalid="ignore"):   # (x = 0 divides by zero above: those rows are overwritten below)
        inv = (np.float32(1.0) / ssum).astype(np.float32)
        out = (out * inv[..., None]).astype(np.float32)
    out[tiny] = 0.0
    out[tiny, 3] = 1.0
    return out


def _axis_taps(dst: int, src: int, scale: float, kind: str):
    """Per destination index: first source index of the tap window and the float coefficients (cv::resize's xofs / alpha)."""
    d = np.arange(dst, dtype=np.float64)
    f = (d + 0.5) * scale - 0.5
    s = np.floor(f).astype(np.int64)
    f = (f - s).astype(np.float32)
    if kind == INTER_LINEAR:
        lo, hi = s < 0, s >= src - 1
        f = np.where(lo | hi, np.float32(0.0), f).astype(np.float32)
        s = np.where(lo, 0, np.where(hi, src - 1, s))
        coef = np.stack([np.float32(1.0) - f, f], -1).astype(np.float32)
        idx = np.stack([s, s + 1], -1)
    else:
        coef = _lanczos4_coeffs(f)
        idx = s[:, None] + np.arange(-3, 5)[None, :]
    return np.clip(idx, 0, src - 1), coef                                   # BORDER_REPLICATE


def resize(img: np.ndarray, dsize: Optional[Tuple[int, int]] = None, fx: Optional[float] = None, fy: Optional[float] = None,
           interpolation: str = INTER_LINEAR) -> np.ndarray:
    """cv2.resize(img, dsize=(w, h)) or cv2.resize(img, None, fx=, fy=) for floating-point images (dataset/train.py:158-173)."""
    assert img.dtype in (np.float32, np.float64), "the reference resizes the composited float image and the float mask"
    src = img if img.ndim == 3 else img[..., None]
    H, W, _ = src.shape
    if dsize is not None:
        w, h = int(dsize[0]), int(dsize[1])
        sx, sy = W / w, H / h
    else:
        w, h = int(round(W * fx)), int(round(H * fy))                       # saturate_cast<int>(cols * fx)
        sx, sy = 1.0 / fx, 1.0 / fy
    if interpolation == INTER_LINEAR and W == 2 * w and H == 2 * h:         # cv::resize: exact 2 x 2 decimation -> INTER_AREA fast path
        out = (src[0::2, 0::2] + src[0::2, 1::2] + src[1::2, 0::2] + src[1::2, 1::2]) * src.dtype.type(0.25)
        return out if img.ndim == 3 else out[..., 0]
    xi, xa = _axis_taps(w, W, sx, interpolation)
    yi, ya = _axis_taps(h, H, sy, interpolation)
    wt = src.dtype.type
    # horizontal pass over every source row, then the vertical pass (the order of cv::resize's generic path)
    tmp = np.zeros((H, w, src.shape[2]), dtype=src.dtype)
    for t in range(xi.shape[1]):
        tmp += src[:, xi[:, t], :] * xa[None, :, t, None].astype(wt)
    out = np.zeros((h, w, src.shape[2]), dtype=src.dtype)
    for t in range(yi.shape[1]):
        out += tmp[yi[:, t], :, :] * ya[:, t, None, None].astype(wt)
    return out if img.ndim == 3 else out[..., 0]


# --------------------------------------------------------------------------------------------------------------------- crop
def crop_image(img: np.ndarray, mask: np.ndarray, K: np.ndarray, crop_size: Tuple[int, int], rng=np.random):
    """dataset/train.py:176-200: a crop_w x crop_h window within 50 pixels of the mask's centroid that holds at least 20 mask units;
    the principal point moves with it."""
    crop_w, crop_h = crop_size
    h, w, _ = img.shape
    h_center, w_center, _ = np.stack(np.nonzero(mask), axis=-1).mean(axis=0).astype(int)
    if h_center + (crop_h + 1) // 2 > h:
        h_center = h - (crop_h + 1) // 2
    if h_center - crop_h // 2 < 0:
        h_center = crop_h // 2
    if w_center + (crop_w + 1) // 2 > w:
        w_center = w - (crop_w + 1) // 2
    if w_center - crop_w // 2 < 0:
        w_center = crop_w // 2
    h_left, w_left = h_center - crop_h // 2, w_center - crop_w // 2
    while True:
        rand_w = rng.randint(max(0, w_left - 50), min(w_left + 50, w - crop_w + 1))
        rand_h = rng.randint(max(0, h_left - 50), min(h_left + 50, h - crop_h + 1))
        crop_mask = mask[rand_h:rand_h + crop_h, rand_w:rand_w + crop_w]
        if np.sum(crop_mask) < 20:
            continue
        K_new = K.copy()
        K_new[0, 2] -= rand_w
        K_new[1, 2] -= rand_h
        return img[rand_h:rand_h + crop_h, rand_w:rand_w + crop_w], crop_mask, K_new
