"""Evaluation metrics of the reference's eval loop (eval.py:86-180, 336-361;
definitions in SURVEY.md App. C), on device:

    to_8b / from_8b   utils/image_util.py:21-22 quantisation round trip
    psnr              eval.py:101-104   -10 log10(mean((p-g)^2)) on the quantised images
    ssim_skimage      eval.py:106-108   skimage 0.18 structural_similarity(multichannel=True)
    ssim_torchmetrics eval.py:157       torchmetrics SSIM(data_range=1)
    lpips x 1000      eval.py:110-116   gomavatar_amd.lpips.LPIPS

SSIM runs in one HIP kernel (csrc/metrics.hip, fp64 accumulation)."""
from __future__ import annotations

import math

import torch

from . import _lib


def to_8b(img: torch.Tensor) -> torch.Tensor:
    """clip to [0,1], x255, truncate to uint8 (image_util.py:21-22)."""
    return (255.0 * img.clamp(0.0, 1.0)).to(torch.uint8)


def from_8b(img8: torch.Tensor) -> torch.Tensor:
    return img8.to(torch.float32) / 255.0


def psnr(pred: torch.Tensor, gt: torch.Tensor) -> float:
    """eval.py:101-104 on images that already went through the 8-bit round trip (eval.py:355-361)."""
    mse = torch.mean((pred.double() - gt.double()) ** 2)
    return float(-10.0 * torch.log10(mse))


def _ssim(pred, gt, win, weights, cov_norm, data_range) -> float:
    lib = _lib.load()
    assert pred.shape == gt.shape and pred.dim() == 3, "HWC images"
    a, b = pred.float().contiguous(), gt.float().contiguous()
    H, W, C = a.shape
    w = weights.to(a.device, torch.float64).contiguous()
    partials = torch.empty(_lib.GOM_LOSS_BLOCKS, dtype=torch.float64, device=a.device)
    c1, c2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    _lib.check(lib.gom_ssim(H, W, C, _lib.ptr(a), _lib.ptr(b), win, _lib.ptr(w), float(cov_norm), c1, c2, _lib.ptr(partials), _lib.stream_ptr()))
    return float(partials.sum() / ((H - win + 1) * (W - win + 1) * C))


def ssim_skimage(pred: torch.Tensor, gt: torch.Tensor) -> float:
    """(H,W,C) float images; 7x7 uniform window, sample covariance, data_range 2 (float input in skimage 0.18)."""
    w = torch.full((7, 7), 1.0 / 49.0, dtype=torch.float64)
    return _ssim(pred, gt, 7, w, 49.0 / 48.0, 2.0)


def ssim_torchmetrics(pred: torch.Tensor, gt: torch.Tensor) -> float:
    """(H,W,C) float images; 11x11 gaussian (sigma 1.5) window, population covariance, data_range 1."""
    k = torch.arange(11, dtype=torch.float64) - 5.0
    g = torch.exp(-(k / 1.5) ** 2 / 2.0)
    g = g / g.sum()
    return _ssim(pred, gt, 11, torch.outer(g, g), 1.0, 1.0)


class Evaluator:
    """eval.py:86-147 (ZJU-MoCap protocol): accumulates mse / psnr / ssim / lpips x 1000 per frame."""

    def __init__(self, lpips_model=None):
        self.lpips_model = lpips_model
        self.mse, self.psnr, self.ssim, self.lpips = [], [], [], []

    def evaluate(self, rgb_pred: torch.Tensor, rgb_gt: torch.Tensor) -> None:
        self.mse.append(float(torch.mean((rgb_pred.double() - rgb_gt.double()) ** 2)))
        self.psnr.append(psnr(rgb_pred, rgb_gt))
        self.ssim.append(ssim_skimage(rgb_pred, rgb_gt))
        if self.lpips_model is not None:
            with torch.no_grad():
                v = self.lpips_model(rgb_pred.float()[None].permute(0, 3, 1, 2) * 2.0 - 1.0, rgb_gt.float()[None].permute(0, 3, 1, 2) * 2.0 - 1.0)
            self.lpips.append(float(v.mean()) * 1000.0)

    def summarize(self):
        mean = lambda v: (sum(v) / len(v)) if v else math.nan
        out = {"mse": mean(self.mse), "psnr": mean(self.psnr), "ssim": mean(self.ssim), "lpips": mean(self.lpips)}
        self.mse, self.psnr, self.ssim, self.lpips = [], [], [], []
        return out
