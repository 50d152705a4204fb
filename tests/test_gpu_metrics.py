"""Evaluation metrics on the GPU against the numpy/scipy restatement (oracle/metrics.py)."""
import numpy as np
import pytest
import torch

from oracle import metrics as om

pytestmark = pytest.mark.gpu


def _pair(seed, H=96, W=80):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:H, 0:W]
    base = 0.5 + 0.4 * np.sin(xx / 7.0)[..., None] * np.cos(yy / 5.0)[..., None] * np.array([1.0, 0.7, 0.4])
    a = np.clip(base + rng.normal(0, 0.05, (H, W, 3)), 0, 1).astype(np.float32)
    b = np.clip(base + rng.normal(0, 0.08, (H, W, 3)), 0, 1).astype(np.float32)
    return a, b


@pytest.mark.parametrize("seed", [0, 1])
def test_ssim_both_definitions_and_psnr(seed):
    from gomavatar_amd import metrics as M
    a, b = _pair(seed)
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    qa, qb = M.from_8b(M.to_8b(ta)), M.from_8b(M.to_8b(tb))          # eval.py:355-361: metrics see 8-bit images
    na, nb = om.to_8b(a) / 255.0, om.to_8b(b) / 255.0
    np.testing.assert_array_equal(M.to_8b(ta).cpu().numpy(), om.to_8b(a))
    assert abs(M.psnr(qa, qb) - om.psnr(na, nb)) < 1e-4
    assert abs(M.ssim_skimage(qa, qb) - om.ssim_skimage(na.astype(np.float32), nb.astype(np.float32))) < 1e-6
    assert abs(M.ssim_torchmetrics(qa, qb) - om.ssim_torchmetrics(na.astype(np.float32), nb.astype(np.float32))) < 1e-6
    assert abs(M.ssim_skimage(qa, qa) - 1.0) < 1e-12


def test_evaluator_accumulates():
    from gomavatar_amd import metrics as M
    a, b = _pair(3, 64, 64)
    ev = M.Evaluator()
    ev.evaluate(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda())
    ev.evaluate(torch.from_numpy(b).cuda(), torch.from_numpy(a).cuda())
    out = ev.summarize()
    assert 10 < out["psnr"] < 40 and 0 < out["ssim"] < 1 and ev.psnr == []
