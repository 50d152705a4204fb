import numpy as np
import pytest
import torch

from oracle import geometry as og

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("with_shade", [False, True])
def test_l1_photometric_matches_reference_formula(with_shade):
    from gomavatar_amd import losses
    torch.manual_seed(0)
    H, W = 70, 90
    pred = torch.rand(4, H, W)
    gt = torch.rand(H, W, 3)
    gm = (torch.rand(H, W) > 0.5).float()
    bg = torch.rand(3)
    shade = torch.rand(H, W) * 2 if with_shade else None
    # reference formula (train.py:53-55, 101-111), fp64
    p = pred.double().requires_grad_()
    s = shade.double().requires_grad_() if with_shade else None
    rgb = p[:3].permute(1, 2, 0)[None] * (s[None, :, :, None] if with_shade else 1.0)
    rgbp = og.unpack(rgb, p[3][None], bg.double()[None])
    l_rgb, l_mask = og.l1_losses(rgbp, p[3][None], gt.double()[None], gm.double()[None])
    (1.0 * l_rgb + 5.0 * l_mask).backward()
    hp = pred.cuda().requires_grad_()
    hs = shade.cuda().requires_grad_() if with_shade else None
    total, hr, hm = losses.l1_photometric(hp, gt.cuda(), gm.cuda(), bg.cuda(), hs, 1.0, 5.0)
    assert abs(hr.item() - l_rgb.item()) < 1e-6 and abs(hm.item() - l_mask.item()) < 1e-6
    (total * 2.0).backward()
    np.testing.assert_allclose(hp.grad.cpu().numpy() / 2.0, p.grad.numpy(), atol=1e-9, rtol=1e-5)
    if with_shade:
        np.testing.assert_allclose(hs.grad.cpu().numpy() / 2.0, s.grad.numpy(), atol=1e-9, rtol=1e-5)
