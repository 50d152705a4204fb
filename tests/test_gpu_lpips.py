"""LPIPS-VGG on the GPU: fused HIP head (csrc/lpips.hip) + library convolutions,
against the golden of the reference class and the CPU oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import lpips as ol

pytestmark = pytest.mark.gpu


def test_head_kernel_matches_formula_forward_and_backward():
    from gomavatar_amd.lpips import _LpipsHead
    g = torch.Generator().manual_seed(1)
    for (B, C, H, W) in ((2, 64, 17, 23), (1, 512, 8, 8), (3, 128, 32, 32)):
        f0 = torch.randn(B, C, H, W, generator=g).abs()
        f1 = torch.randn(B, C, H, W, generator=g).abs()
        f0[:, :, 0, 0] = 0.0                                   # an all-zero feature column: the eps path
        w = torch.rand(C, generator=g)
        a = f0.double().requires_grad_()
        d = (ol.normalize_tensor(a) - ol.normalize_tensor(f1.double())) ** 2
        ref = (d * w.double().view(1, -1, 1, 1)).sum(1).mean([1, 2])
        coef = torch.arange(1, B + 1, dtype=torch.float64)
        (ref * coef).sum().backward()
        x = f0.cuda().requires_grad_()
        out = _LpipsHead.apply(x, f1.cuda(), w.cuda())
        (out * coef.float().cuda()).sum().backward()
        np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), rtol=2e-5, atol=1e-7)
        gr, gg = a.grad.numpy(), x.grad.cpu().numpy()
        finite = np.isfinite(gr)
        assert np.abs(gg[finite] - gr[finite]).max() <= 2e-5 * np.abs(gr[finite]).max()


def test_lpips_matches_reference_golden_and_oracle_gradient(golden_dir):
    from gomavatar_amd.lpips import LPIPS, seeded_trunk
    g = np.load(os.path.join(golden_dir, "lpips_vgg.npz"))
    m = LPIPS(net="vgg", trunk_seed=int(g["trunk_seed"]))
    in0 = torch.from_numpy(g["in0"]).cuda().requires_grad_()
    in1 = torch.from_numpy(g["in1"]).cuda()
    val, res = m(in0, in1, retPerLayer=True)
    assert val.shape == (2, 1, 1, 1)
    np.testing.assert_allclose(val.detach().cpu().numpy(), g["val"], rtol=3e-4)      # fp32 convolutions: library algorithm differs
    for k in range(1, 5):   # (the reference's res[0] aliases the total, see tests/test_oracle_lpips.py)
        np.testing.assert_allclose(res[k].detach().cpu().numpy(), g[f"res{k}"], rtol=5e-4)
    val.sum().backward()
    x = torch.from_numpy(g["in0"]).double().requires_grad_()
    lins = [l.cpu() for l in m.lins]
    ol.lpips_vgg(x, torch.from_numpy(g["in1"]).double(), seeded_trunk(int(g["trunk_seed"])), lins).sum().backward()
    gr, gg = x.grad.numpy(), in0.grad.cpu().numpy()
    assert np.abs(gg - gr).max() <= 2e-3 * np.abs(gr).max()
    assert np.median(np.abs(gg - gr)) <= 1e-5 * np.abs(gr).max()


def test_train_call_pattern_and_bf16_trunk():
    """train.py:113-117: (B,H,W,3) images in [0,1]; the bf16 trunk is an option, within 2 % of fp32."""
    from gomavatar_amd.lpips import LPIPS, lpips_loss
    g = torch.Generator().manual_seed(2)
    pred = torch.rand(1, 64, 64, 3, generator=g).cuda().requires_grad_()
    gt = torch.rand(1, 64, 64, 3, generator=g).cuda()
    l32 = lpips_loss(LPIPS(trunk_seed=3), pred, gt)
    l32.backward()
    assert pred.grad is not None and torch.isfinite(pred.grad).all() and float(pred.grad.abs().max()) > 0
    l16 = lpips_loss(LPIPS(trunk_seed=3, trunk_dtype=torch.bfloat16), pred.detach(), gt)
    assert abs(float(l16) - float(l32.detach())) <= 0.02 * float(l32.detach())
