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


@pytest.mark.parametrize("H,W,k,with_nm", [(64, 64, 7, True), (37, 91, 5, True), (48, 40, 7, False), (33, 35, 1, True)])
def test_l1_terms_match_torch_formulas(H, W, k, with_nm):
    """csrc/loss.hip gom_l1_terms_*: mean|rgb - gt|, mean|mask - gt|, mean|normal_mask - maxpool_k(gt mask)| (train.py:101-111,
    141-149) against torch, values and gradients; then compute_loss's total / dict against the term-by-term composition."""
    import torch.nn.functional as F
    from types import SimpleNamespace as NS
    from gomavatar_amd.losses import l1_terms
    from gomavatar_amd.train_util import compute_loss
    g = torch.Generator().manual_seed(H * 100 + W)
    rgb, rgb_gt = torch.rand(1, H, W, 3, generator=g), torch.rand(1, H, W, 3, generator=g)
    mask, mask_gt = torch.rand(1, H, W, generator=g), (torch.rand(1, H, W, generator=g) > 0.7).float()
    nm = torch.rand(1, H, W, 1, generator=g)
    rgb[0, :3, :3] = rgb_gt[0, :3, :3]                      # exact zeros: sign(0) = 0 like torch's abs backward
    coeff = torch.tensor([1.3, 0.7, 2.1])
    def ref(r, m, n):
        dil = F.max_pool2d(mask_gt.unsqueeze(1), kernel_size=k, stride=1, padding=k // 2).squeeze(1) if k > 1 else mask_gt
        out = [torch.mean(torch.abs(r - rgb_gt)), torch.mean(torch.abs(m - mask_gt))]
        out.append(torch.mean(torch.abs(n[..., 0] - dil)) if with_nm else torch.zeros(()))
        return torch.stack(out)
    leaves_c = [x.clone().requires_grad_() for x in (rgb, mask, nm)]
    vr = ref(*leaves_c); (vr * coeff).sum().backward()
    leaves_g = [x.cuda().requires_grad_() for x in (rgb, mask, nm)]
    vg = l1_terms(leaves_g[0], rgb_gt.cuda(), leaves_g[1], mask_gt.cuda(), leaves_g[2][..., 0] if with_nm else None, k if with_nm else 0)
    (vg * coeff.cuda()).sum().backward()
    assert torch.allclose(vg.cpu(), vr, rtol=2e-6, atol=1e-7), (vg.cpu(), vr)
    for a, b in zip(leaves_g[:3 if with_nm else 2], leaves_c):
        assert torch.allclose(a.grad.cpu(), b.grad, rtol=1e-6, atol=1e-12)
    # compute_loss on device tensors: same dict keys / order as the reference, total = sum of the scaled entries
    cfg = NS(rgb=NS(coeff=1.0), mask=NS(coeff=5.0), lpips=NS(coeff=0.0), laplacian=NS(coeff_canonical=0.0, coeff_observation=0.0),
             normal=NS(coeff_mask=1.5 if with_nm else 0.0, kernel_size=max(k, 1), coeff_consist=0.0), color_consist=NS(coeff=0.0))
    lg = [x.detach().clone().requires_grad_() for x in leaves_g]
    total, losses = compute_loss(lg[0], lg[1], {"normal_mask": lg[2][..., 0]}, rgb_gt.cuda(), mask_gt.cuda(), cfg)
    total.backward()
    assert list(losses) == (["rgb", "mask", "normal_mask"] if with_nm else ["rgb", "mask"])
    exp = vr[0] * 1.0 + vr[1] * 5.0 + (vr[2] * 1.5 if with_nm else 0.0)
    assert abs(float(total.detach()) - float(exp.detach())) <= 2e-6 * abs(float(exp.detach()))
    assert abs(float(losses["mask"]["scaled"].detach()) - 5.0 * float(vr[1].detach())) <= 1e-5
    assert abs(float(losses["rgb"]["unscaled"].detach()) - float(vr[0].detach())) <= 1e-6
    assert torch.allclose(lg[1].grad.cpu(), leaves_c[1].grad * (5.0 / 0.7), rtol=1e-5, atol=1e-12)


@pytest.mark.parametrize("H,W,with_shade", [(40, 56, True), (33, 31, False)])
def test_compose_and_unpack_kernels_match_the_torch_chains(H, W, with_shade):
    """csrc/loss.hip gom_compose_* (model.py:262-287: albedo / mask views of the splat image, rgb = albedo * shading) and gom_unpack_*
    (train.py:53-55) against the slice / permute / multiply chains they replace, values and gradients."""
    from gomavatar_amd.losses import compose
    from gomavatar_amd.train_util import unpack
    g = torch.Generator().manual_seed(H + W)
    img, shade, bg = torch.rand(4, H, W, generator=g), torch.rand(1, H, W, 1, generator=g) * 2, torch.rand(1, 3, generator=g)
    w = [torch.randn(1, H, W, 3, generator=g), torch.randn(1, H, W, generator=g), torch.randn(1, H, W, 3, generator=g)]
    def chain(im, sh, dev):
        albedos, masks = im[:3].permute(1, 2, 0)[None], im[3][None]
        rgbs = albedos * sh if sh is not None else albedos
        out = rgbs * masks.unsqueeze(-1) + bg.to(dev)[:, None, None, :] * (1 - masks).unsqueeze(-1)     # unpack, the torch formula
        return albedos, masks, rgbs, out
    ic, sc = img.clone().requires_grad_(), (shade.clone().requires_grad_() if with_shade else None)
    a, m, r, o = chain(ic, sc, "cpu")
    ((a * w[0]).sum() + (m * w[1]).sum() + (o * w[2]).sum()).backward()
    ig, sg = img.cuda().requires_grad_(), (shade.cuda().requires_grad_() if with_shade else None)
    a2, m2, r2 = compose(ig, sg)
    o2 = unpack(r2, m2, bg.cuda())
    ((a2 * w[0].cuda()).sum() + (m2 * w[1].cuda()).sum() + (o2 * w[2].cuda()).sum()).backward()
    for x, y in ((a2, a), (m2, m), (r2, r), (o2, o)):
        assert x.shape == y.shape and torch.allclose(x.detach().cpu(), y.detach(), rtol=1e-6, atol=1e-7)
    assert torch.allclose(ig.grad.cpu(), ic.grad, rtol=1e-5, atol=1e-6)
    if with_shade:
        assert torch.allclose(sg.grad.cpu(), sc.grad, rtol=1e-5, atol=1e-6)


@pytest.mark.gpu
def test_loss_tail_is_the_sum_per_term_the_scaling_and_the_total_with_their_gradients():
    """losses.loss_tail (csrc/loss.hip gom_loss_tail): partial-sum matrices -> (terms, terms * coefficients, total) in one launch, against the
    torch ops it replaces (a sum per term, cat, multiply, sum) in float64; the gradient of `total` w.r.t. every partial sum is its term's
    coefficient times the factor in front of the row sum, zero on rows that are no terms."""
    from gomavatar_amd.losses import loss_tail
    g = torch.Generator().manual_seed(3)
    mats = [torch.rand(3, 1, generator=g), torch.rand(1, 1280, generator=g), torch.rand(1, 256, generator=g), torch.rand(2, 77, generator=g)]
    used, pre = [2, 1, 1, 2], [1.0, 0.5, 1.0, 1.0]
    coeffs = torch.tensor([1.0, 5.0, 1.0, 10.0, 0.1, 0.05], device="cuda")
    ins = [m.cuda().requires_grad_(True) for m in mats]
    vec, scaled, total = loss_tail(coeffs, ins, used, pre)
    ref_terms = torch.cat([(m.double()[:u].sum(1) * p) for m, u, p in zip(mats, used, pre)])
    ref_scaled = ref_terms * coeffs.cpu().double()
    assert vec.shape == (6,) and total.dim() == 0
    assert float((vec.cpu().double() - ref_terms).abs().max()) <= 1e-6 * float(ref_terms.abs().max())
    assert float((scaled.cpu().double() - ref_scaled).abs().max()) <= 1e-6 * float(ref_scaled.abs().max())
    assert abs(float(total) - float(ref_scaled.sum())) <= 1e-6 * float(ref_scaled.sum())
    total.backward()
    k = 0
    for x, m, u, p in zip(ins, mats, used, pre):
        want = torch.zeros_like(m)
        for r in range(u):
            want[r] = float(coeffs[k + r]) * p
        k += u
        assert torch.allclose(x.grad.cpu(), want, rtol=1e-6, atol=0), (tuple(m.shape), x.grad.cpu()[:, :2], want[:, :2])
    # a gradient arriving at the TERMS (somebody differentiates losses[name]["unscaled"]) reaches the partial sums as well
    ins2 = [m.cuda().requires_grad_(True) for m in mats]
    vec2, _, total2 = loss_tail(coeffs, ins2, used, pre)
    (total2 + 3.0 * vec2[3]).backward()
    assert torch.allclose(ins2[2].grad.cpu(), torch.full((1, 256), 10.0 + 3.0), rtol=1e-6)
    assert torch.allclose(ins2[1].grad.cpu(), torch.full((1, 1280), 0.5), rtol=1e-6)
