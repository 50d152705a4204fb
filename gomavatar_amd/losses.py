"""Photometric losses of the hot path, HIP-backed.

`l1_photometric` fuses the reference's `unpack` (train.py:53-55) with the L1
rgb and L1 mask terms of `compute_loss` (train.py:101-111) and their backward
into one kernel; `compute_loss_l1` keeps the reference's return structure
({'rgb': {'unscaled','scaled'}, 'mask': {...}}).  No CPU fallback.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib


class _L1Photometric(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, shade, gt_rgb, gt_mask, bg, c_rgb, c_mask):
        lib = _lib.load()
        C, H, W = pred.shape
        assert C == 4, "pred must be the rasterizer's (4,H,W) output: albedo rgb + alpha"
        p = pred.contiguous()
        s = shade.contiguous() if shade is not None else None
        dpred = torch.empty_like(p)
        dshade = torch.empty_like(s) if s is not None else None
        partials = torch.empty((_lib.GOM_LOSS_BLOCKS, 2), dtype=torch.float32, device=p.device)
        gt_rgb, gt_mask, bg = gt_rgb.contiguous(), gt_mask.contiguous(), bg.contiguous().float()  # alive until enqueued
        _lib.check(lib.gom_l1_loss(H, W, _lib.ptr(p), _lib.ptr(s), _lib.ptr(gt_rgb), _lib.ptr(gt_mask),
                                   _lib.ptr(bg), float(c_rgb), float(c_mask), 1.0, _lib.ptr(dpred), _lib.ptr(dshade),
                                   _lib.ptr(partials), _lib.stream_ptr()))
        sums = partials.sum(0)
        l_rgb = sums[0] / (3.0 * H * W)
        l_mask = sums[1] / float(H * W)
        ctx.save_for_backward(dpred, dshade if dshade is not None else torch.empty(0, device=p.device))
        ctx.has_shade = s is not None
        total = c_rgb * l_rgb + c_mask * l_mask
        ctx.mark_non_differentiable(l_rgb, l_mask)
        return total, l_rgb, l_mask

    @staticmethod
    def backward(ctx, g_total, _g_rgb, _g_mask):
        dpred, dshade = ctx.saved_tensors
        return dpred * g_total, (dshade * g_total if ctx.has_shade else None), None, None, None, None, None


def l1_photometric(pred_chw: torch.Tensor, gt_rgb: torch.Tensor, gt_mask: torch.Tensor, bgcolor: torch.Tensor,
                   shade: Optional[torch.Tensor] = None, c_rgb: float = 1.0, c_mask: float = 5.0):
    """pred_chw (4,H,W) = rasterizer output (albedo rgb, alpha); gt_rgb (H,W,3);
    gt_mask (H,W); bgcolor (3,); shade (H,W) optional shading factor.
    Returns (c_rgb*L_rgb + c_mask*L_mask, L_rgb, L_mask)."""
    if not pred_chw.is_cuda:
        raise RuntimeError("gomavatar_amd.losses: tensors must be on the HIP device (no CPU fallback)")
    return _L1Photometric.apply(pred_chw, shade, gt_rgb, gt_mask, bgcolor.reshape(-1)[:3], c_rgb, c_mask)


def compute_loss_l1(pred_chw, gt_rgb, gt_mask, bgcolor, loss_cfg=None, shade=None):
    """The rgb + mask entries of the reference's `compute_loss` dict."""
    c_rgb = loss_cfg.rgb.coeff if loss_cfg is not None else 1.0
    c_mask = loss_cfg.mask.coeff if loss_cfg is not None else 5.0
    total, l_rgb, l_mask = l1_photometric(pred_chw, gt_rgb, gt_mask, bgcolor, shade, c_rgb, c_mask)
    losses = {"rgb": {"unscaled": l_rgb, "scaled": l_rgb * c_rgb}, "mask": {"unscaled": l_mask, "scaled": l_mask * c_mask}}
    return total, losses


class _L1Terms(torch.autograd.Function):
    """The three mean-|a - b| terms of compute_loss (rgb, mask, normal mask vs the dilated target mask) on unpacked images:
    csrc/loss.hip gom_l1_terms_forward / _backward.  Returns a (3,) tensor; a term whose prediction is None is 0."""

    @staticmethod
    def forward(ctx, rgb, rgb_gt, mask, mask_gt, normal_mask, dil_k):
        lib = _lib.load()
        H, W = mask_gt.shape[-2:]
        keep = [None if x is None else x.float().contiguous() for x in (rgb, rgb_gt, mask, mask_gt, normal_mask)]
        r, rg, m, mg, nm = keep
        out = torch.empty(3, dtype=torch.float32, device=mg.device)
        partials = torch.empty(4 * _lib.GOM_LOSS_BLOCKS * 3, dtype=torch.float32, device=mg.device)
        _lib.check(lib.gom_l1_terms_forward(H, W, _lib.ptr(r), _lib.ptr(rg), _lib.ptr(m), _lib.ptr(mg), _lib.ptr(nm), int(dil_k), _lib.ptr(out),
                                            _lib.ptr(partials), _lib.stream_ptr()))
        ctx.save_for_backward(*keep)
        ctx.dil_k, ctx.hw = int(dil_k), (H, W)
        ctx.shapes = [None if x is None else (x.shape, x.dtype) for x in (rgb, mask, normal_mask)]
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        r, rg, m, mg, nm = ctx.saved_tensors
        g = g.float().contiguous()
        d = [None if x is None else torch.empty_like(x) for x in (r, m, nm)]
        _lib.check(lib.gom_l1_terms_backward(ctx.hw[0], ctx.hw[1], _lib.ptr(r), _lib.ptr(rg), _lib.ptr(m), _lib.ptr(mg), _lib.ptr(nm), ctx.dil_k,
                                             _lib.ptr(g), _lib.ptr(d[0]), _lib.ptr(d[1]), _lib.ptr(d[2]), _lib.stream_ptr()))
        d = [None if x is None else x.reshape(s[0]).to(s[1]) for x, s in zip(d, ctx.shapes)]
        return d[0], None, d[1], None, d[2], None


def l1_terms(rgb_pred, rgb_gt, mask_pred, mask_gt, normal_mask=None, dilate: int = 0) -> torch.Tensor:
    """(3,) = mean|rgb_pred - rgb_gt|, mean|mask_pred - mask_gt|, mean|normal_mask - maxpool_dilate(mask_gt)| for ONE image
    ((1,H,W,3) / (1,H,W) tensors on the HIP device); value and gradients as torch's abs().mean() (sign(0) = 0)."""
    if not mask_gt.is_cuda:
        raise RuntimeError("gomavatar_amd.losses: tensors must be on the HIP device (no CPU fallback)")
    return _L1Terms.apply(rgb_pred, rgb_gt, mask_pred, mask_gt, normal_mask, dilate)


class _Compose(torch.autograd.Function):
    """(4,H,W) rasterizer output (+ shading (1,H,W,1)) -> albedo (1,H,W,3), mask (1,H,W), rgb = albedo * shading (1,H,W,3)."""

    @staticmethod
    def forward(ctx, img, shade):
        lib = _lib.load()
        _, H, W = img.shape
        im = img.float().contiguous()
        sh = shade.float().contiguous() if shade is not None else None
        albedo = torch.empty(1, H, W, 3, dtype=torch.float32, device=im.device)
        mask = torch.empty(1, H, W, dtype=torch.float32, device=im.device)
        rgb = torch.empty_like(albedo) if sh is not None else None
        _lib.check(lib.gom_compose_forward(H, W, _lib.ptr(im), _lib.ptr(sh), _lib.ptr(albedo), _lib.ptr(mask), _lib.ptr(rgb), _lib.stream_ptr()))
        ctx.save_for_backward(im, sh)
        ctx.set_materialize_grads(False)    # (an output nobody differentiates -- the albedo -- arrives as None, not as a zero image filled for it)
        ctx.shade_shape = None if shade is None else shade.shape
        if rgb is None:
            return albedo, mask
        return albedo, mask, rgb

    @staticmethod
    def backward(ctx, d_albedo, d_mask, d_rgb=None):
        im, sh = ctx.saved_tensors
        lib = _lib.load()
        _, H, W = im.shape
        keep = [None if g is None else g.float().contiguous() for g in (d_albedo, d_mask, d_rgb)]
        d_img = torch.empty_like(im)
        d_shade = torch.empty_like(sh) if sh is not None and ctx.needs_input_grad[1] else None
        _lib.check(lib.gom_compose_backward(H, W, _lib.ptr(im), _lib.ptr(sh), _lib.ptr(keep[0]), _lib.ptr(keep[1]), _lib.ptr(keep[2]), _lib.ptr(d_img),
                                            _lib.ptr(d_shade), _lib.stream_ptr()))
        return d_img, (d_shade.reshape(ctx.shade_shape) if d_shade is not None else None)


def compose(img_chw: torch.Tensor, shade: Optional[torch.Tensor] = None):
    """model.py:262-287 around the splat image: (albedos (1,H,W,3), masks (1,H,W), rgbs (1,H,W,3) = albedos * shade); without a
    shading the rgbs ARE the albedos."""
    if not img_chw.is_cuda:
        raise RuntimeError("gomavatar_amd.losses: tensors must be on the HIP device (no CPU fallback)")
    out = _Compose.apply(img_chw, shade)
    return out if len(out) == 3 else (out[0], out[1], out[0])


class _Unpack(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rgbs, masks, bg):
        lib = _lib.load()
        B, H, W, _ = rgbs.shape
        r, m, k = rgbs.float().contiguous(), masks.float().contiguous(), bg.float().contiguous()
        out = torch.empty_like(r)
        _lib.check(lib.gom_unpack_forward(B, H, W, _lib.ptr(r), _lib.ptr(m), _lib.ptr(k), _lib.ptr(out), _lib.stream_ptr()))
        ctx.save_for_backward(r, m, k)
        return out

    @staticmethod
    def backward(ctx, g):
        r, m, k = ctx.saved_tensors
        lib = _lib.load()
        B, H, W, _ = r.shape
        g = g.float().contiguous()
        d_r, d_m = torch.empty_like(r), torch.empty_like(m)
        _lib.check(lib.gom_unpack_backward(B, H, W, _lib.ptr(r), _lib.ptr(m), _lib.ptr(k), _lib.ptr(g), _lib.ptr(d_r), _lib.ptr(d_m), _lib.stream_ptr()))
        return d_r, d_m, None


def unpack_fused(rgbs, masks, bgcolors):
    """train.py:53-55 as one kernel each way: rgbs (B,H,W,3), masks (B,H,W), bgcolors (B,3)."""
    return _Unpack.apply(rgbs, masks, bgcolors)


class _LossTail(torch.autograd.Function):
    """compute_loss's tail (train.py:98-163) as ONE launch each way (csrc/loss.hip gom_loss_tail): the terms' partial sums -> (vector of terms,
    vector * coefficients, total).  inputs: 2-D fp32 tensors of partial sums; the first used[i] rows of input i are terms, term = pre[i] * row sum.
    Backward: d input_i[r, :] = pre[i] * (g_vec + coeff * (g_scaled + g_total)) of its term, as expanded views of one small vector."""

    @staticmethod
    def forward(ctx, coeffs, used, pre, *inputs):
        import ctypes
        lib = _lib.load()
        n = len(inputs)
        keep = [x.float().contiguous() for x in inputs]
        assert all(x.dim() == 2 for x in keep) and len(used) == n and len(pre) == n
        K = int(sum(used))
        dev = keep[0].device
        vec, scaled = torch.empty(K, dtype=torch.float32, device=dev), torch.empty(K, dtype=torch.float32, device=dev)
        total = torch.empty((), dtype=torch.float32, device=dev)
        ptrs = (ctypes.c_void_p * n)(*[x.data_ptr() for x in keep])
        rows = (ctypes.c_int32 * n)(*[x.shape[0] for x in keep])
        cols = (ctypes.c_int32 * n)(*[x.shape[1] for x in keep])
        usd = (ctypes.c_int32 * n)(*[int(u) for u in used])
        pr = (ctypes.c_float * n)(*[float(p) for p in pre])
        _lib.check(lib.gom_loss_tail(n, ptrs, rows, cols, usd, pr, _lib.ptr(coeffs), _lib.ptr(vec), _lib.ptr(scaled), _lib.ptr(total), _lib.stream_ptr()))
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(coeffs)
        ctx.meta = ([tuple(x.shape) for x in keep], tuple(int(u) for u in used), tuple(float(p) for p in pre), [x.dtype for x in inputs])
        return vec, scaled, total

    @staticmethod
    def backward(ctx, g_vec, g_scaled, g_total):
        (coeffs,) = ctx.saved_tensors
        shapes, used, pre, dtypes = ctx.meta
        g = None                                        # d L / d term, per term
        if g_total is not None:
            g = coeffs * g_total
        if g_scaled is not None:
            g = coeffs * g_scaled if g is None else g + coeffs * g_scaled
        if g_vec is not None:
            g = g_vec if g is None else g + g_vec
        out, k = [], 0
        for shp, u, p, dt in zip(shapes, used, pre, dtypes):
            if g is None:
                out.append(None)
            else:
                gi = g[k:k + u] if p == 1.0 else g[k:k + u] * p
                if u < shp[0]:                           # (rows that are no terms)
                    gi = torch.cat([gi, gi.new_zeros(shp[0] - u)])
                out.append(gi.unsqueeze(1).expand(shp).to(dt))
            k += u
        return (None, None, None, *out)


def loss_tail(coeffs: torch.Tensor, inputs, used, pre):
    """-> (terms (K,), terms * coeffs (K,), total ()), differentiable w.r.t. the partial-sum matrices `inputs`."""
    return _LossTail.apply(coeffs, tuple(used), tuple(pre), *inputs)
