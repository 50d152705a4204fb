#!/usr/bin/env python
"""The three frames/s figures SURVEY.md 8(d) asks for, at the metric workload (55 104 Gaussians, 512x512), single frame
at a time (batch 1, the reference's own semantics) unless stated:

  (i)   raster only, fwd+bwd:  one fused 4-channel pass  |  the reference's pattern: two 3-channel calls (gaussian.py:77-94)
  (ii)  render path fwd+bwd = FK + LBS + face Gaussians + raster + L1 losses (bench.py's step), batch 1 and batch 8
  (iii) (ii) + LPIPS-VGG (library fp32 / bf16 trunk, hand-written bf16 MFMA trunk) + Adam on the four parameter tensors
        [the mesh / shadow branch of the whole Model step: scripts/train_synthetic.py]

Prints one JSON object.  Not the graded benchmark (that is bench.py)."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gomavatar_amd import synthetic as syn, rasterizer as R, _lib  # noqa: E402
from gomavatar_amd.pipeline import RenderStep  # noqa: E402
from gomavatar_amd.lpips import LPIPS, lpips_loss  # noqa: E402
from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer  # noqa: E402


def timeit(fn, n=200, warm=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return n / (time.perf_counter() - t0)


def main():
    img, dev = 512, torch.device("cuda", 0)
    body = syn.make_body(1)
    N, F = body["canonical_vertex"].shape[0], body["faces"].shape[0]
    w = torch.from_numpy(body["canonical_lbs_weights"]).T
    w25 = torch.cat([w, torch.zeros(1, N)], 0).contiguous()
    faces = torch.from_numpy(body["faces"])
    gp = syn.make_gaussian_params(F, 1)
    params = dict(vertices=torch.from_numpy(body["canonical_vertex"]).T.contiguous().to(dev), so3=torch.from_numpy(gp["so3"]).to(dev),
                  scale=torch.from_numpy(gp["scale"]).to(dev), appearance=torch.from_numpy(gp["appearance"]).to(dev))
    fr = syn.make_frame(0, img)
    frame = {k: torch.from_numpy(fr[k][0]).contiguous().to(dev) for k in ("cnl_gtfms", "dst_Rs", "dst_Ts")}
    bg = torch.from_numpy(fr["bgcolor"][0]).to(dev)
    out = {}
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        step = RenderStep(faces, N, (img, img), w25, device=dev)
        step.set_camera(fr["K"][0], fr["E"][0])
        gt_rgb = torch.rand(img, img, 3, device=dev)
        gt_mask = (torch.rand(img, img, device=dev) > 0.5).float()
        step.forward_backward(params, frame, gt_rgb, gt_mask, bg)
        torch.cuda.synchronize()
        # (ii) render path
        out["render_path_batch1_fps"] = round(timeit(lambda: step.forward_backward(params, frame, gt_rgb, gt_mask, bg, graph=True)), 1)
        B = 8
        stepB = RenderStep(faces, N, (img, img), w25, device=dev, batch=B)
        stepB.set_cameras([fr["K"][0]] * B, [fr["E"][0]] * B)
        frB = {k: v[None].repeat(B, *([1] * v.dim())).contiguous() for k, v in frame.items()}
        gB, mB, bB = gt_rgb[None].repeat(B, 1, 1, 1).contiguous(), gt_mask[None].repeat(B, 1, 1).contiguous(), bg[None].repeat(B, 1).contiguous()
        out["render_path_batch8_fps"] = round(B * timeit(lambda: stepB.forward_backward(params, frB, gB, mB, bB, graph=True), n=100, warm=10), 1)
        # (i) raster only, on the Gaussians of that frame
        xyz, cov6, feat, op = step.xyz.clone(), step.cov6.clone(), step.feat.clone(), step.opacity.clone()
        cam = step.cam
        wimg = torch.randn(4, img, img, device=dev)

        def raster4():
            a = [t.detach().requires_grad_() for t in (xyz, cov6, feat)]
            o, _ = R.rasterize(a[0], a[1], a[2], op, cam)
            (o * wimg).sum().backward()
        out["raster_only_fused4_fps"] = round(timeit(raster4, n=100, warm=10), 1)
        rs = GaussianRasterizationSettings(image_height=img, image_width=img, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, bg=torch.zeros(4, device=dev),
                                           scale_modifier=1.0, viewmatrix=torch.tensor(list(cam.view), device=dev).view(4, 4),
                                           projmatrix=torch.tensor(list(cam.proj), device=dev).view(4, 4), sh_degree=0,
                                           campos=torch.zeros(3, device=dev), prefiltered=False, debug=False)
        rast = GaussianRasterizer(None)
        rast.raster_settings = rs
        feat6 = torch.cat([feat, feat[:, :2]], -1)

        def raster2x3():
            a = [t.detach().requires_grad_() for t in (xyz, cov6, feat6)]
            m2d = torch.zeros_like(a[0], requires_grad=True)
            outs = [rast(means3D=a[0], means2D=m2d, colors_precomp=a[2][:, i:i + 3], shs=None, opacities=op[:, None], scales=None, rotations=None,
                         cov3D_precomp=a[1])[0] for i in (0, 3)]
            (torch.cat(outs, 0)[:4] * wimg).sum().backward()
        out["raster_only_reference_2x3_fps"] = round(timeit(raster2x3, n=100, warm=10), 1)

        # (iii) + LPIPS + Adam (autograd glue between the HIP pipeline pieces and the conv trunk)
        from gomavatar_amd.geometry import MeshTopology, posed_face_gaussians
        from gomavatar_amd.losses import compute_loss_l1
        topo = MeshTopology(faces, N, device=dev)
        from gomavatar_amd.lpips import LPIPSMatrixCore
        for name, dt in (("fp32", torch.float32), ("bf16", torch.bfloat16), ("bf16_matrix_core", None)):
            lp = LPIPS(trunk_seed=0, trunk_dtype=dt, device=dev) if dt is not None else LPIPSMatrixCore(trunk_seed=0, device=dev, precision="bf16")
            P = {k: v.clone().requires_grad_() for k, v in params.items()}
            opt = torch.optim.Adam(list(P.values()), lr=1e-4)

            def full():
                opt.zero_grad(set_to_none=True)
                x, c6, _ = posed_face_gaussians(P["vertices"], P["so3"], P["scale"], frame["dst_Rs"], frame["dst_Ts"], frame["cnl_gtfms"], w25.to(dev), topo, 1e-3)
                f4 = torch.cat([P["appearance"].T, torch.ones(F, 1, device=dev)], 1)
                o, _ = R.rasterize(x, c6, f4, op, cam)
                total, _ = compute_loss_l1(o, gt_rgb, gt_mask, bg)          # c_rgb * L_rgb + c_mask * L_mask (train.py:101-111)
                rgb, mask = o[:3].permute(1, 2, 0), o[3]
                unpacked = rgb * mask[..., None] + bg * (1 - mask[..., None])          # train.py:53-55
                ll = lpips_loss(lp, unpacked[None], gt_rgb[None]) if dt is not None else lp.loss(unpacked[None], gt_rgb[None])
                loss = total + 1.0 * ll                                                 # train.py:113-121
                loss.backward()
                opt.step()
            try:
                out[f"full_step_lpips_{name}_adam_fps"] = round(timeit(full, n=30, warm=5), 1)
            except Exception as e:  # report, do not hide
                out[f"full_step_lpips_{name}_adam_fps"] = f"failed: {type(e).__name__}: {e}"
        # (iii) at batch 8 through the native path: forward half -> LPIPS on 8 unpacked images -> backward half -> Adam on the summed
        # gradients (RenderStep.lpips_hook); the same "one optimizer step on 8 frames" the 8-GPU frame-parallel run takes.
        lp8 = LPIPSMatrixCore(trunk_seed=0, device=dev, precision="bf16")
        P8 = {k: v.clone() for k, v in params.items()}
        for k, v in P8.items():
            v.grad = stepB.grads[k]
        opt8 = torch.optim.Adam(list(P8.values()), lr=1e-4)
        hook8 = stepB.lpips_hook(lp8, gB, bB, coeff=1.0)

        def full8():
            stepB.forward_backward(P8, frB, gB, mB, bB, graph=True, image_grad_hook=hook8)
            opt8.step()
        out["full_step_lpips_bf16_matrix_core_adam_batch8_fps"] = round(B * timeit(full8, n=30, warm=5), 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
