#!/usr/bin/env python
"""Records tests/golden/train_steps.npz: three seeded training steps (train.py:309-349) and one eval frame (eval.py:336-361)
of the CPU oracle (oracle/train_step.py, float64) on the synthetic S body (13 776 Gaussians / 6 890 verts) at 128 x 128 --
SURVEY.md 8(c)'s harness goldens.  Per step: every loss term, the checksums of the rendered rgb / mask, the gradient norm of
every parameter group, the parameter norms after the Adam step; for the eval frame the 8-bit PSNR against the teacher's render.

    python scripts/make_train_goldens.py        (CPU only, ~10 minutes; tests/test_gpu_train_golden.py replays it on the GPU)"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gomavatar_amd import synthetic as syn                    # noqa: E402  (seeded input generators + constant data only)
from gomavatar_amd.lpips import seeded_trunk                  # noqa: E402
from oracle import geometry as og, raster as orast, train_step as ots   # noqa: E402

IMG, STEPS = 128, 3


def shadow_weights(seed=3):
    g = torch.Generator().manual_seed(seed)
    dims = [(128, 39), (128, 128), (128, 128), (1, 128)]
    wb = []
    for i, (o, n) in enumerate(dims):
        bound = (6.0 / (o + n)) ** 0.5                         # xavier_uniform, as initseq does
        w = (torch.rand(o, n, generator=g, dtype=torch.float64) * 2 - 1) * bound
        if i == 3:
            w = torch.randn(o, n, generator=g, dtype=torch.float64) * 0.3     # a shading that actually varies (the reference starts at 1e-5)
        wb += [w, torch.zeros(o, dtype=torch.float64)]
    return wb


def params(body, seed, scale_mul=1.0):
    F = body["faces"].shape[0]
    gp = syn.make_gaussian_params(F, seed)
    return dict(vertices=torch.from_numpy(body["canonical_vertex"]).T.contiguous(), so3=torch.from_numpy(gp["so3"]),
                scale=torch.from_numpy(gp["scale"]) * scale_mul, appearance=torch.from_numpy(gp["appearance"]))


def main():
    orast.set_threads(os.cpu_count() or 1)
    body = syn.make_body(0)
    wb = shadow_weights()
    teacher = ots.OracleAvatar(body, IMG, params(body, 2), wb)
    student = ots.OracleAvatar(body, IMG, params(body, 1), wb)
    trunk = [t.double() for t in seeded_trunk(0)]
    lins = [torch.from_numpy(v).double() for v in np.load(os.path.join(ROOT, "gomavatar_amd", "data", "lpips_vgg_lin_v0.1.npz")).values()]
    frames = []
    for i in range(STEPS + 1):
        fr = {k: torch.from_numpy(v) for k, v in syn.make_frame(i, IMG).items()}
        with torch.no_grad():
            rgbs, masks, _ = teacher.forward(fr, training=False)
            fr["target_rgbs"] = og.unpack(rgbs, masks, fr["bgcolor"].double()).clamp(0, 1)
            fr["target_masks"] = masks.clone()
        frames.append(fr)
    opt = torch.optim.Adam(student.param_groups(), betas=(0.9, 0.999))
    out = {"img": np.int64(IMG), "steps": np.int64(STEPS), **{f"shadow_wb{i}": t.numpy() for i, t in enumerate(wb)}}
    for it in range(STEPS):
        t0 = time.time()
        fr = frames[it]
        opt.zero_grad(set_to_none=True)
        rgbs, masks, o = student.forward(fr)
        rgb = og.unpack(rgbs, masks, fr["bgcolor"].double())
        total, L = student.compute_loss(rgb, masks, o, fr["target_rgbs"], fr["target_masks"], trunk, lins)
        total.backward()
        for k, v in L.items():
            out[f"s{it}_loss_{k}"] = np.float64(v.detach())
        out[f"s{it}_loss_total"] = np.float64(total.detach())
        out[f"s{it}_rgb_mean"] = np.float64(rgb.detach().mean()); out[f"s{it}_mask_mean"] = np.float64(masks.detach().mean())
        out[f"s{it}_rgb_l2"] = np.float64(rgb.detach().norm()); out[f"s{it}_normal_mask_mean"] = np.float64(o["normal_mask"].detach().mean())
        for gi, g in enumerate(opt.param_groups):
            out[f"s{it}_gradnorm_{gi}_{g['name']}"] = np.float64(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in g["params"])))
        opt.step()
        ots.update_lr(opt, it)
        for gi, g in enumerate(opt.param_groups):
            out[f"s{it}_paramnorm_{gi}_{g['name']}"] = np.float64(torch.sqrt(sum((p.detach().double() ** 2).sum() for p in g["params"])))
        print(f"step {it}: total {float(total):.6f}  " + "  ".join(f"{k} {float(v):.3e}" for k, v in L.items()) + f"   ({time.time() - t0:.0f} s)", flush=True)
    # eval frame (eval.py:336-361): white background as cfg.bgcolor = [255, 255, 255] (configs/default.yaml), 8-bit PSNR against the teacher
    fr = frames[STEPS]
    white = torch.ones(1, 3, dtype=torch.float64)
    with torch.no_grad():
        rgbs, masks, _ = student.forward(fr, training=False)
        pred = og.unpack(rgbs, masks, white)
        t_rgbs, t_masks, _ = teacher.forward(fr, training=False)
        truth = og.unpack(t_rgbs, t_masks, white)
    out["eval_psnr"] = np.float64(ots.psnr_8bit(pred[0], truth[0]))
    out["eval_pred_8b_sum"] = np.int64(ots.to_8b(pred[0]).long().sum()); out["eval_truth_8b"] = ots.to_8b(truth[0]).numpy()
    print("eval PSNR", out["eval_psnr"])
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "train_steps.npz"), **out)


if __name__ == "__main__":
    main()
