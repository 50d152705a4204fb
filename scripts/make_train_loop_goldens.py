#!/usr/bin/env python
"""Records tests/golden/train_loop.npz: BASELINE configs[1] in miniature -- the reference's training loop (train.py:309-349) at
512 x 512 run by the CPU oracle (oracle/train_step.py, float64): N1 iterations on the S body (13 776 Gaussians), `subdivide()`
with the optimizer rebuilt as train.py:341-346 does (children 4f .. 4f+3 inherit so3 / scale / appearance, Adam moments reset),
then N2 iterations on the M body (55 104 Gaussians, the metric workload's size).  Every loss term of exps/zju-mocap_377.yaml
except LPIPS (coefficient 0 here: the float64 VGG trunk at 512 x 512 is what the oracle cannot afford; LPIPS has its own goldens).
Per iteration: loss terms, total, 8-bit PSNR of the prediction against the target frame (eval.py:101-104,355-361), image
checksums, gradient norm per parameter group, parameter norms after the step.  tests/test_gpu_train_loop.py replays it through
gomavatar_amd.model.Model + train_util.train_iteration on the GPU ("PSNR-vs-iteration of HIP vs oracle-trained").

    python scripts/make_train_loop_goldens.py [N1 N2]      (CPU only; the mesh branch uses the exact tile-culled evaluation)"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gomavatar_amd import synthetic as syn                    # noqa: E402  (seeded input generators only)
from oracle import geometry as og, raster as orast, train_step as ots   # noqa: E402
from make_train_goldens import shadow_weights, params         # noqa: E402

IMG = 512


def main():
    N1, N2 = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (20, 10)
    orast.set_threads(os.cpu_count() or 1)
    torch.set_num_threads(os.cpu_count() or 1)
    ots.LOSS["lpips"] = 0.0
    body = syn.make_body(0)
    wb = shadow_weights()
    teacher = ots.OracleAvatar(body, IMG, params(body, 2), wb)
    student = ots.OracleAvatar(body, IMG, params(body, 1), wb)
    teacher.tiled_mesh = student.tiled_mesh = True
    opt = torch.optim.Adam(student.param_groups(), betas=(0.9, 0.999))
    out = {"img": np.int64(IMG), "n1": np.int64(N1), "n2": np.int64(N2), **{f"shadow_wb{i}": t.numpy() for i, t in enumerate(wb)}}
    keys = ("rgb", "mask", "laplacian_observation", "normal_mask", "normal_consist", "color_consist")
    rec = {k: [] for k in keys + ("total", "psnr8", "rgb_mean", "mask_mean", "rgb_l2", "normal_mask_mean", "n_faces")}
    gn, pn = [], []
    for it in range(N1 + N2):
        t0 = time.time()
        fr = {k: torch.from_numpy(v) for k, v in syn.make_frame(it, IMG).items()}
        with torch.no_grad():
            rgbs, masks, _ = teacher.forward(fr, training=False)
            fr["target_rgbs"] = og.unpack(rgbs, masks, fr["bgcolor"].double()).clamp(0, 1)
            fr["target_masks"] = masks.clone()
        opt.zero_grad(set_to_none=True)
        rgbs, masks, o = student.forward(fr)
        rgb = og.unpack(rgbs, masks, fr["bgcolor"].double())
        total, L = student.compute_loss(rgb, masks, o, fr["target_rgbs"], fr["target_masks"])
        total.backward()
        for k in keys:
            rec[k].append(float(L[k].detach()))
        rec["total"].append(float(total.detach()))
        rec["psnr8"].append(ots.psnr_8bit(rgb.detach()[0], fr["target_rgbs"][0]))
        rec["rgb_mean"].append(float(rgb.detach().mean())); rec["mask_mean"].append(float(masks.detach().mean()))
        rec["rgb_l2"].append(float(rgb.detach().norm())); rec["normal_mask_mean"].append(float(o["normal_mask"].detach().mean()))
        rec["n_faces"].append(float(student.faces.shape[0]))
        gn.append([float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in g["params"]))) for g in opt.param_groups])
        opt.step()
        if it == N1 - 1:                                            # train.py:341-346 (cfg.model.subdivide_iters)
            student.subdivide()
            opt = torch.optim.Adam(student.param_groups(), betas=(0.9, 0.999))
        ots.update_lr(opt, it)
        pn.append([float(torch.sqrt(sum((p.detach().double() ** 2).sum() for p in g["params"]))) for g in opt.param_groups])
        print(f"iter {it}: F {student.faces.shape[0]} total {rec['total'][-1]:.6f} psnr8 {rec['psnr8'][-1]:.3f}  " +
              "  ".join(f"{k} {rec[k][-1]:.3e}" for k in keys) + f"   ({time.time() - t0:.0f} s)", flush=True)
    for k, v in rec.items():
        out[k] = np.asarray(v, np.float64)
    out["gradnorm"] = np.asarray(gn, np.float64); out["paramnorm"] = np.asarray(pn, np.float64)
    out["group_names"] = np.asarray([0, 1, 2, 2, 3], np.int64)      # appearance, canonical_geometry_xyz, canonical_geometry x 2, shadow
    # the closing eval frame (eval.py:336-361, white background) on the subdivided student
    fr = {k: torch.from_numpy(v) for k, v in syn.make_frame(N1 + N2, IMG).items()}
    white = torch.ones(1, 3, dtype=torch.float64)
    with torch.no_grad():
        rgbs, masks, _ = student.forward(fr, training=False)
        t_rgbs, t_masks, _ = teacher.forward(fr, training=False)
    out["eval_psnr"] = np.float64(ots.psnr_8bit(og.unpack(rgbs, masks, white)[0], og.unpack(t_rgbs, t_masks, white)[0]))
    print("eval PSNR", out["eval_psnr"])
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "train_loop.npz"), **out)


if __name__ == "__main__":
    main()
