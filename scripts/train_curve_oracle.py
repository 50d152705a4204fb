#!/usr/bin/env python
"""TEST INFRASTRUCTURE (imports oracle/): the first N iterations of BASELINE configs[1] trained by the CPU ORACLE (oracle/train_step.py, float64,
every loss term of exps/zju-mocap_377.yaml INCLUDING LPIPS on the seeded trunk) from the initialisation scripts/train_synthetic.py starts from
-> profiles/<tag>_train_curve_oracle.json: per iteration every loss term, the total and the 8-bit PSNR of the prediction against the target.
The HIP run's curve (scripts/train_synthetic.py -> profiles/<tag>_train_curve.json) lies beside it.  CPU only, ~10 s per iteration on 8 cores.

    python scripts/train_curve_oracle.py [iterations = 150] [tag = r05]"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import train_curve_common as C                                   # noqa: E402
from gomavatar_amd.lpips import seeded_trunk                     # noqa: E402  (seeded weights only)
from oracle import geometry as og, raster as orast, train_step as ots   # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    tag = sys.argv[2] if len(sys.argv) > 2 else "r05"
    orast.set_threads(os.cpu_count() or 1); torch.set_num_threads(os.cpu_count() or 1)
    ots.LR.update({k: v for k, v in C.LR.items() if k in ots.LR}); ots.LR_DECAY_STEPS = C.LR_DECAY_STEPS
    body, tp, sp, twb, swb = C.setup(0)
    teacher, student = ots.OracleAvatar(body, C.IMG, tp, twb), ots.OracleAvatar(body, C.IMG, sp, swb)
    teacher.tiled_mesh = student.tiled_mesh = True
    trunk = seeded_trunk(0)
    lin = np.load(os.path.join(ROOT, "gomavatar_amd", "data", "lpips_vgg_lin_v0.1.npz"))
    lins = [torch.from_numpy(lin[f"lin{k}"]) for k in range(5)]
    opt = torch.optim.Adam(student.param_groups(), betas=(0.9, 0.999))
    targets, log, t0 = {}, [], time.time()
    out_path = os.path.join(ROOT, "profiles", f"{tag}_train_curve_oracle.json")
    for it in range(n):
        fr = {k: torch.from_numpy(v) for k, v in C.frame(it).items()}
        key = it % C.N_VIEWS
        if key not in targets:
            with torch.no_grad():
                rgbs, masks, _ = teacher.forward(fr, training=False)
                targets[key] = (og.unpack(rgbs, masks, fr["bgcolor"].double()).clamp(0, 1), masks.clone())
        gt_rgb, gt_mask = targets[key]
        opt.zero_grad(set_to_none=True)
        rgbs, masks, o = student.forward(fr)
        rgb = og.unpack(rgbs, masks, fr["bgcolor"].double())
        total, L = student.compute_loss(rgb, masks, o, gt_rgb, gt_mask, lpips_trunk=trunk, lpips_lins=lins)
        total.backward()
        opt.step()
        ots.update_lr(opt, it + 1)                               # train.py:341: n_iters counts from 1
        log.append({"iter": it + 1, "total": float(total.detach()), "psnr": round(ots.psnr_8bit(rgb.detach()[0], gt_rgb[0]), 4),
                    **{k: float(v.detach()) for k, v in L.items()}})
        print(log[-1], f"({time.time() - t0:.0f} s)", flush=True)
        if (it + 1) % 10 == 0 or it + 1 == n:
            json.dump({"what": "CPU oracle (oracle/train_step.py, float64) training BASELINE configs[1] from the initialisation of scripts/train_synthetic.py: "
                               "13 776 Gaussians, 512 x 512, 8 views, every loss term incl. LPIPS (seeded trunk), torch.optim.Adam + update_lr",
                       "iterations": it + 1, "seconds": round(time.time() - t0, 1), "cpu_threads": os.cpu_count(), "log": log}, open(out_path, "w"), indent=0)


if __name__ == "__main__":
    main()
