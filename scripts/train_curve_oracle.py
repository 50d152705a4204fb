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


def from_state(path, n, tag, subdivide_at=1000):
    """ROUND 6: the oracle CONTINUES a HIP run -- parameters and Adam state saved by `scripts/train_synthetic.py --save-state-at K` -- for n iterations
    ACROSS the subdivision (train.py:330-346: subdivide at `subdivide_at`, optimizer rebuilt), every loss term incl. LPIPS, float64.
    -> profiles/<tag>_train_curve_oracle_from_hip_state.json: what "HIP vs oracle-trained" looks like after the subdivision, from the same state."""
    orast.set_threads(os.cpu_count() or 1); torch.set_num_threads(os.cpu_count() or 1)
    ots.LR.update({k: v for k, v in C.LR.items() if k in ots.LR}); ots.LR_DECAY_STEPS = C.LR_DECAY_STEPS
    st = np.load(path)
    k0 = int(st["iterations_done"])
    body, tp, sp, twb, swb = C.setup(0)
    sp = {k: torch.from_numpy(st[k]) for k in ("vertices", "so3", "scale", "appearance")}
    swb = [torch.from_numpy(st[f"shadow{i}"]) for i in range(8)]
    teacher, student = ots.OracleAvatar(body, C.IMG, tp, twb), ots.OracleAvatar(body, C.IMG, sp, swb)
    teacher.tiled_mesh = student.tiled_mesh = True
    trunk = seeded_trunk(0)
    lin = np.load(os.path.join(ROOT, "gomavatar_amd", "data", "lpips_vgg_lin_v0.1.npz"))
    lins = [torch.from_numpy(lin[f"lin{k}"]) for k in range(5)]
    opt = torch.optim.Adam(student.param_groups(), betas=(0.9, 0.999))
    names = {id(student.p[k]): k for k in ("vertices", "so3", "scale", "appearance")} | {id(t): f"shadow{i}" for i, t in enumerate(student.shadow)}
    for g in opt.param_groups:                                      # the HIP run's moments and step counts, in float64
        for p in g["params"]:
            nm = names[id(p)]
            opt.state[p] = {"step": torch.tensor(float(st[nm + ".step"])), "exp_avg": torch.from_numpy(st[nm + ".exp_avg"]).double().reshape(p.shape).clone(),
                            "exp_avg_sq": torch.from_numpy(st[nm + ".exp_avg_sq"]).double().reshape(p.shape).clone()}
    ots.update_lr(opt, k0)                                          # the rates iteration k0 + 1 runs with (set by update_lr at the end of iteration k0)
    targets, log, t0 = {}, [], time.time()
    out_path = os.path.join(ROOT, "profiles", f"{tag}_train_curve_oracle_from_hip_state.json")
    for it in range(k0, k0 + n):
        if it == subdivide_at:
            student.subdivide()
            opt = torch.optim.Adam(student.param_groups(), betas=(0.9, 0.999))   # (configured rates, like train.py:341-346: update_lr follows the step)
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
        ots.update_lr(opt, it + 1)
        log.append({"iter": it + 1, "total": float(total.detach()), "psnr": round(ots.psnr_8bit(rgb.detach()[0], gt_rgb[0]), 4), "faces": int(student.faces.shape[0]),
                    **{k: float(v.detach()) for k, v in L.items()}})
        print(log[-1], f"({time.time() - t0:.0f} s)", flush=True)
        if (it + 1) % 5 == 0 or it + 1 == k0 + n:
            json.dump({"what": "CPU oracle (oracle/train_step.py, float64) CONTINUING the HIP run of scripts/train_synthetic.py from its saved state (parameters + Adam moments + step "
                               f"counts after {k0} iterations) across the subdivision at iteration {subdivide_at}: every loss term incl. LPIPS (seeded trunk), torch.optim.Adam + update_lr",
                       "from_state": os.path.basename(path), "start_iterations_done": k0, "iterations": it + 1 - k0, "subdivide_at": subdivide_at,
                       "seconds": round(time.time() - t0, 1), "cpu_threads": os.cpu_count(), "log": log}, open(out_path, "w"), indent=0)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--from-state":
        return from_state(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 70, sys.argv[4] if len(sys.argv) > 4 else "r06")
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
