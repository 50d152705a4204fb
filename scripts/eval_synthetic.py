#!/usr/bin/env python
"""The reference's evaluation loop (eval.py:334-365) on synthetic data: no_grad forward in eval mode -> unpack on a
fixed background -> 8-bit quantisation -> mse / psnr / ssim / lpips x 1000 per frame (`gomavatar_amd.metrics.Evaluator`),
optional PNG dump.  A briefly trained student is evaluated against the teacher that rendered its targets.

    python scripts/eval_synthetic.py --train-iters 150 --img 256 [--out /tmp/eval_frames]"""
import argparse, json, os, sys
from types import SimpleNamespace as NS
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gomavatar_amd import synthetic as syn, metrics as M
from gomavatar_amd.model import Model
from gomavatar_amd.train_util import compute_loss, unpack
from gomavatar_amd.lpips import LPIPS

ap = argparse.ArgumentParser()
ap.add_argument("--train-iters", type=int, default=150); ap.add_argument("--img", type=int, default=256); ap.add_argument("--out", default=None)
a = ap.parse_args()
img = a.img
cfg = NS(img_size=(img, img), canonical_geometry=NS(sigma=1e-3, radius_scale=1.0, deform_so3=True, deform_scale=True), appearance=NS(color_init=0.5),
         normal_renderer=NS(sigma=1e-5, soft_mask=True), shadow_module=NS(name="basic", multires=6, mlp_width=128, mlp_depth=3, skips=(4,)),
         lbs_weights=NS(refine=False))
loss_cfg = NS(rgb=NS(coeff=1.0), mask=NS(coeff=5.0), lpips=NS(coeff=0.0), laplacian=NS(coeff_canonical=0.0, coeff_observation=10.0),
              normal=NS(coeff_mask=1.0, kernel_size=7, coeff_consist=0.1), color_consist=NS(coeff=0.05))
lr = NS(lr=NS(appearance=5e-3, canonical_geometry=5e-4, canonical_geometry_xyz=5e-5, shadow=5e-4))
body = syn.make_body(0)
teacher, student = Model(cfg, body).train(), Model(cfg, body).train()
with torch.no_grad():
    teacher.appearance.copy_(torch.rand(teacher.appearance.shape, device="cuda", generator=torch.Generator(device="cuda").manual_seed(0)))
bg = torch.tensor([[0.0, 0.0, 0.0]], device="cuda")                     # cfg.bgcolor of the evaluation configs
frames = []
for i in range(12):                                                      # 8 training views + 4 held-out ones
    fr = {k: torch.from_numpy(v).cuda() for k, v in syn.make_frame(i, img).items()}
    with torch.no_grad():
        rgbs, masks, _ = teacher(fr["K"], fr["E"], fr["cnl_gtfms"], fr["dst_Rs"], fr["dst_Ts"])
        fr["gt_rgb_train"], fr["gt_mask"] = unpack(rgbs, masks, fr["bgcolor"]).clamp(0, 1), masks.clone()
        fr["gt_rgb_eval"] = unpack(rgbs, masks, bg).clamp(0, 1)
    frames.append(fr)
opt = torch.optim.Adam(student.get_param_groups(lr))
for it in range(a.train_iters):
    fr = frames[it % 8]
    opt.zero_grad(set_to_none=True)
    rgbs, masks, out = student(fr["K"], fr["E"], fr["cnl_gtfms"], fr["dst_Rs"], fr["dst_Ts"], i_iter=it)
    total, _ = compute_loss(unpack(rgbs, masks, fr["bgcolor"]), masks, out, fr["gt_rgb_train"], fr["gt_mask"], loss_cfg)
    total.backward(); opt.step()
student.eval()
ev = M.Evaluator(lpips_model=LPIPS(net="vgg"))
if a.out:
    os.makedirs(a.out, exist_ok=True)
for i, fr in enumerate(frames[8:]):
    with torch.no_grad():
        pred, mask, _ = student(fr["K"], fr["E"], fr["cnl_gtfms"], fr["dst_Rs"], fr["dst_Ts"])
        pred = unpack(pred, mask, bg)
    p8, g8 = M.to_8b(pred[0]), M.to_8b(fr["gt_rgb_eval"][0])               # eval.py:355-361
    ev.evaluate(M.from_8b(p8), M.from_8b(g8))
    if a.out:
        from PIL import Image
        Image.fromarray(np.concatenate([p8.cpu().numpy(), g8.cpu().numpy()], 1)).save(os.path.join(a.out, f"frame_{i:06d}.png"))
print(json.dumps({k: round(v, 4) for k, v in ev.summarize().items()}))
