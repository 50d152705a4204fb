#!/usr/bin/env python
"""The reference's training iteration (train.py:309-349: zero_grad -> forward -> unpack -> compute_loss -> backward ->
Adam step) on synthetic data, through `gomavatar_amd.model.Model` + `train_util.compute_loss` (+ LPIPS on the matrix
cores).  A teacher avatar renders the targets; the student starts from the reference's initialisation (grey colours,
unit scales).  Prints PSNR of the student's renders against the targets while it trains, and iterations/s.

    python scripts/train_synthetic.py --iters 300 --img 256 --subdivide-at 150"""
import argparse, json, os, sys, time
from types import SimpleNamespace as NS
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gomavatar_amd import synthetic as syn
from gomavatar_amd.model import Model
from gomavatar_amd.train_util import GraphedTrainStep, compute_loss, unpack
from gomavatar_amd.lpips import LPIPSMatrixCore
from gomavatar_amd import metrics as M

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=300); ap.add_argument("--img", type=int, default=256)
ap.add_argument("--level", type=int, default=0, help="SMPL-like body subdivisions (0: 13 776 faces)")
ap.add_argument("--subdivide-at", type=int, default=-1); ap.add_argument("--no-lpips", action="store_true")
ap.add_argument("--graph", action="store_true", help="capture the whole iteration in one HIP graph (train_util.GraphedTrainStep)")
ap.add_argument("--fused-adam", action="store_true", help="torch.optim.Adam(fused=True): one kernel per step instead of ~40")
ap.add_argument("--no-host-sync", action="store_true", help="Model.capture_safe without a graph: no device->host read per iteration, the host runs ahead")
a = ap.parse_args()
img = a.img
cfg = NS(img_size=(img, img), canonical_geometry=NS(sigma=1e-3, radius_scale=1.0, deform_so3=True, deform_scale=True), appearance=NS(color_init=0.5),
         normal_renderer=NS(sigma=1e-5, soft_mask=True), shadow_module=NS(name="basic", multires=6, mlp_width=128, mlp_depth=3, skips=(4,)),
         lbs_weights=NS(refine=False))
loss_cfg = NS(rgb=NS(coeff=1.0), mask=NS(coeff=5.0), lpips=NS(coeff=0.0 if a.no_lpips else 1.0), laplacian=NS(coeff_canonical=0.0, coeff_observation=10.0),
              normal=NS(coeff_mask=1.0, kernel_size=7, coeff_consist=0.1), color_consist=NS(coeff=0.05))
lr = NS(lr=NS(appearance=5e-3, canonical_geometry=5e-4, canonical_geometry_xyz=5e-5, shadow=5e-4))
body = syn.make_body(a.level)
teacher, student = Model(cfg, body).train(), Model(cfg, body).train()
with torch.no_grad():
    g = torch.Generator(device="cuda").manual_seed(0)
    teacher.appearance.copy_(torch.rand(teacher.appearance.shape, device="cuda", generator=g))
frames = []
for i in range(8):
    fr = {k: torch.from_numpy(v).cuda() for k, v in syn.make_frame(i, img).items()}
    with torch.no_grad():
        rgbs, masks, _ = teacher(fr["K"], fr["E"], fr["cnl_gtfms"], fr["dst_Rs"], fr["dst_Ts"])
        fr["gt_rgb"], fr["gt_mask"] = unpack(rgbs, masks, fr["bgcolor"]).clamp(0, 1), masks.clone()
    frames.append(fr)
lp = None if a.no_lpips else LPIPSMatrixCore(trunk_seed=0)
adam_kw = dict(fused=True, capturable=a.graph) if a.fused_adam else dict(capturable=a.graph)
opt = torch.optim.Adam(student.get_param_groups(lr), **adam_kw)
for fr in frames:
    fr["target_rgbs"], fr["target_masks"] = fr["gt_rgb"], fr["gt_mask"]                  # the reference's key names (dataset/train.py:272-275)
student.capture_safe = a.no_host_sync or a.graph
gstep = GraphedTrainStep(student, opt, loss_cfg, lp) if a.graph else None
log, t0, t_warm, n_warm = [], time.perf_counter(), None, min(20, a.iters // 2)
for it in range(a.iters):
    if it == n_warm:                      # steady state: lazy kernel loading, graph capture and allocator growth are behind us
        torch.cuda.synchronize(); t_warm = time.perf_counter()
    if it == a.subdivide_at:
        student.subdivide(); opt = torch.optim.Adam(student.get_param_groups(lr), **adam_kw)   # train.py:330-340 rebuilds the optimizer
        gstep = GraphedTrainStep(student, opt, loss_cfg, lp) if a.graph else None        # new topology: new capture
    fr = frames[it % 8]
    if gstep is not None:
        total = gstep(fr, i_iter=it)
        if it % 50 == 0 or it == a.iters - 1:
            with torch.no_grad():
                rgbs, masks, _ = student(fr["K"], fr["E"], fr["cnl_gtfms"], fr["dst_Rs"], fr["dst_Ts"], i_iter=it)
                pred = unpack(rgbs, masks, fr["bgcolor"])
    else:
        opt.zero_grad(set_to_none=True)
        rgbs, masks, out = student(fr["K"], fr["E"], fr["cnl_gtfms"], fr["dst_Rs"], fr["dst_Ts"], i_iter=it)
        pred = unpack(rgbs, masks, fr["bgcolor"])
        total, losses = compute_loss(pred, masks, out, fr["gt_rgb"], fr["gt_mask"], loss_cfg, lpips_func=lp)
        total.backward(); opt.step()
    if it % 50 == 0 or it == a.iters - 1:
        with torch.no_grad():
            p8, g8 = M.from_8b(M.to_8b(pred[0])), M.from_8b(M.to_8b(fr["gt_rgb"][0]))
            log.append({"iter": it, "loss": round(float(total), 5), "psnr": round(M.psnr(p8, g8), 2), "faces": int(student.faces.shape[0])})
            print(log[-1], flush=True)
torch.cuda.synchronize()
t1 = time.perf_counter()
print(json.dumps({"iters_per_s": round((a.iters - n_warm) / (t1 - t_warm), 1), "iters_per_s_including_first_%d" % n_warm: round(a.iters / (t1 - t0), 1),
                  "img": img, "log": log}))
