#!/usr/bin/env python
"""BASELINE configs[1] as the reference runs it, alone: N iterations of train.py:309-349 through the drop-in Model (55 104 Gaussians, 512^2, every
loss term incl. LPIPS on the bf16x3 trunk) -- for rocprofv3 kernel traces of the iteration (scripts/model_iter_prof.sh) and quick timings.
usage: python scripts/model_iter.py [iters] [torch|gom|graph] [bf16x3|bf16]"""
import os, sys, time
from types import SimpleNamespace as NS
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gomavatar_amd.workload import MetricWorkload
from gomavatar_amd.model import Model
from gomavatar_amd import train_util as tu
from gomavatar_amd.lpips import LPIPSMatrixCore
from gomavatar_amd.optim import GomAdam

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
which = sys.argv[2] if len(sys.argv) > 2 else "gom"
prec = sys.argv[3] if len(sys.argv) > 3 else "bf16x3"
dev = "cuda"
wl = MetricWorkload(dev, subdiv=1, img=512, n_frames=4)
cfg = NS(img_size=(512, 512), canonical_geometry=NS(sigma=1e-3, radius_scale=1.0, deform_so3=True, deform_scale=True), appearance=NS(color_init=0.5),
         normal_renderer=NS(sigma=1e-5, soft_mask=True), shadow_module=NS(name="basic", multires=6, mlp_width=128, mlp_depth=3, skips=(4,)),
         lbs_weights=NS(refine=False))
tcfg = NS(lr=NS(lbs_weights=0.0, appearance=0.0005, canonical_geometry=0.0005, canonical_geometry_xyz=0.0005, shadow=0.0005), lr_decay_steps=100000,
          losses=NS(rgb=NS(coeff=1.0), mask=NS(coeff=5.0), lpips=NS(coeff=1.0), laplacian=NS(coeff_canonical=0.0, coeff_observation=10.0),
                    normal=NS(coeff_mask=1.0, kernel_size=7, coeff_consist=0.1), color_consist=NS(coeff=0.05)))
model = Model(cfg, wl.body).train()
if os.environ.get("CAPTURE_SAFE"):
    model.capture_safe = True   # (device-resident camera: no host read of K / E per iteration)
mcl = LPIPSMatrixCore(trunk_seed=0, device=dev, precision=prec)
groups = model.get_param_groups(tcfg)
opt = torch.optim.Adam(groups, betas=(0.9, 0.999)) if which == "torch" else GomAdam(groups, betas=(0.9, 0.999))
frames = []
for i in range(4):
    fr = {k: torch.from_numpy(v).to(dev) for k, v in wl.frames_np[i].items()}
    fr["target_rgbs"], fr["target_masks"] = wl.frames[i]["gt_rgb"][None], wl.frames[i]["gt_mask"][None]
    if os.environ.get("BINARY_MASKS"):   # masks of {0, 1} like the data sets' segmentation masks (the synthetic targets are rendered: soft everywhere)
        fr["target_masks"] = (fr["target_masks"] > 0.5).float()
    frames.append(fr)
if which == "graph":          # the same iteration as ONE HIP graph (train_util.GraphedTrainStep: Adam capturable, lr frozen at capture)
    opt = GomAdam(groups, betas=(0.9, 0.999), capturable=True)
    gstep = tu.GraphedTrainStep(model, opt, tcfg.losses, mcl)
    step = lambda it: gstep(frames[it % 4], i_iter=1)
else:
    step = lambda it: tu.train_iteration(model, opt, frames[it % 4], tcfg, it + 1, lpips_func=mcl)
for it in range(10):
    step(it)
torch.cuda.synchronize()
t0 = time.perf_counter()
for it in range(iters):
    step(10 + it)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / iters
print(f"model train iteration ({which} Adam, {prec}): {dt * 1e3:.3f} ms = {1 / dt:.1f} it/s")
