#!/usr/bin/env python
"""Kernel launches per phase of the whole-`Model` training iteration (forward / loss / backward / optimizer step), by the aten op or
autograd function that issued them, and by autograd node for the backward: the tool behind the launch diet in DESIGN.md section 6.
Same setup as scripts/model_iter.py (BASELINE configs[1]: 55 104 Gaussians, 512^2, LPIPS bf16x3, GomAdam); ten iterations under torch.profiler.

    python scripts/count_launches.py"""
import collections
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

iters, which, prec = 10, "gom", "bf16x3"
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
    frames.append(fr)
from torch.profiler import profile, ProfilerActivity, record_function


def one(it):
    data = frames[it % 4]
    if hasattr(mcl, "prefetch_target"):
        mcl.prefetch_target(data["target_rgbs"])
    opt.zero_grad()
    with record_function("PH_forward"):
        rgb, mask, outputs = model(data["K"], data["E"], data["cnl_gtfms"], data["dst_Rs"], data["dst_Ts"], dst_posevec=data.get("dst_posevec"),
                                   canonical_joints=data.get("dst_tpose_joints"), i_iter=it + 1, bgcolor=data.get("bgcolor"))
        rgb = tu.unpack(rgb, mask, data["bgcolor"])
    with record_function("PH_loss"):
        loss, items = tu.compute_loss(rgb, mask, outputs, data["target_rgbs"], data["target_masks"], tcfg.losses, data, it + 1, lpips_func=mcl)
    with record_function("PH_backward"):
        loss.backward()
    with record_function("PH_step"):
        opt.step()
        tu.update_lr(opt, it + 1, tcfg)


for it in range(15):
    one(it)
torch.cuda.synchronize()
N = 10
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for it in range(N):
        one(it)
    torch.cuda.synchronize()
ev = prof.events()
phases = [e for e in ev if e.name.startswith("PH_")]
cnt = collections.defaultdict(lambda: collections.Counter())
for e in ev:
    if e.device_type == torch.autograd.DeviceType.CPU and e.kernels:
        for p in phases:
            if p.time_range.start <= e.time_range.start <= p.time_range.end:
                cnt[p.name][e.name] += len(e.kernels)
                break
for ph, c in cnt.items():
    print("==", ph, "launches/it", sum(c.values()) / N, " host ms/it", sum(p.cpu_time_total for p in phases if p.name == ph) / N / 1e3)
    for k, v in c.most_common(30):
        print("    %-60s %5.1f" % (k[:60], v / N))
# autograd node names for backward ops: which backward functions issue the small ops
bw = collections.Counter()
for e in ev:
    if e.device_type == torch.autograd.DeviceType.CPU and e.name.startswith("autograd::engine::evaluate_function"):
        nk = 0
        stack = [e]
        while stack:
            x = stack.pop()
            nk += len(x.kernels)
            stack.extend(x.cpu_children)
        bw[e.name.replace("autograd::engine::evaluate_function: ", "")] += nk
print("== backward nodes (launches/it)")
for k, v in bw.most_common(40):
    print("    %-60s %5.1f" % (k[:60], v / N))
# the small aten ops, by the chain of ops / autograd nodes that issued them
chains = collections.Counter()
for e in ev:
    if e.device_type == torch.autograd.DeviceType.CPU and e.kernels and e.name.startswith("aten::"):
        names, p = [e.name], e.cpu_parent
        while p is not None and len(names) < 6:
            if not p.name.startswith("PH_"):
                names.append(p.name.replace("autograd::engine::evaluate_function: ", "bw:"))
            p = p.cpu_parent
        chains[" < ".join(names)] += len(e.kernels)
print("== aten launches by issuing chain (launches/it)")
for k, v in chains.most_common(60):
    print("    %5.1f  %s" % (v / N, k[:200]))
