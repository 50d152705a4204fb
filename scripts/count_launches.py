#!/usr/bin/env python
"""Kernel launches per phase of the whole-`Model` training iteration (forward / loss / backward / optimizer step), by the aten op or
autograd function that issued them, and by autograd node for the backward: the tool behind the launch diet in DESIGN.md §6.
Runs scripts/train_synthetic.py's setup (level 1, 512x512), then profiles ten iterations with torch.profiler.

    python scripts/count_launches.py"""
import sys, collections, torch
sys.path.insert(0, "/root/repo")
sys.argv = ["x", "--iters", "30", "--img", "512", "--level", "1"]
src = open("/root/repo/scripts/train_synthetic.py").read()
head = src[:src.index("log, t0, t_warm")]
exec(head)
from torch.profiler import profile, ProfilerActivity, record_function
def one(it):
    fr = frames[it % 8]
    opt.zero_grad(set_to_none=True)
    with record_function("PH_forward"):
        rgbs, masks, out = student(fr["K"], fr["E"], fr["cnl_gtfms"], fr["dst_Rs"], fr["dst_Ts"], i_iter=it)
        pred = unpack(rgbs, masks, fr["bgcolor"])
    with record_function("PH_loss"):
        total, losses = compute_loss(pred, masks, out, fr["gt_rgb"], fr["gt_mask"], loss_cfg, lpips_func=lp)
    with record_function("PH_backward"):
        total.backward()
    with record_function("PH_step"):
        opt.step()
for it in range(25): one(it)
torch.cuda.synchronize()
N = 10
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for it in range(N): one(it)
    torch.cuda.synchronize()
ev = prof.events()
phases = [e for e in ev if e.name.startswith("PH_")]
cnt = collections.defaultdict(lambda: collections.Counter())
for e in ev:
    if e.device_type == torch.autograd.DeviceType.CPU and e.kernels:
        for p in phases:
            if p.time_range.start <= e.time_range.start <= p.time_range.end:
                cnt[p.name][e.name] += len(e.kernels); break
for ph, c in cnt.items():
    print("==", ph, "launches/it", sum(c.values()) / N, " host ms/it", sum(p.cpu_time_total for p in phases if p.name == ph) / N / 1e3)
    for k, v in c.most_common(24): print("    %-60s %5.1f" % (k[:60], v / N))
# autograd node names for backward ops: which backward functions issue the small ops
bw = collections.Counter()
for e in ev:
    if e.device_type == torch.autograd.DeviceType.CPU and e.name.startswith("autograd::engine::evaluate_function"):
        nk = 0
        stack = [e]
        while stack:
            x = stack.pop(); nk += len(x.kernels); stack.extend(x.cpu_children)
        bw[e.name.replace("autograd::engine::evaluate_function: ", "")] += nk
print("== backward nodes (launches/it)")
for k, v in bw.most_common(40): print("    %-60s %5.1f" % (k[:60], v / N))
