#!/usr/bin/env python
"""Is ONE Model training iteration (forward, every loss term incl. LPIPS, backward) repeatable when the caching allocator's free blocks hold different garbage?
The same parameters and frame every run; between runs the free blocks are poisoned (random / NaN / Inf / 0x7f7f7f7f); losses and every gradient compared bitwise
with the first run's.  A difference = some kernel reads memory nobody wrote.  usage: python scripts/iter_repeat.py [img] [subdiv] [runs]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gomavatar_amd.workload import MetricWorkload, zju_cfg, model_frames, build_model
from gomavatar_amd import train_util as tu
from gomavatar_amd.lpips import LPIPSMatrixCore

img = int(sys.argv[1]) if len(sys.argv) > 1 else 256
subdiv = int(sys.argv[2]) if len(sys.argv) > 2 else 0
runs = int(sys.argv[3]) if len(sys.argv) > 3 else 8
dev = "cuda"
wl = MetricWorkload(dev, subdiv=subdiv, img=img, n_frames=2)
mcfg, tcfg = zju_cfg(img)
model = build_model(wl, mcfg, with_mlps=False)
with torch.no_grad():
    model.shadow_module.block_mlps[-1].weight.normal_(0, 0.3)
lp = LPIPSMatrixCore(trunk_seed=0, device=dev)
fr = model_frames(wl)[0]
params = [p for p in model.parameters() if p.requires_grad]

def poison(seed, kind):
    g = torch.Generator(device=dev).manual_seed(seed)
    blocks = []
    for n in (1 << 20, 1 << 22, 1 << 18, 1 << 24, 3 << 19, 1 << 16, 5 << 20):
        t = torch.empty(n, device=dev)
        if kind == 0: t.normal_(generator=g)
        elif kind == 1: t.fill_(float("nan"))
        elif kind == 2: t.fill_(float("inf"))
        else: t.view(torch.int32).fill_(0x7f7f7f7f)
        blocks.append(t)
    del blocks
    torch.cuda.synchronize()

ref = None
names = [n for n, p in model.named_parameters() if p.requires_grad]
for run in range(runs):
    poison(run, run % 4)
    for p in params: p.grad = None
    rgbs, masks, out = model(fr["K"], fr["E"], fr["cnl_gtfms"], fr["dst_Rs"], fr["dst_Ts"], i_iter=1)
    total, losses = tu.compute_loss(tu.unpack(rgbs, masks, fr["bgcolor"]), masks, out, fr["target_rgbs"], fr["target_masks"], tcfg.losses, lpips_func=lp)
    total.backward()
    torch.cuda.synchronize()
    got = [total.detach().clone(), rgbs.detach().clone(), masks.detach().clone()] + [p.grad.clone() for p in params]
    if ref is None:
        ref = got
        print("first run finite:", all(bool(torch.isfinite(t).all()) for t in got), "total", float(total))
    else:
        same = [bool(torch.equal(a, b)) for a, b in zip(ref, got)]
        if not all(same):
            bad = [(["total", "rgbs", "masks"] + names)[i] + f" {float((ref[i] - got[i]).abs().max()):.2e}" for i, s in enumerate(same) if not s]
            print(f"run {run} (poison kind {run % 4}): NOT bitwise:", bad)
        else:
            print(f"run {run} (poison kind {run % 4}): bitwise")
