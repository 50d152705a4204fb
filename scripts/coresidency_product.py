#!/usr/bin/env python
"""Is the DEFAULT product path exposed to the co-residency fault of LABBOOK R6.8 (a `v_pk_fma_f32 ... op_sel:[0,1,0]` losing a term in lanes 48..63 while waves of
another kernel feed matrix-core instructions from LDS on the same SIMD)?  One process, two streams: a side stream runs an AGGRESSOR back to back -- the LPIPS bf16x3
trunk (the product's only matrix-core kernels), or the opt-in matrix-core shadow MLP as the positive control -- while the current stream repeats ONE Model training
iteration (or the native frame step) on fixed inputs; every gradient is compared bitwise with a run made alone.
usage: python scripts/coresidency_product.py [lpips|mc|none] [runs] [img] [subdiv]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gomavatar_amd import _lib
from gomavatar_amd.workload import MetricWorkload, zju_cfg, model_frames, build_model
from gomavatar_amd import train_util as tu
from gomavatar_amd.lpips import LPIPSMatrixCore
from gomavatar_amd.model import ShadowModule, _ShadeUnderMesh

aggr = sys.argv[1] if len(sys.argv) > 1 else "lpips"
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 200
img = int(sys.argv[3]) if len(sys.argv) > 3 else 256
subdiv = int(sys.argv[4]) if len(sys.argv) > 4 else 0
dev = "cuda"
wl = MetricWorkload(dev, subdiv=subdiv, img=img, n_frames=2)
mcfg, tcfg = zju_cfg(img)
model = build_model(wl, mcfg, with_mlps=False)
with torch.no_grad():
    model.shadow_module.block_mlps[-1].weight.normal_(0, 0.3)
lp = LPIPSMatrixCore(trunk_seed=0, device=dev)
lp2 = LPIPSMatrixCore(trunk_seed=0, device=dev)          # the aggressor's own object (its own workspaces)
fr = model_frames(wl)[0]
params = [p for p in model.parameters() if p.requires_grad]
names = [n for n, p in model.named_parameters() if p.requires_grad]
side = torch.cuda.Stream()
a_img, b_img = torch.rand(1, img, img, 3, device=dev), torch.rand(1, img, img, 3, device=dev)
sm = ShadowModule().to(dev)
lin = [m for m in sm.block_mlps if isinstance(m, torch.nn.Linear)]
sm_params = [p for m in lin for p in (m.weight, m.bias)]
nrm = torch.nn.functional.normalize(torch.randn(img * img, 3, device=dev), dim=-1)

def aggress(k):
    with torch.cuda.stream(side):
        for _ in range(k):
            if aggr == "lpips":
                x = a_img.clone().requires_grad_()
                lp2.loss(x, b_img).sum().backward()
            elif aggr == "mc":
                _ShadeUnderMesh.matrix_cores = True
                x = nrm.clone().requires_grad_()
                _ShadeUnderMesh.apply(x, sm.multires, *sm_params).sum().backward()
                _ShadeUnderMesh.matrix_cores = False

def iteration():
    for p in params: p.grad = None
    rgbs, masks, out = model(fr["K"], fr["E"], fr["cnl_gtfms"], fr["dst_Rs"], fr["dst_Ts"], i_iter=1)
    total, losses = tu.compute_loss(tu.unpack(rgbs, masks, fr["bgcolor"]), masks, out, fr["target_rgbs"], fr["target_masks"], tcfg.losses, lpips_func=lp)
    total.backward()
    return [total.detach().clone(), rgbs.detach().clone()] + [p.grad.clone() for p in params]

ref = iteration(); torch.cuda.synchronize()
again = iteration(); torch.cuda.synchronize()
assert all(torch.equal(a, b) for a, b in zip(ref, again)), "not repeatable even alone"
bad = 0
for run in range(runs):
    aggress(6 if aggr == "lpips" else 40)
    got = iteration()
    torch.cuda.synchronize()
    same = [bool(torch.equal(a, b)) for a, b in zip(ref, got)]
    if not all(same):
        bad += 1
        if bad <= 6:
            print(f"run {run}: NOT bitwise:", [(["total", "rgbs"] + names)[i] + f" {float((ref[i] - got[i]).abs().max()):.2e}" for i, s in enumerate(same) if not s], flush=True)
print(f"aggressor on a side stream: {aggr}; Model iteration ({img}^2, subdiv {subdiv}) x {runs}: {bad} runs differ from the run made alone")
