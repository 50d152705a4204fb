#!/usr/bin/env python
"""Development: how far two valid evaluations of the bf16x3 LPIPS gradient are apart when only the fp32 summation order of the split-K layers changes
(GOM_CONV_SPLIT_MODE=0 / 1 in two processes: `save NAME`, then `cmp A B`), and against the target-prefetch path."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gomavatar_amd.lpips import LPIPSMatrixCore
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
os.makedirs(out, exist_ok=True)
if sys.argv[1] == "save":
    g = torch.Generator().manual_seed(21)
    pred = torch.rand(1, 256, 256, 3, generator=g).cuda()
    gt = (pred.cpu() + 0.2 * torch.randn(1, 256, 256, 3, generator=g)).clamp(0, 1).cuda()
    mc = LPIPSMatrixCore(trunk_seed=5, precision="bf16x3")
    v, gr = mc.value_and_grad(pred, gt)
    mc.prefetch_target(gt)
    v2, gr2 = mc.value_and_grad(pred, gt)
    torch.save({"v": float(v), "g": gr.cpu(), "v2": float(v2), "g2": gr2.cpu()}, os.path.join(out, f"lpips_{sys.argv[2]}.pt"))
else:
    a, b = (torch.load(os.path.join(out, f"lpips_{n}.pt")) for n in sys.argv[2:4])
    rel = lambda x, y: float((x - y).norm() / y.norm())
    print(f"batched {sys.argv[2]} vs batched {sys.argv[3]}: value {abs(a['v'] - b['v']) / b['v']:.2e} gradient rel L2 {rel(a['g'], b['g']):.2e}")
    print(f"prefetch vs batched ({sys.argv[2]}): value {abs(a['v2'] - a['v']) / a['v']:.2e} gradient {rel(a['g2'], a['g']):.2e};  ({sys.argv[3]}): {rel(b['g2'], b['g']):.2e}")
    print(f"prefetch {sys.argv[2]} vs prefetch {sys.argv[3]}: gradient {rel(a['g2'], b['g2']):.2e}")
