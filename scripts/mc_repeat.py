#!/usr/bin/env python
"""Is the matrix-core shadow MLP (GOM_MLP_MATRIX_CORES=1) repeatable run to run?  _ShadeUnderMesh forward + backward on fixed inputs, the allocator's free
blocks poisoned with different garbage between runs; every output compared bitwise with the first run's.  usage: python scripts/mc_repeat.py [img]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gomavatar_amd.model import _ShadeUnderMesh, ShadowModule

img = int(sys.argv[1]) if len(sys.argv) > 1 else 128
torch.manual_seed(0)
dev = "cuda"
sm = ShadowModule().to(dev)
with torch.no_grad():
    sm.block_mlps[-1].weight.normal_(0, 0.3)
HW = img * img
normal = torch.zeros(HW, 3, device=dev)
idx = torch.randperm(HW, device=dev)[: HW // 6]
normal[idx] = torch.nn.functional.normalize(torch.randn(idx.numel(), 3, device=dev), dim=-1)
w = torch.randn(HW, 1, device=dev)
lin = [m for m in sm.block_mlps if isinstance(m, torch.nn.Linear)]
params = [p for m in lin for p in (m.weight, m.bias)]

def poison(seed, kind):
    g = torch.Generator(device=dev).manual_seed(seed)
    blocks = []
    for n in (1 << 20, 1 << 22, 1 << 18, 1 << 24, 3 << 19):
        t = torch.empty(n, device=dev)
        if kind == 0: t.normal_(generator=g)
        elif kind == 1: t.fill_(float("nan"))
        elif kind == 2: t.fill_(float("inf"))
        else: t.view(torch.int32).fill_(0x7f7f7f7f)
        blocks.append(t)
    del blocks
    torch.cuda.synchronize()

for mc in (False, True):
    _ShadeUnderMesh.matrix_cores = mc
    ref = None
    for run in range(8):
        poison(run, run % 4)
        x = normal.clone().requires_grad_()
        for p in params: p.grad = None
        out = _ShadeUnderMesh.apply(x, sm.multires, *params)
        (out * w).sum().backward()
        torch.cuda.synchronize()
        got = [out.detach().clone(), x.grad.clone()] + [p.grad.clone() for p in params]
        if ref is None:
            ref = got
            print(f"matrix_cores={mc}: finite {all(bool(torch.isfinite(t).all()) for t in got)}")
        else:
            same = [bool(torch.equal(a, b)) for a, b in zip(ref, got)]
            if not all(same):
                d = [float((a - b).abs().max()) for a, b in zip(ref, got)]
                print(f"  run {run} (poison kind {run % 4}): NOT bitwise: {same}  max|d| {d}")
    print(f"matrix_cores={mc}: done")
