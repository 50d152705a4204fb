#!/usr/bin/env python
"""Who corrupts k_mlp3_wgrad_partial?  (LABBOOK R6.8)  One VICTIM process repeats the weight-gradient kernels of the shadow MLP on fixed inputs and compares dW bitwise
with its first result; AGGRESSOR processes beside it on the same device repeat one kind of kernel.  usage: python scripts/mc_forensics.py victim|mc|valu|lpips [seconds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gomavatar_amd import _lib
from gomavatar_amd.model import ShadowModule

role = sys.argv[1]
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
img = 256
torch.manual_seed(0)
dev = "cuda"
lib = _lib.load()
P, st = _lib.ptr, _lib.stream_ptr()
if role == "lpips":
    from gomavatar_amd.lpips import LPIPSMatrixCore
    lp = LPIPSMatrixCore(trunk_seed=0, device=dev)
    a, b = torch.rand(1, 256, 256, 3, device=dev), torch.rand(1, 256, 256, 3, device=dev)
    t0 = time.time(); n = 0
    while time.time() - t0 < secs:
        x = a.clone().requires_grad_()
        lp.loss(x, b).sum().backward(); n += 1
    torch.cuda.synchronize(); print(f"lpips aggressor: {n} evaluations"); sys.exit(0)
sm = ShadowModule().to(dev)
with torch.no_grad():
    sm.block_mlps[-1].weight.normal_(0, 0.3)
HW = img * img
normal = torch.zeros(HW, 3, device=dev)
idx = torch.randperm(HW, device=dev)[: HW // 6]
normal[idx] = torch.nn.functional.normalize(torch.randn(idx.numel(), 3, device=dev), dim=-1)
g = torch.randn(HW, device=dev)
lin = [m for m in sm.block_mlps if isinstance(m, torch.nn.Linear)]
ps = [p.detach().float().contiguous() for m in lin for p in (m.weight, m.bias)]
W1, b1, W2, b2, W3, b3, W4, b4 = ps
L, H = sm.multires, 128
D0 = 3 + 6 * L
ws = torch.zeros(lib.gom_shade_workspace_ints(HW), dtype=torch.int32, device=dev)
pos = torch.empty(HW, dtype=torch.int32, device=dev)
pe = torch.zeros(HW + 1, D0, device=dev); hs = torch.zeros(3, HW + 1, H, device=dev); out = torch.zeros(HW + 1, device=dev)
g_rows = torch.zeros(HW + 1, device=dev); dz = torch.zeros(3, HW + 1, H, device=dev); dz4 = torch.zeros(HW + 1, device=dev); dpe = torch.zeros(HW + 1, D0, device=dev)
pack = torch.empty(lib.gom_mlp3_pack_elems(), dtype=torch.int16, device=dev)
_lib.check(lib.gom_shade_select(HW, L, P(normal), P(pos), P(pe), P(ws), st))

def fwd(pk):
    _lib.check(lib.gom_mlp3_forward_rows(HW, P(ws), D0, H, P(pe), *[P(t) for t in ps], P(hs[0]), P(hs[1]), P(hs[2]), P(out), P(pk), st))
def bwd(pk):
    _lib.check(lib.gom_mlp3_backward_rows(HW, P(ws), D0, H, P(g_rows), P(out), P(hs[0]), P(hs[1]), P(hs[2]), P(W1), P(W2), P(W3), P(W4), P(dz4), P(dz[2]), P(dz[1]), P(dz[0]), P(dpe), P(pk), st))
fwd(None)
_lib.check(lib.gom_shade_backward_gather(HW, P(pos), P(g), P(ws), 2.0, P(g_rows), st))
bwd(None)
torch.cuda.synchronize()
t0 = time.time(); n = 0
if role in ("mc", "valu"):
    pk = pack if role == "mc" else None
    while time.time() - t0 < secs:
        for _ in range(20):
            fwd(pk); bwd(pk)
        torch.cuda.synchronize(); n += 20
    print(f"{role} aggressor: {n} forward + backward pairs"); sys.exit(0)
wws = torch.empty(4 * lib.gom_linear_wgrad_slices() * 129 * 128, dtype=torch.float32, device=dev)
ref, bad = None, 0
if role.startswith("inproc"):      # ONE process, two streams: the aggressor kernels (inproc_mc / inproc_valu) on a side stream beside the victim's
    side = torch.cuda.Stream()
    pk = pack if role == "inproc_mc" else None
    hs2, out2, dz2, dz42, dpe2 = torch.zeros_like(hs), torch.zeros_like(out), torch.zeros_like(dz), torch.zeros_like(dz4), torch.zeros_like(dpe)
    sp = side.cuda_stream
    def aggress(k):
        for _ in range(k):
            _lib.check(lib.gom_mlp3_forward_rows(HW, P(ws), D0, H, P(pe), *[P(t) for t in ps], P(hs2[0]), P(hs2[1]), P(hs2[2]), P(out2), P(pk), sp))
            _lib.check(lib.gom_mlp3_backward_rows(HW, P(ws), D0, H, P(g_rows), P(out2), P(hs2[0]), P(hs2[1]), P(hs2[2]), P(W1), P(W2), P(W3), P(W4), P(dz42), P(dz2[2]), P(dz2[1]), P(dz2[0]), P(dpe2), P(pk), sp))
    while time.time() - t0 < secs:
        aggress(30)
        for _ in range(30):
            grads = [torch.empty_like(p) for p in ps]
            _lib.check(lib.gom_mlp3_wgrad_rows(HW, P(ws), D0, H, P(pe), P(hs[0]), P(hs[1]), P(hs[2]), P(dz[0]), P(dz[1]), P(dz[2]), P(dz4), *[P(t) for t in grads], P(wws), st))
            n += 1
            if ref is None:
                torch.cuda.synchronize(); ref = grads
            elif not all(torch.equal(a, b) for a, b in zip(ref, grads)):
                bad += 1
        torch.cuda.synchronize()
    print(f"{role}: victim on the current stream, aggressor on a side stream of the SAME process: {n} runs, {bad} differ from the first"); sys.exit(0)
while time.time() - t0 < secs:
    grads = [torch.empty_like(p) for p in ps]
    _lib.check(lib.gom_mlp3_wgrad_rows(HW, P(ws), D0, H, P(pe), P(hs[0]), P(hs[1]), P(hs[2]), P(dz[0]), P(dz[1]), P(dz[2]), P(dz4), *[P(t) for t in grads], P(wws), st))
    torch.cuda.synchronize(); n += 1
    if ref is None:
        ref = grads
    elif not all(torch.equal(a, b) for a, b in zip(ref, grads)):
        bad += 1
print(f"victim (wgrad kernels alone in its process): {n} runs, {bad} differ from the first")
