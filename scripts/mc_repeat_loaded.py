#!/usr/bin/env python
"""The matrix-core shadow MLP under load: the library calls of _ShadeUnderMesh (forward + backward) on fixed inputs, RUNS times, every INTERMEDIATE buffer compared
bitwise with the first run's; run several of these at once on one device (scripts/mc_repeat_loaded.sh) -- a timing-dependent result shows as a mismatch, and its
PATTERN (which buffer, rows, channels) says where.  usage: python scripts/mc_repeat_loaded.py [img] [runs] [mc]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gomavatar_amd import _lib
from gomavatar_amd.model import ShadowModule

img = int(sys.argv[1]) if len(sys.argv) > 1 else 256
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 300
mc = (sys.argv[3] if len(sys.argv) > 3 else "1") != "0"
torch.manual_seed(0)
dev = "cuda"
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
lib = _lib.load()
P, st = _lib.ptr, _lib.stream_ptr()
L, H = sm.multires, 128
D0 = 3 + 6 * L
n_rows = int((normal != 0).any(-1).sum()) + 1

def one():
    x = normal
    ws = torch.zeros(lib.gom_shade_workspace_ints(HW), dtype=torch.int32, device=dev)
    pos = torch.empty(HW, dtype=torch.int32, device=dev)
    pe = torch.full((HW + 1, D0), 7.0, device=dev)
    hs = torch.full((3, HW + 1, H), 7.0, device=dev)
    out = torch.full((HW + 1,), 7.0, device=dev)
    shading = torch.empty(HW, 1, device=dev)
    _lib.check(lib.gom_shade_select(HW, L, P(x), P(pos), P(pe), P(ws), st))
    pack = torch.empty(lib.gom_mlp3_pack_elems(), dtype=torch.int16, device=dev) if mc else None
    _lib.check(lib.gom_mlp3_forward_rows(HW, P(ws), D0, H, P(pe), *[P(t) for t in ps], P(hs[0]), P(hs[1]), P(hs[2]), P(out), P(pack), st))
    _lib.check(lib.gom_shade_scatter(HW, P(pos), P(out), P(ws), 2.0, P(shading), st))
    g_rows = torch.full((HW + 1,), 7.0, device=dev)
    pack2 = torch.empty(lib.gom_mlp3_pack_elems(), dtype=torch.int16, device=dev) if mc else None
    dz = torch.full((3, HW + 1, H), 7.0, device=dev)
    dz4 = torch.full((HW + 1,), 7.0, device=dev)
    dpe = torch.full((HW + 1, D0), 7.0, device=dev)
    _lib.check(lib.gom_shade_backward_gather(HW, P(pos), P(g), P(ws), 2.0, P(g_rows), st))
    _lib.check(lib.gom_mlp3_backward_rows(HW, P(ws), D0, H, P(g_rows), P(out), P(hs[0]), P(hs[1]), P(hs[2]), P(W1), P(W2), P(W3), P(W4), P(dz4), P(dz[2]), P(dz[1]), P(dz[0]), P(dpe), P(pack2), st))
    wws = torch.empty(4 * lib.gom_linear_wgrad_slices() * 129 * 128, dtype=torch.float32, device=dev)
    grads = [torch.empty_like(p) for p in ps]
    _lib.check(lib.gom_mlp3_wgrad_rows(HW, P(ws), D0, H, P(pe), P(hs[0]), P(hs[1]), P(hs[2]), P(dz[0]), P(dz[1]), P(dz[2]), P(dz4), *[P(t) for t in grads], P(wws), st))
    torch.cuda.synchronize()
    d = {"pe": pe[:n_rows], "h1": hs[0, :n_rows], "h2": hs[1, :n_rows], "h3": hs[2, :n_rows], "out": out[:n_rows], "shading": shading, "g_rows": g_rows[:n_rows],
         "dz4": dz4[:n_rows], "dz3": dz[2, :n_rows], "dz2": dz[1, :n_rows], "dz1": dz[0, :n_rows], "dpe": dpe[:n_rows]}
    if pack is not None:
        d["pack_fwd"], d["pack_bwd"] = pack, pack2
    for nm, t in zip(("dW1", "db1", "dW2", "db2", "dW3", "db3", "dW4", "db4"), grads):
        d[nm] = t
    return {k: v.clone() for k, v in d.items()}

ref, bad = None, 0
for run in range(runs):
    got = one()
    if ref is None:
        ref = got
        continue
    for nm in ref:
        a, b = ref[nm], got[nm]
        if torch.equal(a, b):
            continue
        bad += 1
        ne = (a != b)
        nz = ne.nonzero()
        msg = f"run {run} {nm} {tuple(a.shape)}: {int(ne.sum())} differ, max|d| {float((a.float() - b.float()).abs().max()):.3e}; dim0 {nz[:, 0].unique().tolist()[:10]}"
        if nz.shape[1] > 1:
            msg += f"; dim1 {nz[:, 1].unique().tolist()[:40]}"
        print(msg, flush=True)
print(f"matrix_cores={mc}: {runs} runs, {bad} mismatching buffers", flush=True)
