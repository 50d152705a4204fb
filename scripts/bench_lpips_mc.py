#!/usr/bin/env python
"""LPIPS value+gradient at 512x512: hand-written MFMA trunk vs library convolutions; per-layer conv TFLOP/s."""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gomavatar_amd import _lib
from gomavatar_amd.lpips import LPIPS, LPIPSMatrixCore, lpips_loss

def timeit(fn, n=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

H = W = int(os.environ.get("IMG", 512))
pred = torch.rand(1, H, W, 3, device="cuda"); gt = torch.rand(1, H, W, 3, device="cuda")
mc = LPIPSMatrixCore(trunk_seed=0, precision="bf16")
bufs = (torch.empty((5, 1, _lib.GOM_LOSS_BLOCKS), device="cuda"), torch.empty((1, H, W, 3), device="cuda"))
stream = torch.cuda.Stream()
with torch.cuda.stream(stream):
    graph_ms = round(timeit(lambda: mc.value_and_grad(pred, gt, out=bufs)), 3)
out = {"matrix_core_value_and_grad_graph_ms": graph_ms, "matrix_core_value_and_grad_ms": round(timeit(lambda: mc.value_and_grad(pred, gt)), 3),
       "matrix_core_value_only_ms": round(timeit(lambda: mc.value_and_grad(pred, gt, want_grad=False)), 3)}
for name, dt in (() if os.environ.get("NO_LIBRARY") else (("fp32", torch.float32), ("bf16", torch.bfloat16))):
    m = LPIPS(trunk_seed=0, trunk_dtype=dt)
    def f():
        p = pred.clone().requires_grad_(); lpips_loss(m, p, gt).backward()
    out[f"library_{name}_value_and_grad_ms"] = round(timeit(f, n=10), 3)
# per-layer forward conv
lib = _lib.load(); layers = {}
h, w = H, W
x = torch.randn(1, h, w, 32, device="cuda").to(torch.bfloat16)
from gomavatar_amd.lpips import POOL_BEFORE_CONV
for i in range(13):
    if i in POOL_BEFORE_CONV: h //= 2; w //= 2
    cin, cout = mc.cin[i], mc.cout[i]
    x = torch.randn(1, h, w, cin, device="cuda").to(torch.bfloat16)
    o = torch.empty(1, h, w, cout, device="cuda", dtype=torch.bfloat16)
    sp = lib.gom_conv3x3_splits(1, h, w, cin, cout)
    ws = torch.empty(sp * h * w * cout, device="cuda") if sp > 1 else None
    fn = lambda: _lib.check(lib.gom_conv3x3_bf16_splitk(1, h, w, cin, cout, _lib.ptr(x), _lib.ptr(mc.w_fwd[i]), _lib.ptr(mc.bias[i]), 0, _lib.ptr(o), 1, sp,
                                                        _lib.ptr(ws), _lib.stream_ptr()))
    ms = timeit(fn, n=30, warm=5)
    layers[f"conv{i}_{h}x{w}_{cin}->{cout}_split{sp}"] = {"us": round(ms * 1e3, 1), "TFLOPs": round(2 * 9 * cin * cout * h * w / (ms * 1e-3) / 1e12, 1)}
out["layers"] = layers
print(json.dumps(out, indent=1))
