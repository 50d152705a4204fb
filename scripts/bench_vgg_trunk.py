#!/usr/bin/env python
"""Timing of the LPIPS VGG16 trunk (library convolutions) under different dtype / layout / autotune settings:
forward of two 512x512 images + backward to the first one (what one training step needs, ~480 GFLOP)."""
import itertools, json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gomavatar_amd.lpips import seeded_trunk, trunk_features

def run(dtype, channels_last, benchmark, n=10):
    torch.backends.cudnn.benchmark = benchmark
    dev = torch.device("cuda", 0)
    wb = [t.to(dev, dtype) for t in seeded_trunk(0)]
    if channels_last:
        wb = [t.contiguous(memory_format=torch.channels_last) if t.dim() == 4 else t for t in wb]
    x0 = torch.rand(1, 3, 512, 512, device=dev, dtype=dtype)
    x1 = torch.rand(1, 3, 512, 512, device=dev, dtype=dtype)
    if channels_last:
        x0, x1 = x0.contiguous(memory_format=torch.channels_last), x1.contiguous(memory_format=torch.channels_last)
    def step():
        a = x0.detach().requires_grad_()
        with torch.no_grad():
            f1 = trunk_features(x1, wb)
        f0 = trunk_features(a, wb)
        sum((u.float() - v.float()).square().mean() for u, v in zip(f0, f1)).backward()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

out = {}
for dtype, cl, bm in itertools.product((torch.float32, torch.bfloat16, torch.float16), (False, True), (False, True)):
    key = f"{str(dtype).split('.')[1]}{'_nhwc' if cl else '_nchw'}{'_tuned' if bm else ''}"
    try:
        out[key] = round(run(dtype, cl, bm), 2)
    except Exception as e:
        out[key] = f"failed: {type(e).__name__}"
    print(key, out[key], flush=True)
print(json.dumps(out))
