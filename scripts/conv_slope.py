#!/usr/bin/env python
"""Development: k_conv3x3_bf16_v2 time against the number of 32-channel chunks (slope = cost of the K loop, intercept = prologue + epilogue + launch)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gomavatar_amd import _lib
lib = _lib.load()
def timeit(fn, n=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for name, B, hw, cout in (("128^2 x 256 co, B 2 (512 WGs)", 2, 128, 256), ("256^2 x 128 co, B 2 (1024 WGs)", 2, 256, 128), ("512^2 x 64 co, B 2 (2048 WGs)", 2, 512, 64), ("128^2 x 256 co, B 1 (256 WGs)", 1, 128, 256)):
    row = []
    for chunks in (1, 2, 4, 8, 16, 24, 48):
        cin = 32 * chunks
        x = torch.randn(B, hw, hw, cin, device="cuda").to(torch.bfloat16)
        w = (torch.randn(chunks, 9, cout, 32, device="cuda") * 0.02).to(torch.bfloat16)
        b = torch.zeros(cout, device="cuda")
        o = torch.empty(B, hw, hw, cout, device="cuda", dtype=torch.bfloat16)
        us = timeit(lambda: _lib.check(lib.gom_conv3x3_bf16_splitk(B, hw, hw, cin, cout, _lib.ptr(x), _lib.ptr(w), _lib.ptr(b), 0, _lib.ptr(o), 1, 1, 0, _lib.stream_ptr())))
        row.append(f"{chunks}: {us:.1f}")
    print(name, " | ".join(row))
