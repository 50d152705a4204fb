#!/usr/bin/env python
"""Development: which torch ops (and which autograd nodes) an eager cfg-2 iteration launches besides the library's kernels (torch profiler, 5 iterations)."""
import os, sys
sys.argv = [sys.argv[0], "0"]
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "model_iter.py")).read().split("for it in range(10):")[0])
from torch.profiler import profile, ProfilerActivity
for it in range(5):
    step(it)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False) as prof:
    for it in range(5):
        step(it)
    torch.cuda.synchronize()
rows = [(e.key, e.count / 5, e.device_time_total / 5) for e in prof.key_averages() if e.device_time_total > 0 or "Backward" in e.key]
for k, c, t in sorted(rows, key=lambda r: -r[1])[:70]:
    print(f"{k[:80]:80s} {c:6.1f} calls/it {t:8.1f} us/it")
