#!/usr/bin/env python
"""Development aid: cycles per phase of k_seg_bwd from a -DGOM_PHASE_PROF build (scripts/exp_build.py prof -DGOM_PHASE_PROF).
    GOM_HIP_LIB=gomavatar_amd/_variants/libgom_hip_prof.so python scripts/phase_prof.py [bench args]"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gomavatar_amd import _lib
import bench
lib = ctypes.CDLL(_lib.LIB_PATH)
sys.argv = ["bench.py", "--inflight", "1", "--no-graph", "--no-cpu-baseline", "--steps", "20", "--warmup", "4"] + sys.argv[1:]
bench.main()
ph = np.zeros(16, np.uint64); wg = np.zeros(4096, np.uint64)
lib.gom_debug_phase_counters(ph.ctypes.data_as(ctypes.c_void_p), wg.ctypes.data_as(ctypes.c_void_p), 0)
names = ["dead-task", "prologue (desc, nmax, clear, n_contrib, wave max)", "checkpoint + entry loads, cull", "entry loop", "sync before flush", "flush", "kernel (per wave)"]
tot = float(ph[6])
for i, n in enumerate(names if tot else []):
    print(f"{n:55s} {float(ph[i]) / tot * 100:6.2f} % of wave time")
print("dead tasks (waves)", ph[8], "live tasks (waves)", ph[9], "waves entering the loop", ph[10], "kept entries / wave", float(ph[11]) / max(float(ph[10]), 1),
      "reduced entries / wave", float(ph[12]) / max(float(ph[10]), 1), "lanes with alpha > 0 per reduced entry", float(ph[13]) / max(float(ph[12]), 1))
t0 = np.zeros(4096, np.uint64); t1 = np.zeros(4096, np.uint64)
lib.gom_debug_wg_timeline(t0.ctypes.data_as(ctypes.c_void_p), t1.ctypes.data_as(ctypes.c_void_p))
a, b = t0.astype(np.float64) - float(t0.min()), t1.astype(np.float64) - float(t0.min())
span = b.max()
print("last launch: span %.0f ticks of 10 ns; sum of workgroup busy / span = %.1f workgroups in flight on average; start times: 25/50/75/100 %% of the workgroups started by %.2f / %.2f / %.2f / %.2f of the span"
      % (span, (b - a).sum() / span, *(np.quantile(a, q) / span for q in (0.25, 0.5, 0.75, 1.0))))
print("workgroups still running in the last 10 / 20 / 30 %% of the span:", [(b > span * f).sum() for f in (0.9, 0.8, 0.7)])
dur = b - a
print("workgroup duration / span: mean %.3f max %.3f; by (blockIdx & 3) = sub-range:" % (dur.mean() / span, dur.max() / span), [round(dur[i::4].mean() / span, 3) for i in range(4)])
w = wg[wg > 0].astype(np.float64)
print("workgroup busy cycles: min %.3g mean %.3g max %.3g (max/mean %.2f)" % (w.min(), w.mean(), w.max(), w.max() / w.mean()))
