#!/usr/bin/env python
"""Development aid: where a k_seg_T task's time goes, per wave, from a -DGOM_PHASE_PROF build (scripts/exp_build.py prof -DGOM_PHASE_PROF).
    GOM_HIP_LIB=gomavatar_amd/_variants/libgom_hip_prof.so python scripts/phase_prof_T.py"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gomavatar_amd import _lib
import bench
lib = ctypes.CDLL(_lib.LIB_PATH)
sys.argv = ["bench.py", "--inflight", "1", "--no-graph", "--no-cpu-baseline", "--no-modes", "--steps", "20", "--warmup", "4"]
bench.main()
ph = np.zeros((8192 * 4, 8), np.uint64)
lib.gom_debug_wave_phases(ph.ctypes.data_as(ctypes.c_void_p))
ph = ph[ph[:, 6] > 0].astype(np.float64)
names = ["task id + descriptor", "entry loads (+cull)", "ballot, staging, loop", "stores, LDS, queue head", "barrier", "after the last task"]
tasks = ph[:, 6].sum()
print("waves", len(ph), "tasks per wave: mean %.1f" % ph[:, 6].mean(), "cycles per wave-task: %.0f" % (ph[:, :5].sum() / tasks))
for i, n in enumerate(names):
    print(f"{n:28s} {ph[:, i].sum() / tasks:9.0f} cycles per wave-task   {ph[:, i].sum() / ph[:, :6].sum() * 100:5.1f} %")
w = ph.reshape(-1, 8)
