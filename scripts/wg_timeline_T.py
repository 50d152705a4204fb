"""Development aid: per-workgroup lifetime, number of tasks, longest and last task of a persistent segment kernel.
    python scripts/exp_build.py prof -DGOM_PHASE_PROF=1      # k_seg_T (resident grid 2048); =2: k_seg_bwd_pair (GOM_TL_GRID=1280)
    GOM_HIP_LIB=gomavatar_amd/_variants/libgom_hip_prof.so [GOM_TL_GRID=1280] python scripts/wg_timeline_T.py"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from gomavatar_amd import _lib
import bench
lib = ctypes.CDLL(_lib.LIB_PATH)
sys.argv = ["bench.py", "--inflight", "1", "--no-cpu-baseline", "--no-modes", "--steps", "20", "--warmup", "4"]
bench.main()
n = int(os.environ.get("GOM_TL_GRID", "2048"))
t0 = np.zeros(4096 * 4, np.uint64); t1 = np.zeros(4096 * 4, np.uint64); wg = np.zeros(4096 * 4, np.uint64)
lib.gom_debug_wg_timeline(t0.ctypes.data_as(ctypes.c_void_p), t1.ctypes.data_as(ctypes.c_void_p))
lib.gom_debug_phase_counters(None, wg.ctypes.data_as(ctypes.c_void_p), 0)
a, b, w = t0[:n].astype(np.float64), t1[:n].astype(np.float64), wg[:n]
o = a.min(); a = (a - o) / 100; b = (b - o) / 100
tasks = (w & np.uint64(0xffff)).astype(np.float64); lastsurv = ((w >> np.uint64(16)) & np.uint64(0xff)).astype(int); maxsurv = ((w >> np.uint64(24)) & np.uint64(0xff)).astype(int)
last = ((w >> np.uint64(32)) & np.uint64(0xffff)).astype(np.float64) / 100; mx = (w >> np.uint64(48)).astype(np.float64) / 100
print("end quantiles (us)", [round(float(np.quantile(b, q)), 1) for q in (0.01, 0.1, 0.5, 0.9, 1.0)])
print("tasks per workgroup mean %.1f; mean task %.1f us; LAST task: mean %.1f us, 90%% %.1f, max %.1f; LONGEST task per workgroup: mean %.1f, max %.1f" % (
    tasks.mean(), ((b - a) / tasks).mean(), last.mean(), np.quantile(last, .9), last.max(), mx.mean(), mx.max()))
late = b > np.quantile(b, 0.9)
print("latest 10 %%: tasks %.1f, last task %.1f us (survivors of wave 0: %.1f), longest %.1f us (survivors %.1f)" % (tasks[late].mean(), last[late].mean(), lastsurv[late].mean(), mx[late].mean(), maxsurv[late].mean()))
print("all: survivors (wave 0) of last task %.1f, of the longest task %.1f" % (lastsurv.mean(), maxsurv.mean()))
print("start of the last task (us): quantiles", [round(float(np.quantile(b - last, q)), 1) for q in (0.01, 0.1, 0.5, 0.9, 1.0)])
