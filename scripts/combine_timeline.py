"""Development aid: when every workgroup of k_combine_fwd starts and ends (riders = the first eight).
    python scripts/exp_build.py prof4 -DGOM_PHASE_PROF=4
    GOM_HIP_LIB=gomavatar_amd/_variants/libgom_hip_prof4.so python scripts/combine_timeline.py"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from gomavatar_amd import _lib
import bench
lib = ctypes.CDLL(_lib.LIB_PATH)
sys.argv = ["bench.py", "--inflight", "1", "--no-cpu-baseline", "--no-modes", "--steps", "20", "--warmup", "4"]
bench.main()
t0 = np.zeros(4096 * 4, np.uint64); t1 = np.zeros(4096 * 4, np.uint64)
lib.gom_debug_wg_timeline(t0.ctypes.data_as(ctypes.c_void_p), t1.ctypes.data_as(ctypes.c_void_p))
n = 2056
a, b = t0[:n].astype(np.float64), t1[:n].astype(np.float64)
o = a[a > 0].min(); a = (a - o) / 100; b = (b - o) / 100
print("riders: start", np.round(a[:8], 1), "end", np.round(b[:8], 1))
ta, tb = a[8:], b[8:]
ok = t0[8:n] > 0
print("tile workgroups: start quantiles", [round(float(np.quantile(ta[ok], q)), 1) for q in (0, .5, 1)], "end quantiles", [round(float(np.quantile(tb[ok], q)), 1) for q in (0.1, .5, .9, 1)],
      "duration mean %.1f max %.1f" % ((tb - ta)[ok].mean(), (tb - ta)[ok].max()))
