"""Development: what one 8-frame launch of k_seg_bwd_blk (GOM_OPT_BWD_MODE 2) does on the metric workload -- tasks, wave trips, skipped trips,
survivors, useful lanes.  Needs the stats build:  python scripts/exp_build.py blkstats -DGOM_BLK_STATS ; GOM_HIP_LIB=gomavatar_amd/_variants/libgom_hip_blkstats.so python scripts/blk_stats.py"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gomavatar_amd import _lib
from gomavatar_amd.workload import MetricWorkload
lib = _lib.load()
wl = MetricWorkload("cuda", subdiv=1, img=512, n_frames=8)
step = wl.step(8); bt = wl.batches(step)[0]
MODE = int(sys.argv[1]) if len(sys.argv) > 1 else 2
step.state.set_option(_lib.OPT_BWD_MODE, MODE)
stream = torch.cuda.Stream()
with torch.cuda.stream(stream):
    step.cam = bt["cam"]; step.cams_dev.copy_(bt["cams_dev"])
    step.forward_backward(wl.params, bt, bt["gt_rgb"], bt["gt_mask"], bt["bg"])
    torch.cuda.synchronize()
    lib.gom_debug_blk_stats(None, 1)
    step.forward_backward(wl.params, bt, bt["gt_rgb"], bt["gt_mask"], bt["bg"])
    torch.cuda.synchronize()
out = np.zeros(16, np.uint64)
lib.gom_debug_blk_stats(out.ctypes.data_as(ctypes.c_void_p), 0)
tasks, trips, skipped, surv, multi, nmax_sum, act_lanes = [int(x) for x in out[:7]]
print(f"live sub-range tasks {tasks}; wave trips {trips} ({trips / max(tasks, 1) / 4:.1f} per wave and task), skipped (no lane alive) {skipped} = {skipped / max(trips, 1):.2f}")
print(f"row items' survivors {surv} -> perfectly packed {surv / 4:.0f} wave trips: packing {surv / 4 / max(trips, 1):.2f}; sum of the longest list per task {nmax_sum} (x4 waves = {4 * nmax_sum})")
print(f"tasks with a list longer than one chunk: {multi}; lanes with alpha > 0 in the evaluated trips: {act_lanes} = {act_lanes / max((trips - skipped) * 64, 1):.3f} of the lanes")
if MODE == 0:
    pieces, surv, act, lanes = [int(x) for x in out[8:12]]
    print(f"pair kernel: live (sub-range, quadrant) pieces {pieces}; survivors of the stored cull evaluated {surv} ({surv / max(pieces, 1):.1f} per piece); with a lane alive {act} = {act / max(surv, 1):.2f}; "
          f"lanes alive in those {lanes} = {lanes / max(act * 64, 1):.3f} of 64; of all evaluated lanes {lanes / max(surv * 64, 1):.3f}")
    print(f"k_seg_T evaluates {int(out[12])} survivors over ALL pieces; the pieces some pixel is still alive in hold {surv}")
    tot, cur, lpt = [int(x) for x in out[13:16]]
    print(f"pieces of a task over its four waves: survivors {tot}; busiest wave as assigned (w, 3 - w) {cur} = {4 * cur / max(tot, 1):.2f} x the mean; dealt largest first {lpt} = {4 * lpt / max(tot, 1):.2f} x")
