"""On the GPU box: per-frame and per-256-Gaussian-block tile counts of the metric workload (what k_preprocess_bwd's blocks own)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gomavatar_amd import _lib
from gomavatar_amd.workload import MetricWorkload
wl = MetricWorkload("cuda", subdiv=1, img=512, n_frames=8)
B = 8
step = wl.step(B); bt = wl.batches(step)[0]
step.cam = bt["cam"]; step.cams_dev.copy_(bt["cams_dev"])
step.forward_backward(wl.params, bt, bt["gt_rgb"], bt["gt_mask"], bt["bg"], graph=False)
torch.cuda.synchronize()
P = wl.params["appearance"].shape[1] if wl.params["appearance"].shape[0] == 3 else wl.params["appearance"].shape[0]
tt = step.state.export(_lib.BUF_TILES_TOUCHED, torch.empty(B * P, dtype=torch.int32, device="cuda")).cpu().numpy().reshape(B, P).astype(np.int64)
for b in range(B):
    t = tt[b]; pad = (-P) % 256
    s = np.pad(t, (0, pad)).reshape(-1, 256).sum(1)
    print("frame", b, "D", t.sum(), "max nt", t.max(), "nt>64:", int((t > 64).sum()), "block sum pcts 50/90/99/max", np.percentile(s, [50, 90, 99, 100]).astype(int))
