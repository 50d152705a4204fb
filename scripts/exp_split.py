#!/usr/bin/env python
"""Experiment (round 6): ONE step of 8 frames as K concurrent launch sequences (gom_split_forward_backward) against the single 8-frame launch
sequence: bitwise the same gradients, and the rate with Adam behind each step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gomavatar_amd.workload import MetricWorkload
from gomavatar_amd.parallel import FrameParallel, FlatAdam, shapes_for_model

dev = torch.device("cuda:0")
wl = MetricWorkload(dev, subdiv=1, img=512, n_frames=8)
B = 8
stream = torch.cuda.Stream()

def make(K, graph=True, adam=True, pct=0):
    fp = FrameParallel(shapes_for_model(wl.N, wl.F), dev)
    for name in ("vertices", "so3", "scale", "appearance"):
        fp.params[name].copy_(wl.params[name])
    opt = FlatAdam(fp, {"default": 1e-5})
    params = dict(fp.params.items())
    st = wl.step(B, split=K)
    if pct:
        from gomavatar_amd import _lib
        st.state.set_option(_lib.OPT_TASK_GRID_PCT, pct)
    for name in ("vertices", "so3", "scale", "appearance"):
        st.grads[name] = fp.grads[name]
    bt = wl.batches(st)[0]
    def step():
        with torch.cuda.stream(stream):
            st.cam = bt["cam"]; st.cams_dev = bt["cams_dev"]
            st.forward_backward(params, bt, bt["gt_rgb"], bt["gt_mask"], bt["bg"], graph=graph)
            if adam:
                opt.step(1.0 / B)
    return step, st, fp

def rate(step, n=1500, warm=60):
    for _ in range(warm): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize()
    return n / (time.perf_counter() - t0)

# bitwise: gradients, images, losses, radii of the split step against the single launch sequence (no Adam: same parameters)
ref = None
for K in (1, 2, 4, (3, 3, 2), (5, 3), (2, 2, 2, 2)):
    step, st, fp = make(K, adam=False)
    for _ in range(3): step()
    torch.cuda.synchronize()
    got = (fp.grads.flat.clone(), st.image.clone(), st.loss_partials.clone(), st.radii.clone())
    if ref is None:
        ref = got
    else:
        print(f"K={K}: bitwise equal to one launch sequence:", [bool(torch.equal(a, b)) for a, b in zip(ref, got)], flush=True)
for K, pct in ((1, 0), (2, 0), ((3, 3, 2), 0), ((5, 3), 0), ((6, 2), 0), ((4, 2, 2), 0), (1, 0), (2, 0), ((3, 3, 2), 0), ((5, 3), 0)):
    step, st, fp = make(K, pct=pct)
    r = rate(step)
    print(f"K={K} grid {pct or 100} %: {r:.1f} steps/s = {r * B / 1e3:.2f} k frames/s ({1e3 / r:.4f} ms per 8-frame step)", flush=True)
