#!/usr/bin/env python
"""Novel-view / novel-pose rendering rate of the whole avatar (the number the reference's paper quotes as "43 FPS"): `Model` in
eval mode under no_grad -- skinning, face Gaussians, splat rasterizer, mesh normal map, shadow MLP, composition on the
background (eval.py:334-350) -- one frame per call, a different pose and camera each time.

    python scripts/bench_inference.py [--img 512] [--level 1] [--frames 300]"""
import argparse, json, os, sys, time
from types import SimpleNamespace as NS
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gomavatar_amd import synthetic as syn
from gomavatar_amd.model import Model
from gomavatar_amd.train_util import unpack

ap = argparse.ArgumentParser()
ap.add_argument("--img", type=int, default=512); ap.add_argument("--level", type=int, default=1); ap.add_argument("--frames", type=int, default=300)
ap.add_argument("--graph", action="store_true", help="capture the frame once in a HIP graph (device-resident camera, fixed-capacity shadow list) and replay it")
ap.add_argument("--overlap", action="store_true", help="mesh branch and splat rasterizer on two streams (Model.overlap_branches)")
ap.add_argument("--subdivide", action="store_true", help="one mesh subdivision first (the reference subdivides during training: 4x the faces)")
a = ap.parse_args()
cfg = NS(img_size=(a.img, a.img), canonical_geometry=NS(sigma=1e-3, radius_scale=1.0, deform_so3=True, deform_scale=True), appearance=NS(color_init=0.5),
         normal_renderer=NS(sigma=1e-5, soft_mask=True), shadow_module=NS(name="basic", multires=6, mlp_width=128, mlp_depth=3, skips=(4,)),
         lbs_weights=NS(refine=False))
model = Model(cfg, syn.make_body(a.level))
with torch.no_grad():
    model.appearance.copy_(torch.rand(model.appearance.shape, device="cuda", generator=torch.Generator(device="cuda").manual_seed(0)))
    model.shadow_module.block_mlps[-1].weight.normal_(0, 0.3)
if a.subdivide:
    model.subdivide()
model.eval()
frames = [{k: torch.from_numpy(v).cuda() for k, v in syn.make_frame(i, a.img).items()} for i in range(16)]

def render_eager(fr):
    with torch.no_grad():
        rgbs, masks, _ = model(fr["K"], fr["E"], fr["cnl_gtfms"], fr["dst_Rs"], fr["dst_Ts"])
        return unpack(rgbs, masks, fr["bgcolor"])

model.overlap_branches = a.overlap
if a.graph:
    from gomavatar_amd.train_util import GraphedRender
    # the graph evaluates the shadow MLP on a fixed number of pixel slots: 1.3 x the largest mesh footprint of the sequence
    with torch.no_grad():
        model.shadow_capacity = int(1.3 * max(int((model(fr["K"], fr["E"], fr["cnl_gtfms"], fr["dst_Rs"], fr["dst_Ts"])[1] > 0).sum()) for fr in frames))
    render = GraphedRender(model)
else:
    render = render_eager

for i in range(30):
    out = render(frames[i % 16])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(a.frames):
    out = render(frames[i % 16])
torch.cuda.synchronize()
dt = time.perf_counter() - t0
lat = []
for i in range(50):                       # latency of one frame, nothing else in flight
    torch.cuda.synchronize(); t = time.perf_counter()
    out = render(frames[i % 16]); torch.cuda.synchronize()
    lat.append(time.perf_counter() - t)
lat.sort()
print(json.dumps({"frames_per_s": round(a.frames / dt, 1), "latency_ms_median": round(lat[len(lat) // 2] * 1e3, 3), "img": a.img,
                  "faces": int(model.faces.shape[0]), "graph": bool(a.graph),
                  "max_abs_diff_vs_eager": round(float((render(frames[3]) - render_eager(frames[3])).abs().max()), 6)}))
