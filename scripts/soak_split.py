#!/usr/bin/env python
"""Soak of gom_split_forward_backward: random (B, K, image size, body, graph or not, frames), the split step against the one launch sequence -- bitwise.
usage: python scripts/soak_split.py [seconds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from gomavatar_amd import synthetic as syn
from gomavatar_amd.pipeline import RenderStep, SplitRenderStep

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(time.time()))
t0, n = time.time(), 0
stream = torch.cuda.Stream()
while time.time() - t0 < budget:
    B = int(rng.choice([2, 4, 6, 8]))
    K = int(rng.choice([k for k in (2, 3, 4) if B % k == 0]))
    img = int(rng.choice([64, 96, 128, 176, 256]))
    graph = bool(rng.integers(0, 2))
    body = syn.make_body(0) if rng.random() < 0.3 else syn.icosphere_body(int(rng.integers(2, 4)))
    F, N = body["faces"].shape[0], body["canonical_vertex"].shape[0]
    gp = syn.make_gaussian_params(F, int(rng.integers(0, 100)))
    w = torch.from_numpy(body["canonical_lbs_weights"]).T
    w25 = torch.cat([w, torch.zeros(1, N)], 0).contiguous()
    faces = torch.from_numpy(body["faces"])
    params = {k: v.cuda() for k, v in dict(vertices=torch.from_numpy(body["canonical_vertex"]).T.contiguous(), so3=torch.from_numpy(gp["so3"]),
                                           scale=torch.from_numpy(gp["scale"]) * float(rng.uniform(1, 3)), appearance=torch.from_numpy(gp["appearance"])).items()}
    f0 = int(rng.integers(0, 1000))
    frames = [syn.make_frame(f0 + b, img) for b in range(B)]
    stack = lambda k: torch.from_numpy(np.stack([f[k][0] for f in frames])).contiguous().cuda()
    fr_b = {k: stack(k) for k in ("cnl_gtfms", "dst_Rs", "dst_Ts")}
    bg_b = stack("bgcolor")
    gt_rgb = torch.from_numpy(rng.uniform(0, 1, (B, img, img, 3)).astype(np.float32)).cuda()
    gt_mask = torch.from_numpy((rng.uniform(0, 1, (B, img, img)) > 0.5).astype(np.float32)).cuda()
    res = []
    from gomavatar_amd import _lib
    for st in (RenderStep(faces, N, (img, img), w25, batch=B), SplitRenderStep(faces, N, (img, img), w25, batch=B, split=K)):
        if B // K == 1 and hasattr(st, "parts"):
            st.state.set_option(_lib.OPT_SEG_SHIFT, 8)        # (a one-frame branch would pick 128-entry segments: bitwise equality needs the batch's 256)
        st.set_cameras([f["K"][0] for f in frames], [f["E"][0] for f in frames])
        stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(stream):
            for _ in range(3 if graph else 1):
                st.forward_backward(params, fr_b, gt_rgb, gt_mask, bg_b, graph=graph)
        stream.synchronize()
        assert not st.state.poll()[1]
        res.append([st.image.clone(), st.loss_partials.clone(), st.radii.clone()] + [st.grads[k].clone() for k in ("vertices", "so3", "scale", "appearance")])
        del st
    ok = all(torch.equal(a, b) for a, b in zip(*res))
    n += 1
    if not ok:
        print(f"MISMATCH: B={B} K={K} img={img} graph={graph} F={F} f0={f0}", [bool(torch.equal(a, b)) for a, b in zip(*res)], flush=True)
        sys.exit(1)
print(f"soak_split: {n} random configurations, all bitwise equal ({time.time() - t0:.0f} s)")
