#!/usr/bin/env python
"""bench.py -- rendered frames/sec (fwd+bwd) of the GoMAvatar hot path on MI355X.

One "step" = one batch of B frames (--batch, default 8) per GPU through the
whole hot path, forward AND backward, in ONE sequence of 17 kernel launches:
FK -> LBS -> per-face Gaussians -> splat forward (4-channel) -> fused
unpack+L1(rgb)+L1(mask) loss fwd/bwd -> splat backward -> face backward ->
vertex gather + LBS backward -> sum of the per-frame gradients, producing the
batch gradient for vertices / so3 / scale / appearance.  Workload (BASELINE.json
metric): 512x512, 55 104 Gaussians (SMPL-topology body, one midpoint
subdivision), synthetic poses/cameras/targets already resident in HBM when the
timed region starts.  `value` counts FRAMES per second (B per step per GPU).

N > 1 (launched by torch.distributed.run): frame-parallel data parallelism, B
frames per GPU per step, plus ONE RCCL all-reduce of the flat fp32 gradient
buffer (951 023 floats = the reference model's full parameter count) per step
inside the timed region.  Weak scaling: per-GPU work is fixed.

Prints one JSON line on rank 0 (contract in the task statement) carrying
`roofline` (dominant kernel, HIP-event timed on its own stream) and, at N=1,
`cpu_baseline` (the CPU oracle timed on this box's host cores).
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # the host driver only supports dmabuf IPC (needed by RCCL across processes)

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
MODEL_PARAMS_M = 951_023  # reference model at 55 104 Gaussians: all-reduce payload


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--img", type=int, default=512)
    ap.add_argument("--subdiv", type=int, default=1, help="0: 13 776, 1: 55 104 (metric), 2: 220 416 Gaussians")
    ap.add_argument("--frames", type=int, default=32, help="distinct synthetic frames cycled through")
    ap.add_argument("--batch", type=int, default=8, help="frames per step per GPU, rendered by one batched launch sequence")
    ap.add_argument("--inflight", type=int, default=3,
                    help="independent steps in flight per GPU, each on its own HIP stream with its own scratch; "
                         "1 = strictly one step after the other")
    ap.add_argument("--seg-shift", type=int, default=0, help="log2 of the tile-list segment size (0 = library default)")
    ap.add_argument("--no-graph", action="store_true", help="enqueue the 17 kernels of a frame one by one instead of replaying a hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=6)
    ap.add_argument("--task-grid-pct", type=int, default=0, help="GOM_OPT_TASK_GRID_PCT, 10..100 (0 = library default, 100)")
    return ap.parse_args()


def algorithmic_bytes(P, D, HW, C):
    """SURVEY.md section 8(d) byte model, per kernel, per frame."""
    return {
        "preprocess": P * (12 + 24 + 4 * C + 4) + P * (8 + 4 + 16 + 4 + 4),
        "scan_tiles": 0,
        "emit": 12 * D,
        "sort": 2 * 12 * D,                                   # one read + one write of the (key, value) pairs
        "seg_T": D * (4 + 8 + 16),                             # transmittance pre-pass: list + geometry attributes
        "seg_fwd": D * (4 + 8 + 16 + 4 * C),                  # B_ren_fwd, per-entry part
        "combine": HW * (4 * C + 4 + 4),                      # B_ren_fwd, per-pixel part
        "seg_bwd": HW * (4 * C + 4 + 4) + D * (4 + 8 + 16 + 4 * C) + P * (8 + 12 + 4 + 4 * C),   # B_ren_bwd
        "preprocess_bwd": P * (12 + 24 + 4 + 12 + 8) + P * (12 + 24),
    }


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world} (launch with torch.distributed.run)"
    dev = torch.device("cuda", torch.cuda.current_device())

    from __graft_entry__ import build
    from gomavatar_amd import build as hip_build
    if rank == 0 and hip_build.needs_build():
        build()
    if world > 1:
        dist.barrier()
    from gomavatar_amd import _lib, synthetic as syn
    from gomavatar_amd.pipeline import RenderStep

    # ---------------- workload (synthetic, seeded; resident in HBM) ----------------
    img = args.img
    body = syn.make_body(args.subdiv)
    N, F = body["canonical_vertex"].shape[0], body["faces"].shape[0]
    w = torch.from_numpy(body["canonical_lbs_weights"]).T
    w25 = torch.cat([w, torch.zeros(1, N)], 0).contiguous()
    faces = torch.from_numpy(body["faces"])
    B = max(1, args.batch)
    gen = RenderStep(faces, N, (img, img), w25, device=dev)   # single-frame instance: renders the targets
    step = RenderStep(faces, N, (img, img), w25, device=dev, batch=B)

    def dev_params(seed):
        gp = syn.make_gaussian_params(F, seed)
        return dict(vertices=torch.from_numpy(body["canonical_vertex"]).T.contiguous().to(dev), so3=torch.from_numpy(gp["so3"]).to(dev),
                    scale=torch.from_numpy(gp["scale"]).to(dev), appearance=torch.from_numpy(gp["appearance"]).to(dev))

    # flat fp32 gradient buffer = the all-reduce payload; the hot path's gradients are views into it.
    # Padded to the reference model's full parameter count so the collective moves what a real step moves.
    from gomavatar_amd.parallel import FrameParallel, shapes_for_model
    n_own = 3 * N + 9 * F
    fp = FrameParallel(shapes_for_model(N, F), dev, pad_to=MODEL_PARAMS_M if args.subdiv == 1 else n_own)
    flat = fp.grads.flat
    for k in ("vertices", "so3", "scale", "appearance"):
        step.grads[k] = fp.grads[k]
    params = dev_params(1)
    target_params = dev_params(2)
    frames = []
    for i in range(max(args.frames, B)):
        fr = syn.make_frame(rank * 1000 + i, img)  # each rank renders different frames
        d = {k: torch.from_numpy(fr[k][0]).contiguous().to(dev) for k in ("cnl_gtfms", "dst_Rs", "dst_Ts")}
        d["K"], d["E"], d["bg"] = fr["K"][0], fr["E"][0], torch.from_numpy(fr["bgcolor"][0]).to(dev)
        # target = render of a different parameter set (so gradients are non-zero), produced by the HIP path itself
        gen.set_camera(d["K"], d["E"])
        dummy_rgb = torch.zeros((img, img, 3), device=dev)
        dummy_m = torch.zeros((img, img), device=dev)
        gen.forward_backward(target_params, d, dummy_rgb, dummy_m, d["bg"], backward=False)
        rgb, mask = gen.rgb_mask()
        d["gt_rgb"] = (rgb[0] * mask[0, ..., None] + d["bg"] * (1 - mask[0, ..., None])).contiguous().clone()
        d["gt_mask"] = mask[0].contiguous().clone()
        frames.append(d)
    torch.cuda.synchronize()
    del gen
    # batches of B consecutive frames: stacked per-frame inputs + one device camera array each
    batches = []
    for j in range(len(frames) // B):
        grp = frames[j * B:(j + 1) * B]
        bt = {k: torch.stack([g[k] for g in grp]).contiguous() for k in ("cnl_gtfms", "dst_Rs", "dst_Ts", "gt_rgb", "gt_mask", "bg")}
        if B == 1:
            bt = {k: v[0] for k, v in bt.items()}
        step.set_cameras([g["K"] for g in grp], [g["E"] for g in grp]) if B > 1 else step.set_camera(grp[0]["K"], grp[0]["E"])
        torch.cuda.synchronize()
        bt["cams_dev"], bt["cam"] = step.cams_dev.clone(), step.cam
        batches.append(bt)

    # steps in flight: slot k owns a stream, a RenderStep (scratch + intermediates) and a gradient buffer
    S = max(1, args.inflight)
    slots = [dict(step=step, fp=fp, stream=torch.cuda.Stream(device=dev))]  # (the legacy NULL stream cannot be graph-captured)
    for k in range(1, S):
        st_k = RenderStep(faces, N, (img, img), w25, device=dev, batch=B)
        fp_k = FrameParallel(shapes_for_model(N, F), dev, pad_to=flat.numel())
        for name in ("vertices", "so3", "scale", "appearance"):
            st_k.grads[name] = fp_k.grads[name]
        slots.append(dict(step=st_k, fp=fp_k, stream=torch.cuda.Stream(device=dev)))

    if args.seg_shift:
        for sl in slots:
            sl["step"].state.set_option(_lib.OPT_SEG_SHIFT, args.seg_shift)
    # GOM_OPT_TASK_GRID_PCT: with several steps in flight, giving every step's persistent task-queue grids HALF of the workgroup
    # slots lets kernels of different steps run side by side (+4 % frames/s: 13.1-13.2 k) at the price of every kernel running longer
    # (k_seg_bwd 360 us instead of 230) -- the default keeps full grids so that the per-kernel durations behind `roofline` stay those
    # of the kernels themselves.
    if args.task_grid_pct:
        for sl in slots:
            sl["step"].state.set_option(_lib.OPT_TASK_GRID_PCT, args.task_grid_pct)

    def run_step(i):
        bt = batches[i % len(batches)]
        sl = slots[i % S]
        with torch.cuda.stream(sl["stream"]):
            sl["step"].cam = bt["cam"]
            if B > 1:
                sl["step"].cams_dev.copy_(bt["cams_dev"], non_blocking=True)   # this step's cameras (device array read by the kernels)
            sl["step"].forward_backward(params, bt, bt["gt_rgb"], bt["gt_mask"], bt["bg"], graph=not args.no_graph)
            sl["fp"].all_reduce_grads()  # no-op at world size 1

    torch.cuda.synchronize()
    for i in range(args.warmup):
        run_step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        run_step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    n_pairs, overflow = step.state.poll()
    assert not overflow, "pair buffer overflow during the benchmark"
    assert all(torch.isfinite(g).all() for g in step.grads.values()), "non-finite gradients"

    # ---------------- per-kernel times: HIP events recorded by the library on each launch stream ----------------
    # Same loop, same frames in flight as the timed region (profiling brackets every launch with events, so the
    # frame is enqueued kernel by kernel instead of replayed from its graph).  The rocprofv3 --kernel-trace --stats
    # summary of this very command (profiles/) must agree with these averages.
    for sl in slots:
        sl["step"].state.set_option(_lib.OPT_PROFILE, 1)
    acc, n_prof = {}, 0
    rounds = max(4, 48 // S)
    for r in range(rounds):
        for k in range(S):
            run_step(r * S + k)
        for sl in slots:
            for kname, v in sl["step"].state.kernel_times_ms().items():
                acc[kname] = acc.get(kname, 0.0) + v
            nd, _ = sl["step"].state.poll()
            acc["D"] = acc.get("D", 0) + nd
            n_prof += 1
    # the same kernels with ONE step in flight (nothing else on the GPU): what a launch costs when it owns the chip
    iso, n_iso = {}, 8
    for r in range(n_iso):
        torch.cuda.synchronize()
        run_step(r * S)          # slot 0
        torch.cuda.synchronize()
        for kname, v in slots[0]["step"].state.kernel_times_ms().items():
            iso[kname] = iso.get(kname, 0.0) + v / n_iso
    for sl in slots:
        sl["step"].state.set_option(_lib.OPT_PROFILE, 0)
    torch.cuda.synchronize()
    kt = {k: acc[k] / n_prof for k in _lib.KERNEL_NAMES}
    D_avg = acc["D"] / n_prof
    abytes = algorithmic_bytes(F * B, D_avg, img * img * B, 4)   # per launch: B frames (D_avg already counts all B)
    # Dominant kernel = the one that costs most when it owns the chip.  (With several steps in flight a kernel's duration also
    # counts the time it spends sharing CUs with the other steps' kernels -- a property of the mix, not of the kernel; the
    # few long tile lists of k_sort, two launches under one event pair, stretch most that way.)  Both durations are reported.
    dom = max(iso, key=iso.get)
    achieved = abytes[dom] / (kt[dom] * 1e-3) / 1e9
    # HBM traffic of that kernel from the PMC passes (FETCH_SIZE / WRITE_SIZE collected separately, FETCH doubled per
    # MI355X_MICROARCH.md), recorded by scripts/collect_profiles.sh into profiles/<round>_traffic.json
    traffic = None
    tj = os.path.join(ROOT, "profiles", "r01_traffic.json")
    if os.path.exists(tj) and args.subdiv == 1 and img == 512:
        try:
            tjd = json.load(open(tj))
            traffic = (int(tjd.get("k_" + dom, {}).get("hbm_bytes")) or None) if tjd.get("batch", 1) == B else None
        except Exception:
            traffic = None
    roofline = {"kernel": "k_" + dom, "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "avg_us": round(kt[dom] * 1e3, 2),
                "algorithmic_bytes": int(abytes[dom]),
                "all_kernels_us": {k: round(v * 1e3, 2) for k, v in kt.items()}, "pairs_D": int(D_avg),
                # same launch with the GPU to itself (one step in flight): duration and the fraction it would reach
                "alone": {"avg_us": round(iso[dom] * 1e3, 2), "frac": round(abytes[dom] / (iso[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                          "all_kernels_us": {k: round(iso[k] * 1e3, 2) for k in _lib.KERNEL_NAMES}}}

    out = {
        "metric": "rendered frames/sec (fwd+bwd) at 512x512, ~50k Gaussians; PSNR vs ref",
        "value": round(world * B * args.steps / elapsed, 2),
        "unit": "frames/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"GoMAvatar hot path fwd+bwd, {F} Gaussians / {N} verts, {img}x{img}, {B} frames per GPU per step "
                               "(one batched launch sequence)" + (", + RCCL all-reduce of the flat grad buffer" if world > 1 else ""),
                   "gaussians": F, "image": [img, img], "frames_per_step": world * B, "frames_per_gpu_per_step": B,
                   "parallelism": f"frame-dp{world}", "steps_in_flight_per_gpu": S,
                   "allreduce_floats": int(flat.numel()) if world > 1 else 0},
        "roofline": roofline,
    }

    # ---------------- CPU baseline: the oracle on this box's host cores (rank 0, N=1 only) ----------------
    if world == 1 and not args.no_cpu_baseline:
        from oracle import geometry as og, raster as orast
        cores = os.cpu_count() or 1
        threads = max(1, min(cores, 64))
        torch.set_num_threads(threads)
        orast.set_threads(threads)
        pc = {k: v.detach().cpu() for k, v in params.items()}
        times = []
        check = RenderStep(faces, N, (img, img), w25, device=dev)   # single-frame HIP render of the same frames: PSNR vs the oracle
        mse_sum, mse_n = 0.0, 0
        for i in range(args.cpu_frames + 1):
            fr = {k: torch.from_numpy(v) for k, v in syn.make_frame(i, img).items()}
            po = {k: v.clone().requires_grad_() for k, v in pc.items()}
            t1 = time.perf_counter()
            o_rgb, o_mask, _ = og.render_path(po, fr, faces, w25, img)
            gt = frames[i % len(frames)]
            l1, l2 = og.l1_losses(og.unpack(o_rgb, o_mask, fr["bgcolor"]), o_mask, gt["gt_rgb"].cpu()[None], gt["gt_mask"].cpu()[None])
            (l1 + 5.0 * l2).backward()
            times.append(time.perf_counter() - t1)
            fd = frames[i % len(frames)]
            if i < len(frames):   # frames[i] was generated from make_frame(i) on rank 0: identical inputs on both sides
                check.set_camera(fd["K"], fd["E"])
                check.forward_backward(params, fd, fd["gt_rgb"], fd["gt_mask"], fd["bg"], backward=False)
                h_rgb, h_mask = check.rgb_mask()
                d = torch.cat([h_rgb[0].cpu() - o_rgb[0].detach(), (h_mask[0].cpu() - o_mask[0].detach())[..., None]], -1).double()
                mse_sum += float((d ** 2).mean()); mse_n += 1
        times = times[1:]  # first frame warms caches / page-faults
        if mse_n:
            out["psnr_vs_oracle_db"] = round(-10.0 * math.log10(max(mse_sum / mse_n, 1e-30)), 2)   # "PSNR vs ref" of BASELINE.json's metric
        out["cpu_baseline"] = {"value": round(len(times) / sum(times), 3), "unit": "frames/s", "cores": threads, "kind": "port",
                               "sample": f"{len(times)} frames of the same workload (fwd+bwd) through the CPU oracle, "
                                         f"{threads} threads (torch + OpenMP), host has {cores} logical cores"}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
