#!/usr/bin/env python
"""bench.py -- rendered frames/sec (fwd+bwd) of the GoMAvatar hot path on MI355X.

One "step" = one batch of B frames per GPU (--batch; default 8 at N = 1 = BASELINE configs[1], 1 at N > 1 = configs[3]'s literal
"one frame per GPU") through the whole hot path, forward AND backward, in ONE sequence of kernel launches: FK -> LBS -> per-face Gaussians -> splat forward (4-channel) -> fused unpack + L1(rgb) +
L1(mask) loss fwd/bwd -> splat backward -> face backward -> vertex gather + LBS backward -> sum of the per-frame gradients,
producing the batch gradient for vertices / so3 / scale / appearance -> (N > 1: ONE all-reduce of the flat gradient buffer) -> Adam on the
flat parameter buffer (the reference's optimizer, one native launch), so that step k + 1 renders with the parameters step k produced.  Workload (BASELINE.json metric; built by
gomavatar_amd.workload, the same object the -m gpu parity tests check): 512x512, 55 104 Gaussians (SMPL-topology body, one
midpoint subdivision), synthetic poses / cameras / targets already resident in HBM when the timed region starts.
`value` counts FRAMES per second, with ONE step in flight per GPU by default (what an optimizer loop can do: step k + 1
needs step k's update).  The other operating points are measured in the same run and reported under `modes`.

`python bench.py --gpus N` launches itself: with WORLD_SIZE unset and N > 1 it re-executes under torch.distributed.run
(one rank per GPU, RCCL); with fewer devices than ranks (the 1-GPU development lease) the ranks share device 0 over gloo --
a functional proof of the N > 1 path, labelled as such.  N > 1: frame-parallel data parallelism, ONE frame per GPU per
step by default (BASELINE configs[3]; `modes.b8_*` repeats it with 8), plus ONE all-reduce of the flat fp32 gradient buffer
(951 023 floats = the reference model's parameter count) and the Adam step per step inside the timed region.  Weak scaling:
per-GPU work is fixed; `config.local_only_fps` is the same loop without the collective (what N x one GPU would do).

Timing: W warm-up steps, then EXACTLY K steps between barrier + synchronize on both sides, MAX over ranks.  When such a
region is shorter than 0.25 s (the driver's K = 20 is 17 ms) it is REPEATED -- each repetition again exactly K steps between
the same brackets -- and `ms_per_step` / `value` are taken over all repetitions (`timed_regions` says how many).

Prints one JSON line on rank 0 carrying `roofline` (dominant kernel, HIP-event timed on its own stream), `modes`, and at N=1
`cpu_baseline` (the CPU oracle on this box's host cores, 1 thread and all physical cores, S and M sizes).
"""
from __future__ import annotations

import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # the host driver only supports dmabuf IPC (needed by RCCL across processes)

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
N_SIMD, SCLK_HZ = 256 * 4, 2.4e9     # 256 CUs x 4 SIMDs at the 2.4 GHz engine clock (MI355X_MICROARCH.md)
ISSUE_TICKS_PER_VALU = 3.8           # mean issue ticks per wave-level VALU instruction of the segment kernels' mix at 4 waves / SIMD (profiles/r03_valu_rate.txt)
MODEL_PARAMS_M = 951_023  # reference model at 55 104 Gaussians: all-reduce payload
MIN_TIMED_S = 0.25
PROFILE_TAG = "r06"
ADAM_LR = 1e-9   # the reference's Adam arithmetic and traffic at a rate that leaves the synthetic workload the parity tests check unchanged over 10^4 timed steps


def note(msg):
    """Progress on stderr (never on stdout: the one JSON line is the contract)."""
    if os.environ.get("RANK", "0") == "0":
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--img", type=int, default=512)
    ap.add_argument("--subdiv", type=int, default=1, help="0: 13 776, 1: 55 104 (metric), 2: 220 416 Gaussians")
    ap.add_argument("--frames", type=int, default=32, help="distinct synthetic frames cycled through")
    ap.add_argument("--batch", type=int, default=0, help="frames per step per GPU, rendered by one batched launch sequence (0 = 8 at N = 1, 1 at N > 1)")
    ap.add_argument("--split", type=int, default=0, help="the step's frames as this many CONCURRENT launch sequences forked and joined inside the step "
                    "(gom_split_forward_backward: same bits, tails filled; 0 = 2 for a step of 8 frames at N = 1, else 1)")
    ap.add_argument("--inflight", type=int, default=1,
                    help="independent steps in flight per GPU, each on its own HIP stream with its own scratch; "
                         "1 (default) = strictly one step after the other, as an optimizer loop runs")
    ap.add_argument("--seg-shift", type=int, default=0, help="log2 of the tile-list segment size (0 = library default)")
    ap.add_argument("--no-graph", action="store_true", help="enqueue the kernels of a step one by one instead of replaying a hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-modes", action="store_true", help="skip the other operating points (modes) and the raster-only / full-step figures")
    ap.add_argument("--cpu-frames", type=int, default=5, help="timed frames per CPU-baseline row (median; after 3 warm-up frames at M, 1 at S)")
    ap.add_argument("--no-configs", action="store_true", help="skip the other BASELINE configs (cfg 3 at 1024^2 / 540^2, cfg 5) of the `configs` block")
    ap.add_argument("--attach-adam", action="store_true", help="the optimizer as the last launch of the frame step's recorded graph (measured slower: off)")
    ap.add_argument("--no-adam", action="store_true", help="leave the optimizer step out of the timed loop (round-2 behaviour)")
    ap.add_argument("--task-grid-pct", type=int, default=0, help="GOM_OPT_TASK_GRID_PCT, 10..100 (0 = library default, 100)")
    ap.add_argument("--bwd-mode", type=int, default=-1, help="GOM_OPT_BWD_MODE (-1 = library default)")
    ap.add_argument("--sort-mode", type=int, default=-1, help="GOM_OPT_SORT_MODE (-1 = library default)")
    ap.add_argument("--train-curve", action="store_true", help="run BASELINE configs[1] at its stated shape now (scripts/train_synthetic.py: 3 000 iterations through a "
                                                               "subdivision, ~10 s) and report it under modes.cfg2_train_curve; default: cite the committed profiles/ curve")
    ap.add_argument("--backend", default="auto", help="auto: nccl (= RCCL) with one device per rank, gloo when ranks share a device")
    return ap.parse_args()


def self_launch(args) -> None:
    """`python bench.py --gpus N` with N > 1 and no launcher around it: become the launcher."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd))


def physical_cores() -> int:
    try:
        seen, phys, core = set(), None, None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                core = line.split(":")[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    seen.add((phys, core))
                phys = core = None
        if seen:
            return len(seen)
    except OSError:
        pass
    return os.cpu_count() or 1


PEER_PROBE = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from gomavatar_amd.parallel import PeerAllReduce
dev = int(sys.argv[2]); torch.cuda.set_device(dev)
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
ar = PeerAllReduce(65536 + 3, f"cuda:{dev}", timeout_s=20.0)
ok = True
for epoch in range(3):
    g = torch.Generator().manual_seed(10 * epoch)
    parts = [torch.randn(65536 + 3, generator=g) for _ in range(world)]
    ar.buffer.copy_(parts[rank].cuda())
    out = ar.run(scale=1.0); ar.check()
    ref = parts[0].clone()
    for r in range(1, world):
        ref = ref + parts[r]
    ok = ok and bool(torch.equal(out.cpu(), ref))
ar.close(); dist.destroy_process_group()
sys.exit(0 if ok else 3)
"""


def peer_probe(world, rank, dev_idx):
    """The direct peer exchange touches other processes' device memory from inside running kernels: where that is not possible (IPC mapping refused, no
    peer access between two devices, ...) the failure mode can be a GPU memory fault that takes the PROCESS down -- and the bench line with it.  So the
    exchange is tried first in CHILD processes (one per rank, their own gloo group on another port, the same devices, a small buffer, bitwise check): only
    when every child exits 0 do the parents run the peer phases.  -> (ok on every rank, note)"""
    import torch
    import torch.distributed as dist
    env = {k: v for k, v in os.environ.items() if not k.startswith("TORCHELASTIC") and k not in ("GROUP_RANK", "ROLE_RANK", "ROLE_NAME")}
    env["MASTER_PORT"] = str(int(os.environ.get("MASTER_PORT", "29500")) + 733)
    env["MASTER_ADDR"] = "127.0.0.1"
    err = None
    try:
        r = subprocess.run([sys.executable, "-c", PEER_PROBE, ROOT, str(dev_idx)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=240)
        if r.returncode != 0:
            err = f"probe exit code {r.returncode}: " + (r.stderr.strip().splitlines() or ["?"])[-1][:200]
    except Exception as e:
        err = f"{type(e).__name__}: {e}"
    got = [None] * world
    dist.all_gather_object(got, err)
    bad = [f"rank {i}: {e}" for i, e in enumerate(got) if e]
    return (not bad), (bad[0] if bad else None)


def algorithmic_bytes(P, D, HW, C):
    """SURVEY.md section 8(d) byte model, per kernel, per launch (P Gaussians, D pairs, HW pixels of the whole launch)."""
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
        # depth ranking (not in the reference's pipeline, whose global radix sort it replaces): depth + visibility read twice,
        # 8-byte keys written + read, 48-byte records gathered and written in rank order
        "depth_hist": P * 8,
        "depth_rank": P * (8 + 8 + 8) + P * (36 + 52),
    }


class Runner:
    """S steps in flight of B frames each on one GPU: slot k owns a stream, a RenderStep (scratch + intermediates) and a flat
    gradient buffer (the all-reduce payload; the hot path's gradients are views into it)."""

    def __init__(self, wl, B, S, graph, world, args, impl="collective", split=1):
        import torch
        from gomavatar_amd import _lib
        from gomavatar_amd.parallel import FrameParallel, shapes_for_model
        self.torch, self.wl, self.B, self.S, self.graph, self.world = torch, wl, B, S, graph, world
        self.split = split if (split > 1 and B % split == 0 and B // split >= 1) else 1
        pad = 0   # the native step exchanges what it trains (vertices / so3 / scale / appearance); the reference model's full 951 023 floats are exchanged by `modes.model_parallel` (ModelFrameParallel: real weights)
        self.slots = []
        self.adam = not args.no_adam
        from gomavatar_amd.parallel import FlatAdam
        for k in range(S):
            st = wl.step(B, split=self.split)
            fp = FrameParallel(shapes_for_model(wl.N, wl.F), wl.device, pad_to=pad, impl=impl)
            for name in ("vertices", "so3", "scale", "appearance"):
                st.grads[name] = fp.grads[name]
                fp.params[name].copy_(wl.params[name])   # every slot trains its own replica of the parameters (S > 1: independent steps)
            # (Measured: the Adam launch as the LAST launch of the frame step's own recorded graph -- gom_state_set_frame_optimizer, --attach-adam --
            #  is slower than a plain launch behind the graph: 13.96 k instead of 14.09 k frames/s, 4.65 k instead of 4.84 k at B = 1.  The ~9 us
            #  between two graph replays are there either way, and the plain launch runs inside them.)
            self.attached = self.adam and world == 1 and graph and args.attach_adam
            opt = FlatAdam(fp, {"default": ADAM_LR}, graphable=self.attached) if self.adam else None
            if self.attached:
                opt.attach(st.state, 1.0 / B)
            if args.seg_shift:
                st.state.set_option(_lib.OPT_SEG_SHIFT, args.seg_shift)
            if args.task_grid_pct:
                st.state.set_option(_lib.OPT_TASK_GRID_PCT, args.task_grid_pct)
            if args.bwd_mode >= 0:
                st.state.set_option(_lib.OPT_BWD_MODE, args.bwd_mode)
            if args.sort_mode >= 0:
                st.state.set_option(_lib.OPT_SORT_MODE, args.sort_mode)
            self.slots.append(dict(step=st, fp=fp, opt=opt, params=dict(fp.params.items()), stream=torch.cuda.Stream(device=wl.device)))   # (the legacy NULL stream cannot be graph-captured)
        self.batches = wl.batches(self.slots[0]["step"])
        if os.environ.get("GOM_DEBUG_ADDRS", "0") != "0":    # (development: which buffer does a faulting address belong to)
            for k, sl in enumerate(self.slots):
                for nm, t in [("fp.params.flat", sl["fp"].params.flat), ("fp.grads.flat", sl["fp"].grads.flat)] + [("step.grads." + a, b) for a, b in sl["step"].grads.items()]:
                    print(f"[gom torch pid {os.getpid()}] slot{k}.{nm} {t.data_ptr():#x} .. {t.data_ptr() + t.numel() * t.element_size():#x}", file=sys.stderr)
            for j, bt in enumerate(self.batches):
                for nm, t in bt.items():
                    if hasattr(t, "data_ptr") and t.is_cuda:
                        print(f"[gom torch pid {os.getpid()}] batch{j}.{nm} {t.data_ptr():#x} .. {t.data_ptr() + t.numel() * t.element_size():#x}", file=sys.stderr)
            for nm, t in wl.params.items():
                print(f"[gom torch pid {os.getpid()}] wl.params.{nm} {t.data_ptr():#x} .. {t.data_ptr() + t.numel() * t.element_size():#x}", file=sys.stderr)
        assert self.batches, "not enough frames for one batch"
        # (Measured and dropped: the whole step -- frame step + Adam -- captured by the caller as ONE torch.cuda.CUDAGraph, to close the ~9 us
        #  between the library's graph and the plain Adam launch behind it: 13.55 k instead of 13.8 k frames/s, 4.61 k instead of 4.83 k at
        #  B = 1 -- a torch graph replay brings two small kernels of its own.)
        self.profiling = False
        self.payload = int(self.slots[0]["fp"].grads.flat.numel())

    def run_step(self, i, collective=True):
        torch = self.torch
        bt = self.batches[i % len(self.batches)]
        sl = self.slots[i % self.S]
        with torch.cuda.stream(sl["stream"]):
            sl["step"].cam = bt["cam"]
            if self.B > 1:
                sl["step"].cams_dev = bt["cams_dev"]   # this batch's device camera array: resident like its targets and poses (one recorded graph per batch)
            sl["step"].forward_backward(sl["params"], bt, bt["gt_rgb"], bt["gt_mask"], bt["bg"], graph=self.graph)
            if collective and sl["fp"].peer is not None and sl["opt"] is not None:
                # the peer exchange carries the optimizer in its all-gather kernel (1 / B: mean over the frames of the batch as well)
                sl["fp"].peer.run_adam(sl["opt"], 1.0 / (self.world * self.B), zero1=sl["fp"].impl == "peer-zero1")
                return
            if collective:
                sl["fp"].all_reduce_grads()  # no-op at world size 1
            if sl["opt"] is not None and not self.attached:
                sl["opt"].step(1.0 / self.B)  # mean over the frames of the batch (the collective has averaged over the ranks)

    def region(self, steps, warmup, first=0, collective=True):
        """`warmup` untimed steps, then EXACTLY `steps` steps between barrier + synchronize on both sides; MAX over ranks."""
        torch = self.torch
        for i in range(warmup):
            self.run_step(first + i, collective)
        torch.cuda.synchronize()
        if self.world > 1:
            import torch.distributed as dist
            dist.barrier()
        t0 = time.perf_counter()
        for i in range(steps):
            self.run_step(first + warmup + i, collective)
        torch.cuda.synchronize()
        if self.world > 1:
            dist.barrier()
        el = time.perf_counter() - t0
        if self.world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=self.wl.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el

    def measure(self, steps, warmup, collective=True):
        """-> (total seconds, total steps, regions).  Regions shorter than MIN_TIMED_S are repeated (same brackets, no new warm-up)."""
        el = self.region(steps, warmup, collective=collective)
        total, n = el, 1
        if el < MIN_TIMED_S:
            reps = min(200, int(math.ceil(MIN_TIMED_S * 1.2 / max(el, 1e-6))) - 1)   # el is the MAX over ranks: every rank repeats equally often
            for r in range(reps):
                total += self.region(steps, 0, first=(r + 1) * steps, collective=collective)
                n += 1
        return total, steps * n, n

    def check(self):
        n_pairs, overflow = self.slots[0]["step"].state.poll()
        assert not overflow, "pair buffer overflow during the benchmark"
        if not os.environ.get("GOM_BENCH_TIMING_ONLY"):   # (development: knock-out builds of the library whose results are invalid by construction)
            assert all(self.torch.isfinite(g).all() for g in self.slots[0]["step"].grads.values()), "non-finite gradients"
        return n_pairs

    def kernel_profile(self, rounds):
        """Per-kernel HIP-event durations (ms) with this runner's steps in flight: the library brackets every launch with events
        on the launch stream (the step is then enqueued kernel by kernel instead of replayed from its graph)."""
        from gomavatar_amd import _lib
        torch = self.torch
        for sl in self.slots:
            sl["step"].state.set_option(_lib.OPT_PROFILE, 1)
        self.profiling = True
        acc, n, D = {}, 0, 0
        for r in range(rounds):
            if self.S == 1:
                torch.cuda.synchronize()
            for k in range(self.S):
                self.run_step(r * self.S + k)
            torch.cuda.synchronize()
            for sl in self.slots:
                for kname, v in sl["step"].state.kernel_times_ms().items():
                    acc[kname] = acc.get(kname, 0.0) + v
                D += sl["step"].state.poll()[0]
                n += 1
        for sl in self.slots:
            sl["step"].state.set_option(_lib.OPT_PROFILE, 0)
        self.profiling = False
        torch.cuda.synchronize()
        # (a kernel id that no launch carried this time -- e.g. the depth histogram, which the frame step builds inside k_preprocess -- reports
        #  -1 ms: it is left out)
        return {k: acc[k] / n for k in _lib.KERNEL_NAMES if acc.get(k, -1.0) >= 0.0}, D / n


def timeit(torch, fn, min_s=MIN_TIMED_S, warm=5, chunk=10, windows=1):
    """Calls per second over `windows` timed windows of >= min_s each (the median window: a launch-bound Python loop is exposed to host jitter)."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    rates = []
    for _ in range(windows):
        n, t0 = 0, time.perf_counter()
        while True:
            for _ in range(chunk):
                fn()
            torch.cuda.synchronize()
            n += chunk
            el = time.perf_counter() - t0
            if el >= min_s:
                rates.append(n / el)
                break
    return sorted(rates)[len(rates) // 2]


def extra_figures(torch, wl):
    """(i) raster only fwd+bwd, fused 4-channel and the reference's two 3-channel calls; (iii) full step =
    render path + LPIPS-VGG in the reference's precision (fp32 trunk through the library convolutions) + Adam.  One frame
    at a time (the reference's semantics), launched from Python through autograd."""
    from gomavatar_amd import rasterizer as R
    from gomavatar_amd.geometry import MeshTopology, posed_face_gaussians
    from gomavatar_amd.losses import compute_loss_l1
    from gomavatar_amd.lpips import LPIPS, lpips_loss
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    dev, img, F, N = wl.device, wl.img, wl.F, wl.N
    out = {}
    d = wl.frames[0]
    step = wl.step(1)
    step.set_camera(d["K"], d["E"])
    step.forward_backward(wl.params, d, d["gt_rgb"], d["gt_mask"], d["bg"])
    torch.cuda.synchronize()
    xyz, cov6, feat, op, cam = step.xyz.clone(), step.cov6.clone(), step.feat.clone(), step.opacity.clone(), step.cam
    wimg = torch.randn(4, img, img, device=dev)

    def raster4():
        a = [t.detach().requires_grad_() for t in (xyz, cov6, feat)]
        o, _ = R.rasterize(a[0], a[1], a[2], op, cam)
        (o * wimg).sum().backward()
    out["raster_only_fused4_b1_fps"] = round(timeit(torch, raster4), 1)
    rs = GaussianRasterizationSettings(image_height=img, image_width=img, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, bg=torch.zeros(4, device=dev),
                                       scale_modifier=1.0, viewmatrix=torch.tensor(list(cam.view), device=dev).view(4, 4),
                                       projmatrix=torch.tensor(list(cam.proj), device=dev).view(4, 4), sh_degree=0,
                                       campos=torch.zeros(3, device=dev), prefiltered=False, debug=False)
    rast = GaussianRasterizer(None)
    rast.raster_settings = rs
    feat6 = torch.cat([feat, feat[:, :2]], -1)

    def raster2x3():   # gaussian.py:77-94
        a = [t.detach().requires_grad_() for t in (xyz, cov6, feat6)]
        m2d = torch.zeros_like(a[0], requires_grad=True)
        outs = [rast(means3D=a[0], means2D=m2d, colors_precomp=a[2][:, i:i + 3], shs=None, opacities=op[:, None], scales=None, rotations=None,
                     cov3D_precomp=a[1])[0] for i in (0, 3)]
        (torch.cat(outs, 0)[:4] * wimg).sum().backward()
    out["raster_only_reference_2x3_b1_fps"] = round(timeit(torch, raster2x3), 1)

    topo = MeshTopology(wl.faces, N, device=dev)
    w25 = wl.w25.to(dev)
    lp = LPIPS(trunk_seed=0, trunk_dtype=torch.float32, device=dev)
    P = {k: v.clone().requires_grad_() for k, v in wl.params.items()}
    opt = torch.optim.Adam(list(P.values()), lr=1e-4)

    def full():   # train.py:313-339 without the mesh / shadow branch
        opt.zero_grad(set_to_none=True)
        x, c6, _ = posed_face_gaussians(P["vertices"], P["so3"], P["scale"], d["dst_Rs"], d["dst_Ts"], d["cnl_gtfms"], w25, topo, 1e-3)
        f4 = torch.cat([P["appearance"].T, torch.ones(F, 1, device=dev)], 1)
        o, _ = R.rasterize(x, c6, f4, op, cam)
        total, _ = compute_loss_l1(o, d["gt_rgb"], d["gt_mask"], d["bg"])
        rgb, mask = o[:3].permute(1, 2, 0), o[3]
        unpacked = rgb * mask[..., None] + d["bg"] * (1 - mask[..., None])
        (total + lpips_loss(lp, unpacked[None], d["gt_rgb"][None])).backward()
        opt.step()
    try:
        out["full_step_lpips_fp32_adam_b1_fps"] = round(timeit(torch, full, warm=3, chunk=5), 1)
    except Exception as e:  # report, do not hide
        out["full_step_lpips_fp32_adam_b1_fps"] = f"failed: {type(e).__name__}: {e}"
    del lp, opt, P
    # (iii) again, through the NATIVE path: forward half of the frame step -> LPIPS value + image gradient on the hand-written trunk
    # (RenderStep.lpips_hook -> gom_lpips_vgg_value_and_grad) chained through `unpack` into the image gradient -> backward half ->
    # gom_adam_flat.  bf16x3 = hi + lo bf16 planes, three MFMA passes: the reference's fp32 precision (tests/test_gpu_vgg_bf16.py:
    # <= 1e-5 on the value against the fp32 library convolutions); bf16 = one pass, 3 % on the value.
    from gomavatar_amd.lpips import LPIPSMatrixCore
    from gomavatar_amd.parallel import FlatAdam, FrameParallel, shapes_for_model
    for prec in ("bf16x3", "bf16"):
        for b_ in (1, 8):
            key = f"full_step_lpips_{prec}_native_adam_b{b_}_fps"
            try:
                st = wl.step(b_)
                bt = wl.batches(st)[0]
                fp = FrameParallel(shapes_for_model(N, F), dev)
                for name in ("vertices", "so3", "scale", "appearance"):
                    st.grads[name] = fp.grads[name]
                    fp.params[name].copy_(wl.params[name])
                optn = FlatAdam(fp, {"default": ADAM_LR})
                mc = LPIPSMatrixCore(trunk_seed=0, device=dev, precision=prec)
                hook = st.lpips_hook(mc, bt["gt_rgb"], bt["bg"], coeff=1.0)
                pv = dict(fp.params.items())
                torch.cuda.synchronize()   # (what was set up on the default stream is there before a non-blocking stream reads it)
                stream = torch.cuda.Stream(device=dev)

                def native():
                    with torch.cuda.stream(stream):
                        st.cam = bt["cam"]
                        if b_ > 1:
                            st.cams_dev = bt["cams_dev"]
                        st.forward_backward(pv, bt, bt["gt_rgb"], bt["gt_mask"], bt["bg"], graph=True, image_grad_hook=hook)
                        optn.step(1.0 / b_)
                out[key] = round(b_ * timeit(torch, native, warm=3, chunk=5), 1)
                assert torch.isfinite(fp.grads.flat).all() and torch.isfinite(st.lpips_value)
                del st, mc, fp, optn
            except Exception as e:  # report, do not hide
                out[key] = f"failed: {type(e).__name__}: {e}"
    # BASELINE configs[1] as the reference runs it: ONE iteration of train.py:309-349 through the drop-in `Model` (forward incl. the mesh normal /
    # silhouette branch and the shadow MLP, unpack, every loss term of exps/zju-mocap_377.yaml incl. LPIPS, backward, torch Adam over the
    # reference's parameter groups, update_lr), one frame per iteration, launched from Python like the reference's loop.
    try:
        from types import SimpleNamespace as NS
        from gomavatar_amd.model import Model
        from gomavatar_amd import train_util as tu
        cfg = NS(img_size=(img, img), canonical_geometry=NS(sigma=1e-3, radius_scale=1.0, deform_so3=True, deform_scale=True), appearance=NS(color_init=0.5),
                 normal_renderer=NS(sigma=1e-5, soft_mask=True), shadow_module=NS(name="basic", multires=6, mlp_width=128, mlp_depth=3, skips=(4,)),
                 lbs_weights=NS(refine=False))
        tcfg = NS(lr=NS(lbs_weights=0.0, appearance=0.0005, canonical_geometry=0.0005, canonical_geometry_xyz=0.0005, shadow=0.0005), lr_decay_steps=100000,
                  losses=NS(rgb=NS(coeff=1.0), mask=NS(coeff=5.0), lpips=NS(coeff=1.0), laplacian=NS(coeff_canonical=0.0, coeff_observation=10.0),
                            normal=NS(coeff_mask=1.0, kernel_size=7, coeff_consist=0.1), color_consist=NS(coeff=0.05)))
        for prec in ("bf16x3", "bf16"):
            model = Model(cfg, wl.body).train()
            mcl = LPIPSMatrixCore(trunk_seed=0, device=dev, precision=prec)
            optm = torch.optim.Adam(model.get_param_groups(tcfg), betas=(0.9, 0.999))
            frames = []
            for i in range(4):
                fr = {k_: torch.from_numpy(v_).to(dev) for k_, v_ in wl.frames_np[i].items()}
                fr["target_rgbs"], fr["target_masks"] = wl.frames[i]["gt_rgb"][None], wl.frames[i]["gt_mask"][None]
                frames.append(fr)
            it_ = [0]

            def train_it():
                tu.train_iteration(model, optm, frames[it_[0] % 4], tcfg, it_[0] + 1, lpips_func=mcl)
                it_[0] += 1
            out[f"model_train_iteration_lpips_{prec}_torch_adam_b1_ips"] = round(timeit(torch, train_it, warm=20, chunk=5, windows=3), 1)   # (warm: torch builds its foreach optimizer state and the library its LPIPS buffers in the first iterations)
            # the same iteration with the reference's optimizer as one native launch (gomavatar_amd.optim.GomAdam: a torch.optim.Optimizer over the
            # same param groups, update_lr and checkpoints unchanged) -- the default the drop-in train loop is meant to run with
            from gomavatar_amd.optim import GomAdam
            model = Model(cfg, wl.body).train()
            optm = GomAdam(model.get_param_groups(tcfg), betas=(0.9, 0.999))
            it_[0] = 0
            out[f"model_train_iteration_lpips_{prec}_b1_ips"] = round(timeit(torch, train_it, warm=10, chunk=5, windows=3), 1)
            if prec == "bf16x3":
                # The data sets' target masks are segmentation masks of {0, 1} (dataset/train.py:240-258); the synthetic targets above are RENDERED masks, soft
                # everywhere.  With {0, 1} targets |normal_mask - target|'s gradient is exactly 0 under the body (alpha rounds to 1.0f there) and the mesh
                # rasterizer's backward evaluates the outline's band alone (mesh_raster.hip: k_mesh_backward_entries) -- the same iteration on such targets:
                soft = [f_["target_masks"] for f_ in frames]
                for f_ in frames:
                    f_["target_masks"] = (f_["target_masks"] > 0.5).float()
                torch.cuda.synchronize()
                out["model_train_iteration_lpips_bf16x3_binary_target_masks_b1_ips"] = round(timeit(torch, train_it, warm=6, chunk=5, windows=3), 1)
                for f_, m_ in zip(frames, soft):
                    f_["target_masks"] = m_
            if prec == "bf16x3":   # the same iteration captured once in a HIP graph and replayed per frame (train_util.GraphedTrainStep; Adam capturable, lr frozen at capture)
                model_g = Model(cfg, wl.body).train()
                opt_g = GomAdam(model_g.get_param_groups(tcfg), betas=(0.9, 0.999), capturable=True)
                gstep = tu.GraphedTrainStep(model_g, opt_g, tcfg.losses, mcl)
                jt = [0]

                def train_graphed():
                    gstep(frames[jt[0] % 4], i_iter=1)
                    jt[0] += 1
                out["model_train_iteration_lpips_bf16x3_b1_graphed_ips"] = round(timeit(torch, train_graphed, warm=6, chunk=5, windows=3), 1)
                del model_g, opt_g, gstep
            del model, mcl, optm
    except Exception as e:  # report, do not hide
        out["model_train_iteration_b1_ips"] = f"failed: {type(e).__name__}: {e}"
    return out


VGG_LAYERS = ((3, 64, 1), (64, 64, 1), (64, 128, 2), (128, 128, 2), (128, 256, 4), (256, 256, 4), (256, 256, 4), (256, 512, 8), (512, 512, 8), (512, 512, 8),
              (512, 512, 16), (512, 512, 16), (512, 512, 16))   # (Cin, Cout, downscale) of the 13 3x3 convolutions (pretrained_networks.py:96-134)
MFMA_BF16_PEAK_TFLOPS = 2500.0   # dense bf16 (MI355X_MICROARCH.md); scripts/ubench/mfma_rate.hip reaches 2 031 with random operands on this part


def lpips_roofline(torch, wl):
    """The kernel family that owns the wall clock of cfg 2's iteration: the LPIPS-VGG trunk on the bf16 matrix cores (csrc/vgg_bf16.hip).  One
    training evaluation = trunk forward of the prediction and of the target + backward-data of the prediction = 3 walks over the 13
    convolutions; bf16x3 (the reference's fp32 precision) issues every product three times.  `us` = HIP events in this run around
    `LPIPSMatrixCore.value_and_grad` (both images as one batch, one stream: heads, pools, split-K epilogues and the first layer's im2col are
    INSIDE the bracket), `flops` = 2 * MACs * 3 walks * 3 passes, `frac` = flops / us / the dense bf16 peak."""
    from gomavatar_amd.lpips import LPIPSMatrixCore
    img = wl.img
    macs = sum(ci * co * 9 * (img // d) ** 2 for ci, co, d in VGG_LAYERS)
    out = {}
    for prec, passes in (("bf16x3", 3), ("bf16", 1)):
        mc = LPIPSMatrixCore(trunk_seed=0, device=wl.device, precision=prec)
        gt = wl.frames[0]["gt_rgb"][None].contiguous()
        pred = (gt * 0.9 + 0.05).contiguous()
        torch.cuda.synchronize()   # (what was set up on the default stream is there before a non-blocking stream reads it)
        stream = torch.cuda.Stream(device=wl.device)
        import torch as _t
        from gomavatar_amd import _lib as _l
        bufs = (_t.empty((5, 1, _l.GOM_LOSS_BLOCKS), dtype=_t.float32, device=wl.device), _t.empty((1, img, img, 3), dtype=_t.float32, device=wl.device))
        with torch.cuda.stream(stream):   # persistent buffers on a non-default stream: the call replays its recorded hipGraph (no host launch cost in the bracket)
            for _ in range(5):
                mc.value_and_grad(pred, gt, out=bufs)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 30
            e0.record()
            for _ in range(n):
                v, _ = mc.value_and_grad(pred, gt, out=bufs)
            e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        flops = 2.0 * macs * 3 * passes
        out[prec] = {"us": round(us, 1), "flops": int(flops), "achieved": round(flops / (us * 1e-6) / 1e12, 1), "frac": round(flops / (us * 1e-6) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4),
                     "fp32_grade_tflops": round(2.0 * macs * 3 / (us * 1e-6) / 1e12, 1), "value": round(float(v), 6)}
        del mc
    lp, src = read_profile_json("lpips_launches")
    r = out["bf16x3"]
    return {"kernel": "k_conv3x3_bf16_v2 family (LPIPS-VGG trunk, bf16x3 = fp32-grade)", "bound": "mfma", "achieved": r["achieved"], "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": r["frac"], "us": r["us"], "flops": r["flops"], "walks": "prediction forward + target forward + prediction backward-data, x 3 bf16 passes",
            "timing": "HIP events in this run around value_and_grad (heads, pools, epilogues, im2col inside the bracket)", "fp32_grade_tflops": r["fp32_grade_tflops"],
            "one_pass_bf16": out["bf16"], "launches": (lp or {}).get("launches_per_evaluation"), "conv_only": (lp or {}).get("conv_only"), "launches_source": src,
            "measured_peak_random_operands_tflops": 2031.0}


def render_only_modes(torch, wl):
    """The comparator for the paper's 43 FPS (eval.py:336-361): frames/s of RENDERING alone.  (i) the drop-in `Model` in
    eval mode under no_grad -- mesh branch + shadow MLP included, unpack on the background -- frame by frame, eager and as one replayed HIP graph
    with the two-stream overlap of the mesh branch and the splat rasterizer; (ii) the native forward half of the frame step (FK -> LBS -> face
    Gaussians -> splat forward -> loss values), one frame and 8 per launch sequence."""
    from gomavatar_amd.workload import zju_cfg, model_frames, build_model
    from gomavatar_amd.train_util import GraphedRender, unpack
    out = {}
    mcfg, _ = zju_cfg(wl.img)
    model = build_model(wl, mcfg, with_mlps=False).eval()
    with torch.no_grad():
        for k in ("so3", "scale", "appearance"):
            getattr(model, k).copy_(wl.params[k])
    frames = model_frames(wl, range(min(8, len(wl.frames))))
    it = [0]

    def eager():
        fr = frames[it[0] % len(frames)]; it[0] += 1
        with torch.no_grad():
            rgbs, masks, _ = model(fr["K"], fr["E"], fr["cnl_gtfms"], fr["dst_Rs"], fr["dst_Ts"])
            return unpack(rgbs, masks, fr["bgcolor"])
    try:
        out["render_only_eval_eager_b1_fps"] = round(timeit(torch, eager, warm=10, chunk=10, windows=3), 1)
        model.overlap_branches = True
        with torch.no_grad():
            model.shadow_capacity = int(1.3 * max(int((model(fr["K"], fr["E"], fr["cnl_gtfms"], fr["dst_Rs"], fr["dst_Ts"])[1] > 0).sum()) for fr in frames))
        render = GraphedRender(model)

        def graphed():
            fr = frames[it[0] % len(frames)]; it[0] += 1
            return render(fr)
        out["render_only_eval_b1_fps"] = round(timeit(torch, graphed, warm=10, chunk=10, windows=3), 1)
    except Exception as e:  # report, do not hide
        out["render_only_eval_b1_fps"] = f"failed: {type(e).__name__}: {e}"
    for b_ in (1, 8):
        try:
            st = wl.step(b_)
            bts = wl.batches(st)
            torch.cuda.synchronize()   # (what was set up on the default stream is there before a non-blocking stream reads it)
            stream = torch.cuda.Stream(device=wl.device)
            j = [0]

            def native():
                bt = bts[j[0] % len(bts)]; j[0] += 1
                with torch.cuda.stream(stream):
                    st.cam = bt["cam"]
                    if b_ > 1:
                        st.cams_dev = bt["cams_dev"]
                    st.forward_backward(wl.params, bt, bt["gt_rgb"], bt["gt_mask"], bt["bg"], backward=False, graph=True)
            out[f"render_only_native_forward_b{b_}_fps"] = round(b_ * timeit(torch, native, warm=10, chunk=20, windows=3), 1)
            del st
        except Exception as e:
            out[f"render_only_native_forward_b{b_}_fps"] = f"failed: {type(e).__name__}: {e}"
    return out


def model_parallel_modes(torch, wl, world, rank, impls, iters=40, warm=8):
    """BASELINE configs[3] in the reference's step shape: `Model` + compute_loss (every term, LPIPS bf16x3) + the reference's Adam, one frame per
    rank per step through parallel.ModelFrameParallel (flat parameter / gradient buffers, ONE exchange per step, next iteration's target trunk
    under it).  -> {impl: iterations/s of the whole job (= frames/s: world frames per step)}, `local_only` = the same iteration with GomAdam and no
    exchange (what N independent GPUs do), and the exchanged float count.  Same iteration counts on every rank; MAX over ranks."""
    import torch.distributed as dist
    from gomavatar_amd.workload import zju_cfg, model_frames, build_model
    from gomavatar_amd.parallel import ModelFrameParallel
    from gomavatar_amd.lpips import LPIPSMatrixCore
    from gomavatar_amd.optim import GomAdam
    from gomavatar_amd import train_util as tu
    mcfg, tcfg = zju_cfg(wl.img)
    frames = model_frames(wl, range(min(8, len(wl.frames))))
    mcl = LPIPSMatrixCore(trunk_seed=0, device=wl.device, precision="bf16x3")
    out = {"unit": "iterations/s x ranks = frames/s (whole job)", "what": "Model iteration (mesh branch, shadow MLP, all loss terms, LPIPS bf16x3, Adam + update_lr), one frame per rank per step"}

    def timed(fn):
        for i in range(warm):
            fn(i)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        for i in range(iters):
            fn(warm + i)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=wl.device if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return round(world * iters / el, 1)

    model = build_model(wl, mcfg)
    opt = GomAdam(model.get_param_groups(tcfg), betas=(0.9, 0.999))
    out["local_only_ips"] = timed(lambda i: tu.train_iteration(model, opt, frames[i % len(frames)], tcfg, i + 1, lpips_func=mcl))
    del model, opt
    for impl in impls:
        key = f"model_train_iteration_lpips_bf16x3_{impl.replace('-', '_')}_ips"
        note(f"  ModelFrameParallel impl={impl}")
        ok = torch.ones(1, dtype=torch.int32)
        mfp, err = None, None
        try:
            model = build_model(wl, mcfg)
            mfp = ModelFrameParallel(model, tcfg, impl=impl)
        except Exception as e:   # report, do not hide (e.g. IPC not permitted between these devices)
            err, _ = f"{type(e).__name__}: {e}", ok.zero_()
        if world > 1:
            okd = ok.to(wl.device) if dist.get_backend() == "nccl" else ok
            dist.all_reduce(okd, op=dist.ReduceOp.MIN)
            ok = okd.cpu()
        if int(ok.item()) != 1:
            out[key] = f"not available on some rank ({err})"
            continue
        try:
            out[key] = timed(lambda i: tu.train_iteration(model, None, frames[i % len(frames)], tcfg, i + 1, lpips_func=mcl, frame_parallel=mfp))
            mfp.finish()
            if mfp.fp.peer is not None:
                mfp.fp.peer.check()
            out["param_floats"], out["allreduce_floats"] = mfp.param_floats, mfp.payload_floats
            mfp.overlap = False
            out[key.replace("_ips", "_no_overlap_ips")] = timed(lambda i: tu.train_iteration(model, None, frames[i % len(frames)], tcfg, i + 1, lpips_func=mcl, frame_parallel=mfp))
            mfp.finish()
        except Exception as e:
            out[key] = f"failed: {type(e).__name__}: {e}"
        mfp.close()
        del model, mfp
    return out


def cpu_baseline(torch, args, wl_M, batched_step, batched_batch):
    """The oracle (`kind: port`) on this box's host cores, SURVEY.md 8(d)'s protocol: the render path (geometry + raster + L1 losses)
    forward alone and forward + backward, 1 thread and all physical cores, at S (BASELINE configs[0]: 13 776 Gaussians) and M (the
    metric workload); warm-up frames first, then the MEDIAN of `--cpu-frames` timed frames.  Bounded sample (about 20-30 s)."""
    from oracle import geometry as og, raster as orast
    from gomavatar_amd.workload import MetricWorkload
    import statistics
    phys, logical = physical_cores(), os.cpu_count() or 1
    rows, keep = {}, {}
    n_timed = max(1, args.cpu_frames)
    wl_S = MetricWorkload(wl_M.device, subdiv=0, img=wl_M.img, n_frames=8)
    for tag, wl in (("S", wl_S), ("M", wl_M)):
        n_warm = 3 if tag == "M" else 1
        # thread settings: (raster OpenMP threads, torch threads).  All physical cores for both is NOT the fastest on a 128-core host (torch's
        # pools on the small geometry tensors): 16 / 16 and "raster on all cores, torch on 16" are measured too and the best row is the baseline
        mids = sorted({min(phys, 16)} - {1, phys})
        for k, kt in [(1, 1)] + [(m, m) for m in mids] + ([(phys, m) for m in mids] if phys > 16 else []) + [(phys, phys)]:
            torch.set_num_threads(kt)
            orast.set_threads(k)
            t_fwd, t_all = [], []
            for j in range(n_warm + n_timed):
                i = j % len(wl.frames)
                fr = wl.oracle_frame(i)
                gt = wl.frames[i]
                gt_rgb, gt_mask = gt["gt_rgb"].cpu()[None], gt["gt_mask"].cpu()[None]
                with torch.no_grad():                                          # forward alone
                    t1 = time.perf_counter()
                    o_rgb, o_mask, _ = og.render_path(wl.params_cpu, fr, wl.faces, wl.w25, wl.img)
                    og.l1_losses(og.unpack(o_rgb, o_mask, fr["bgcolor"]), o_mask, gt_rgb, gt_mask)
                    tf = time.perf_counter() - t1
                po = {kk: v.clone().requires_grad_() for kk, v in wl.params_cpu.items()}
                t1 = time.perf_counter()                                       # forward + backward
                o_rgb, o_mask, _ = og.render_path(po, fr, wl.faces, wl.w25, wl.img)
                l1, l2 = og.l1_losses(og.unpack(o_rgb, o_mask, fr["bgcolor"]), o_mask, gt_rgb, gt_mask)
                (l1 + 5.0 * l2).backward()
                ta = time.perf_counter() - t1
                if j >= n_warm:
                    t_fwd.append(tf); t_all.append(ta)
                if tag == "M" and k == phys and kt == phys:
                    keep[i] = (o_rgb[0].detach(), o_mask[0].detach())
            rows[f"{tag}_threads{k}" if k == kt else f"{tag}_threads{k}_torch{kt}"] = {"frames_per_s": round(1.0 / statistics.median(t_all), 3), "fwd_only_frames_per_s": round(1.0 / statistics.median(t_fwd), 3),
                                         "median_ms_fwd_bwd": round(1e3 * statistics.median(t_all), 2), "median_ms_fwd": round(1e3 * statistics.median(t_fwd), 2),
                                         "frames": len(t_all), "warmup_frames": n_warm, "threads": k, "torch_threads": kt, "gaussians": wl.F}
    # "PSNR vs ref": the BATCHED step's own image (what the timed loop renders) against the oracle's render of the same frames
    mse, n = 0.0, 0
    img = batched_step.image.reshape(batched_step.B, 4, wl_M.img, wl_M.img)
    for pos, fi in enumerate(batched_batch["frames"]):
        if fi in keep:
            o_rgb, o_mask = keep[fi]
            h = img[pos].permute(1, 2, 0).cpu()
            dd = torch.cat([h[..., :3] - o_rgb, (h[..., 3] - o_mask)[..., None]], -1).double()
            mse += float((dd ** 2).mean()); n += 1
    head = max((r for kk, r in rows.items() if kk.startswith("M_")), key=lambda r: r["frames_per_s"])
    cb = {"value": head["frames_per_s"], "unit": "frames/s", "cores": head["threads"], "kind": "port",
          "sample": f"median of {head['frames']} frames (after {head['warmup_frames']} warm-up frames) of the metric workload, fwd+bwd: geometry + raster + L1 losses, through the CPU oracle at "
                    f"{head['threads']} OpenMP thread(s) (raster) / {head['torch_threads']} torch thread(s) (geometry), the fastest of the thread settings in `rows` (1, 16, all {phys} physical cores, "
                    f"and raster on all cores with torch on 16); host has {logical} logical CPUs.  `rows`: every setting, "
                    "forward alone and forward + backward, at M (the metric workload) and S (BASELINE configs[0])",
          "rows": rows, "physical_cores": phys, "logical_cpus": logical}
    return cb, (round(-10.0 * math.log10(max(mse / n, 1e-30)), 2) if n else None)


def other_configs(torch, dev, args):
    """The BASELINE configs the headline is not quoted on, each on its own synthetic workload: cfg 3 (55 104 Gaussians at 1024^2, and at
    540^2 -- the size exps/snapshot_f3c.yaml really uses) and cfg 5 (220 416 Gaussians at 1024^2, "HBM-bound stress").  Frames/s one frame
    at a time and 8 per launch sequence (one step in flight, Adam inside), and the roofline fraction of the kernel that dominates there."""
    from gomavatar_amd.workload import MetricWorkload
    out = {}
    for name, subdiv, img in (("cfg3_M_1024", 1, 1024), ("cfg3_M_540", 1, 540), ("cfg5_L_1024", 2, 1024)):
        try:
            wl = MetricWorkload(dev, subdiv=subdiv, img=img, n_frames=8)
            row = {"gaussians": wl.F, "image": [img, img]}
            for b in (1, 8):
                r = Runner(wl, b, 1, not args.no_graph, 1, args)
                el, ns, _ = r.measure(40 if b > 1 else 150, 8)
                r.check()
                iso, D = r.kernel_profile(6)
                ab = algorithmic_bytes(wl.F * b, D, img * img * b, 4)
                dom = max(iso, key=iso.get)

                def frac(k):
                    return round(ab[k] / (iso[k] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if iso.get(k, 0) > 0 and ab[k] else 0.0
                row[f"b{b}"] = {"fps": round(b * ns / el, 1), "ms_per_step": round(1e3 * el / ns, 4), "pairs_D": int(D),
                                "dominant": {"kernel": "k_" + dom, "avg_us": round(iso[dom] * 1e3, 2), "algorithmic_bytes": int(ab[dom]), "frac": frac(dom)},
                                "raster_backward": {"kernel": "k_seg_bwd", "avg_us": round(iso["seg_bwd"] * 1e3, 2), "algorithmic_bytes": int(ab["seg_bwd"]), "frac": frac("seg_bwd")}}
                del r
            # the same 8 frames per step as two concurrent 4-frame launch sequences (what the headline times on the metric workload)
            r = Runner(wl, 8, 1, not args.no_graph, 1, args, split=2)
            el, ns, _ = r.measure(40, 8)
            r.check()
            row["b8_split2"] = {"fps": round(8 * ns / el, 1), "ms_per_step": round(1e3 * el / ns, 4)}
            del r
            out[name] = row
            del wl
            torch.cuda.empty_cache()
        except Exception as e:   # report, do not hide
            out[name] = f"failed: {type(e).__name__}: {e}"
    return out


def read_profile_json(name):
    """A PMC summary committed under profiles/ (collected by scripts/collect_profiles.sh in its OWN rocprofv3 passes, not in this run):
    -> (dict, stamp) with the file's path and sha256 so that the bench line says where `traffic` / `valu` come from."""
    import hashlib
    rel = os.path.join("profiles", f"{PROFILE_TAG}_{name}.json")
    try:
        raw = open(os.path.join(ROOT, rel), "rb").read()
        return json.loads(raw), {"file": rel, "sha256_16": hashlib.sha256(raw).hexdigest()[:16], "measured_in_this_run": False}
    except Exception:
        return None, None


def main():
    args = parse()
    self_launch(args)
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    n_dev = torch.cuda.device_count()
    assert n_dev >= 1, "no HIP device"
    shared = world > n_dev                      # fewer devices than ranks: functional proof of the N > 1 path on a small lease
    dev_idx = local_rank % n_dev
    torch.cuda.set_device(dev_idx)
    backend = None
    if world > 1:
        import torch.distributed as dist
        backend = args.backend if args.backend != "auto" else ("gloo" if shared else "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_idx))
        else:
            dist.init_process_group(backend)
    dev = torch.device("cuda", dev_idx)

    from __graft_entry__ import build
    from gomavatar_amd import build as hip_build
    if rank == 0 and hip_build.needs_build():
        build()
    if world > 1:
        dist.barrier()
    from gomavatar_amd import _lib
    from gomavatar_amd.workload import MetricWorkload

    if args.batch <= 0:
        # WEAK scaling as the contract defines it: the per-GPU work is the SAME at every N -- 8 frames per GPU per step, the metric's operating point
        # (until round 5 N > 1 defaulted to configs[3]'s literal one frame per GPU: a different per-GPU job than N = 1, so value(N) / (N value(1)) said
        # nothing about the exchange).  configs[3]'s literal operating point is `modes.b1_per_gpu` (and `--batch 1`).
        args.batch = 8
    img, B, S = args.img, max(1, args.batch), max(1, args.inflight)
    wl = MetricWorkload(dev, subdiv=args.subdiv, img=img, n_frames=max(args.frames, B), rank=rank)
    F, N = wl.F, wl.N
    # the step's B frames as `split` concurrent launch sequences (same bits: tests/test_gpu_batch.py); 2 x 4 is the fastest cut of 8 frames on MI355X
    split = args.split if args.split > 0 else (2 if (B == 8 and S == 1 and not args.attach_adam) else 1)
    main_run = Runner(wl, B, S, not args.no_graph, world, args, split=split)
    split = main_run.split

    # ---------------- the timed region(s) ----------------
    # (in front of the W warm-up steps of the contract: ~0.3 s of the same steps, untimed -- a fresh box's first launches pay for code-object
    #  loading, graph capture and the clocks' ramp, and W = 5 steps of 0.6 ms do not cover that)
    t_pre, i_pre = time.perf_counter(), 0
    soak = int(os.environ.get("GOM_BENCH_SOAK_PRERUN", "0"))   # (scripts/soak_two_ranks.sh: this many times the pre-run's steps, then exit -- the hunt for LABBOOK R5.8's fault)
    while (time.perf_counter() - t_pre < 0.3) if world == 1 and not soak else (i_pre < 256 * max(1, soak)):   # (N > 1: the same number of collectives on every rank)
        for _ in range(8):
            main_run.run_step(i_pre, collective=not os.environ.get("GOM_BENCH_SOAK_LOCAL")); i_pre += 1
        torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    if soak:
        main_run.check()
        note(f"soak: {i_pre} steps without a fault")
        if world > 1:
            dist.destroy_process_group()
        return
    note("timed region")
    elapsed, n_steps, regions = main_run.measure(args.steps, args.warmup)
    main_run.check()
    value = world * B * n_steps / elapsed

    # ---------------- the collective alone, and the same loop without it (N > 1) ----------------
    allreduce_us = None
    local_only_fps = None
    if world > 1:
        el_l, ns_l, _ = main_run.measure(args.steps, min(args.warmup, 10), collective=False)   # (MAX over ranks, like `value`)
        local_only_fps = round(world * B * ns_l / el_l, 1)
        fp = main_run.slots[0]["fp"]
        for _ in range(5):
            fp.all_reduce_grads()
        torch.cuda.synchronize(); dist.barrier()
        t0 = time.perf_counter()
        for _ in range(50):
            fp.all_reduce_grads()
        torch.cuda.synchronize()
        allreduce_us = round((time.perf_counter() - t0) / 50 * 1e6, 1)

    # ---------------- N > 1: the same job over the direct peer-pointer all-reduce (csrc/frame_parallel.hip), when it comes up ----------------
    peer_info, use_peer, collective_fps = None, False, None
    peer_ok = False
    if world > 1:
        note("peer exchange: probe in child processes")
        peer_ok, probe_err = peer_probe(world, rank, dev_idx)
        note(f"peer exchange: probe {'ok' if peer_ok else 'FAILED: ' + str(probe_err)}")
        peer_info = {"impl": "two-shot reduce-scatter / all-gather over hipIpc-mapped peer buffers, rank-order sum, Adam inside the all-gather kernel (gom_peer_reduce_run_adam)",
                     "probe": "ok" if peer_ok else probe_err}
        ok_t = torch.ones(1, dtype=torch.int32, device=dev if backend == "nccl" else "cpu")
        peer_run, err = None, None
        try:
            if not peer_ok:
                raise RuntimeError("the peer exchange failed its probe in child processes: " + str(probe_err))
            peer_run = Runner(wl, B, 1, not args.no_graph, world, args, impl="peer", split=split)
        except Exception as e:   # report, do not hide (e.g. IPC not permitted between these devices)
            err = f"{type(e).__name__}: {e}"
            ok_t.zero_()
        dist.all_reduce(ok_t, op=dist.ReduceOp.MIN)   # every rank takes the same branch
        if int(ok_t.item()) == 1:
            try:
                el_p, ns_p, _ = peer_run.measure(max(20, args.steps // 2), 10)
                fp_p = peer_run.slots[0]["fp"]
                torch.cuda.synchronize(); dist.barrier()
                t0 = time.perf_counter()
                for _ in range(50):
                    fp_p.all_reduce_grads()
                torch.cuda.synchronize()
                peer_info.update(us=round((time.perf_counter() - t0) / 50 * 1e6, 1), fps=round(world * B * ns_p / el_p, 1), frames_per_gpu_per_step=B)
                fp_p.peer.check()
                peer_info["status"] = "ok"
                # the same exchange with the optimizer sharded over the ranks (ZeRO-1: gom_peer_reduce_run_zero1; bitwise the same replicas)
                for sl in peer_run.slots:
                    sl["fp"].close()
                peer_run = None
                z_run = Runner(wl, B, 1, not args.no_graph, world, args, impl="peer-zero1", split=split)
                el_z, ns_z, _ = z_run.measure(max(20, args.steps // 2), 10)
                z_run.slots[0]["fp"].peer.check()
                peer_info["zero1_fps"] = round(world * B * ns_z / el_z, 1)
                if peer_info["zero1_fps"] > peer_info["fps"]:
                    peer_info["impl"] = "two-shot reduce-scatter / all-gather over hipIpc-mapped peer buffers, rank-order sum, ZeRO-1: Adam of the own slice inside the reduce-scatter kernel, parameters gathered (gom_peer_reduce_run_zero1)"
                    peer_info["replicated_adam_fps"], peer_info["fps"], el_p, ns_p = peer_info["fps"], peer_info["zero1_fps"], el_z, ns_z
                for sl in z_run.slots:
                    sl["fp"].close()
            except Exception as e:
                peer_info["status"] = f"failed: {type(e).__name__}: {e}"
        else:
            peer_info["status"] = f"not available on some rank ({err})" if err else "not available on some rank"
        dist.barrier()
        # the headline of an N > 1 run is the faster of the two exchanges of the SAME job (both reported, the choice named in config.allreduce_impl)
        collective_fps = value
        if peer_info.get("status") == "ok" and peer_info["fps"] > value:
            value, elapsed, n_steps = peer_info["fps"], el_p, ns_p
            use_peer = True

    # ---------------- per-kernel times, one step in flight (the kernels own the chip) ----------------
    note("per-kernel profile")
    # (the roofline describes a launch that OWNS the chip: one sequence of B frames, one step in flight -- as in every earlier round)
    alone_run = main_run if (S == 1 and split == 1) else Runner(wl, B, 1, not args.no_graph, world, args)
    iso, D_avg = alone_run.kernel_profile(12)
    abytes = algorithmic_bytes(F * B, D_avg, img * img * B, 4)   # per launch: B frames (D_avg already counts all B)
    dom = max(iso, key=iso.get)                                   # dominant kernel = the one that costs most when it owns the chip
    traffic_j, traffic_src = read_profile_json("traffic") if (args.subdiv == 1 and img == 512) else (None, None)
    valu_j, valu_src = read_profile_json("valu") if (args.subdiv == 1 and img == 512) else (None, None)
    # a summary collected for another batch size, or before a kernel of today's step existed, says nothing about this run: dropped
    if traffic_j and (traffic_j.get("batch") != B or ("k_" + dom) not in traffic_j or "k_seg_bwd" not in traffic_j):
        traffic_j, traffic_src = None, None
    if valu_j and (valu_j.get("batch", B) != B or ("k_" + dom) not in valu_j):
        valu_j, valu_src = None, None

    kavg_j, kavg_src = read_profile_json("kernel_avg") if (args.subdiv == 1 and img == 512) else (None, None)
    if kavg_j and kavg_j.get("batch") != B:
        kavg_j, kavg_src = None, None

    def kernel_row(name, us):
        gbs = abytes[name] / (us * 1e-6) / 1e9 if us > 0 and abytes[name] else 0.0
        tr = None
        if traffic_j and traffic_j.get("batch") == B and ("k_" + name) in traffic_j:
            tr = int(traffic_j["k_" + name].get("hbm_bytes") or 0) or None
        return {"kernel": "k_" + name, "avg_us": round(us, 2), "algorithmic_bytes": int(abytes[name]), "achieved": round(gbs, 2),
                "frac": round(gbs / HBM_PEAK_GBS, 5), "traffic": tr}
    dr = kernel_row(dom, iso[dom] * 1e3)
    # The bound that HOLDS for this kernel is VALU instruction issue, not HBM (round-5 review): `achieved / peak / frac` stay
    # SURVEY.md 8(d)'s HBM figures (algorithmic bytes / launch duration / 8 TB/s: comparable across rounds and with north_star's 60 % target, which
    # this design does not reach), `issue` says how far the launch is from the bound it really runs against.
    vrow = (valu_j or {}).get("k_" + dom) or {}
    n_valu = vrow.get("valu_insts_per_launch")
    issue = None
    if n_valu:
        floor_us = n_valu * ISSUE_TICKS_PER_VALU / (N_SIMD * SCLK_HZ) * 1e6
        issue = {"valu_wave_insts": int(n_valu), "valu_wave_insts_source": valu_src,
                 "issue_floor_us": round(floor_us, 1), "frac_of_issue_floor": round(floor_us / dr["avg_us"], 4) if dr["avg_us"] else None,
                 "how": f"wave-level VALU instructions per launch (SQ_INSTS_VALU of the committed PMC pass: static for a given pair count D) x {ISSUE_TICKS_PER_VALU} issue "
                        f"ticks per instruction (scripts/ubench/valu_rate.hip at 4 waves per SIMD, this kernel's mix of plain / packed / DPP / transcendental instructions) "
                        f"/ ({N_SIMD} SIMDs x {SCLK_HZ / 1e9:.1f} GHz); lanes alive per evaluated entry ~13 %: the instruction count, not the byte count, is what a better design must cut"}
    roofline = {"kernel": dr["kernel"], "bound": "valu_issue", "bound_of_the_figures_below": "hbm (SURVEY.md 8(d) algorithmic bytes over 8 TB/s)", "issue": issue,
                "achieved": dr["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": dr["frac"],
                "traffic": dr["traffic"], "avg_us": dr["avg_us"], "algorithmic_bytes": dr["algorithmic_bytes"], "pairs_D": int(D_avg),
                "steps_in_flight": 1, "all_kernels_us": {k: round(v * 1e3, 2) for k, v in iso.items()},
                "all_kernels_note": "HIP-event brackets around every launch of an UN-GRAPHED pass (the library enqueues the step kernel by kernel for this): each reads "
                                    "~6-8 % long against the graph-replayed launches of the timed loop, so their sum exceeds ms_per_step; avg_us_rocprof = the rocprofv3 "
                                    "kernel-trace average of the graph-replayed launches (committed profile, stamped)",
                "avg_us_rocprof": (kavg_j or {}).get(dr["kernel"]), "avg_us_rocprof_source": kavg_src,
                "raster_backward": kernel_row("seg_bwd", iso["seg_bwd"] * 1e3) | {"with_preprocess_bwd_frac": round(
                    (abytes["seg_bwd"] + abytes["preprocess_bwd"]) / ((iso["seg_bwd"] + iso["preprocess_bwd"]) * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)},
                # These kernels are VALU-issue bound, not bandwidth bound: the second axis, from the SQ counters
                # of the PMC pass (SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES and useful lanes), when profiles/ holds it
                "valu": (valu_j or {}).get("k_" + dom), "traffic_source": traffic_src, "valu_source": valu_src}

    if split > 1:
        # the launches of the TIMED configuration (B / split frames each), bracketed one branch after the other (GOM_OPT_PROFILE serialises the
        # branches: every launch owns the chip): sums over the step's `split` launches per kernel
        iso_s, D_s = main_run.kernel_profile(12)
        ab_s = algorithmic_bytes(F * B, D_s, img * img * B, 4)
        roofline["timed_configuration"] = {
            "what": f"the timed step runs {split} concurrent launch sequences of {B // split} frames each (gom_split_forward_backward: fork / join inside one hipGraph, one frame "
                    f"sum over all {B} frames; gradients bitwise those of one {B}-frame sequence).  The figures above are for the one-sequence launch of {B} frames, which owns "
                    "the chip (comparable with earlier rounds, and with profiles collected with --split 1); below: the timed configuration's own launches, bracketed one "
                    "branch after the other (alone on the chip), per launch; in the timed loop they overlap, which is the point",
            "split": split, "frames_per_launch": B // split,
            "kernel": "k_" + dom, "avg_us_per_launch_alone": round(iso_s[dom] * 1e3 / split, 2), "algorithmic_bytes_per_launch": int(ab_s[dom] / split),
            "frac_alone": round(ab_s[dom] / (iso_s[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
            "all_kernels_us_per_step_alone": {k: round(v * 1e3, 2) for k, v in iso_s.items()},
            "sum_of_kernels_us_alone": round(sum(iso_s.values()) * 1e3, 1), "ms_per_step_timed": round(1e3 * elapsed / n_steps, 4)}

    out = {
        "metric": "rendered frames/sec (fwd+bwd) at 512x512, ~50k Gaussians; PSNR vs ref",
        "value": round(value, 2),
        "unit": "frames/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / n_steps, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "timed_regions": regions,
        "config": {"workload": f"GoMAvatar hot path fwd+bwd, {F} Gaussians / {N} verts, {img}x{img}, {B} frames per GPU per step "
                               + ("(one batched launch sequence)" if split == 1 else f"({split} concurrent launch sequences of {B // split} frames, forked and joined inside the step's one hipGraph)")
                               + f", {S} step(s) in flight per GPU"
                               + (", + all-reduce of the flat grad buffer" if world > 1 else "") + ("" if args.no_adam else ", + Adam"),
                   "gaussians": F, "image": [img, img], "frames_per_step": world * B, "frames_per_gpu_per_step": B,
                   "parallelism": f"frame-dp{world}", "steps_in_flight_per_gpu": S, "launch_sequences_per_step": split,
                   "optimizer": (f"Adam on the flat parameter buffer inside the timed loop (gom_adam_flat, lr {ADAM_LR:g}: the reference's arithmetic, a rate that "
                                 "keeps the synthetic workload fixed)" + ("; the last launch of the frame step's recorded graph" if main_run.attached else "")
                                 if not args.no_adam else None),
                   "allreduce_floats": main_run.payload if world > 1 else 0, "allreduce_us": allreduce_us, "allreduce_impl": (None if world == 1 else peer_info["impl"] if use_peer else
                                      "torch.distributed all_reduce (" + ("RCCL, ReduceOp.AVG" if backend == "nccl" else backend + " through pinned host memory") + ")"),
                   "allreduce_collective_fps": round(collective_fps, 1) if collective_fps is not None else None,
                   "local_only_fps": local_only_fps, "allreduce_peer": peer_info,
                   "note": (None if world == 1 else
                            f"weak scaling: {B} frame(s) per GPU per step at every N (the N = 1 line's per-GPU job), one exchange of the mean gradient + Adam per step; "
                            "`local_only_fps` = the same loop without the exchange (N independent GPUs), `modes.b1_per_gpu` = BASELINE configs[3] as written (one frame "
                            "per GPU per step)"),
                   "backend": (("rccl" if backend == "nccl" else backend) + (f" ({world} ranks share {n_dev} device(s): functional proof, not a scaling number)" if shared else ""))
                   if world > 1 else None},
        "roofline": roofline,
    }

    # The headline's runners go before the other operating points are measured: with them alive (three states, their recorded graphs, the library's side streams)
    # the eager `Model` iteration below read 367 it/s against 394 without (same box, A/B; mechanism not pinned down: LABBOOK R6.5).  The closing PSNR check
    # builds a fresh step.
    import gc
    main_run.slots, alone_run = [], None
    gc.collect(); torch.cuda.empty_cache()

    # ---------------- N > 1: BASELINE configs[3]'s literal operating point -- ONE frame per GPU per step, the exchange + Adam behind every frame ----------------
    if world > 1 and not args.no_modes and B != 1:
        r1 = Runner(wl, 1, 1, not args.no_graph, world, args)
        el1, ns1, _ = r1.measure(max(40, args.steps // 2), 10)
        el1l, ns1l, _ = r1.measure(max(40, args.steps // 2), 5, collective=False)
        out["modes"] = {"unit": "frames/s", "what": "whole job, b = frames per GPU per step; local_only = the same loop without the exchange (N independent GPUs)",
                        f"b{B}_per_gpu": round(value, 1), f"b{B}_per_gpu_local_only": local_only_fps,
                        "b1_per_gpu": round(world * ns1 / el1, 1), "b1_per_gpu_local_only": round(world * ns1l / el1l, 1),
                        "b1_per_gpu_is": "BASELINE configs[3] as written: batch = N frames, one per GPU, gradient all-reduce + Adam after every frame"}
        del r1
    # ---------------- N > 1: configs[3] in the reference's own step shape (Model + LPIPS + Adam, one frame per rank, ONE exchange) ----------------
    if world > 1 and not args.no_modes:
        note("Model frame-parallel modes")
        torch.cuda.empty_cache()
        mp_ = model_parallel_modes(torch, wl, world, rank, ("collective", "peer", "peer-zero1") if peer_ok else ("collective",))
        out.setdefault("modes", {})["model_parallel"] = mp_
        out["config"]["model_allreduce_floats"] = mp_.get("allreduce_floats")
        out["config"]["model_param_floats"] = mp_.get("param_floats")

    # ---------------- the other operating points (N = 1) ----------------
    if world == 1 and not args.no_modes:
        modes = {f"b{B}_inflight{S}" + (f"_split{split}" if split > 1 else ""): round(value, 1)}
        for (b_, s_) in ((1, 1), (8, 1), (8, 3)):
            key = f"b{b_}_inflight{s_}"      # (one launch sequence per step: the operating points of rounds 2-5)
            if key in modes:
                continue
            r_ = Runner(wl, b_, s_, not args.no_graph, 1, args)
            el, ns, _ = r_.measure(100 if b_ > 1 else 400, 20)
            r_.check()
            modes[key] = round(b_ * ns / el, 1)
            if (b_, s_) == (8, 3):   # the kernel that dominates when three steps share the chip (k_sort's long lists stretch most)
                mix, _ = r_.kernel_profile(max(4, 24 // s_))
                dm = max(mix, key=mix.get)
                roofline["dominant_in_mix"] = kernel_row(dm, mix[dm] * 1e3) | {"steps_in_flight": 3, "all_kernels_us": {k: round(v * 1e3, 2) for k, v in mix.items()}}
            del r_
        out["modes"] = {"unit": "frames/s", "what": "render path fwd+bwd (b = frames per launch sequence, inflight = independent steps on separate streams)", **modes}
        note("extra figures")
        out["modes"].update(extra_figures(torch, wl))
        torch.cuda.empty_cache()
        note("render-only / flat Model iteration / LPIPS roofline")
        out["modes"].update(render_only_modes(torch, wl))
        # the Model iteration over the FLAT buffers (parallel.ModelFrameParallel at world size 1: what every rank of configs[3] runs, without a peer):
        # parameters and .grad seated on one buffer each, the reference's Adam as ONE segment launch, both optional MLPs held (951 023 parameters)
        try:
            mp1 = model_parallel_modes(torch, wl, 1, 0, ("collective",), iters=60, warm=10)
            out["modes"]["model_train_iteration_lpips_bf16x3_flat_b1_ips"] = mp1.get("model_train_iteration_lpips_bf16x3_collective_ips")
            out["modes"]["model_train_iteration_lpips_bf16x3_with_mlps_gomadam_b1_ips"] = mp1.get("local_only_ips")
            out["config"]["model_param_floats"] = mp1.get("param_floats")
        except Exception as e:  # report, do not hide
            out["modes"]["model_train_iteration_lpips_bf16x3_flat_b1_ips"] = f"failed: {type(e).__name__}: {e}"
        try:
            out["roofline_lpips"] = lpips_roofline(torch, wl)
        except Exception as e:
            out["roofline_lpips"] = f"failed: {type(e).__name__}: {e}"

    # ---------------- BASELINE configs[1] at its stated shape: S -> subdivide at 1 000 -> M, 3 000 iterations, every loss term + LPIPS bf16x3 + GomAdam ----------------
    if world == 1 and not args.no_modes:
        curve, src = None, None
        if args.train_curve:
            note("training curve (scripts/train_synthetic.py)")
            tmp = os.path.join(ROOT, "gpurun_out", "bench_train_curve.json")
            os.makedirs(os.path.dirname(tmp), exist_ok=True)
            r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "train_synthetic.py"), "--iters", "3000", "--subdivide-at", "1000", "--out", tmp],
                               stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
            if r.returncode == 0:
                curve, src = json.load(open(tmp)), {"measured_in_this_run": True}
            else:
                out["modes"]["cfg2_train_curve"] = "failed: " + r.stdout[-300:]
        else:
            curve, src = read_profile_json("train_curve")
        if curve:
            orc, osrc = read_profile_json("train_curve_oracle")
            row = {k: curve.get(k) for k in ("iterations", "subdivide_at", "wall_seconds_including_logging", "iterations_per_s", "peak_memory_MB", "final_psnr_mean_8_views")}
            row["psnr_at"] = {str(r_["iter"]): r_["psnr"] for r_ in curve["log"] if r_["iter"] in (1, 50, 100, 150, 500, 1000, 2000, 3000)}
            if orc:   # the CPU oracle trained from the same initialisation (scripts/train_curve_oracle.py): PSNR of the same iterations, side by side
                hip = {r_["iter"]: r_["psnr"] for r_ in curve["log"]}
                both = [(r_["iter"], hip[r_["iter"]], r_["psnr"]) for r_ in orc["log"] if r_["iter"] in hip]
                row["vs_oracle_trained"] = {"iterations_compared": len(both), "max_abs_psnr_difference_db": round(max(abs(a - b) for _, a, b in both), 3) if both else None,
                                            "mean_abs_psnr_difference_db": round(sum(abs(a - b) for _, a, b in both) / max(len(both), 1), 4), "source": osrc}
                # the float64 oracle against ITSELF: an earlier run of the same script (thread scheduling = summation order differs) -- the chaos floor the comparison above sits on
                orc0, osrc0 = read_profile_json("train_curve_oracle_first_run")
                if orc0:
                    o1 = {r_["iter"]: r_["psnr"] for r_ in orc["log"]}
                    two = [(r_["iter"], r_["psnr"], o1[r_["iter"]]) for r_ in orc0["log"] if r_["iter"] in o1]
                    if two:
                        row["vs_oracle_trained"]["oracle_against_its_own_earlier_run"] = {"iterations_compared": len(two), "max_abs_psnr_difference_db": round(max(abs(a - b) for _, a, b in two), 3),
                                                                                          "mean_abs_psnr_difference_db": round(sum(abs(a - b) for _, a, b in two) / len(two), 4), "source": osrc0}
            # ... and the oracle CONTINUING this run from its saved state across the subdivision (round 6: scripts/train_curve_oracle.py --from-state)
            orc2, osrc2 = read_profile_json("train_curve_oracle_from_hip_state")
            if orc2:
                hip = {r_["iter"]: r_ for r_ in curve["log"]}
                both = [(r_["iter"], hip[r_["iter"]]["psnr"], r_["psnr"], hip[r_["iter"]]["faces"]) for r_ in orc2["log"] if r_["iter"] in hip and hip[r_["iter"]].get("faces") == r_.get("faces")]
                after = [b for b in both if b[0] > orc2.get("subdivide_at", 1000)]
                stat = lambda rows: {"iterations_compared": len(rows), "max_abs_psnr_difference_db": round(max(abs(a - b) for _, a, b, _ in rows), 3) if rows else None,
                                     "mean_abs_psnr_difference_db": round(sum(abs(a - b) for _, a, b, _ in rows) / max(len(rows), 1), 4)}
                row["vs_oracle_continuing_from_hip_state"] = {"from_iterations_done": orc2.get("start_iterations_done"), "subdivide_at": orc2.get("subdivide_at"), **stat(both),
                                                              "after_the_subdivision": stat(after), "source": osrc2}
            row["source"] = src
            out["modes"]["cfg2_train_curve"] = row

    # ---------------- the other BASELINE configs (N = 1) ----------------
    if world == 1 and not args.no_configs and not args.no_modes:
        out["configs"] = {"unit": "frames/s", "what": "render path fwd+bwd + Adam, one step in flight; frac = SURVEY 8(d) bytes / HIP-event duration / 8 TB/s",
                          **other_configs(torch, dev, args)}

    # ---------------- CPU baseline: the oracle on this box's host cores (rank 0, N=1 only) ----------------
    if world == 1 and not args.no_cpu_baseline:
        note("cpu baseline")
        st0 = wl.step(B, split=split)
        bt0 = wl.batches(st0)[0]
        with torch.cuda.stream(torch.cuda.Stream(device=wl.device)):
            st0.cam = bt0["cam"]
            if B > 1:
                st0.cams_dev = bt0["cams_dev"]
            st0.forward_backward(wl.params, bt0, bt0["gt_rgb"], bt0["gt_mask"], bt0["bg"], graph=not args.no_graph)
        torch.cuda.synchronize()
        out["cpu_baseline"], psnr = cpu_baseline(torch, args, wl, st0, bt0)
        if psnr is not None:
            out["psnr_vs_oracle_db"] = psnr   # "PSNR vs ref" of BASELINE.json's metric, on the batched step's own image
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        # leave together: a rank that tears its sockets down while another is still inside its last collective takes that one down with it
        torch.cuda.synchronize()
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
