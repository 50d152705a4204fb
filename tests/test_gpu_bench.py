"""bench.py end to end on the GPU box: the contract of the JSON line, and `--gpus 2` launching itself (WORLD_SIZE unset ->
re-exec under torch.distributed.run; on a 1-GPU box the two ranks share device 0 over gloo: a functional proof that process-group
init, the collective inside `torch.cuda.stream(...)` and the MAX-reduced timing work before an 8-GPU node runs them over RCCL)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*extra, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *extra], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                       timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_line_contract_single_gpu():
    d = _run("--steps", "20", "--warmup", "5", "--no-modes", "--cpu-frames", "1")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5 and d["unit"] == "frames/s" and d["vs_baseline"] is None
    assert d["config"]["gaussians"] == 55104 and d["config"]["image"] == [512, 512] and d["config"]["steps_in_flight_per_gpu"] == 1
    # 20 steps are a few milliseconds: the region is repeated until 0.25 s are covered, every repetition exactly 20 steps
    assert d["timed_regions"] > 1 and abs(d["value"] - 8 * 1e3 / d["ms_per_step"]) <= 1e-3 * d["value"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel"):
        assert k in r, k
    assert r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    # round 6: the bound that holds is named (VALU issue), the HBM figures stay SURVEY 8(d)'s; the timed step is two concurrent 4-frame launch
    # sequences (gom_split_forward_backward), the roofline describes the one-sequence 8-frame launch and carries the timed configuration's own launches
    assert r["bound"] == "valu_issue" and "hbm" in r["bound_of_the_figures_below"]
    assert r["issue"] is None or (r["issue"]["issue_floor_us"] > 0 and 0 < r["issue"]["frac_of_issue_floor"] <= 1.0)
    assert d["config"]["launch_sequences_per_step"] == 2 and "2 concurrent launch sequences of 4 frames" in d["config"]["workload"]
    tc = r["timed_configuration"]
    assert tc["split"] == 2 and tc["frames_per_launch"] == 4 and tc["avg_us_per_launch_alone"] > 0 and 0 < tc["frac_alone"] < 1
    assert tc["sum_of_kernels_us_alone"] > 1e3 * tc["ms_per_step_timed"]          # the launches overlap in the timed loop: that is the point
    assert r["traffic_source"] is None or (r["traffic_source"]["measured_in_this_run"] is False and len(r["traffic_source"]["sha256_16"]) == 16)
    assert d["config"]["optimizer"]
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and set(c["rows"]) >= {"S_threads1", "M_threads1"}
    assert c["rows"]["M_threads1"]["fwd_only_frames_per_s"] > c["rows"]["M_threads1"]["frames_per_s"] > 0
    assert d["psnr_vs_oracle_db"] > 80.0


def test_bench_launches_itself_for_two_ranks():
    # (Until round 5 this test retried once on "Connection closed by peer": a rank aborting under the other.  That was no transport hiccup but a
    #  race of this library's own -- counters zeroed by hipMemset on the NULL stream, met unzeroed by the first frame's kernels on a non-blocking
    #  stream when a second process held the device (gom_api.hip: zero_now; LABBOOK R5.9; 0 of 100 start-ups since the fix).  No retry any more.)
    d = _run("--gpus", "2", "--steps", "6", "--warmup", "2")
    # WEAK scaling (round 6): the per-GPU job at N > 1 is the N = 1 line's -- 8 frames per GPU per step as two concurrent launch sequences --, one exchange of
    # the mean gradient + Adam per step inside the timed loop; BASELINE configs[3]'s literal point (one frame per GPU per step) is modes.b1_per_gpu
    # (the native render step exchanges what it trains -- vertices / so3 / scale / appearance of the metric workload, no padding: 3 * 27 554 + 9 * 55 104)
    assert d["n_gpus"] == 2 and d["config"]["frames_per_step"] == 16 and d["config"]["frames_per_gpu_per_step"] == 8 and d["config"]["allreduce_floats"] == 578598
    assert d["config"]["launch_sequences_per_step"] == 2 and d["scaling"] == "weak"
    assert d["config"]["allreduce_us"] > 0 and d["config"]["parallelism"] == "frame-dp2" and d["value"] > 0
    assert d["config"]["optimizer"] and d["config"]["local_only_fps"] >= d["value"] * 0.5 and d["modes"]["b1_per_gpu"] > 0 and d["modes"]["b1_per_gpu_local_only"] > 0
    # the direct peer-pointer all-reduce comes up between two processes on the one device and carries the same loop
    pr = d["config"]["allreduce_peer"]
    assert pr["probe"] == "ok" and pr["status"] == "ok" and pr["us"] > 0 and pr["fps"] > 0, pr      # (probe: the exchange tried in child processes first)
    # BASELINE configs[3] in the reference's step shape: the Model iteration frame-parallel (parallel.ModelFrameParallel), the reference model's REAL parameter count
    mp = d["modes"]["model_parallel"]
    assert d["config"]["model_param_floats"] == 951023 and d["config"]["model_allreduce_floats"] == 951029
    for k in ("local_only_ips", "model_train_iteration_lpips_bf16x3_collective_ips", "model_train_iteration_lpips_bf16x3_peer_ips", "model_train_iteration_lpips_bf16x3_peer_zero1_ips"):
        assert isinstance(mp[k], float) and mp[k] > 0, (k, mp[k])
    assert "cpu_baseline" not in d     # rank 0 at N = 1 only


def test_bench_eight_ranks_end_to_end_on_one_device():
    """BASELINE configs[3]'s launch line -- `python -m torch.distributed.run --nproc-per-node 8 bench.py --gpus 8`, what the driver runs on an 8-GPU
    node -- end to end with the eight ranks SHARING this box's one device (gloo + the hipIpc peer exchange; S body at 256^2 so that eight processes
    fit a test): rendezvous on 127.0.0.1, 8 frames per rank per step (the N = 1 line's per-GPU job: weak scaling), the exchange + Adam inside the timed loop, MAX-over-ranks timing, ONE JSON
    line from rank 0 with the N > 1 schema (allreduce_us, local_only_fps, the peer block, modes.b1_per_gpu = configs[3] as written, modes.model_parallel over collective /
    peer / ZeRO-1).  A functional proof of every code path the first real 8-GPU run takes except RCCL's own transport (DESIGN.md section 7 holds the
    numbers that run will be judged against)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1", "--master-port", "29547",
           os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "12", "--warmup", "3", "--subdiv", "0", "--img", "256", "--no-configs"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]           # rank 0 alone prints
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["steps"] == 12 and d["warmup"] == 3 and d["scaling"] == "weak" and d["value"] > 0
    c = d["config"]
    assert c["frames_per_step"] == 64 and c["frames_per_gpu_per_step"] == 8 and c["parallelism"] == "frame-dp8" and c["launch_sequences_per_step"] == 2
    assert c["allreduce_floats"] == 3 * 6890 + 9 * 13776 and c["allreduce_us"] > 0 and c["local_only_fps"] > 0 and "8 ranks share 1 device" in c["backend"]
    assert abs(d["value"] - 64 * 1e3 / d["ms_per_step"]) <= 1e-3 * d["value"]           # whole-job frames / MAX-over-ranks time
    pr = c["allreduce_peer"]
    assert pr["probe"] == "ok" and pr["status"] == "ok" and pr["fps"] > 0 and pr["zero1_fps"] > 0, pr
    assert d["modes"]["b1_per_gpu"] > 0 and d["modes"]["b8_per_gpu"] > 0 and d["modes"]["b1_per_gpu_local_only"] > 0
    mp = d["modes"]["model_parallel"]
    for k in ("local_only_ips", "model_train_iteration_lpips_bf16x3_collective_ips", "model_train_iteration_lpips_bf16x3_peer_ips", "model_train_iteration_lpips_bf16x3_peer_zero1_ips"):
        assert isinstance(mp[k], float) and mp[k] > 0, (k, mp[k])
    assert c["model_allreduce_floats"] >= c["model_param_floats"] > 0 and "cpu_baseline" not in d


def test_flat_adam_is_torch_adam():
    """gom_adam_flat (one launch over the flat parameter buffer, per-tensor learning rates, update_lr's decay) against
    torch.optim.Adam with the same groups -- the reference's optimizer (train.py:263-267) -- over several steps."""
    import torch
    from gomavatar_amd.parallel import FlatAdam, FrameParallel, shapes_for_model
    shapes = shapes_for_model(1001, 2003, extra=[("mlp.w", (7, 13)), ("mlp.b", (5,))])
    fp = FrameParallel(shapes, "cuda", pad_to=3 * 1001 + 9 * 2003 + 96 + 33)
    g = torch.Generator(device="cuda").manual_seed(0)
    fp.params.flat.copy_(torch.randn(fp.params.numel, device="cuda", generator=g))
    lrs = {"vertices": 5e-4, "so3": 1e-3, "default": 2e-3}
    ref_p = {k: v.detach().clone().requires_grad_() for k, v in fp.params.items()}
    ref = torch.optim.Adam([{"params": [ref_p[k]], "lr": lrs.get(k, lrs["default"]), "name": k} for k in ref_p], betas=(0.9, 0.999))
    opt = FlatAdam(fp, lrs)
    pad0 = fp.grads.flat[fp.params.numel:].clone()
    for it in range(6):
        fp.grads.flat.copy_(torch.randn(fp.grads.numel, device="cuda", generator=g) * (10.0 ** (it - 3)))
        for k in ref_p:
            ref_p[k].grad = fp.grads[k].clone() * 0.5
        for pg in ref.param_groups:
            pg["lr"] = lrs.get(pg["name"], lrs["default"]) * 0.1 ** (it / 100.0)
        opt.decay(it, 100.0)
        opt.step(grad_scale=0.5)
        ref.step()
        for k in ref_p:
            a, b = fp.params[k], ref_p[k].detach()
            assert float((a - b).abs().max()) <= 2e-6 * float(b.abs().max()), (it, k, float((a - b).abs().max()))
    assert opt.t == 6 and torch.isfinite(fp.params.flat).all()
    # the graphable variant (step count + learning-rate schedule on the device), captured ONCE and replayed, takes the same six steps
    fp2 = FrameParallel(shapes, "cuda", pad_to=3 * 1001 + 9 * 2003 + 96 + 33)
    g2 = torch.Generator(device="cuda").manual_seed(0)
    fp2.params.flat.copy_(torch.randn(fp2.params.numel, device="cuda", generator=g2))
    opt2 = FlatAdam(fp2, lrs, graphable=True, lr_decay_steps=100.0)
    gs = [torch.randn(fp2.grads.numel, device="cuda", generator=g2) * (10.0 ** (it - 3)) for it in range(6)]
    stream = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    stream.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(stream):
        fp2.grads.flat.copy_(gs[0])
        with torch.cuda.graph(graph, stream=stream):
            opt2.step(grad_scale=0.5)
        # (capture does not execute: the replays below are the six steps)
        for it in range(6):
            fp2.grads.flat.copy_(gs[it])
            graph.replay()
    stream.synchronize()
    assert int(opt2.step_dev[0]) == 6 and int(opt2.step_dev[1]) == 0
    for k in ref_p:
        a, b = fp2.params[k], ref_p[k].detach()
        assert float((a - b).abs().max()) <= 2e-6 * float(b.abs().max()), (k, float((a - b).abs().max()))


def test_optimizer_attached_to_the_frame_graph_takes_the_same_steps():
    """gom_state_set_frame_optimizer: the Adam launch as the last launch of the frame step's recorded graph (forward + backward + optimizer =
    one graph replay) against the frame step followed by a separate gom_adam_flat: same parameters and moments, bit for bit, over three steps
    in which step k + 1 renders with what step k wrote."""
    import torch
    from gomavatar_amd import _lib
    from gomavatar_amd.parallel import FlatAdam, FrameParallel, shapes_for_model
    from gomavatar_amd.workload import MetricWorkload
    if not _lib.has_lab():
        pytest.skip("laboratory entry point (include/gom_hip_lab.h): run with GOM_HIP_LIB pointing at a -DGOM_LAB build (scripts/exp_build.py lab -DGOM_LAB)")
    wl = MetricWorkload("cuda", subdiv=0, img=128, n_frames=4)
    runs = []
    for attached in (False, True):
        st = wl.step(2)
        bt = wl.batches(st)[0]
        fp = FrameParallel(shapes_for_model(wl.N, wl.F), "cuda")
        for k in ("vertices", "so3", "scale", "appearance"):
            st.grads[k] = fp.grads[k]
            fp.params[k].copy_(wl.params[k])
        opt = FlatAdam(fp, {"default": 1e-3}, graphable=attached, lr_decay_steps=50.0)
        if attached:
            opt.attach(st.state, 0.5)
        pv = dict(fp.params.items())
        stream = torch.cuda.Stream()
        stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(stream):
            st.cam = bt["cam"]; st.cams_dev = bt["cams_dev"]
            for it in range(3):
                st.forward_backward(pv, bt, bt["gt_rgb"], bt["gt_mask"], bt["bg"], graph=True)
                if not attached:
                    opt.decay(it + 1, 50.0)          # update_lr's schedule as the device variant derives it: step t uses base * 0.1^((t - 1) / D)
                    opt.lr = [b * 0.1 ** (it / 50.0) for b in opt.base_lr]
                    opt.step(0.5)
        stream.synchronize()
        runs.append((fp.params.flat.clone(), opt.exp_avg.clone(), opt.exp_avg_sq.clone(), int(opt.step_dev[0]) if attached else opt.t))
    (p0, m0, v0, t0), (p1, m1, v1, t1) = runs
    assert t0 == t1 == 3
    assert float((p0 - wl_flat(wl, p0)).abs().max()) > 0          # the parameters really moved
    # (host-side float(lr / bc1) against the device's double arithmetic: the last bit of step_size may differ -> a few ulp on the parameters)
    assert float((p0 - p1).abs().max()) <= 2e-6 * float(p0.abs().max()) and torch.allclose(m0, m1, rtol=1e-6, atol=0) and torch.allclose(v0, v1, rtol=1e-6, atol=0)


def wl_flat(wl, like):
    import torch
    return torch.cat([wl.params[k].reshape(-1) for k in ("vertices", "so3", "scale", "appearance")])[: like.numel()]
