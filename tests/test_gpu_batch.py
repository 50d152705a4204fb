"""Batched launches (gom_batch_forward_backward): B frames through the same 12
kernels must reproduce B single-frame calls BITWISE (images, losses, binning)
and sum the gradients in frame order."""
import numpy as np
import pytest
import torch

from gomavatar_amd import synthetic as syn

pytestmark = pytest.mark.gpu


def _scene(img, B, smpl_like=False):
    body = syn.make_body(0) if smpl_like else syn.icosphere_body(3)
    F, N = body["faces"].shape[0], body["canonical_vertex"].shape[0]
    gp = syn.make_gaussian_params(F)
    w = torch.from_numpy(body["canonical_lbs_weights"]).T
    w25 = torch.cat([w, torch.zeros(1, N)], 0).contiguous()
    faces = torch.from_numpy(body["faces"])
    params = dict(vertices=torch.from_numpy(body["canonical_vertex"]).T.contiguous(), so3=torch.from_numpy(gp["so3"]),
                  scale=torch.from_numpy(gp["scale"]) * 3.0, appearance=torch.from_numpy(gp["appearance"]))
    params = {k: v.cuda() for k, v in params.items()}
    frames = [syn.make_frame(b + 1, img) for b in range(B)]
    rng = np.random.default_rng(3)
    gt_rgb = torch.from_numpy(rng.uniform(0, 1, (B, img, img, 3)).astype(np.float32)).cuda()
    gt_mask = torch.from_numpy((rng.uniform(0, 1, (B, img, img)) > 0.5).astype(np.float32)).cuda()
    return faces, N, w25, params, frames, gt_rgb, gt_mask


@pytest.mark.parametrize("B,img,graph,smpl_like", [(3, 128, False, False), (4, 96, True, False), (2, 256, False, True), (5, 256, True, False)])
def test_batch_equals_single_frames_bitwise(B, img, graph, smpl_like):
    """(5, 256): hundreds of Gaussians over more than 32 tiles per frame -- every frame must make the same choices (k_preprocess_bwd's
    rider waves) alone and in a batch; a list shared by the batch once overflowed where the single frames' did not (scripts/soak.py).
    smpl_like: the 13 776-face body at 256x256 has tile lists on both sides of the 2048-entry split between the
    two k_sort instantiations of a batched launch."""
    from gomavatar_amd.pipeline import RenderStep
    faces, N, w25, params, frames, gt_rgb, gt_mask = _scene(img, B, smpl_like)
    stack = lambda k: torch.from_numpy(np.stack([f[k][0] for f in frames])).contiguous().cuda()
    fr_b = {k: stack(k) for k in ("cnl_gtfms", "dst_Rs", "dst_Ts")}
    bg_b = stack("bgcolor")
    bg4 = (0.1, 0.2, 0.3, 0.0)

    from gomavatar_amd import _lib
    single = RenderStep(faces, N, (img, img), w25)
    single.state.set_option(_lib.OPT_SEG_SHIFT, 8)      # a batch uses 256-entry segments: bitwise equality needs the same split
    imgs, losses, grads, radii = [], [], [], []
    for b in range(B):
        single.set_camera(frames[b]["K"][0], frames[b]["E"][0], bg4)
        fr = {k: fr_b[k][b].contiguous() for k in fr_b}
        single.forward_backward(params, fr, gt_rgb[b].contiguous(), gt_mask[b].contiguous(), bg_b[b].contiguous())
        torch.cuda.synchronize()
        imgs.append(single.image.clone()); losses.append(single.loss_partials.clone()); radii.append(single.radii.clone())
        grads.append({k: v.clone() for k, v in single.grads.items()})

    batch = RenderStep(faces, N, (img, img), w25, batch=B)
    batch.set_cameras([f["K"][0] for f in frames], [f["E"][0] for f in frames], bg4)
    stream = torch.cuda.Stream()
    stream.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(stream):
        for _ in range(3 if graph else 1):   # graph: capture, then replays
            batch.forward_backward(params, fr_b, gt_rgb, gt_mask, bg_b, graph=graph)
    stream.synchronize()
    assert torch.equal(batch.image, torch.stack(imgs))
    assert torch.equal(batch.loss_partials, torch.stack(losses))
    assert torch.equal(batch.radii, torch.stack(radii))
    D, overflow = batch.state.poll()
    assert not overflow and D > 0
    # default single-frame segment size (128): same images up to fp32 association
    auto = RenderStep(faces, N, (img, img), w25)
    auto.set_camera(frames[0]["K"][0], frames[0]["E"][0], bg4)
    auto.forward_backward(params, {k: fr_b[k][0].contiguous() for k in fr_b}, gt_rgb[0].contiguous(), gt_mask[0].contiguous(), bg_b[0].contiguous())
    torch.cuda.synchronize()
    assert float((auto.image - imgs[0]).abs().max()) < 2e-6
    if smpl_like:
        tb = batch.state.export(_lib.BUF_TILE_BASE, torch.empty(B * (img // 16) ** 2 + 1, dtype=torch.int32, device="cuda")).cpu().numpy()
        cnt = np.diff(tb.astype(np.int64))
        assert cnt.max() > 2048 and (cnt[cnt > 0] <= 2048).any()
    for k in ("vertices", "so3", "scale", "appearance"):
        acc = grads[0][k].clone()
        for b in range(1, B):
            acc = acc + grads[b][k]            # same order as k_sum_frames
        assert torch.equal(batch.grads[k], acc), k
        assert float(acc.abs().max()) > 0


def test_batch_new_cameras_need_no_recapture():
    """The cameras live in device memory: a replayed graph renders the new views."""
    from gomavatar_amd.pipeline import RenderStep
    B, img = 2, 96
    faces, N, w25, params, frames, gt_rgb, gt_mask = _scene(img, B + 1)
    stack = lambda k, idx: torch.from_numpy(np.stack([frames[i][k][0] for i in idx])).contiguous().cuda()
    batch = RenderStep(faces, N, (img, img), w25, batch=B)
    stream = torch.cuda.Stream()
    outs = []
    bgc, rgb_t, mask_t = stack("bgcolor", (0, 1)), gt_rgb[:B].contiguous(), gt_mask[:B].contiguous()
    for idx in ((0, 1), (2, 1)):
        fr = {k: stack(k, idx) for k in ("cnl_gtfms", "dst_Rs", "dst_Ts")}
        with torch.cuda.stream(stream):
            batch.set_cameras([frames[i]["K"][0] for i in idx], [frames[i]["E"][0] for i in idx])
            # same device buffers every time so that the captured pointers stay valid
            if not outs:
                pose = {k: v.clone() for k, v in fr.items()}
            else:
                for k in pose:
                    pose[k].copy_(fr[k])
            batch.forward_backward(params, pose, rgb_t, mask_t, bgc, graph=True)
        stream.synchronize()
        outs.append(batch.image.clone())
    assert torch.equal(outs[0][1], outs[1][1])          # frame 1 unchanged
    assert not torch.equal(outs[0][0], outs[1][0])      # frame 0 got the new pose + camera


def test_batch_with_nothing_in_view_and_then_a_normal_one():
    """No (tile, Gaussian) pair at all: the task queues of the segment kernels find zero tasks, every image is the background,
    every gradient zero -- and the same state then renders a normal batch (queue heads were reset)."""
    from gomavatar_amd.pipeline import RenderStep
    B, img = 2, 96
    faces, N, w25, params, frames, gt_rgb, gt_mask = _scene(img, B)
    stack = lambda k: torch.from_numpy(np.stack([f[k][0] for f in frames])).contiguous().cuda()
    fr_b = {k: stack(k) for k in ("cnl_gtfms", "dst_Rs", "dst_Ts")}
    bg_b = stack("bgcolor")
    batch = RenderStep(faces, N, (img, img), w25, batch=B)
    far = []
    for f in frames:
        E = f["E"][0].copy()
        E[2, 3] = -50.0                      # the whole body behind the camera
        far.append(E)
    batch.set_cameras([f["K"][0] for f in frames], far, (0.1, 0.2, 0.3, 0.0))
    batch.forward_backward(params, fr_b, gt_rgb, gt_mask, bg_b)
    torch.cuda.synchronize()
    D, overflow = batch.state.poll()
    assert D == 0 and not overflow
    assert torch.equal(batch.image[:, 3], torch.zeros_like(batch.image[:, 3]))
    for c, v in enumerate((0.1, 0.2, 0.3)):
        assert float((batch.image[:, c] - v).abs().max()) == 0.0
    for k in ("vertices", "so3", "scale", "appearance"):
        assert float(batch.grads[k].abs().max()) == 0.0
    batch.set_cameras([f["K"][0] for f in frames], [f["E"][0] for f in frames], (0.1, 0.2, 0.3, 0.0))
    batch.forward_backward(params, fr_b, gt_rgb, gt_mask, bg_b)
    torch.cuda.synchronize()
    D, overflow = batch.state.poll()
    assert D > 0 and not overflow and float(batch.image[:, 3].max()) > 0.9
    single = RenderStep(faces, N, (img, img), w25)
    from gomavatar_amd import _lib
    single.state.set_option(_lib.OPT_SEG_SHIFT, 8)
    single.set_camera(frames[1]["K"][0], frames[1]["E"][0], (0.1, 0.2, 0.3, 0.0))
    single.forward_backward(params, {k: fr_b[k][1].contiguous() for k in fr_b}, gt_rgb[1].contiguous(), gt_mask[1].contiguous(), bg_b[1].contiguous())
    torch.cuda.synchronize()
    assert torch.equal(batch.image[1], single.image)


@pytest.mark.parametrize("B", [1, 2])
def test_split_call_with_lpips_hook_matches_the_autograd_composition(B):
    """RenderStep(forward half -> LPIPS value + image gradient chained through unpack -> backward half) against the same step
    written with the autograd pieces (posed_face_gaussians -> rasterize -> compute_loss_l1 + LPIPSMatrixCore.loss), summed over
    the frames of the batch."""
    from gomavatar_amd.pipeline import RenderStep
    from gomavatar_amd import rasterizer as R
    from gomavatar_amd.geometry import MeshTopology, posed_face_gaussians
    from gomavatar_amd.losses import compute_loss_l1
    from gomavatar_amd.lpips import LPIPSMatrixCore
    img = 96
    faces, N, w25, params, frames, gt_rgb, gt_mask = _scene(img, B)
    stack = lambda k: torch.from_numpy(np.stack([f[k][0] for f in frames])).contiguous().cuda()
    fr_b = {k: stack(k) for k in ("cnl_gtfms", "dst_Rs", "dst_Ts")}
    bg_b = stack("bgcolor")
    lp = LPIPSMatrixCore(trunk_seed=0, precision="bf16")
    step = RenderStep(faces, N, (img, img), w25, batch=B)
    sq = (lambda t: t) if B > 1 else (lambda t: t[0].contiguous())
    if B > 1:
        step.set_cameras([f["K"][0] for f in frames], [f["E"][0] for f in frames])
    else:
        step.set_camera(frames[0]["K"][0], frames[0]["E"][0])
    step.forward_backward(params, {k: sq(v) for k, v in fr_b.items()}, sq(gt_rgb), sq(gt_mask), sq(bg_b),
                          image_grad_hook=step.lpips_hook(lp, sq(gt_rgb), sq(bg_b), coeff=0.7))
    torch.cuda.synchronize()
    # reference composition, one frame at a time
    topo = MeshTopology(faces, N, device="cuda")
    F = faces.shape[0]
    acc = {k: torch.zeros_like(v) for k, v in params.items()}
    lp_vals = []
    single = RenderStep(faces, N, (img, img), w25)
    if B > 1:
        from gomavatar_amd import _lib
        single.state.set_option(_lib.OPT_SEG_SHIFT, 8)     # a batch uses 256-entry segments: same split -> bitwise the same render
    for b in range(B):
        P = {k: v.clone().requires_grad_() for k, v in params.items()}
        single.set_camera(frames[b]["K"][0], frames[b]["E"][0])
        x, c6, _ = posed_face_gaussians(P["vertices"], P["so3"], P["scale"], fr_b["dst_Rs"][b], fr_b["dst_Ts"][b], fr_b["cnl_gtfms"][b], w25.cuda(), topo, 1e-3)
        f4 = torch.cat([P["appearance"].T, torch.ones(F, 1, device="cuda")], 1)
        o, _ = R.rasterize(x, c6, f4, torch.ones(F, device="cuda"), single.cam, state=single.state)
        total, _ = compute_loss_l1(o, gt_rgb[b], gt_mask[b], bg_b[b])
        rgb, mask = o[:3].permute(1, 2, 0), o[3]
        unpacked = rgb * mask[..., None] + bg_b[b] * (1 - mask[..., None])
        ll = lp.loss(unpacked[None], gt_rgb[b][None])
        (total + 0.7 * ll).backward()
        lp_vals.append(float(ll.detach()))
        for k in acc:
            acc[k] += P[k].grad
    assert abs(float(step.lpips_value) - np.mean(lp_vals)) <= 1e-4 * max(1.0, abs(np.mean(lp_vals)))
    for k in acc:
        d = float((step.grads[k] - acc[k]).norm()) / max(float(acc[k].norm()), 1e-20)
        # Without the hook the two paths agree bitwise; with it the image gradients agree to 2e-8 (the order of three additions)
        # and the geometry gradients -- sums of large cancelling terms under a noise-like LPIPS image gradient -- to 3e-3.
        # (B = 2: the trunk picks other split-K factors for a 4-image batch, and bf16 activations turn a last-bit difference of a
        #  partial sum into a flipped rounding somewhere downstream: 1e-3 .. 1e-2 on every gradient, depending on which layers split --
        #  this is the ONE-PASS bf16 trunk, whose own distance to the float64 gradient is 0.2: tests/test_gpu_vgg_bf16.py)
        assert d < (1e-5 if k == "appearance" and B == 1 else 3e-2), (k, d)


def test_batch_with_small_segments_equals_single_frames_bitwise():
    """GOM_OPT_SEG_SHIFT = 7 in a batched launch: the task queue then hands out 128-entry segments / 32-entry sub-ranges."""
    from gomavatar_amd.pipeline import RenderStep
    from gomavatar_amd import _lib
    B, img = 2, 128
    faces, N, w25, params, frames, gt_rgb, gt_mask = _scene(img, B)
    stack = lambda k: torch.from_numpy(np.stack([f[k][0] for f in frames])).contiguous().cuda()
    fr_b = {k: stack(k) for k in ("cnl_gtfms", "dst_Rs", "dst_Ts")}
    bg_b = stack("bgcolor")
    single = RenderStep(faces, N, (img, img), w25)          # single frames default to 128-entry segments
    imgs, grads = [], []
    for b in range(B):
        single.set_camera(frames[b]["K"][0], frames[b]["E"][0])
        single.forward_backward(params, {k: fr_b[k][b].contiguous() for k in fr_b}, gt_rgb[b].contiguous(), gt_mask[b].contiguous(), bg_b[b].contiguous())
        torch.cuda.synchronize()
        imgs.append(single.image.clone()); grads.append({k: v.clone() for k, v in single.grads.items()})
    batch = RenderStep(faces, N, (img, img), w25, batch=B)
    batch.state.set_option(_lib.OPT_SEG_SHIFT, 7)
    batch.set_cameras([f["K"][0] for f in frames], [f["E"][0] for f in frames])
    batch.forward_backward(params, fr_b, gt_rgb, gt_mask, bg_b)
    torch.cuda.synchronize()
    assert torch.equal(batch.image, torch.stack(imgs))
    for k in ("vertices", "so3", "scale", "appearance"):
        assert torch.equal(batch.grads[k], grads[0][k] + grads[1][k]), k


@pytest.mark.parametrize("B", [1, 3])
def test_face_frame_inside_the_per_gaussian_kernels_matches_the_separate_face_kernels(B):
    """GOM_OPT_FUSE_FACE (default 1): the per-face Gaussian frame and its backward run inside k_preprocess / k_preprocess_bwd.
    Same device functions as the stand-alone face kernels (geom_face.hpp, contraction pinned there): the two launch sequences
    agree bitwise."""
    from gomavatar_amd.pipeline import RenderStep
    from gomavatar_amd import _lib
    img = 128
    faces, N, w25, params, frames, gt_rgb, gt_mask = _scene(img, B, smpl_like=True)
    stack = lambda k: torch.from_numpy(np.stack([f[k][0] for f in frames])).contiguous().cuda()
    fr_b = {k: stack(k) for k in ("cnl_gtfms", "dst_Rs", "dst_Ts")}
    bg_b = stack("bgcolor")
    sq = (lambda t: t[0].contiguous()) if B == 1 else (lambda t: t)
    out = []
    for fuse in (0, 1):
        step = RenderStep(faces, N, (img, img), w25, batch=B)
        step.state.set_option(_lib.OPT_FUSE_FACE, fuse)
        if B == 1:
            step.set_camera(frames[0]["K"][0], frames[0]["E"][0], (0.1, 0.2, 0.3, 0.0))
        else:
            step.set_cameras([f["K"][0] for f in frames], [f["E"][0] for f in frames], (0.1, 0.2, 0.3, 0.0))
        for graph in (False, True, True):
            step.forward_backward(params, {k: sq(v) for k, v in fr_b.items()}, sq(gt_rgb), sq(gt_mask), sq(bg_b), graph=graph)
            torch.cuda.synchronize()
        out.append(dict(image=step.image.clone(), radii=step.radii.clone(), loss=step.loss_partials.clone(), D=step.state.poll()[0],
                        grads={k: v.clone() for k, v in step.grads.items()}))
    a, b = out
    assert a["D"] == b["D"] and torch.equal(a["radii"], b["radii"])
    d_img = float((a["image"] - b["image"]).abs().max())
    print(f"\n[fused face kernels, B={B}] max |image difference| {d_img:.2e}  bitwise: {torch.equal(a['image'], b['image'])}")
    assert torch.equal(a["image"], b["image"]) and torch.equal(a["loss"], b["loss"])
    for k in a["grads"]:
        ga, gb = a["grads"][k], b["grads"][k]
        rel = float((ga - gb).abs().max() / ga.abs().max())
        print(f"[fused face kernels, B={B}] d{k}: max |difference| / max |g| = {rel:.2e}  bitwise: {torch.equal(ga, gb)}")
        assert float(ga.abs().max()) > 0 and torch.equal(ga, gb)


@pytest.mark.parametrize("B,img", [(1, 128), (3, 128), (1, 104), (2, 104)])   # (104: six and a half tiles a side -- the loss riders' partial tiles)
def test_development_switches_do_not_change_results(B, img, monkeypatch):
    """GOM_LOSS_SKIP=0 (the loss kernel reads and writes the pixels of empty tiles too), GOM_BWD_ORDER=0 (the backward's tasks in list
    order) and GOM_FUSE_LOSS=0 (GOM_OPT_FUSE_LOSS: the loss as a launch of its own instead of riding in k_emit / k_combine_fwd) are A/B
    switches: image and every gradient must be BITWISE those of the default path -- the skipped pixels hold the background and no list
    entry reads their gradient; the order of the tasks is not the order of any sum; the loss kernel and the riders share one per-pixel
    function with every rounding spelled out (l1_pixel.hpp).  The loss SUMS are bitwise too where the launch sequence is the same
    (GOM_BWD_ORDER) and agree to fp32 summation order where the stand-alone kernel sums strided blocks and the riders sum tiles."""
    from gomavatar_amd.pipeline import RenderStep
    faces, N, w25, params, frames, gt_rgb, gt_mask = _scene(img, B)
    stack = lambda k: torch.from_numpy(np.stack([f[k][0] for f in frames])).contiguous().cuda()
    fr_b = {k: stack(k) for k in ("cnl_gtfms", "dst_Rs", "dst_Ts")}
    bg_b = stack("bgcolor")
    bg4 = (0.1, 0.2, 0.3, 0.0)

    def run():
        step = RenderStep(faces, N, (img, img), w25, batch=B)   # (the switches are read when the state is created)
        if B > 1:
            step.set_cameras([f["K"][0] for f in frames], [f["E"][0] for f in frames], bg4)
            step.forward_backward(params, fr_b, gt_rgb, gt_mask, bg_b)
        else:
            step.set_camera(frames[0]["K"][0], frames[0]["E"][0], bg4)
            step.forward_backward(params, {k: v[0].contiguous() for k, v in fr_b.items()}, gt_rgb[0].contiguous(), gt_mask[0].contiguous(), bg_b[0].contiguous())
        torch.cuda.synchronize()
        return step.image.clone(), step.loss_partials.clone(), {k: v.clone() for k, v in step.grads.items()}

    ref_img, ref_loss, ref_grads = run()
    assert float(ref_img[..., 3, :, :].max()) > 0.5 and all(float(g.abs().max()) > 0 for g in ref_grads.values())
    for name in ("GOM_LOSS_SKIP", "GOM_BWD_ORDER", "GOM_FUSE_LOSS"):
        monkeypatch.setenv(name, "0")
        im, lo, gr = run()
        monkeypatch.delenv(name)
        assert torch.equal(im, ref_img), name
        if name == "GOM_BWD_ORDER":
            assert torch.equal(lo, ref_loss), name
        else:   # (GOM_LOSS_SKIP=0 also takes the stand-alone kernel)
            a, b = lo.double().sum(-2), ref_loss.double().sum(-2)
            print(f"\n[{name}=0, B={B}] loss sums {a.flatten().tolist()} vs riders {b.flatten().tolist()}")
            assert float(b.abs().min()) > 0 and float(((a - b).abs() / b.abs()).max()) < 2e-6, name
        for k in ref_grads:
            assert torch.equal(gr[k], ref_grads[k]), (name, k)


def test_recorded_step_survives_a_larger_frame_on_the_same_state():
    """A recorded launch sequence (GOM_FRAME_USE_GRAPH) holds the addresses of the state's buffers.  A larger frame on the SAME state
    re-allocates them: the old recording must be dropped and re-recorded, not replayed on freed memory."""
    from gomavatar_amd.pipeline import RenderStep
    B = 2
    faces, N, w25, params, frames, gt_rgb, gt_mask = _scene(64, B)
    stack = lambda fs, k: torch.from_numpy(np.stack([f[k][0] for f in fs])).contiguous().cuda()
    fr_b = {k: stack(frames, k) for k in ("cnl_gtfms", "dst_Rs", "dst_Ts")}
    bg_b = stack(frames, "bgcolor")
    small = RenderStep(faces, N, (64, 64), w25, batch=B)
    small.set_cameras([f["K"][0] for f in frames], [f["E"][0] for f in frames])
    stream = torch.cuda.Stream()
    stream.wait_stream(torch.cuda.current_stream())

    def run_small():
        with torch.cuda.stream(stream):
            for _ in range(2):   # record, replay
                small.forward_backward(params, fr_b, gt_rgb, gt_mask, bg_b, graph=True)
        stream.synchronize()
        return small.image.clone(), small.loss_partials.clone(), {k: v.clone() for k, v in small.grads.items()}

    ref = run_small()
    # a 4x larger image through the same state: every per-tile / per-pixel / pair buffer is re-allocated
    _, _, _, _, frames2, gt_rgb2, gt_mask2 = _scene(256, B)
    big = RenderStep(faces, N, (256, 256), w25, batch=B)
    big.state = small.state
    big.set_cameras([f["K"][0] for f in frames2], [f["E"][0] for f in frames2])
    fr2 = {k: stack(frames2, k) for k in ("cnl_gtfms", "dst_Rs", "dst_Ts")}
    with torch.cuda.stream(stream):
        big.forward_backward(params, fr2, gt_rgb2, gt_mask2, stack(frames2, "bgcolor"), graph=True)
    stream.synchronize()
    assert float(big.image[:, 3].max()) > 0.5
    again = run_small()
    assert torch.equal(again[0], ref[0]) and torch.equal(again[1], ref[1])
    for k in ref[2]:
        assert torch.equal(again[2][k], ref[2][k]), k


@pytest.mark.parametrize("B,K,img,graph", [(4, 2, 96, True), (4, 4, 96, False), (6, 2, 128, True), (8, 2, 256, True), (8, (3, 3, 2), 128, True), (5, (4, 1), 96, False)])
def test_split_step_equals_one_launch_sequence_bitwise(B, K, img, graph):
    """pipeline.SplitRenderStep / gom_split_forward_backward (ABI 11): ONE step of B frames as K CONCURRENT launch sequences of B / K frames,
    (K a tuple: sequences of those sizes) each on its own state and stream between a fork and a join, closed by one frame sum over all B frames in frame order -- against
    RenderStep(batch=B), the one launch sequence: images, loss partials, radii and all four gradients BITWISE equal (a frame's results do not
    depend on the launch it rides in; the sum adds the same slices in the same order).  (4, 4): one frame per branch -- the single-frame
    kernels with per-frame gradient slices.  graph: capture, then replays of the recorded fork / join."""
    from gomavatar_amd.pipeline import RenderStep, SplitRenderStep
    faces, N, w25, params, frames, gt_rgb, gt_mask = _scene(img, B)
    stack = lambda k: torch.from_numpy(np.stack([f[k][0] for f in frames])).contiguous().cuda()
    fr_b = {k: stack(k) for k in ("cnl_gtfms", "dst_Rs", "dst_Ts")}
    bg_b = stack("bgcolor")
    bg4 = (0.1, 0.2, 0.3, 0.0)
    from gomavatar_amd import _lib
    one = RenderStep(faces, N, (img, img), w25, batch=B)
    split = SplitRenderStep(faces, N, (img, img), w25, batch=B, split=K)
    if 1 in split.sizes:
        split.state.set_option(_lib.OPT_SEG_SHIFT, 8)     # (a one-frame branch would pick 128-entry segments: bitwise equality needs the batch's 256)
    stream = torch.cuda.Stream()
    stream.wait_stream(torch.cuda.current_stream())
    res = []
    for st in (one, split):
        st.set_cameras([f["K"][0] for f in frames], [f["E"][0] for f in frames], bg4)
        with torch.cuda.stream(stream):
            for _ in range(3 if graph else 1):
                st.forward_backward(params, fr_b, gt_rgb, gt_mask, bg_b, graph=graph)
        stream.synchronize()
        assert st.state.poll()[0] > 0 and not st.state.poll()[1]
        res.append((st.image.clone(), st.loss_partials.clone(), st.radii.clone(), st.d_image.clone(), {k: v.clone() for k, v in st.grads.items()}))
    for a, b in zip(res[0][:4], res[1][:4]):
        assert a.shape == b.shape and torch.equal(a, b)
    for k in res[0][4]:
        assert torch.equal(res[0][4][k], res[1][4][k]), k
        assert float(res[0][4][k].abs().max()) > 0
    assert one.state.poll()[0] == split.state.poll()[0]      # the same number of (tile, Gaussian) pairs in all
    # a branch state replaced (the old one destroyed, its address possibly handed out again): the lead's recording that names it is neither replayed nor
    # dereferenced (recordings remember their states by uid), the step is recorded anew and gives the same bits
    if graph:
        from gomavatar_amd.rasterizer import RasterState
        old_state = split.parts[-1].state
        split.parts[-1].state = RasterState()
        if 1 in split.sizes:
            split.parts[-1].state.set_option(_lib.OPT_SEG_SHIFT, 8)
        split.state.states[-1] = split.parts[-1].state
        del old_state
        for v in split.grads.values():
            v.zero_()
        with torch.cuda.stream(stream):
            for _ in range(2):
                split.forward_backward(params, fr_b, gt_rgb, gt_mask, bg_b, graph=True)
        stream.synchronize()
        assert torch.equal(split.image, res[0][0])
        for k in res[0][4]:
            assert torch.equal(split.grads[k], res[0][4][k]), k
    # refusals: loud, not silent
    with pytest.raises(NotImplementedError):
        split.forward_backward(params, fr_b, gt_rgb, gt_mask, bg_b, backward=False)
