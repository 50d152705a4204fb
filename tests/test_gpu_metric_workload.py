"""Parity ON THE METRIC WORKLOAD ITSELF (BASELINE.json: 55 104 Gaussians at 512x512; gomavatar_amd.workload, the very object
bench.py times): one frame at a time (B = 1) and the batched launch of 8 frames bench.py measures (configs[1] late phase and
the per-GPU work of configs[3]).

  * batched RenderStep(batch=8): images / losses / radii bitwise equal to 8 single-frame calls, gradients = their ordered sum;
  * every integer output of the binning of each of the 8 frames bit-exact against the C oracle (raster parity unpinned: the
    oracle restates the un-vendored CUDA rasterizer, see oracle/raster_oracle.c);
  * images within 1e-4 per pixel up to threshold flips, whose COUNT is printed (run with -s) and bounded;
  * gradients of the 8-frame step against the fp64 oracle, summed over the frames."""
import numpy as np
import pytest
import torch

from oracle import geometry as og, raster as orast

pytestmark = pytest.mark.gpu

IMG_TOL, FLIP_RATE, MEAN_TOL = 1e-4, 2e-4, 2e-6     # as tests/test_gpu_raster.py
MARGIN = 2e-5                                       # as tests/test_gpu_raster.py: relative distance to a branch threshold below which a pixel may flip
# |err| / max|g| bounds (median, q99, q99.9, max), <= 3x the values measured on MI355X (printed with -s), per parameter:
#   BOUND_VS_FP32    HIP against the fp32 build of the oracle, same image gradient
#   BOUND_STEP       whole step including the L1 loss's sign(), against the float64 oracle (round 2's bounds here were 1e-5 / 2e-3 / 5e-2 / 0.25)
BOUND_VS_FP32 = dict(vertices=(5e-8, 6e-6, 7e-4, 1e-2), so3=(3e-9, 1.2e-6, 9e-5, 8e-2), scale=(1.2e-9, 4e-7, 9e-5, 0.15), appearance=(4e-8, 5e-6, 2e-5, 2e-2))
BOUND_SUM = dict(vertices=(6e-7, 3e-4, 2.4e-3, 7.5e-2), so3=(3e-8, 2.4e-5, 3e-4, 6.5e-2), scale=(2.1e-8, 2.1e-5, 3e-4, 0.18), appearance=(2.1e-7, 6e-6, 1e-4, 2e-2))   # the sum over the 8 frames of the batch
BOUND_STEP = dict(vertices=(7e-8, 1e-4, 2.2e-3, 7.5e-2), so3=(4e-9, 1e-5, 2.6e-4, 6.5e-2), scale=(2e-9, 9e-6, 2.5e-4, 0.18), appearance=(7e-8, 6e-6, 1e-4, 2e-2))


@pytest.fixture(scope="module")
def wl():
    from gomavatar_amd.workload import MetricWorkload
    import os
    orast.set_threads(min(os.cpu_count() or 1, 64))
    return MetricWorkload("cuda", subdiv=1, img=512, n_frames=8)


def _export_batch(step, B, P, H, W):
    from gomavatar_amd import _lib
    st = step.state
    gx, gy = (W + 15) // 16, (H + 15) // 16
    D, overflow = st.poll()
    assert not overflow
    i32 = lambda n: torch.empty(n, dtype=torch.int32, device="cuda")
    e = dict(D=D)
    e["depth"] = st.export(_lib.BUF_DEPTH, torch.empty(B * P, device="cuda")).cpu().numpy().reshape(B, P)
    e["xy"] = st.export(_lib.BUF_XY, torch.empty((B * P, 2), device="cuda")).cpu().numpy().reshape(B, P, 2)
    e["conic_opacity"] = st.export(_lib.BUF_CONIC_OPACITY, torch.empty((B * P, 4), device="cuda")).cpu().numpy().reshape(B, P, 4)
    e["tiles_touched"] = st.export(_lib.BUF_TILES_TOUCHED, i32(B * P)).cpu().numpy().view(np.uint32).reshape(B, P)
    e["rect"] = st.export(_lib.BUF_RECT, torch.empty((B * P, 4), dtype=torch.int16, device="cuda")).cpu().numpy().view(np.uint16).astype(np.int32).reshape(B, P, 4)
    e["tile_base"] = st.export(_lib.BUF_TILE_BASE, i32(B * gx * gy + 1)).cpu().numpy().view(np.uint32)
    e["keys"] = st.export(_lib.BUF_KEYS, torch.empty(max(D, 1), dtype=torch.int64, device="cuda")).cpu().numpy().view(np.uint64)[:D]
    e["point_list"] = st.export(_lib.BUF_POINT_LIST, i32(max(D, 1))).cpu().numpy().view(np.uint32)[:D]
    e["final_T"] = st.export(_lib.BUF_FINAL_T, torch.empty((B, H, W), device="cuda")).cpu().numpy()
    e["n_contrib"] = st.export(_lib.BUF_N_CONTRIB, i32(B * H * W)).cpu().numpy().view(np.uint32).reshape(B, H, W)
    return e


def _oracle_forward(wl, i, dtype=np.float32):
    with torch.no_grad():
        p = {k: (v.double() if dtype == np.float64 else v) for k, v in wl.params_cpu.items()}
        fr = wl.oracle_frame(i)
        if dtype == np.float64:
            fr = {k: v.double() for k, v in fr.items()}
        _, _, aux = og.render_path(p, fr, wl.faces, wl.w25.double() if dtype == np.float64 else wl.w25, wl.img)
    return aux


@pytest.mark.parametrize("B", [1, 8])
def test_metric_workload_matches_oracle_and_batch_matches_singles(wl, B, capsys):
    from gomavatar_amd import _lib
    img, P = wl.img, wl.F
    gx = gy = img // 16
    T = gx * gy
    step = wl.step(B)
    bt = wl.batches(step)[0]
    stream = torch.cuda.Stream()
    stream.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(stream):
        step.cam = bt["cam"]
        if B > 1:
            step.cams_dev.copy_(bt["cams_dev"])
        for _ in range(3):   # capture + replays, as in the timed loop of bench.py
            step.forward_backward(wl.params, bt, bt["gt_rgb"], bt["gt_mask"], bt["bg"], graph=True)
    stream.synchronize()
    image = step.image.reshape(B, 4, img, img).clone()
    radii = step.radii.reshape(B, P).cpu().numpy()
    grads = {k: v.clone() for k, v in step.grads.items()}
    dimg = step.d_image.reshape(B, 4, img, img).clone()           # dL/d(image) the backward consumed (zero on empty tiles: never written, never read)
    lr, lm = step.losses()
    lr, lm = lr.reshape(B).cpu().numpy(), lm.reshape(B).cpu().numpy()
    e = _export_batch(step, B, P, img, img)

    # ---- (a) a batch is bitwise the frames one by one (same segment size), gradients their frame-ordered sum
    frame_grads = None
    if B > 1:
        single = wl.step(1)
        single.state.set_option(_lib.OPT_SEG_SHIFT, 8)
        acc, frame_grads = None, []
        for b in range(B):
            d = wl.frames[b]
            single.set_camera(d["K"], d["E"])
            single.forward_backward(wl.params, d, d["gt_rgb"], d["gt_mask"], d["bg"])
            torch.cuda.synchronize()
            assert torch.equal(single.image, image[b]), b
            assert torch.equal(single.radii, step.radii[b]) and torch.equal(single.loss_partials, step.loss_partials[b])
            frame_grads.append({k: v.clone() for k, v in single.grads.items()})
            acc = {k: v.clone() for k, v in single.grads.items()} if acc is None else {k: acc[k] + single.grads[k] for k in acc}
        for k in acc:
            assert torch.equal(grads[k], acc[k]), k

    # ---- (b) per frame: integer state bit-exact vs the C oracle, image parity with the flip count
    flips_total, ncontrib_mismatch, worst, fragile_total, ill_total, hipform_total = 0, 0, 0.0, 0, 0, 0
    for b in range(B):
        aux = _oracle_forward(wl, b)
        feat = torch.cat([wl.params_cpu["appearance"].T, torch.ones(P, 1)], -1).numpy()
        # the geometry kernels feed the rasterizer means / covariances that differ from the torch oracle's in the last bits;
        # bit-exactness of the BINNING is therefore asserted on the rasterizer's own inputs: re-run the oracle on the HIP inputs
        xyz_h, cov_h = step.xyz.reshape(B, P, 3)[b].cpu().numpy(), step.cov6.reshape(B, P, 6)[b].cpu().numpy()
        f = orast.forward(aux["cam"], xyz_h, cov_h, feat, np.ones(P, np.float32), margin=True)
        np.testing.assert_array_equal(radii[b], f["radii"])
        np.testing.assert_array_equal(e["tiles_touched"][b], f["tiles_touched"])
        rect = e["rect"][b].copy(); rect[:, 1] -= b * gy; rect[:, 3] -= b * gy          # stacked tile rows
        vis = f["radii"] > 0
        np.testing.assert_array_equal(rect[vis], f["rect"][vis])
        for name in ("depth", "xy", "conic_opacity"):
            np.testing.assert_array_equal(e[name][b].view(np.uint32), f[name].astype(np.float32).view(np.uint32))
        tb = e["tile_base"][b * T:(b + 1) * T + 1].astype(np.int64)
        cnt = (f["ranges"][:, 1] - f["ranges"][:, 0]).astype(np.int64)
        np.testing.assert_array_equal(np.diff(tb), cnt)
        lo, hi = int(tb[0]), int(tb[-1])
        assert hi - lo == f["D"]
        np.testing.assert_array_equal(e["point_list"][lo:hi].astype(np.int64) - b * P, f["point_list"].astype(np.int64))
        np.testing.assert_array_equal((e["keys"][lo:hi] >> np.uint64(32)).astype(np.uint32), (f["keys"] & np.uint64(0xffffffff)).astype(np.uint32))
        err = np.abs(image[b].cpu().numpy() - f["color"])
        bad = err.max(axis=0) > IMG_TOL
        flips_total += int(bad.sum()); worst = max(worst, float(err.max()))
        assert err.mean() <= MEAN_TOL and bad.sum() <= max(1, int(FLIP_RATE * bad.size)) and err.max() <= 1.5e-2, (b, err.mean(), int(bad.sum()), err.max())
        mism = int((e["n_contrib"][b] != f["n_contrib"]).sum())
        ncontrib_mismatch += mism
        assert mism <= max(1, int(FLIP_RATE * bad.size)), (b, mism)
        # ... and every one of them IS a threshold flip (tests/test_gpu_raster.py assert_flips_are_threshold_margins): ZERO pixels with a
        # comfortable margin deviate, in value, n_contrib or final T
        # the oracle in the HIP path's own formulation of alpha (oracle/raster_oracle.c A.3, form 1): ZERO pixels beyond 1e-4, ZERO n_contrib mismatches
        fh = orast.forward(aux["cam"], xyz_h, cov_h, feat, np.ones(P, np.float32), form="hip")
        dev_hip = int(((np.abs(image[b].cpu().numpy() - fh["color"]).max(axis=0) > IMG_TOL) | (e["n_contrib"][b] != fh["n_contrib"])).sum())
        hipform_total += dev_hip
        assert dev_hip == 0, (b, dev_hip)
        firm = f["margin"] >= MARGIN
        ill = firm & (f["roundoff"] > 2e-5)                      # needle-shaped conics far from their centre: ill-conditioned in fp32 in any formulation
        solid = firm & ~ill
        fragile_total += int((~firm).sum()); ill_total += int(ill.sum())
        e1 = err.max(axis=0)
        assert np.array_equal(e["n_contrib"][b][firm], f["n_contrib"][firm]), b
        assert float(e1[solid].max()) <= 5e-5 and float(np.abs(e["final_T"][b] - f["final_T"])[solid].max()) <= 5e-5, (b, float(e1[solid].max()))   # (measured 7e-6 .. 3e-5)
        assert not ill.any() or float((e1[ill] - (2e-5 + 3.0 * f["roundoff"][ill])).max()) <= 0.0, b
        assert (~firm).sum() <= 3e-3 * firm.size and ill.sum() <= 3e-2 * firm.size
    with capsys.disabled():
        print(f"\n[metric workload, B={B}] pairs D={e['D']}  threshold-flip pixels (|d|>1e-4): {flips_total} of {B * img * img}"
              f"  n_contrib mismatches: {ncontrib_mismatch}  max |d|={worst:.2e}   pixels with a branch margin < {MARGIN:g}: {fragile_total}, ill-conditioned in fp32: {ill_total} -- every other pixel within 5e-5, same n_contrib;  against the oracle in the HIP formulation of alpha: {hipform_total} pixels beyond 1e-4 or with another n_contrib")

    # ---- (c) losses and gradients against the fp64 oracle, every frame on its own (the batch is bitwise their ordered sum, (a)) and
    # the sum over the frames.  Two comparisons:
    #   (c1) the rasterizer + geometry backward GIVEN THE SAME IMAGE GRADIENT (the oracle is driven with the HIP path's own
    #        dL/d(image)): what remains is fp32 arithmetic and the rasterizer's threshold flips -- and the flips are ATTRIBUTED: the
    #        Gaussians that reach (alpha >= 1/255) a pixel where the two forwards took different branches (|d image| > 1e-4 or another
    #        n_contrib) are set aside, the rest must agree to 2e-4 of the largest gradient; the ones set aside are bounded separately;
    #   (c2) the whole step including the loss: L1's sign() adds its own discontinuity (inside the body both masks sit within 1e-4 of
    #        1, so sign(mask - target) is decided in the last bits, fp32 against float64) -- bounded at 3x what was measured.
    def grad_stats(got, ref, keep=None):
        scale = float(ref.abs().max())
        err = (got - ref).abs()
        if keep is not None:
            err = err[..., keep] if err.shape[-1] == keep.shape[0] else err
        err = err.flatten()
        sub = err[torch.randperm(err.numel(), generator=torch.Generator().manual_seed(0))[:1_000_000]] if err.numel() > 1_000_000 else err
        return float(err.median()) / scale, float(torch.quantile(sub, 0.99)) / scale, float(torch.quantile(sub, 0.999)) / scale, float(err.max()) / scale

    names = list(wl.params_cpu)
    ref2 = {k: torch.zeros_like(v, dtype=torch.float64) for k, v in wl.params_cpu.items()}
    worst1 = {k: (0.0, 0.0, 0.0, 0.0) for k in names}; worst1_clean = {k: 0.0 for k in names}; worst32_clean = {k: 0.0 for k in names}; worst2 = {k: (0.0, 0.0, 0.0, 0.0) for k in names}
    n_flip_px = n_set_aside = 0
    worst32 = {k: (0.0, 0.0, 0.0, 0.0) for k in names}; worst_h32 = {k: (0.0, 0.0, 0.0, 0.0) for k in names}; worst_ratio = {k: 0.0 for k in names}; diag = []
    faces_np = wl.faces.numpy()
    for b in range(B):
        got_b = frame_grads[b] if frame_grads is not None else grads
        po = {k: v.double().requires_grad_() for k, v in wl.params_cpu.items()}
        fr = {k: (v.double() if v.dtype == torch.float32 else v) for k, v in wl.oracle_frame(b).items()}
        o_rgb, o_mask, aux = og.render_path(po, fr, wl.faces, wl.w25.double(), img)
        # (c1) same upstream gradient
        aux["img"].backward(gradient=dimg[b].cpu().double(), retain_graph=True)
        g1 = {k: po[k].grad.clone() for k in names}
        for k in names:
            po[k].grad = None
        # the yardstick: the fp32 build of the ORACLE, same inputs, same image gradient -- what fp32 arithmetic alone does to these gradients
        p32 = {k: v.clone().requires_grad_() for k, v in wl.params_cpu.items()}
        _, _, aux32 = og.render_path(p32, wl.oracle_frame(b), wl.faces, wl.w25, img)
        aux32["img"].backward(gradient=dimg[b].cpu())
        g32 = {k: p32[k].grad.double() for k in names}
        f64 = orast.forward(aux["cam"], aux["xyz"].detach().numpy(), aux["cov6"].detach().numpy(), aux["feat"].detach().numpy(), np.ones(P), dtype=np.float64, margin=True)
        # pixels where a branch MAY differ between implementations (margin below ~100 ulp; the oracle alone decides this) or DID differ
        flip = (f64["margin"] < MARGIN) | (np.abs(image[b].cpu().numpy().astype(np.float64) - f64["color"]).max(axis=0) > IMG_TOL) | (e["n_contrib"][b] != f64["n_contrib"])
        ys, xs = np.nonzero(flip)
        n_flip_px += len(ys)
        # the Gaussians (= faces) that reach such a pixel with alpha >= 1/255 -- the entries of its tile list whose blend weight or
        # whose transmittance behind the flip the pixel's gradient carries -- and the vertices of those faces
        bad_face = np.zeros(P, bool)
        T_ = (img // 16)
        for y_, x_ in zip(ys.tolist(), xs.tolist()):
            t_ = (y_ // 16) * T_ + (x_ // 16)
            lst = f64["point_list"][f64["ranges"][t_, 0]:f64["ranges"][t_, 1]].astype(np.int64)
            lst = lst[:int(max(f64["n_contrib"][y_, x_], e["n_contrib"][b][y_, x_])) + 2]     # entries behind the last contributor of either side carry nothing
            dx, dy = f64["xy"][lst, 0] - x_, f64["xy"][lst, 1] - y_
            co = f64["conic_opacity"][lst]
            power = -0.5 * (co[:, 0] * dx * dx + co[:, 2] * dy * dy) - co[:, 1] * dx * dy
            bad_face[lst[(power <= 0) & (co[:, 3] * np.exp(power) >= 0.5 / 255.0)]] = True      # (half the threshold: the entries AT the threshold are the flips)
        bad_vert = np.zeros(wl.params_cpu["vertices"].shape[1], bool)
        bad_vert[faces_np[bad_face].reshape(-1)] = True
        n_set_aside += int(bad_face.sum())
        for k in names:
            got = got_b[k].cpu().double()
            keep = torch.from_numpy(~(bad_vert if k == "vertices" else bad_face))
            st = grad_stats(got, g1[k])
            worst1[k] = tuple(max(a, c) for a, c in zip(worst1[k], st))
            clean = float((got - g1[k]).abs()[:, keep].max()) / float(g1[k].abs().max())
            worst1_clean[k] = max(worst1_clean[k], clean)
            worst32_clean[k] = max(worst32_clean[k], float((g32[k] - g1[k]).abs()[:, keep].max()) / float(g1[k].abs().max()))
            st32 = grad_stats(g32[k], g1[k])
            worst32[k] = tuple(max(a, c) for a, c in zip(worst32[k], st32))
            sth = grad_stats(got, g32[k])
            worst_h32[k] = tuple(max(a, c) for a, c in zip(worst_h32[k], sth))
            # where the HIP gradient is furthest from float64: what the fp32 oracle does at that very element
            e_h, e_o = (got - g1[k]).abs(), (g32[k] - g1[k]).abs()
            j = int(e_h.argmax()); ch, idx = divmod(j, e_h.shape[1])
            diag.append(f"    frame {b} d{k}[{ch},{idx}]: fp64 {float(g1[k].flatten()[j]):+.4e}  HIP {float(got.flatten()[j]):+.4e}  fp32 oracle {float(g32[k].flatten()[j]):+.4e}"
                        f"   (max|g| {float(g1[k].abs().max()):.3e}; set aside: {bool(~keep[idx])})")
            ratio = float((e_h.flatten().topk(min(2000, e_h.numel())).values.sum()) / max(float(e_o.flatten().topk(min(2000, e_o.numel())).values.sum()), 1e-300))
            worst_ratio[k] = max(worst_ratio[k], ratio)
        # (c2) the whole step with the oracle's own loss
        d = wl.frames[b]
        l1, l2 = og.l1_losses(og.unpack(o_rgb, o_mask, fr["bgcolor"]), o_mask, d["gt_rgb"].cpu().double()[None], d["gt_mask"].cpu().double()[None])
        (l1 + 5.0 * l2).backward()
        assert abs(float(l1.detach()) - float(lr[b])) <= 1e-5 and abs(float(l2.detach()) - float(lm[b])) <= 1e-5, (b, float(l1.detach()), float(lr[b]), float(l2.detach()), float(lm[b]))
        for k in names:
            ref2[k] += po[k].grad
            st = grad_stats(got_b[k].cpu().double(), po[k].grad)
            worst2[k] = tuple(max(a, c) for a, c in zip(worst2[k], st))
    with capsys.disabled():
        print(f"[metric workload, B={B}] pixels where HIP and the fp64 oracle took different branches: {n_flip_px} of {B * img * img}; Gaussians set aside: {n_set_aside} of {B * P}")
        for k in names:
            st = grad_stats(grads[k].cpu().double(), ref2[k])
            print(f"[metric workload, B={B}] d{k}: |err|/max|g|  same image gradient, worst frame: median {worst1[k][0]:.1e} q99 {worst1[k][1]:.1e} q99.9 {worst1[k][2]:.1e} max {worst1[k][3]:.1e}"
                  f"  max WITHOUT the set-aside Gaussians {worst1_clean[k]:.1e} (fp32 oracle, same elements: {worst32_clean[k]:.1e})   |  whole step incl. L1 sign(), worst frame: median {worst2[k][0]:.1e} q99 {worst2[k][1]:.1e} q99.9 {worst2[k][2]:.1e} max {worst2[k][3]:.1e}"
                  f"  sum of frames: q99.9 {st[2]:.1e} max {st[3]:.1e}")
            print(f"[metric workload, B={B}] d{k}: the fp32 ORACLE against the float64 one, same image gradient, worst frame: median {worst32[k][0]:.1e} q99 {worst32[k][1]:.1e} q99.9 {worst32[k][2]:.1e} max {worst32[k][3]:.1e}"
                  f"   sum of the 2000 largest errors, HIP / fp32 oracle: {worst_ratio[k]:.2f}"
                  f"   |  HIP against the fp32 ORACLE: median {worst_h32[k][0]:.1e} q99 {worst_h32[k][1]:.1e} q99.9 {worst_h32[k][2]:.1e} max {worst_h32[k][3]:.1e}")
        print("\n".join(diag[:16]))
    for k in names:
        # (c1) against the float64 oracle with the same image gradient.  The yardstick is the fp32 build of the oracle itself: HIP, the
        # fp32 oracle and the float64 oracle are three realisations whose tails are mutually alike (measured: same quantiles, and at most
        # of the worst elements HIP and the fp32 oracle agree with EACH OTHER to four digits) -- elements whose value hangs on a
        # threshold deep in a tile list or on a near-singular 2D covariance.  Bulk quantiles <= 3x the yardstick's, the 2 000 largest
        # errors together <= 5x the yardstick's (measured <= 2.6x), the single worst element <= 3x the measured one.
        for q in range(3):
            assert worst1[k][q] <= 3.0 * worst32[k][q] + 1e-9, (k, q, worst1[k], worst32[k])
        assert worst_ratio[k] <= 5.0, (k, worst_ratio[k])
        # the single worst element, FLIP-AWARE (round 6): away from the Gaussians that blend into a pixel whose branch margin is below MARGIN
        # (or that did deviate), every element agrees with float64 to 1e-3 of the largest gradient (measured 5e-6 .. 6e-4) and to twice
        # what the fp32 build of the ORACLE ITSELF misses float64 by on the same elements (measured: HIP is the closer of the two: what is
        # left is the conditioning of the projected covariance, which the reference's formulas in fp32 share);
        # the elements set aside are threshold flips and only bounded by what a flip can carry
        assert worst1_clean[k] <= 1e-3 and worst1_clean[k] <= 2.0 * worst32_clean[k] + 1e-5, (k, worst1_clean[k], worst32_clean[k])
        assert worst1[k][3] <= 0.2, (k, worst1[k])
        # like for like (fp32 against fp32), same image gradient
        assert all(a <= c for a, c in zip(worst_h32[k], BOUND_VS_FP32[k])), (k, worst_h32[k])
        # (c2): <= 3x the measured quantiles of the whole step
        assert all(a <= c for a, c in zip(worst2[k], BOUND_STEP[k])), (k, worst2[k])
        assert all(a <= c for a, c in zip(grad_stats(grads[k].cpu().double(), ref2[k]), BOUND_STEP[k] if B == 1 else BOUND_SUM[k])), (k, grad_stats(grads[k].cpu().double(), ref2[k]))
    # Attribution of the tail to threshold flips, tested rather than asserted in a comment: it holds for the COLOUR gradient (a Gaussian's
    # colour gradient is a plain sum of alpha T dL/dC over its pixels: setting aside the Gaussians that reach a flipped pixel takes the
    # largest error from 1.5e-3 / 5.9e-3 to 8e-6 / 3e-5 of the largest gradient), and it does NOT hold for vertices / so3 / scale: their
    # worst elements are not under any pixel whose image or n_contrib differs, the fp32 oracle misses the float64 value there by the same
    # amount (yardstick above) -- conditioning of the projected covariance, not a branch.
    # ROUND 6: with the pixels flagged by the oracle's branch MARGIN (not only the ones whose image / n_contrib did differ) the attribution holds
    # for all four tensors -- asserted per tensor above.
