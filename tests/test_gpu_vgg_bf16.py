"""The hand-written bf16 MFMA trunk of LPIPS (csrc/vgg_bf16.hip) against torch convolutions on the same
bf16-rounded data, and the whole LPIPS value / image gradient against the fp32 path."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _bf(x):
    return x.to(torch.bfloat16).float()


# small shapes run the 8-row kernel; (2, 200, 136, ...) is large enough for the pipelined 16-row kernel (LDS-DMA staging) with
# ragged right / bottom tiles; (1, 64, 64, 256, 256) takes its split-K form
@pytest.mark.parametrize("B,H,W,Cin,Cout,relu", [(1, 16, 32, 32, 64, True), (2, 24, 40, 64, 128, False), (1, 9, 21, 128, 64, True),
                                                 (2, 200, 136, 64, 128, True), (1, 64, 64, 256, 256, False), (1, 129, 257, 32, 64, True)])
def test_conv3x3_matches_torch(B, H, W, Cin, Cout, relu):
    from gomavatar_amd import _lib
    from gomavatar_amd.lpips import pack_conv_weight, pack_conv_weight_backward
    lib = _lib.load()
    g = torch.Generator().manual_seed(B * 100 + H)
    x = _bf(torch.randn(B, H, W, Cin, generator=g)).cuda()
    w = _bf(torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (Cin * 9)) ** 0.5).cuda()
    b = torch.randn(Cout, generator=g).cuda()
    ref = F.conv2d(x.permute(0, 3, 1, 2), w, b, padding=1)
    ref = (F.relu(ref) if relu else ref).permute(0, 2, 3, 1)
    out = torch.empty(B, H, W, Cout, dtype=torch.bfloat16, device="cuda")
    xb, wp = x.to(torch.bfloat16).contiguous(), pack_conv_weight(w)
    sp = lib.gom_conv3x3_splits(B, H, W, Cin, Cout)
    ws = torch.empty(sp * B * H * W * Cout, device="cuda") if sp > 1 else None
    _lib.check(lib.gom_conv3x3_bf16_splitk(B, H, W, Cin, Cout, _lib.ptr(xb), _lib.ptr(wp), _lib.ptr(b), 0, _lib.ptr(out), 1 if relu else 0, sp,
                                           _lib.ptr(ws), _lib.stream_ptr()))
    torch.cuda.synchronize()
    err = (out.float() - ref).abs().max() / ref.abs().max()
    assert float(err) < 6e-3, float(err)           # one bf16 rounding of the output
    # backward-data = the same kernel on rotated / transposed weights, with the ReLU mask of the layer below fused
    gy = _bf(torch.randn(B, H, W, Cout, generator=g)).cuda()
    below = torch.randn(B, H, W, Cin, generator=g).cuda()
    xr = x.permute(0, 3, 1, 2).clone().requires_grad_()
    F.conv2d(xr, w, None, padding=1).backward(gy.permute(0, 3, 1, 2))
    ref_dx = xr.grad.permute(0, 2, 3, 1) * (below > 0)
    dx = torch.empty(B, H, W, max(64, Cin), dtype=torch.bfloat16, device="cuda")
    wb = pack_conv_weight_backward(w)
    mask = torch.zeros(B, H, W, max(64, Cin), dtype=torch.bfloat16, device="cuda")
    mask[..., :Cin] = below.to(torch.bfloat16)
    gyb = gy.to(torch.bfloat16).contiguous()
    _lib.check(lib.gom_conv3x3_bf16(B, H, W, Cout, max(64, Cin), _lib.ptr(gyb), _lib.ptr(wb), 0, _lib.ptr(mask), _lib.ptr(dx), 0, _lib.stream_ptr()))
    torch.cuda.synchronize()
    err = (dx.float()[..., :Cin] - ref_dx).abs().max() / ref_dx.abs().max()
    assert float(err) < 6e-3, float(err)


def test_maxpool_forward_backward():
    from gomavatar_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(3)
    x = _bf(torch.randn(2, 8, 12, 64, generator=g)).cuda()
    x[0, :2, :2, :] = -x[0, :2, :2, :].abs()          # one all-negative window: its routed gradient is masked ([x > 0])
    xb = x.to(torch.bfloat16).contiguous()
    y = torch.empty(2, 4, 6, 64, dtype=torch.bfloat16, device="cuda")
    _lib.check(lib.gom_maxpool2x2_bf16(2, 8, 12, 64, _lib.ptr(xb), _lib.ptr(y), _lib.stream_ptr()))
    xr = x.permute(0, 3, 1, 2).clone().requires_grad_()
    ref = F.max_pool2d(xr, 2, 2)
    assert torch.equal(y.float(), ref.detach().permute(0, 2, 3, 1))
    dy = _bf(torch.randn(2, 4, 6, 64, generator=g)).cuda()
    ref.backward(dy.permute(0, 3, 1, 2))
    base = _bf(torch.randn(2, 8, 12, 64, generator=g)).cuda()
    dx = base.to(torch.bfloat16).contiguous()
    dyb = dy.to(torch.bfloat16).contiguous()
    _lib.check(lib.gom_maxpool2x2_backward_bf16(2, 8, 12, 64, _lib.ptr(xb), _lib.ptr(dyb), _lib.ptr(dx), 1, _lib.stream_ptr()))
    want = _bf(base + xr.grad.permute(0, 2, 3, 1) * (x > 0))
    assert torch.allclose(dx.float(), want, atol=2e-2, rtol=1e-2)


def test_lpips_matrix_core_matches_fp32_path():
    from gomavatar_amd.lpips import LPIPS, LPIPSMatrixCore, lpips_loss
    g = torch.Generator().manual_seed(11)
    pred = torch.rand(2, 64, 96, 3, generator=g).cuda()
    gt = (pred.cpu() + 0.2 * torch.randn(2, 64, 96, 3, generator=g)).clamp(0, 1).cuda()
    ref_model = LPIPS(trunk_seed=5)
    p = pred.clone().requires_grad_()
    ref = lpips_loss(ref_model, p, gt)
    ref.backward()
    mc = LPIPSMatrixCore(trunk_seed=5, precision="bf16")
    val, grad = mc.value_and_grad(pred, gt)
    torch.cuda.synchronize()
    assert abs(float(val) - float(ref.detach())) <= 0.03 * float(ref.detach()), (float(val), float(ref.detach()))      # bf16 activations
    a, b = grad.flatten().double(), p.grad.flatten().double()
    cos = float((a @ b) / (a.norm() * b.norm()))
    assert cos > 0.98, cos
    assert abs(float(a.norm() / b.norm()) - 1.0) < 0.05
    # autograd wrapper: same value, gradient scaled by the upstream factor
    q = pred.clone().requires_grad_()
    (3.0 * mc.loss(q, gt)).backward()
    assert torch.allclose(q.grad, 3.0 * grad, rtol=1e-6, atol=0)
    v2, none = mc.value_and_grad(pred, gt, want_grad=False)
    assert none is None and abs(float(v2) - float(val)) <= 2e-6 * float(val)      # (value-only: the head FORWARD kernels; with a gradient: out of the backward ones, another summation order)


@pytest.mark.parametrize("shape", [(2, 64, 96), (1, 256, 256)])
def test_lpips_bf16x3_trunk_has_the_precision_of_the_fp32_path(shape):
    """GOM_LPIPS_PRECISION_BF16X3: activations, gradients and weights as hi + lo bf16 planes, three MFMA passes per product.  Against the
    fp32 library convolutions (the reference's precision, utils/lpips/pretrained_networks.py:96-134): value <= 1e-5 relative (the plain
    bf16 trunk: 3 %).  The image gradient passes thirteen ReLU masks and four max-pool argmaxes, each a branch on an activation: two fp32
    implementations differ at isolated pixels by whole gradient contributions.  Measured against the FLOAT64 trunk (printed, with the
    fp32 library's own distance beside it): bf16x3 3e-3 .. 6e-3 relative L2 where the plain bf16 trunk is 0.2 and the fp32 library
    2e-6 .. 4e-4 -- exactly what storing the activations with 16-17 mantissa bits costs when everything else is exact
    (scripts/lpips_storage_precision.py, CPU float64: 1.3e-3 at 17 bits, 3.0e-3 at 16).  Bounds: 3x the measured values."""
    from gomavatar_amd.lpips import LPIPS, LPIPSMatrixCore, lpips_loss
    B, H, W = shape
    g = torch.Generator().manual_seed(11)
    pred = torch.rand(B, H, W, 3, generator=g).cuda()
    gt = (pred.cpu() + 0.2 * torch.randn(B, H, W, 3, generator=g)).clamp(0, 1).cuda()

    def library(dtype):
        m = LPIPS(trunk_seed=5, trunk_dtype=dtype)
        m.lins = [l.to(dtype) for l in m.lins] if dtype == torch.float64 else m.lins
        p = pred.to(dtype).clone().requires_grad_()
        if dtype == torch.float64:   # the fp32 head kernels are not the subject here: plain torch head in float64
            m.shift, m.scale = m.shift.double(), m.scale.double()
            f0 = m.features(2 * p.permute(0, 3, 1, 2) - 1)
            with torch.no_grad():
                f1 = m.features(2 * gt.double().permute(0, 3, 1, 2) - 1)
            val = 0
            for k in range(5):
                n0 = f0[k] / (f0[k].pow(2).sum(1, keepdim=True).sqrt() + 1e-10)
                n1 = f1[k] / (f1[k].pow(2).sum(1, keepdim=True).sqrt() + 1e-10)
                val = val + ((n0 - n1) ** 2 * m.lins[k].view(1, -1, 1, 1)).sum(1).mean((1, 2))
            val = val.mean()
        else:
            val = lpips_loss(m, p, gt)
        val.backward()
        return float(val.detach()), p.grad.double()
    v64, g64 = library(torch.float64)
    v32, g32 = library(torch.float32)
    mc = LPIPSMatrixCore(trunk_seed=5, precision="bf16x3")
    val, grad = mc.value_and_grad(pred, gt)
    torch.cuda.synchronize()

    def stats(a):
        e = (a.double() - g64).abs().flatten()
        return float(e.norm() / g64.norm()), float(torch.quantile(e[:1_000_000], 0.999)) / float(g64.abs().max()), float(e.max()) / float(g64.abs().max())
    sx, s32 = stats(grad), stats(g32)
    rel, rel32 = abs(float(val) - v64) / v64, abs(v32 - v64) / v64
    print(f"\n[bf16x3 {shape}] LPIPS value vs float64: bf16x3 {rel:.2e}, fp32 library {rel32:.2e};  image gradient vs float64 (rel L2, q99.9 / max, max / max): "
          f"bf16x3 {sx[0]:.2e} {sx[1]:.2e} {sx[2]:.2e}   fp32 library {s32[0]:.2e} {s32[1]:.2e} {s32[2]:.2e}")
    assert abs(float(val) - v32) <= 1e-5 * v32 and rel <= 1e-5, (float(val), v32, v64)
    assert sx[0] <= 1.8e-2 and sx[1] <= 4e-2 and sx[2] <= 0.2, (sx, s32)
    plain = LPIPSMatrixCore(trunk_seed=5, precision="bf16")                                  # the one-pass bf16 trunk on the same inputs
    vp, gp = plain.value_and_grad(pred, gt)
    sp = stats(gp)
    print(f"[bf16   {shape}] value vs float64 {abs(float(vp) - v64) / v64:.2e}; gradient rel L2 {sp[0]:.2e}")
    assert sx[0] <= sp[0] / 20.0 and rel <= abs(float(vp) - v64) / v64 / 100.0
    val2, grad2 = mc.value_and_grad(pred, gt)
    assert float(val2) == float(val) and torch.equal(grad, grad2)          # reproducible


@pytest.mark.parametrize("precision", ["bf16x3", "bf16"])
def test_target_features_on_a_second_stream_give_the_same_lpips(precision):
    """LPIPSMatrixCore.prefetch_target: the target image's half of the trunk on another stream before the prediction exists
    (gom_lpips_vgg_target_features + GOM_LPIPS_TARGET_READY).  Same arithmetic per image; the batch of one image set picks other split-K
    counts than the batch of two, i.e. fp32 sums in another order: value to 2e-6 relative (measured 8e-8); the gradient moves by 4.6e-3 of
    its norm -- exactly what ANY change of the split-K summation order does to it (scripts/lpips_split_sensitivity.py: the batched path
    with the round-3 split rule against today's: 4.5e-3; last-bit differences of activations flip ReLU masks and pool argmaxes at isolated
    pixels), inside the 3e-3 .. 6e-3 this trunk keeps to the float64 one (test above).  Bound 1.4e-2.  Bitwise equal from prefetch to
    prefetch over 12 rounds with other work on both streams in between.  A prefetch for ANOTHER target tensor is ignored (the batched path runs)."""
    from gomavatar_amd.lpips import LPIPSMatrixCore
    B, H, W = 1, 256, 256
    g = torch.Generator().manual_seed(21)
    pred = torch.rand(B, H, W, 3, generator=g).cuda()
    gts = [(pred.cpu() + 0.2 * torch.randn(B, H, W, 3, generator=g)).clamp(0, 1).cuda() for _ in range(2)]
    mc = LPIPSMatrixCore(trunk_seed=5, precision=precision)
    base = [mc.value_and_grad(pred, gt) for gt in gts]
    junk = torch.empty(1 << 22, device="cuda")
    first = {}
    for rnd in range(12):
        k = rnd % 2
        mc.prefetch_target(gts[k])
        junk.normal_()                                     # (the frame's forward would be enqueued here)
        val, grad = mc.value_and_grad(pred, gts[k])
        assert mc._target is None
        if k not in first:
            first[k] = (float(val), grad.clone())
            assert abs(float(val) - float(base[k][0])) <= (2e-6 if precision == "bf16x3" else 2e-4) * float(base[k][0]), (float(val), float(base[k][0]))   # (bf16: 1.7e-5)
            tol = 1.4e-2 if precision == "bf16x3" else 0.3   # (one-pass bf16: a rounding of an activation is 2^-9; its distance to float64 is 0.2)
            assert float((grad - base[k][1]).norm()) <= tol * float(base[k][1].norm()), float((grad - base[k][1]).norm() / base[k][1].norm())
        else:
            assert float(val) == first[k][0] and torch.equal(grad, first[k][1]), rnd
    mc.prefetch_target(gts[0])
    val, grad = mc.value_and_grad(pred, gts[1])            # not the prefetched tensor: both images walk the trunk together
    assert float(val) == float(base[1][0]) and torch.equal(grad, base[1][1])


@pytest.mark.parametrize("precision", ["bf16x3", "bf16"])
def test_first_layer_from_the_image_is_bitwise_the_two_kernel_form(precision):
    """k_conv1_1_image builds conv1_1's im2col rows in registers; GOM_LPIPS_FIRST_LAYER_FUSED=0 writes them out and runs the 1 x 1 convolution
    over them (round 4's first form).  Same values through the same MFMA sequence: LPIPS value and gradient equal bit for bit, with and
    without the target prefetch (one image set alone / both as a batch)."""
    import os
    from gomavatar_amd.lpips import LPIPSMatrixCore
    g = torch.Generator().manual_seed(31)
    pred = torch.rand(2, 64, 96, 3, generator=g).cuda()
    gt = (pred.cpu() + 0.2 * torch.randn(2, 64, 96, 3, generator=g)).clamp(0, 1).cuda()
    mc = LPIPSMatrixCore(trunk_seed=7, precision=precision)
    res = {}
    try:
        for fused in ("1", "0"):
            os.environ["GOM_LPIPS_FIRST_LAYER_FUSED"] = fused
            v, gr = mc.value_and_grad(pred, gt)
            mc.prefetch_target(gt)
            v2, gr2 = mc.value_and_grad(pred, gt)
            res[fused] = (float(v), gr.clone(), float(v2), gr2.clone())
    finally:
        os.environ.pop("GOM_LPIPS_FIRST_LAYER_FUSED", None)
    assert res["1"][0] == res["0"][0] and torch.equal(res["1"][1], res["0"][1])
    assert res["1"][2] == res["0"][2] and torch.equal(res["1"][3], res["0"][3])
    assert res["1"][0] > 0 and float(res["1"][1].abs().max()) > 0
    # with a gradient wanted the taps' values come out of the head BACKWARD kernels (one read of the feature maps); without, out of the forward ones:
    # the same per-pixel terms summed in another order
    v_only, none = mc.value_and_grad(pred, gt, want_grad=False)
    assert none is None and abs(float(v_only) - res["1"][0]) <= 2e-6 * res["1"][0], (float(v_only), res["1"][0])


@pytest.mark.parametrize("precision", ["bf16x3", "bf16"])
@pytest.mark.parametrize("shape", [(2, 64, 96), (1, 48, 80), (1, 256, 256)])
def test_first_layer_backward_in_one_kernel_matches_the_two_kernel_form(precision, shape):
    """k_conv1_1_bwd_image: conv1_1's backward-data pass (1 x 1 convolution 64 -> 32 im2col columns) and the col2im gather in one kernel, the
    im2col gradient rows kept in LDS as fp32.  GOM_LPIPS_FIRST_LAYER_BWD_FUSED=0 is the two-kernel form, which rounds those rows to the
    storage planes in between (two bf16 planes = 16 mantissa bits; one plane = 8 in the plain mode): same fragments, same MFMA order, so the
    image gradient differs by that rounding alone -- relative L2 <= 2e-5 (bf16x3) / 6e-3 (bf16) -- and the value not at all."""
    import os
    from gomavatar_amd.lpips import LPIPSMatrixCore
    B, H, W = shape
    g = torch.Generator().manual_seed(37)
    pred = torch.rand(B, H, W, 3, generator=g).cuda()
    gt = (pred.cpu() + 0.2 * torch.randn(B, H, W, 3, generator=g)).clamp(0, 1).cuda()
    mc = LPIPSMatrixCore(trunk_seed=7, precision=precision)
    res = {}
    try:
        for fused in ("1", "0"):
            os.environ["GOM_LPIPS_FIRST_LAYER_BWD_FUSED"] = fused
            v, gr = mc.value_and_grad(pred, gt)
            res[fused] = (float(v), gr.clone())
    finally:
        os.environ.pop("GOM_LPIPS_FIRST_LAYER_BWD_FUSED", None)
    assert res["1"][0] == res["0"][0] and res["1"][0] > 0
    a, b = res["1"][1].double(), res["0"][1].double()
    rel = float((a - b).norm() / b.norm())
    worst = float((a - b).abs().max() / b.abs().max())
    print(f"\n[conv1_1 backward in one kernel, {precision}, {shape}] gradient rel L2 {rel:.2e}, worst element / max |g| {worst:.2e}")
    assert float(b.norm()) > 0 and rel <= (2e-5 if precision == "bf16x3" else 6e-3), rel
    v2, gr2 = mc.value_and_grad(pred, gt)
    assert float(v2) == res["1"][0] and torch.equal(gr2, res["1"][1])   # reproducible


@pytest.mark.parametrize("shape", [(2, 64, 96), (1, 512, 512)])
def test_pooling_inside_the_convolutions_is_bitwise_the_pool_kernel(shape):
    """The four max-pools of the trunk ride in the epilogue of the convolution in front of them (conv_store: a lane's two rows and its neighbour lane;
    k_splitk_epilogue<true>: a thread per window) -- the maximum of the STORED values, i.e. what k_maxpool2_fwd (GOM_LPIPS_FUSED_POOL=0) reads back.
    LPIPS value and gradient equal bit for bit; (2, 64, 96) takes the split-K epilogue everywhere, 512^2 the plain epilogue on the first two pools."""
    import os
    from gomavatar_amd.lpips import LPIPSMatrixCore
    B, H, W = shape
    g = torch.Generator().manual_seed(41)
    pred = torch.rand(B, H, W, 3, generator=g).cuda()
    gt = (pred.cpu() + 0.2 * torch.randn(B, H, W, 3, generator=g)).clamp(0, 1).cuda()
    for precision in ("bf16x3", "bf16"):
        mc = LPIPSMatrixCore(trunk_seed=3, precision=precision)
        res = {}
        try:
            for fused in ("1", "0"):
                os.environ["GOM_LPIPS_FUSED_POOL"] = fused
                v, gr = mc.value_and_grad(pred, gt)
                mc.prefetch_target(gt)
                v2, gr2 = mc.value_and_grad(pred, gt)
                res[fused] = (float(v), gr.clone(), float(v2), gr2.clone())
        finally:
            os.environ.pop("GOM_LPIPS_FUSED_POOL", None)
        assert res["1"][0] == res["0"][0] and torch.equal(res["1"][1], res["0"][1]), precision
        assert res["1"][2] == res["0"][2] and torch.equal(res["1"][3], res["0"][3]), precision


def test_pipelined_conv_is_race_free_over_many_launches():
    """The 16-row kernel orders its LDS-DMA staging by counted vmcnt waits and one barrier per stage: a misplaced wait would show
    up as rare, timing-dependent wrong tiles.  300 back-to-back launches (other launches in between to perturb timing) must
    reproduce the first result bit for bit."""
    from gomavatar_amd import _lib
    from gomavatar_amd.lpips import pack_conv_weight
    lib = _lib.load()
    g = torch.Generator().manual_seed(5)
    for (B, H, W, Cin, Cout) in ((1, 256, 256, 128, 128), (2, 200, 136, 64, 128), (1, 64, 64, 256, 256)):
        x = torch.randn(B, H, W, Cin, generator=g).to(torch.bfloat16).cuda()
        w = pack_conv_weight(_bf(torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (Cin * 9)) ** 0.5).cuda())
        b = torch.randn(Cout, generator=g).cuda()
        sp = lib.gom_conv3x3_splits(B, H, W, Cin, Cout)
        ws = torch.empty(sp * B * H * W * Cout, device="cuda") if sp > 1 else None
        outs = [torch.empty(B, H, W, Cout, dtype=torch.bfloat16, device="cuda") for _ in range(11)]
        junk = torch.empty(1 << 22, device="cuda")
        def run(o):
            _lib.check(lib.gom_conv3x3_bf16_splitk(B, H, W, Cin, Cout, _lib.ptr(x), _lib.ptr(w), _lib.ptr(b), 0, _lib.ptr(o), 1, sp, _lib.ptr(ws),
                                                   _lib.stream_ptr()))
        run(outs[0])
        for rnd in range(30):           # 10 launches back to back (a bandwidth-heavy neighbour now and then), every result checked
            for k in range(1, 11):
                if (rnd + k) % 4 == 0:
                    junk.normal_()
                run(outs[k])
            for k in range(1, 11):
                assert torch.equal(outs[0], outs[k]), (B, H, W, Cin, Cout, rnd, k)
                outs[k].zero_()
