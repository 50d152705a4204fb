"""Parity of the HIP splat rasterizer (through the C ABI) against the CPU
oracle.  Bar (BASELINE.json north_star): tile binning / indices bit-exact,
images <= 1e-4 per-pixel L1, gradients to fp32 round-off."""
import numpy as np
import pytest
import torch

from oracle import raster as orast
from helpers import small_scene, body_scene
from oracle import geometry as og

pytestmark = pytest.mark.gpu

IMG_TOL = 1e-4             # per-pixel L1, stated by north_star
# The algorithm is discontinuous by construction (App. A.3: `alpha < 1/255 -> skip`, `T < 1e-4 -> stop`): an exp()
# that differs by one ulp between glibc and v_exp_f32 flips such a test for a handful of (pixel, entry) pairs and
# moves that pixel by up to alpha*T ~ 4e-3.  Any two implementations (CUDA included) disagree there, so parity is:
# every pixel within IMG_TOL except a vanishing fraction, mean error far below IMG_TOL, contributor counts equal
# except at the same rare flips.
FLIP_RATE = 2e-4
MEAN_TOL = 2e-6


def _flips_allowed(n):
    """Threshold flips (alpha >= 1/255, alpha capped at 0.99, T < 1e-4 under a 1-ulp different exp) are a rate; an image of a
    few thousand pixels is allowed the one flipped pixel a soak over 3 600 random scenes turned up in 0.3 % of them."""
    return max(1, int(FLIP_RATE * n))


def assert_image_parity(img, ref):
    err = np.abs(img - ref)
    assert err.mean() <= MEAN_TOL, err.mean()
    bad = err.max(axis=0) > IMG_TOL
    assert np.count_nonzero(bad) <= _flips_allowed(bad.size), (np.count_nonzero(bad), bad.size, err.max())
    assert err.max() <= 1.5e-2   # a flip is bounded by alpha*T*|colour| right at the 1/255 and 0.99 thresholds


MARGIN = 2e-5              # relative distance of a branch-deciding quantity to its threshold (oracle/raster_oracle.c A.3 "MARGIN")
FLIP_LOG = []              # (pixels, fragile, deviating, deviating against the oracle's HIP formulation) per comparison: printed by the last test


def assert_flips_are_threshold_margins(img, n_contrib, final_T, f, cam_inputs=None, max_fragile=3e-3):
    """The flip ARGUMENT as a test (round-5 review).  The oracle reports, per pixel, how close any branch-deciding quantity came to its
    threshold (`margin`: alpha against 1/255, T (1 - alpha) against 1e-4, the exponent's sign; relative, in units of the quantity's own
    fp32 uncertainty) and what one fp32 rounding of every exponent does to the pixel (`roundoff`: large only under needle-shaped conics far
    from their centre, where the exponent is a difference of terms ~1e4 -- the reference's own expression in fp32 misses float64 there by as
    much as ours).  Held with ZERO exceptions:
      * margin >= MARGIN (~100 ulp) and roundoff <= 2e-5 ("solid", > 98 % of the pixels): within north_star's 1e-4 per pixel -- in fact within
        5e-5 --, the oracle's n_contrib and final T;
      * margin >= MARGIN, roundoff > 2e-5 (ill-conditioned in fp32): the oracle's n_contrib, and within 2e-5 + 3 x the pixel's own estimate;
      * margin < MARGIN: a threshold flip may happen (any two fp32 implementations may differ there): bounded by what one flip carries.
    So every pixel beyond 1e-4 IS a threshold flip or is bounded by its own conditioning; the flagged pixels are a vanishing fraction."""
    firm = f["margin"] >= MARGIN
    ill = firm & (f["roundoff"] > 2e-5)
    solid = firm & ~ill
    n_frag, n_ill = int(np.count_nonzero(~firm)), int(np.count_nonzero(ill))
    assert n_frag <= max_fragile * firm.size + 2 and n_ill <= 3e-2 * firm.size, (n_frag, n_ill, firm.size)
    err = np.abs(img - f["color"]).max(axis=0)
    assert np.array_equal(n_contrib[firm], f["n_contrib"][firm]), (int(np.count_nonzero((n_contrib != f["n_contrib"]) & firm)), f["margin"][(n_contrib != f["n_contrib"]) & firm][:8].tolist())
    bad = solid & (err > 5e-5)
    assert not np.any(bad), (int(np.count_nonzero(bad)), float(err[solid].max()), np.argwhere(bad)[:4].tolist(), f["margin"][bad][:8].tolist(), f["roundoff"][bad][:8].tolist())
    if n_ill:
        over = err[ill] - (2e-5 + 3.0 * f["roundoff"][ill])
        assert float(over.max()) <= 0.0, (float(over.max()), float((err[ill] / f["roundoff"][ill]).max()))
    assert float(np.abs(final_T - f["final_T"])[solid].max()) <= 5e-5
    dev = (err > IMG_TOL) | (n_contrib != f["n_contrib"])
    n_dev_hip = None
    if cam_inputs is not None:      # the same comparison against the oracle built with the HIP path's own FORMULATION of alpha (reported)
        fh = orast.forward(*cam_inputs, form="hip")
        n_dev_hip = int(np.count_nonzero((np.abs(img - fh["color"]).max(axis=0) > IMG_TOL) | (n_contrib != fh["n_contrib"])))
        # ZERO pixels beyond 1e-4 and ZERO n_contrib mismatches once the oracle evaluates alpha the way the HIP path does (pre-scaled conic, fma chain,
        # exp2): what separates the HIP path from the reference's expression is the FORMULATION's rounding, not an algorithmic difference.  (Measured on
        # MI355X: 0 of 2.5 M pixels over every comparison of this file, where the reference formulation has 7.  glibc's exp2f / log2f against v_exp_f32 /
        # v_log_f32 could still flip a pixel whose margin is within an ulp: deterministic for these seeded scenes on this hardware, and zero.)
        assert n_dev_hip == 0, n_dev_hip
    FLIP_LOG.append((firm.size, n_frag, int(np.count_nonzero(dev)), n_dev_hip, n_ill, int(np.count_nonzero(dev & ill)), float(err[solid].max())))


def _compare_forward(cam, means, cov6, colors, op, sort_cap=None, max_fragile=3e-3):
    """Both ways of ordering the tile lists -- merge sort per tile, depth ranking + bitmap pass (csrc/raster_rank.hip) -- against the
    oracle and against each other (bitwise)."""
    from gpu_util import hip_forward, export_state, assert_binning_bit_exact
    from gomavatar_amd import _lib, rasterizer as R
    f = orast.forward(cam, means, cov6, colors, op, margin=True)
    prev = None
    for mode in (_lib.SORT_TILE_MERGE, _lib.SORT_DEPTH_RANK):
        st = R.RasterState()
        st.set_option(_lib.OPT_SORT_MODE, mode)
        if sort_cap is not None:
            st.set_option(_lib.OPT_SORT_CAP, sort_cap)
        out, radii, st, _ = hip_forward(cam, means, cov6, colors, op, state=st)
        e = export_state(st, means.shape[0], cam["H"], cam["W"])
        assert_binning_bit_exact(e, f)
        if prev is not None:
            assert torch.equal(out, prev[0]) and torch.equal(radii, prev[1])
            for k in ("keys", "point_list", "n_contrib", "final_T"):
                np.testing.assert_array_equal(e[k], prev[2][k])
        prev = (out, radii, e)
    np.testing.assert_array_equal(radii.cpu().numpy(), f["radii"])
    img = out.cpu().numpy()
    assert_image_parity(img, f["color"])
    same = e["n_contrib"] == f["n_contrib"]
    assert np.count_nonzero(~same) <= _flips_allowed(same.size), (np.count_nonzero(~same), same.size)
    # same last contributor: T agrees, except where an entry in the middle of the list flipped (T moves by <= alpha * T there)
    dT = np.abs(e["final_T"][same] - f["final_T"][same])
    assert np.count_nonzero(dT > 1e-5) <= _flips_allowed(same.size), (np.count_nonzero(dT > 1e-5), float(dT.max()))
    assert dT.size == 0 or float(dT.max()) <= 1.5e-2
    assert_flips_are_threshold_margins(img, e["n_contrib"], e["final_T"], f, (cam, means, cov6, colors, op), max_fragile)
    return img, f, e


@pytest.mark.parametrize("C", [3, 4])
@pytest.mark.parametrize("shape", [(64, 80), (33, 47), (128, 128)])
def test_forward_random_scene(C, shape):
    cam, means, cov6, colors, op = small_scene(seed=11 + C, P=1500, H=shape[0], W=shape[1], opacity=(0.2, 1.0), C=C)
    cam["bg"] = np.array([0.1, 0.7, 0.3, 0.0], np.float32)
    _compare_forward(cam, means, cov6, colors, op)


def test_forward_dense_tiles_lds_and_fallback_sort_agree():
    # ~2000 gaussians land in a handful of tiles: exercises long lists; then force the global-memory sort
    cam, means, cov6, colors, op = small_scene(seed=21, P=6000, H=64, W=64, spread=0.12, scale=0.02, opacity=1.0)
    img_a, f, e = _compare_forward(cam, means, cov6, colors, op)
    assert (np.diff(e["tile_base"].astype(np.int64))).max() > 1024
    for cap in (64, 512):   # lists longer than the chunk: chunk sorts + global merge passes
        img_b, _, _ = _compare_forward(cam, means, cov6, colors, op, sort_cap=cap)
        np.testing.assert_array_equal(img_a, img_b)


def test_forward_equal_depth_ties_keep_gaussian_order():
    cam, means, cov6, colors, op = small_scene(seed=22, P=900, H=48, W=48, opacity=0.6)
    cam["viewmatrix"] = np.eye(4, dtype=np.float32); cam["viewmatrix"][3, 2] = 3.0   # E = [I | (0,0,3)]
    K = np.array([[60.0, 0, 24], [0, 60.0, 24], [0, 0, 1]], np.float32)
    E = np.eye(4, dtype=np.float32); E[2, 3] = 3.0
    cam = og.camera_from_KE(K, E, 48, 48)
    means[:, 2] = np.round(means[:, 2] * 2) / 2        # many exactly equal depths
    _, f, e = _compare_forward(cam, means, cov6, colors, op)
    k = e["keys"]
    assert np.count_nonzero((k[1:] >> np.uint64(32)) == (k[:-1] >> np.uint64(32))) > 50


def test_forward_all_depths_equal_one_bucket_takes_the_chunked_sort():
    """Every Gaussian at the same depth: the depth ranking's bucket map puts all of them into ONE bucket, far longer than a sort
    chunk (-> chunk sorts + global merge passes), and the order inside every tile list is the Gaussian index alone."""
    cam, means, cov6, colors, op = small_scene(seed=25, P=5000, H=64, W=64, opacity=0.5, scale=0.02)
    K = np.array([[60.0, 0, 32], [0, 60.0, 32], [0, 0, 1]], np.float32)
    E = np.eye(4, dtype=np.float32); E[2, 3] = 3.0
    cam = og.camera_from_KE(K, E, 64, 64)
    means[:, 2] = 0.25
    _, f, e = _compare_forward(cam, means, cov6, colors, op)
    assert np.unique(e["keys"] >> np.uint64(32)).size == 1
    tb = e["tile_base"].astype(np.int64)
    for t in np.nonzero(np.diff(tb) > 1)[0][:50]:
        assert np.all(np.diff(e["point_list"][tb[t]:tb[t + 1]].astype(np.int64)) > 0)
    # two depth values, the nearer one on the higher indices
    means[2500:, 2] = 0.2
    _compare_forward(cam, means, cov6, colors, op)


def test_forward_empty_and_culled():
    from gpu_util import hip_forward
    cam, means, cov6, colors, op = small_scene(seed=23, P=64, H=32, W=32)
    cam["bg"] = np.array([0.25, 0.5, 0.75, 0.0], np.float32)
    out, radii, st, _ = hip_forward(cam, means - np.array([0, 0, 10], np.float32), cov6, colors, op)
    assert st.poll() == (0, False)
    assert (radii == 0).all()
    img = out.cpu().numpy()
    for ch in range(3):
        assert np.all(img[ch] == cam["bg"][ch])
    # P = 0
    out, radii, st, _ = hip_forward(cam, np.zeros((0, 3), np.float32), np.zeros((0, 6), np.float32), np.zeros((0, 4), np.float32), np.zeros(0, np.float32))
    assert out.shape == (4, 32, 32) and radii.numel() == 0 and torch.all(out[0] == 0.25)


def test_pair_buffer_overflow_is_loud():
    from gpu_util import hip_forward
    from gomavatar_amd import _lib, rasterizer as R
    cam, means, cov6, colors, op = small_scene(seed=24, P=2000, H=64, W=64)
    st = R.RasterState()
    st.set_option(_lib.OPT_PAIR_CAPACITY, 100)
    out, radii, st, _ = hip_forward(cam, means, cov6, colors, op, state=st)
    D, overflow = st.poll()
    assert overflow and D > 100
    assert torch.isnan(out).all()


def test_record_buffer_overflow_poisons_the_gradients_loudly():
    """GOM_OPT_BWD_MODE 3: the forward's per-(pixel, entry) records live in a buffer of 8 records per unit of pair capacity.  Forty Gaussians that
    cover the whole 64 x 64 image need 164 k records where an 8 192-pair state has room for 65 k: the IMAGE is the usual one (the records are a
    by-product), the overflow is reported (gom_state_poll bit 1) and every gradient is NaN -- loud, not silently short; the same state with
    its capacity back to automatic works again."""
    from gpu_util import hip_forward
    from gomavatar_amd import _lib, rasterizer as R
    if not _lib.has_lab():      # round 5: the records backward (measured slower than the replay) left the product library; the product build REFUSES the mode
        with pytest.raises(RuntimeError):
            R.RasterState().set_option(_lib.OPT_BWD_MODE, 3)
        pytest.skip("records backward: -DGOM_LAB builds only (include/gom_hip_lab.h)")
    cam, means, cov6, colors, op = small_scene(seed=61, P=40, H=64, W=64, opacity=(0.05, 0.2), spread=0.2, scale=0.6, C=4)
    ref, _, _, _ = hip_forward(cam, means, cov6, colors, op)
    st = R.RasterState()
    st.set_option(_lib.OPT_BWD_MODE, 3)
    st.set_option(_lib.OPT_PAIR_CAPACITY, 8192)      # (pairs: eight shards of 1 024, 640 needed; records: 32 shards of 2 048)
    out, radii, st, t = hip_forward(cam, means, cov6, colors, op, requires_grad=True, state=st)
    D, overflow = st.poll()
    assert overflow and 0 < D <= 1024
    assert st.poll_flags() == 2                       # bit 1 = the records, bit 0 (pair buffers, poisoned image) clear
    assert torch.isfinite(out).all() and float((out - ref).abs().max()) < 1e-5
    out.sum().backward()
    assert all(torch.isnan(x.grad).any() for x in t[:3])
    st.set_option(_lib.OPT_PAIR_CAPACITY, 0)
    out2, _, st, t2 = hip_forward(cam, means, cov6, colors, op, requires_grad=True, state=st)
    out2.sum().backward()
    assert not st.poll()[1] and all(torch.isfinite(x.grad).all() for x in t2)
    # ... and they are the replay's gradients to round-off
    o3, _, _, t3 = hip_forward(cam, means, cov6, colors, op, requires_grad=True)
    o3.sum().backward()
    for a, b in zip(t2, t3):
        assert float((a.grad - b.grad).abs().max()) <= 1e-4 * float(b.grad.abs().max()) + 1e-12


def test_overflow_with_long_lists_in_rank_mode_stays_inside_its_buffers():
    """The scan kernel lists one k_tile_rank work item per 2 048 list positions of a tile from counts that keep growing AFTER the pair
    buffer has overflowed: 5 000 Gaussians over all 16 tiles are 48 items for a 19-item buffer (capacity 4 096 pairs).  The store is
    guarded by the buffer's capacity; the overflow is reported, the image poisoned, and the state is intact afterwards (a normal
    frame on the same state is bit-identical to one on a fresh state)."""
    from gpu_util import hip_forward
    from gomavatar_amd import _lib, rasterizer as R
    cam, means, cov6, colors, op = small_scene(seed=5, P=5000, H=64, W=64, spread=0.2, scale=0.6, opacity=0.05)
    st = R.RasterState()
    st.set_option(_lib.OPT_SORT_MODE, 2)
    st.set_option(_lib.OPT_PAIR_CAPACITY, 4096)
    out, radii, st, _ = hip_forward(cam, means, cov6, colors, op, state=st)
    D, overflow = st.poll()
    assert overflow and D > 16 * 2048 * 2, D           # every tile's list is longer than two windows
    assert torch.isnan(out).all()
    st.set_option(_lib.OPT_PAIR_CAPACITY, 0)
    small = small_scene(seed=24, P=2000, H=64, W=64)
    a, ra, st, _ = hip_forward(*small, state=st)
    fresh = R.RasterState()
    fresh.set_option(_lib.OPT_SORT_MODE, 2)
    b, rb, _, _ = hip_forward(*small, state=fresh)
    assert not st.poll()[1] and torch.equal(a, b) and torch.equal(ra, rb)


@pytest.mark.parametrize("sort_mode", [1, 2])
@pytest.mark.parametrize("C", [3, 4])
def test_backward_matches_oracle(C, sort_mode):
    from gpu_util import hip_forward
    from gomavatar_amd import _lib, rasterizer as R
    cam, means, cov6, colors, op = small_scene(seed=31, P=1200, H=64, W=64, opacity=(0.3, 1.0), scale=0.05, C=C)
    cam["bg"] = np.array([0.2, 0.5, 0.1, 0.4], np.float32)
    rng = np.random.default_rng(1)
    wimg = rng.normal(size=(C, 64, 64)).astype(np.float32)
    st0 = R.RasterState()
    st0.set_option(_lib.OPT_SORT_MODE, sort_mode)
    out, radii, st, t = hip_forward(cam, means, cov6, colors, op, requires_grad=True, state=st0, means2D=True)
    (out * torch.from_numpy(wimg).cuda()).sum().backward()
    f = orast.forward(cam, means, cov6, colors, op, dtype=np.float64)
    g = orast.backward(f, wimg.astype(np.float64))
    # the fifth gradient of the boundary: dL/dmeans2D, (P, 3) with the third column zero as upstream's float3 -- the reference renderer
    # discards it (models/modules/renderer/gaussian.py:69-75 only retains the tensor), but it is on the ABI
    m2 = t[4].grad
    assert m2.shape == (means.shape[0], 3) and float(m2[:, 2].abs().max()) == 0.0
    for name, got, ref in (("means3D", t[0].grad, g["dL_dmeans3D"]), ("cov6", t[1].grad, g["dL_dcov6"]),
                           ("colors", t[2].grad, g["dL_dcolors"]), ("opacity", t[3].grad, g["dL_dopacity"]), ("means2D", m2[:, :2], g["dL_dmeans2D"])):
        got = got.cpu().numpy().astype(np.float64)
        scale = np.abs(ref).max()
        err = np.abs(got - ref)
        # fp32 kernel vs fp64 oracle: relative to the largest gradient of the tensor
        assert scale > 0 and np.quantile(err, 0.999) <= 2e-4 * scale, (name, np.quantile(err, 0.999), scale)
        assert np.median(err) <= 1e-6 * scale, (name, np.median(err), scale)


@pytest.mark.parametrize("P", [300, 2600])
def test_backward_with_gaussians_over_many_tiles(P):
    """Gaussians touching more than 32 tiles get a whole wave each in k_preprocess_bwd (rider blocks fed from a list the forward
    builds, 512 per frame); a frame with more keeps the per-lane walk for all of them.  Both paths against the float64 oracle."""
    from gpu_util import hip_forward
    from gomavatar_amd import _lib
    cam, means, cov6, colors, op = small_scene(seed=77, P=P, H=256, W=256, opacity=(0.02, 0.2), spread=0.5, scale=0.12, C=3)
    rng = np.random.default_rng(2)
    wimg = rng.normal(size=(3, 256, 256)).astype(np.float32)
    out, radii, st, t = hip_forward(cam, means, cov6, colors, op, requires_grad=True)
    (out * torch.from_numpy(wimg).cuda()).sum().backward()
    tt = st.export(_lib.BUF_TILES_TOUCHED, torch.empty(P, dtype=torch.int32, device="cuda")).cpu().numpy()
    n_big = int((tt > 32).sum())
    assert (n_big > 512) == (P > 2048) and n_big > 0.7 * P, n_big
    f = orast.forward(cam, means, cov6, colors, op, dtype=np.float64)
    g = orast.backward(f, wimg.astype(np.float64))
    for name, got, ref in (("means3D", t[0].grad, g["dL_dmeans3D"]), ("cov6", t[1].grad, g["dL_dcov6"]),
                           ("colors", t[2].grad, g["dL_dcolors"]), ("opacity", t[3].grad, g["dL_dopacity"])):
        got = got.cpu().numpy().astype(np.float64)
        scale = np.abs(ref).max()
        err = np.abs(got - ref)
        assert np.quantile(err, 0.999) <= 2e-4 * scale, (name, np.quantile(err, 0.999), scale)
        assert np.median(err) <= 2e-6 * scale, (name, np.median(err), scale)


def test_backward_is_bitwise_reproducible():
    from gpu_util import hip_forward
    cam, means, cov6, colors, op = small_scene(seed=32, P=1500, H=64, W=64, opacity=(0.3, 1.0))
    grads = []
    for _ in range(2):
        out, radii, st, t = hip_forward(cam, means, cov6, colors, op, requires_grad=True)
        out.square().sum().backward()
        grads.append([x.grad.clone() for x in t])
    for a, b in zip(*grads):
        assert torch.equal(a, b)


def test_reference_style_two_calls_reuse_binning():
    """gaussian.py:77-94: features padded to 6 channels, two 3-channel calls on the
    same geometry.  The second call re-uses binning; results equal one 4-channel pass."""
    from gpu_util import gom_camera, dev
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    from gomavatar_amd import rasterizer as R
    cam, means, cov6, colors, op = small_scene(seed=33, P=1000, H=64, W=64, opacity=1.0, C=4)
    colors[:, 3] = 1.0
    rs = GaussianRasterizationSettings(image_height=64, image_width=64, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], bg=torch.zeros(4).cuda(),
                                       scale_modifier=1.0, viewmatrix=dev(cam["viewmatrix"]), projmatrix=dev(cam["projmatrix"]), sh_degree=0,
                                       campos=dev(cam["campos"]), prefiltered=False, debug=False)
    rast = GaussianRasterizer(None)
    rast.raster_settings = rs
    xyz = dev(means.T.copy()).requires_grad_()            # (3,F): the reference passes the transposed VIEW
    feat = torch.cat([dev(colors), dev(colors[:, :2])], -1).requires_grad_()
    cov_t = dev(cov6).requires_grad_()
    opac = dev(op)[:, None]
    means2D = torch.zeros_like(xyz.T, requires_grad=True)
    preds = []
    for i in (0, 3):
        pred, radii = rast(means3D=xyz.T, means2D=means2D, colors_precomp=feat[:, i:i + 3], shs=None, opacities=opac, scales=None,
                           rotations=None, cov3D_precomp=cov_t)
        assert radii.dtype == torch.int32 and radii.shape == (1000,)
        preds.append(pred)
    pred6 = torch.cat(preds, 0)[:4]
    one, _ = R.rasterize(dev(means), dev(cov6), dev(colors), dev(op), gom_camera(cam))
    assert torch.equal(pred6, one)
    w = torch.randn_like(pred6)
    (pred6 * w).sum().backward()
    a = [dev(means).requires_grad_(), dev(cov6).requires_grad_(), dev(colors).requires_grad_()]
    one, _ = R.rasterize(a[0], a[1], a[2], dev(op), gom_camera(cam))
    (one * w).sum().backward()
    def close(x, y, tol=1e-5):   # two 3-channel passes sum the same terms in a different order than one 4-channel pass
        return float((x - y).abs().max()) <= tol * float(y.abs().max())
    assert close(xyz.grad.T, a[0].grad) and close(cov_t.grad, a[1].grad)
    g6 = feat.grad
    assert close(g6[:, :3], a[2].grad[:, :3]) and close(g6[:, 3], a[2].grad[:, 3])
    with pytest.raises(Exception):
        rast(means3D=xyz.T, means2D=means2D, colors_precomp=None, shs=None, opacities=opac, cov3D_precomp=cov_t)


def test_body_frame_512_properties():
    """Full-size case (BASELINE cfg: 13 776 Gaussians, 512x512): bit-exact
    binning against the oracle plus size-independent invariants."""
    from gpu_util import hip_forward, export_state, assert_binning_bit_exact
    sc = body_scene(0, frame=2)
    rgb, mask, aux = og.render_path(sc["params"], sc["frame"], sc["faces"], sc["lbs_weights"], 512)
    F = sc["faces"].shape[0]
    colors = np.concatenate([sc["params"]["appearance"].numpy().T, np.ones((F, 1), np.float32)], 1)
    xyz, cov6 = aux["xyz"].detach().numpy(), aux["cov6"].detach().numpy()
    out, radii, st, _ = hip_forward(aux["cam"], xyz, cov6, colors, np.ones(F, np.float32))
    f = orast.forward(aux["cam"], xyz, cov6, colors, np.ones(F, np.float32), margin=True)
    e = export_state(st, F, 512, 512)
    assert_binning_bit_exact(e, f)
    img = out.cpu().numpy()
    assert_image_parity(img, f["color"])
    assert_flips_are_threshold_margins(img, e["n_contrib"], e["final_T"], f, (aux["cam"], xyz, cov6, colors, np.ones(F, np.float32)))
    np.testing.assert_allclose(img[3] + e["final_T"], 1.0, atol=3e-6)      # sum alpha_i T_i + T_final = 1
    k = e["keys"]
    for t in np.nonzero(np.diff(e["tile_base"].astype(np.int64)))[0][:50]:
        a, b = e["tile_base"][t], e["tile_base"][t + 1]
        assert np.all(k[a + 1:b] > k[a:b - 1])                               # strictly sorted per tile


@pytest.mark.parametrize("subdiv,img", [(1, 1024), (2, 1024)])
def test_body_frame_large_configs(subdiv, img):
    """BASELINE configs[3]/[5] shapes (55 104 and 220 416 Gaussians at 1024x1024): the longest tile lists exceed one
    8192-key sort chunk at 220k, so this also covers the chunked merge path at full size.  Forward bit-exact binning
    + image parity, backward against the fp64 oracle."""
    from gpu_util import hip_forward, export_state, assert_binning_bit_exact
    sc = body_scene(subdiv, frame=1, img=img)
    with torch.no_grad():
        Rs, Ts = og.fk_global_RTs(sc["frame"]["cnl_gtfms"], sc["frame"]["dst_Rs"], sc["frame"]["dst_Ts"])
        v_obs = og.lbs(sc["params"]["vertices"].unsqueeze(0), Rs, Ts, sc["lbs_weights"])[0]
        xyz_t, cov_t = og.face_gaussians(v_obs, sc["faces"], sc["params"]["so3"], sc["params"]["scale"], 1e-3)
    cam = og.camera_from_KE(sc["frame"]["K"][0].numpy(), sc["frame"]["E"][0].numpy(), img, img)
    F = sc["faces"].shape[0]
    xyz, cov6 = xyz_t.numpy(), og.pack_cov6(cov_t).numpy()
    colors = np.concatenate([sc["params"]["appearance"].numpy().T, np.ones((F, 1), np.float32)], 1)
    op = np.ones(F, np.float32)
    out, radii, st, t = hip_forward(cam, xyz, cov6, colors, op, requires_grad=True)
    f = orast.forward(cam, xyz, cov6, colors, op, margin=True)
    e = export_state(st, F, img, img)
    assert_binning_bit_exact(e, f)
    if subdiv == 2:
        assert np.diff(e["tile_base"].astype(np.int64)).max() > 8192
    imgs = out.detach().cpu().numpy()
    assert_image_parity(imgs, f["color"])
    assert_flips_are_threshold_margins(imgs, e["n_contrib"], e["final_T"], f, (cam, xyz, cov6, colors, op))
    np.testing.assert_allclose(imgs[3] + e["final_T"], 1.0, atol=3e-6)
    rng = np.random.default_rng(5)
    wimg = rng.normal(size=(4, img, img)).astype(np.float32)
    (out * torch.from_numpy(wimg).cuda()).sum().backward()
    f64 = orast.forward(cam, xyz, cov6, colors, op, dtype=np.float64)
    g = orast.backward(f64, wimg.astype(np.float64))
    for name, got, ref in (("means3D", t[0].grad, g["dL_dmeans3D"]), ("cov6", t[1].grad, g["dL_dcov6"]), ("colors", t[2].grad, g["dL_dcolors"])):
        got = got.cpu().numpy().astype(np.float64)
        scale = np.abs(ref).max()
        err = np.abs(got - ref)
        assert np.quantile(err, 0.999) <= 2e-4 * scale, (name, np.quantile(err, 0.999), scale)
        assert np.median(err) <= 1e-6 * scale, (name, np.median(err), scale)


@pytest.mark.parametrize("shift", [7, 8])
def test_segment_size_option_matches_oracle(shift):
    """GOM_OPT_SEG_SHIFT: 128-entry (single frame) and 256-entry (batched launch) segments on a single frame."""
    from gpu_util import hip_forward
    from gomavatar_amd import _lib, rasterizer as R
    cam, means, cov6, colors, op = small_scene(seed=41, P=5000, H=64, W=64, spread=0.15, scale=0.03, opacity=(0.2, 0.9))
    cam["bg"] = np.array([0.3, 0.1, 0.6, 0.2], np.float32)
    st = R.RasterState()
    st.set_option(_lib.OPT_SEG_SHIFT, shift)
    out, radii, st, t = hip_forward(cam, means, cov6, colors, op, state=st, requires_grad=True)
    f = orast.forward(cam, means, cov6, colors, op)
    assert (f["ranges"][:, 1] - f["ranges"][:, 0]).max() > 700          # several segments per tile
    assert_image_parity(out.detach().cpu().numpy(), f["color"])
    rng = np.random.default_rng(2)
    wimg = rng.normal(size=(4, 64, 64)).astype(np.float32)
    (out * torch.from_numpy(wimg).cuda()).sum().backward()
    g = orast.backward(orast.forward(cam, means, cov6, colors, op, dtype=np.float64), wimg.astype(np.float64))
    for name, got, ref in (("means3D", t[0].grad, g["dL_dmeans3D"]), ("cov6", t[1].grad, g["dL_dcov6"]), ("colors", t[2].grad, g["dL_dcolors"]),
                           ("opacity", t[3].grad, g["dL_dopacity"])):
        got = got.cpu().numpy().astype(np.float64)
        scale = np.abs(ref).max()
        err = np.abs(got - ref)
        assert np.quantile(err, 0.999) <= 2e-4 * scale and np.median(err) <= 1e-6 * scale, (name, np.quantile(err, 0.999), scale)


@pytest.mark.parametrize("seed", range(10))
def test_fuzz_shapes_scales_and_depths(seed):
    """Random image sizes (not multiples of the tile), Gaussian counts, footprints from sub-pixel to a quarter of the image,
    points close to / behind the 0.2 near cut and far off screen: bit-exact binning + image parity + backward sanity."""
    rng = np.random.default_rng(1000 + seed)
    H, W = int(rng.integers(17, 120)), int(rng.integers(17, 120))
    P = int(rng.integers(1, 3000))
    C = int(rng.choice([3, 4]))
    cam, means, cov6, colors, op = small_scene(seed=2000 + seed, P=P, H=H, W=W, C=C, opacity=(0.05, 1.0), spread=float(rng.uniform(0.2, 1.5)),
                                               scale=float(10 ** rng.uniform(-3, -0.7)), z=float(rng.uniform(0.5, 4.0)))
    cam["bg"] = rng.uniform(0, 1, 4).astype(np.float32)
    # (footprints from sub-pixel to a quarter of the image on images of a few thousand pixels: up to 6e-3 of them sit near a threshold)
    img, f, e = _compare_forward(cam, means, cov6, colors, op, max_fragile=1.5e-2)
    from gpu_util import hip_forward
    out, radii, st, t = hip_forward(cam, means, cov6, colors, op, requires_grad=True)
    wimg = rng.normal(size=(C, H, W)).astype(np.float32)
    (out * torch.from_numpy(wimg).cuda()).sum().backward()
    g = orast.backward(orast.forward(cam, means, cov6, colors, op, dtype=np.float64), wimg.astype(np.float64))
    for name, got, ref in (("means3D", t[0].grad, g["dL_dmeans3D"]), ("cov6", t[1].grad, g["dL_dcov6"]), ("colors", t[2].grad, g["dL_dcolors"])):
        got = got.cpu().numpy().astype(np.float64)
        assert np.isfinite(got).all(), name
        scale = max(np.abs(ref).max(), 1e-30)
        err = np.abs(got - ref)
        # (a flipped pixel moves the gradients of every Gaussian under it: compare below the tail those few produce.  With a few dozen
        #  Gaussians the 99 % quantile IS the worst row -- seed 9890: 58 splats a third of the image wide, one row 0.8 % off in fp32, the
        #  round-1 library the same -- so small scenes are held to the 90 % quantile and a bound on the worst row)
        if err.size >= 1000:
            assert np.quantile(err, 0.99) <= 1e-3 * scale, (name, np.quantile(err, 0.99), scale)
        else:
            assert np.quantile(err, 0.90) <= 1e-3 * scale and err.max() <= 5e-2 * scale, (name, np.quantile(err, 0.90), err.max(), scale)


@pytest.mark.parametrize("seed", [0, 1])
def test_fuzz_indefinite_covariances(seed):
    """Covariances that are NOT positive semi-definite (a caller's bug, or an optimizer step gone wrong): the screen-space conic of some
    Gaussians is indefinite, the exponent is positive on part of the image and the reference skips exactly those pixels (`power > 0.f ->
    continue`).  The HIP path folds that rule into a 0 / 1 factor (alpha_eval: m2 = clamp(1 - 2^126 pw), exact except
    for a DENORMAL positive exponent, 0 < pw < 2^-126, where the factor is fractional and the reference skips -- csrc/raster_render.hip);
    its conservative cull steps aside for such entries (cull_entry: "not positive definite: exact path").  Held: bit-exact binning, image
    parity, and gradients against the float64 oracle."""
    from gpu_util import hip_forward
    rng = np.random.default_rng(7000 + seed)
    H, W, C, P = 176, 208, 4, 4000   # (36 608 pixels: the flip allowance of _compare_forward -- 2e-4 of the pixels -- is 7, not 1)
    cam, means, cov6, colors, op = small_scene(seed=7100 + seed, P=P, H=H, W=W, C=C, opacity=(0.2, 1.0), spread=0.5, scale=0.06)
    # Sigma - s v v^T, |v| = 1: exactly ONE negative eigenvalue can appear, so the projected 2 x 2 covariance has at most one (interlacing)
    # and its larger eigenvalue -- the radius -- stays positive.  (TWO negative eigenvalues make sqrt(max(lambda)) a NaN: upstream then
    # converts NaN to a radius of 0 yet counts the Gaussian's tiles, i.e. reads unwritten keys -- undefined there, not a case to restate.)
    bad = rng.random(P) < 0.4
    v = rng.normal(size=(P, 3)); v /= np.linalg.norm(v, axis=1, keepdims=True)
    sh = (rng.uniform(1.0, 6.0, P) * 0.06 ** 2 * bad)[:, None, None] * (v[:, :, None] * v[:, None, :])
    for k, (r, c_) in enumerate(((0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2))):
        cov6[:, k] -= sh[:, r, c_].astype(np.float32)
    cam["bg"] = rng.uniform(0, 1, 4).astype(np.float32)
    # (40 % indefinite conics: the sign of the exponent is a third kind of branch, 4e-3 of the pixels sit near one)
    img, f, e = _compare_forward(cam, means, cov6, colors, op, max_fragile=1.5e-2)
    co = f["conic_opacity"].astype(np.float64)
    vis = f["radii"] > 0
    n_indef = int(((co[:, 0] * co[:, 2] - co[:, 1] ** 2 <= 0) & vis).sum())
    assert n_indef >= 20, n_indef            # the case is really there
    assert np.isfinite(f["conic_opacity"]).all() and (f["radii"] >= 0).all()
    out, radii, st, t = hip_forward(cam, means, cov6, colors, op, requires_grad=True)
    wimg = rng.normal(size=(C, H, W)).astype(np.float32)
    (out * torch.from_numpy(wimg).cuda()).sum().backward()
    g = orast.backward(orast.forward(cam, means, cov6, colors, op, dtype=np.float64), wimg.astype(np.float64))
    for name, got, ref in (("means3D", t[0].grad, g["dL_dmeans3D"]), ("cov6", t[1].grad, g["dL_dcov6"]), ("colors", t[2].grad, g["dL_dcolors"]),
                           ("opacity", t[3].grad, g["dL_dopacity"])):
        got = got.cpu().numpy().astype(np.float64)
        assert np.isfinite(got).all(), name
        scale = max(np.abs(ref).max(), 1e-30)
        err = np.abs(got - ref)
        assert np.quantile(err, 0.99) <= 1e-3 * scale, (name, np.quantile(err, 0.99), scale)


def test_device_camera_entry_points_are_bitwise_the_host_camera_path():
    """gom_raster_forward_dcam / gom_raster_backward_dcam read the same 160 camera bytes from device memory instead of the
    launch arguments: identical image, radii and gradients."""
    import ctypes
    from gpu_util import hip_forward, gom_camera, dev
    from gomavatar_amd import rasterizer as R
    cam, means, cov6, colors, op = small_scene(seed=77, P=3000, H=80, W=112, opacity=(0.3, 1.0), scale=0.05, C=4)
    cam["bg"] = np.array([0.3, 0.1, 0.6, 0.2], np.float32)
    out, radii, st, t = hip_forward(cam, means, cov6, colors, op, requires_grad=True)
    w = torch.linspace(-1.0, 1.0, out.numel(), device="cuda").reshape(out.shape)
    (out * w).sum().backward()
    host = gom_camera(cam)
    dcam = R.DeviceCamera(cam["H"], cam["W"], "cuda")
    raw = np.frombuffer(ctypes.string_at(ctypes.addressof(host), ctypes.sizeof(host)), dtype=np.float32).copy()
    assert raw.size == 40
    dcam.data.copy_(torch.from_numpy(raw))
    t2 = [dev(means).requires_grad_(), dev(cov6).requires_grad_(), dev(colors).requires_grad_(), dev(op).requires_grad_()]
    out2, radii2 = R.rasterize(t2[0], t2[1], t2[2], t2[3], dcam)
    (out2 * w).sum().backward()
    assert torch.equal(out, out2) and torch.equal(radii, radii2)
    for a, b in zip(t, t2):
        assert torch.equal(a.grad, b.grad)


@pytest.mark.parametrize("seg_shift", [7, 8])
def test_backward_task_shapes_agree(seg_shift):
    """GOM_OPT_BWD_MODE: two sub-ranges between barriers (opposite quadrants per wave) against one sub-range per barrier: the same
    per-quadrant sums folded in the same order -> bitwise the same gradients; the block-row kernel (mode 2) within round-off; all
    against the fp64 oracle."""
    from gpu_util import hip_forward
    from gomavatar_amd import _lib, rasterizer as R
    cam, means, cov6, colors, op = small_scene(seed=41, P=4000, H=96, W=96, opacity=(0.3, 1.0), spread=0.25, scale=0.03, C=4)
    cam["bg"] = np.array([0.2, 0.5, 0.1, 0.4], np.float32)
    rng = np.random.default_rng(2)
    wimg = rng.normal(size=(4, 96, 96)).astype(np.float32)
    f = orast.forward(cam, means, cov6, colors, op, dtype=np.float64)
    g = orast.backward(f, wimg.astype(np.float64))
    assert (f["ranges"][:, 1] - f["ranges"][:, 0]).max() > 600          # several segments per tile
    got = []
    lab = _lib.has_lab()   # modes 2 and 3 live in -DGOM_LAB builds only (include/gom_hip_lab.h); without it modes 0 / 1 stand in (then trivially equal)
    for mode in (0, 1, 2 if lab else 0, 2 if lab else 0, 3 if lab else 1, 3 if lab else 1):
        st = R.RasterState()
        st.set_option(_lib.OPT_BWD_MODE, mode)
        st.set_option(_lib.OPT_SEG_SHIFT, seg_shift)
        out, radii, st, t = hip_forward(cam, means, cov6, colors, op, requires_grad=True, state=st)
        (out * torch.from_numpy(wimg).cuda()).sum().backward()
        got.append([x.grad.cpu().numpy().astype(np.float64) for x in t])
        for name, a, ref in zip(("means3D", "cov6", "colors", "opacity"), got[-1], (g["dL_dmeans3D"], g["dL_dcov6"], g["dL_dcolors"], g["dL_dopacity"])):
            scale = np.abs(ref).max()
            err = np.abs(a - ref)
            assert np.quantile(err, 0.999) <= 2e-4 * scale and np.median(err) <= 1e-6 * scale, (mode, name, np.quantile(err, 0.999), np.median(err), scale)
    for a, b in zip(got[0], got[1]):
        np.testing.assert_array_equal(a, b)
    # mode 2 -- (sub-range, 4x4 block) items, one per DPP row (csrc/lab/seg_bwd_blk.hpp) -- sums a Gaussian's pixels block by block instead of
    # quadrant by quadrant: fp32 round-off away from the other two (both are held to the fp64 oracle above), bitwise equal to itself
    for a, b, c in zip(got[0], got[2], got[3]):
        np.testing.assert_array_equal(b, c)
        assert np.abs(a - b).max() <= 2e-5 * np.abs(a).max()
    # mode 3 -- a lane per (pixel, entry) record the forward left (csrc/lab/rec_bwd.hpp): the suffix colour comes from a difference of the piece's
    # totals instead of the replay's recurrence: round-off away from the replay kernels, bitwise equal to itself (one summation order per entry)
    for a, b, c in zip(got[0], got[4], got[5]):
        np.testing.assert_array_equal(b, c)
        assert np.abs(a - b).max() <= 2e-5 * np.abs(a).max()


def test_zz_flip_census(capsys):
    """Runs last in this file: what the flip-aware comparisons above saw (run with -s)."""
    if not FLIP_LOG:
        pytest.skip("no forward comparison ran in this session")
    px, frag, dev, ill, dev_ill = (sum(r[i] for r in FLIP_LOG) for i in (0, 1, 2, 4, 5))
    hip = [(r[2], r[3]) for r in FLIP_LOG if r[3] is not None]
    with capsys.disabled():
        print(f"\n[flip census] {len(FLIP_LOG)} comparisons, {px} pixels: branch margin < {MARGIN:g} on {frag} ({frag / px:.1e}), ill-conditioned in fp32 (round-off estimate > 2e-5) {ill} ({ill / px:.1e});"
              f" beyond 1e-4 or other n_contrib: {dev}, of which {dev_ill} bounded by their conditioning, the rest on flagged pixels; largest deviation of a solid pixel {max(r[6] for r in FLIP_LOG):.1e};"
              f" beyond 1e-4 against the oracle in the HIP formulation: {sum(h[1] for h in hip)} (reference formulation, same comparisons: {sum(h[0] for h in hip)})")
    assert dev <= frag + dev_ill


def gaussians_under_flagged_pixels(f, flagged, n_contrib_other=None):
    """Gaussians that reach (alpha >= half the 1/255 threshold, within the list positions either side blended) a pixel of the boolean map `flagged`:
    the ones whose gradients a threshold flip or an ill-conditioned exponent at that pixel can move."""
    H, W = flagged.shape
    gx = (W + 15) // 16
    bad = np.zeros(f["xy"].shape[0], bool)
    for y_, x_ in zip(*np.nonzero(flagged)):
        t_ = (int(y_) // 16) * gx + int(x_) // 16
        lst = f["point_list"][f["ranges"][t_, 0]:f["ranges"][t_, 1]].astype(np.int64)
        last = int(f["n_contrib"][y_, x_]) if n_contrib_other is None else int(max(f["n_contrib"][y_, x_], n_contrib_other[y_, x_]))
        lst = lst[:last + 2]
        dx, dy = f["xy"][lst, 0].astype(np.float64) - x_, f["xy"][lst, 1].astype(np.float64) - y_
        co = f["conic_opacity"][lst].astype(np.float64)
        power = -0.5 * (co[:, 0] * dx * dx + co[:, 2] * dy * dy) - co[:, 1] * dx * dy
        bad[lst[(power <= 1e-6) & (co[:, 3] * np.exp(np.minimum(power, 0.0)) >= 0.5 / 255.0)]] = True
    return bad


@pytest.mark.parametrize("seed,P,shape,opacity", [(51, 1500, (96, 112), (0.3, 1.0)), (52, 4000, (128, 128), (0.05, 0.6)), (53, 800, (64, 64), 1.0)])
def test_raster_boundary_gradients_are_round_off_away_from_flagged_pixels(seed, P, shape, opacity):
    """The render backward's OWN outputs -- dL/dcolors, dL/dopacity, dL/dmeans2D: plain sums over a Gaussian's pixels, in front of the ill-conditioned
    conic -> covariance step -- against the float64 oracle, FLIP-AWARE (round-5 review): the Gaussians that blend into a pixel whose branch margin is below
    MARGIN, whose round-off estimate is above 2e-5, or that did deviate are set aside; every other element agrees to 1e-5 of the tensor's largest gradient
    (the review's bound; measured below) -- round-off, with ZERO exceptions.  The gradients behind the covariance step (means3D, cov6) keep the quantile bounds of
    test_backward_matches_oracle: their tail is conditioning, which the fp32 build of the oracle shares (tests/test_gpu_metric_workload.py)."""
    from gpu_util import hip_forward, export_state
    H, W = shape
    cam, means, cov6, colors, op = small_scene(seed=seed, P=P, H=H, W=W, opacity=opacity, scale=0.05, C=4)
    cam["bg"] = np.array([0.2, 0.5, 0.1, 0.4], np.float32)
    wimg = np.random.default_rng(seed).normal(size=(4, H, W)).astype(np.float32)
    out, radii, st, t = hip_forward(cam, means, cov6, colors, op, requires_grad=True, means2D=True)
    (out * torch.from_numpy(wimg).cuda()).sum().backward()
    e = export_state(st, P, H, W)
    f32 = orast.forward(cam, means, cov6, colors, op, margin=True)
    f64 = orast.forward(cam, means, cov6, colors, op, dtype=np.float64, margin=True)
    g = orast.backward(f64, wimg.astype(np.float64))
    img = out.detach().cpu().numpy()
    flagged = (f32["margin"] < MARGIN) | (f64["margin"] < MARGIN) | (f32["roundoff"] > 2e-5) | (np.abs(img - f64["color"]).max(axis=0) > IMG_TOL) | (e["n_contrib"] != f64["n_contrib"])
    aside = gaussians_under_flagged_pixels(f64, flagged, e["n_contrib"])
    assert flagged.mean() <= 5e-2 and aside.mean() <= 0.5, (float(flagged.mean()), float(aside.mean()))      # (not vacuous: most Gaussians are held to the tight bound)
    for name, got, ref in (("colors", t[2].grad, g["dL_dcolors"]), ("opacity", t[3].grad, g["dL_dopacity"]), ("means2D", t[4].grad[:, :2], g["dL_dmeans2D"])):
        got = got.cpu().numpy().astype(np.float64).reshape(ref.shape)
        scale = np.abs(ref).max()
        err = np.abs(got - ref)[~aside]
        print(f"[boundary gradients, seed {seed}] d{name}: max |err| / max|g| away from the flagged pixels' Gaussians {float(err.max()) / scale:.1e} ({int(aside.sum())} of {P} set aside, {int(flagged.sum())} pixels flagged)")
        assert scale > 0 and err.size > 0.5 * P and float(err.max()) <= 1e-5 * scale, (name, float(err.max()) / scale, int(aside.sum()), int(flagged.sum()))
