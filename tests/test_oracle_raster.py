"""Self-consistency of the C raster oracle (the reference ships no vectors for
this third-party op: parity unpinned, see oracle/raster_oracle.c).  What can be
pinned: structural invariants of the binning, the compositing identity, and the
hand-written backward against fp64 finite differences of the forward."""
import numpy as np
import pytest

from oracle import raster as orast
from helpers import small_scene


def test_binning_invariants():
    cam, means, cov6, colors, op = small_scene(seed=3, P=600, H=70, W=90)
    f = orast.forward(cam, means, cov6, colors, op)
    gx, gy = (90 + 15) // 16, (70 + 15) // 16
    assert f["ranges"].shape == (gx * gy, 2)
    assert f["D"] == int(f["tiles_touched"].sum()) == int(f["offsets"][-1])
    keys = f["keys"]
    assert np.all(keys[1:] >= keys[:-1])                       # globally sorted by (tile, depth bits)
    tiles = (keys >> np.uint64(32)).astype(np.int64)
    for t in range(gx * gy):
        a, b = f["ranges"][t]
        assert np.all(tiles[a:b] == t)
        assert (b - a) == np.count_nonzero(tiles == t)
    # ties keep ascending gaussian order (stable sort); list entries reference visible gaussians only
    same = keys[1:] == keys[:-1]
    assert np.all(f["point_list"][1:][same] > f["point_list"][:-1][same])
    assert np.all(f["radii"][f["point_list"]] > 0)
    # depth bits in the key are the float32 bits of the view-space depth
    d = f["depth"][f["point_list"]].astype(np.float32).view(np.uint32)
    assert np.array_equal((keys & np.uint64(0xffffffff)).astype(np.uint32), d)
    # rect area == tiles_touched, rects inside the grid
    r = f["rect"]
    assert np.array_equal((r[:, 2] - r[:, 0]) * (r[:, 3] - r[:, 1]), f["tiles_touched"].astype(np.int64))
    assert r[:, 2].max() <= gx and r[:, 3].max() <= gy


def test_compositing_identity_and_background():
    cam, means, cov6, colors, op = small_scene(seed=4, P=500, H=64, W=64, opacity=(0.2, 1.0))
    colors[:, 3] = 1.0
    cam["bg"] = np.array([0.3, 0.6, 0.9, 0.0], np.float32)
    f = orast.forward(cam, means, cov6, colors, op)
    # channel 3 carries sum(alpha_i T_i): with the leftover transmittance it sums to one
    np.testing.assert_allclose(f["color"][3] + f["final_T"], 1.0, atol=2e-6)
    empty = f["n_contrib"] == 0
    assert empty.any()
    for ch in range(3):
        np.testing.assert_allclose(f["color"][ch][empty], cam["bg"][ch] * f["final_T"][empty], atol=1e-7)
    assert np.all(f["final_T"] >= 1e-4 - 1e-9)


def test_edge_cases():
    cam, means, cov6, colors, op = small_scene(seed=5, P=50, H=33, W=47)   # not multiples of 16
    f = orast.forward(cam, means, cov6, colors, op)
    assert f["color"].shape == (4, 33, 47)
    # everything behind the near plane -> nothing rendered
    f2 = orast.forward(cam, means - np.array([0, 0, 10], np.float32), cov6, colors, op)
    assert f2["D"] == 0 and np.all(f2["radii"] == 0) and np.all(f2["n_contrib"] == 0) and np.all(f2["final_T"] == 1)
    # a single huge gaussian covers every tile
    big = np.array([[0.0, 0.0, 0.0]], np.float32)
    c6 = np.array([[4.0, 0, 0, 4.0, 0, 4.0]], np.float32)
    f3 = orast.forward(cam, big, c6, np.ones((1, 3), np.float32), np.ones(1, np.float32))
    assert f3["tiles_touched"][0] == ((47 + 15) // 16) * ((33 + 15) // 16)
    b = orast.backward(f3, np.ones((3, 33, 47), np.float32))
    assert np.isfinite(b["dL_dmeans3D"]).all() and np.isfinite(b["dL_dcov6"]).all()


def test_f32_matches_f64():
    cam, means, cov6, colors, op = small_scene(seed=6, P=800, H=96, W=96, opacity=(0.3, 1.0))
    f32 = orast.forward(cam, means, cov6, colors, op, dtype=np.float32)
    f64 = orast.forward(cam, means, cov6, colors, op, dtype=np.float64)
    # ulp-level flips of the integer state are possible but must be rare
    assert np.mean(f32["radii"] != f64["radii"]) < 5e-3
    assert np.mean(np.abs(f32["color"] - f64["color"])) < 1e-5


@pytest.mark.parametrize("C", [3, 4])
def test_backward_against_finite_differences(C):
    cam, means, cov6, colors, op = small_scene(seed=7, P=120, H=48, W=48, opacity=(0.3, 0.8), scale=0.06, C=C)
    cam["bg"] = np.array([0.2, 0.5, 0.1, 0.4], np.float32)
    means, cov6, colors, op = (a.astype(np.float64) for a in (means, cov6, colors, op))
    rng = np.random.default_rng(0)
    wimg = rng.normal(size=(C, 48, 48))

    def loss(m, c6, col, o):
        return float((orast.forward(cam, m, c6, col, o, dtype=np.float64)["color"] * wimg).sum())

    f = orast.forward(cam, means, cov6, colors, op, dtype=np.float64)
    g = orast.backward(f, wimg)
    vis = np.nonzero(f["radii"] > 0)[0]
    assert len(vis) > 30
    checked = 0
    for name, arr, grad in (("means", means, g["dL_dmeans3D"]), ("cov6", cov6, g["dL_dcov6"]), ("colors", colors, g["dL_dcolors"]), ("opacity", op, g["dL_dopacity"])):
        for _ in range(12):
            i = int(rng.choice(vis))
            idx = (i,) if arr.ndim == 1 else (i, int(rng.integers(arr.shape[1])))
            args = {"means": means, "cov6": cov6, "colors": colors, "opacity": op}
            an = float(grad[idx])
            errs = []
            # the forward has measure-zero jumps (alpha < 1/255 cut, T < 1e-4 stop, ceil of the radius); a step that
            # straddles one is detected by disagreeing step sizes, so accept the best of three.
            for eps in (1e-6, 1e-7, 3e-8):
                hi = {k: v.copy() for k, v in args.items()}
                lo = {k: v.copy() for k, v in args.items()}
                hi[name][idx] += eps
                lo[name][idx] -= eps
                fd = (loss(hi["means"], hi["cov6"], hi["colors"], hi["opacity"]) - loss(lo["means"], lo["cov6"], lo["colors"], lo["opacity"])) / (2 * eps)
                errs.append(abs(fd - an) / max(1.0, abs(fd), abs(an)))
            assert min(errs) <= 1e-4, (name, idx, an, errs)
            checked += 1
    assert checked == 48


def test_thread_count_changes_no_bit():
    """The oracle's per-Gaussian stages and its (stable, chunked) radix sort are OpenMP loops since round 4 -- bench.py's `cpu_baseline`
    row for all cores is then a real multi-core number.  Every forward output, integer or float, must be the 1-thread one bit for bit."""
    cam, means, cov6, colors, op = small_scene(seed=5, P=30000, H=128, W=128, opacity=(0.05, 1.0), spread=0.3, scale=0.02, C=4)
    w = np.random.default_rng(1).normal(size=(4, 128, 128))
    outs = []
    for nt in (1, 8):
        orast.set_threads(nt)
        f = orast.forward(cam, means, cov6, colors, op, dtype=np.float32)
        g = orast.backward(f, w.astype(np.float32))
        outs.append((f, g))
    orast.set_threads(1)
    (f1, g1), (f8, g8) = outs
    assert f1["D"] == f8["D"] and f1["D"] > 20000      # (enough pairs for the sort to really run on several threads)
    for k in ("radii", "tiles_touched", "ranges", "point_list", "n_contrib", "color", "final_T", "xy", "conic_opacity", "depth"):
        if k in f1:
            np.testing.assert_array_equal(f1[k], f8[k], err_msg=k)
    # (the render backward adds a Gaussian's per-tile terms with `omp atomic` in whatever order the tiles finish: that sum, not the new loops,
    #  moves by round-off with the thread count -- as it did before)
    for k in ("dL_dmeans3D", "dL_dcov6", "dL_dopacity"):
        assert np.abs(g1[k] - g8[k]).max() <= 2e-5 * np.abs(g1[k]).max(), k


MARGIN = 2e-5     # relative distance of a branch-deciding quantity to its threshold above which fp32 implementations take the same branch


@pytest.mark.parametrize("seed", [31, 32, 33, 34])
def test_two_formulations_of_alpha_differ_only_at_threshold_margins(seed):
    """The oracle's two fp32 FORMULATIONS of the same alpha (form="ref": the reference's expression; form="hip": the HIP path's pre-scaled
    conic + fma chain + exp2) are two faithful fp32 implementations of App. A.3.  Claim (tests/test_gpu_raster.py relies on it for the HIP
    path itself): wherever the per-pixel MARGIN is comfortable, they agree to round-off -- same n_contrib, image within 1e-5 --, i.e. every
    larger deviation sits on a pixel the margin flags as a threshold flip; and flagged pixels are rare."""
    from helpers import small_scene
    cam, means, cov6, colors, op = small_scene(seed=seed, P=2500, H=96, W=112, opacity=(0.15, 1.0), C=4)
    a = orast.forward(cam, means, cov6, colors, op, margin=True)
    b = orast.forward(cam, means, cov6, colors, op, form="hip")
    for k in ("radii", "tiles_touched", "keys", "point_list", "ranges"):       # the binning does not depend on the formulation
        np.testing.assert_array_equal(a[k], b[k])
    solid = a["margin"] >= MARGIN
    assert np.count_nonzero(~solid) <= 1e-3 * solid.size + 2, np.count_nonzero(~solid)
    d = np.abs(a["color"] - b["color"]).max(0)
    assert d[solid].max() <= 1e-5 and np.array_equal(a["n_contrib"][solid], b["n_contrib"][solid])
    assert np.abs(a["final_T"] - b["final_T"])[solid].max() <= 1e-5
    # gradients of the two formulations, away from the flagged pixels' Gaussians: fp32 round-off of sums
    g = np.random.default_rng(seed).normal(size=a["color"].shape).astype(np.float32)
    ga, gb = orast.backward(a, g), orast.backward(b, g)
    for k in ("dL_dcolors", "dL_dmeans2D", "dL_dconic", "dL_dopacity"):
        assert np.abs(ga[k] - gb[k]).max() <= 2e-5 * np.abs(ga[k]).max(), k


def test_margin_flags_a_constructed_threshold_pixel():
    """One Gaussian whose alpha at a chosen pixel is EXACTLY at 1/255 up to an ulp: the margin of that pixel is ~1e-7, its neighbours' is not."""
    from oracle import geometry as og
    K = np.array([[80.0, 0, 16], [0, 80.0, 16], [0, 0, 1]], np.float32)
    E = np.eye(4, dtype=np.float32); E[2, 3] = 2.0
    cam = og.camera_from_KE(K, E, 32, 32)
    means = np.array([[0.0, 0.0, 0.0]], np.float32)
    cov6 = np.array([[2e-3, 0, 0, 2e-3, 0, 2e-3]], np.float32)
    col = np.ones((1, 3), np.float32)
    f = orast.forward(cam, means, cov6, col, np.ones(1, np.float32), margin=True)
    co, (cx, cy) = f["conic_opacity"][0], f["xy"][0]
    # choose the opacity so that alpha(20, 16) = 1/255 exactly (in real arithmetic)
    dx, dy = cx - 20.0, cy - 16.0
    power = -0.5 * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy
    o = np.float32((1.0 / 255.0) / np.exp(np.float64(power)))
    f = orast.forward(cam, means, cov6, col, np.array([o], np.float32), margin=True)
    assert f["margin"][16, 20] < 1e-6 and f["margin"][16, 18] > 1e-3 and f["margin"][14, 20] > 1e-3, (f["margin"][16, 20], f["margin"][16, 18])
