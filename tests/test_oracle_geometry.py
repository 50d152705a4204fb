"""The torch restatement of FK / LBS / Steiner / camera conversion against the
golden vectors produced by the reference's own functions
(scripts/make_goldens.py)."""
import math
import os

import numpy as np
import torch

from oracle import geometry as og


def _g(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_fk_matches_reference(golden_dir):
    g = _g(golden_dir, "geometry_fk_lbs.npz")
    for f in (0, 1, 2):
        R, T = og.fk_global_RTs(torch.from_numpy(g[f"f{f}_cnl_gtfms"])[None], torch.from_numpy(g[f"f{f}_dst_Rs"])[None],
                                torch.from_numpy(g[f"f{f}_dst_Ts"])[None])
        np.testing.assert_allclose(R.numpy(), g[f"f{f}_global_Rs"], rtol=0, atol=1e-6)
        np.testing.assert_allclose(T.numpy(), g[f"f{f}_global_Ts"], rtol=0, atol=1e-6)
    R, T = og.fk_global_RTs(torch.from_numpy(g["gen_cnl_gtfms"])[None], torch.from_numpy(g["f1_dst_Rs"])[None],
                            torch.from_numpy(g["f1_dst_Ts"])[None])
    np.testing.assert_allclose(R.numpy(), g["gen_global_Rs"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(T.numpy(), g["gen_global_Ts"], rtol=0, atol=1e-6)


def test_lbs_matches_reference(golden_dir):
    g = _g(golden_dir, "geometry_fk_lbs.npz")
    xyz = torch.from_numpy(g["xyz"]).T.contiguous()[None]
    w = torch.from_numpy(g["lbs_weights25"])
    for f in (0, 1, 2):
        out = og.lbs(xyz, torch.from_numpy(g[f"f{f}_global_Rs"]), torch.from_numpy(g[f"f{f}_global_Ts"]), w)
        np.testing.assert_allclose(out.numpy(), g[f"f{f}_v_obs"], rtol=0, atol=1e-6)


def test_steiner_matches_reference(golden_dir):
    g = _g(golden_dir, "geometry_steiner.npz")
    A = og.steiner_frame(torch.from_numpy(g["tri"]), float(g["sigma"]))
    np.testing.assert_allclose(A.numpy(), g["A"], rtol=1e-5, atol=1e-9)


def test_steiner_ellipse_property():
    # the Steiner inellipse touches the three edge midpoints: |A^-1 (m - c)| has unit in-plane norm / 2
    torch.manual_seed(0)
    tri = torch.randn(50, 3, 3, dtype=torch.float64)
    A = og.steiner_frame(tri, 1e-3)
    c = tri.mean(1)
    for a, b in ((0, 1), (1, 2), (2, 0)):
        m = 0.5 * (tri[:, a] + tri[:, b]) - c
        loc = torch.linalg.solve(A, m[..., None])[..., 0]
        r = (loc[:, 0] ** 2 + loc[:, 1] ** 2).sqrt()
        assert torch.allclose(r, torch.full_like(r, 0.5), atol=1e-9)
        assert loc[:, 2].abs().max() < 1e-6


def test_so3_exp_is_matrix_exponential():
    torch.manual_seed(1)
    v = torch.randn(64, 3, dtype=torch.float64) * 0.7
    K = torch.zeros(64, 3, 3, dtype=torch.float64)
    K[:, 0, 1], K[:, 0, 2], K[:, 1, 0], K[:, 1, 2], K[:, 2, 0], K[:, 2, 1] = -v[:, 2], v[:, 1], v[:, 2], -v[:, 0], -v[:, 1], v[:, 0]
    assert torch.allclose(og.so3_exp(v), torch.matrix_exp(K), atol=1e-12)
    # clamp branch (theta^2 < 1e-4): still a first-order rotation
    small = torch.tensor([[1e-3, -2e-3, 5e-4]], dtype=torch.float64)
    R = og.so3_exp(small)
    assert torch.allclose(R @ R.transpose(1, 2), torch.eye(3, dtype=torch.float64)[None], atol=1e-5)


def test_camera_block_matches_reference(golden_dir):
    g = _g(golden_dir, "renderer_camera.npz")
    cam = og.camera_from_KE(g["K"][0] if g["K"].ndim == 3 else g["K"], g["E"][0] if g["E"].ndim == 3 else g["E"], 512, 512)
    assert cam["tanfovx"] == float(g["tanfovx"]) and cam["tanfovy"] == float(g["tanfovy"])
    np.testing.assert_array_equal(cam["viewmatrix"], g["viewmatrix"])
    np.testing.assert_array_equal(cam["projmatrix"], g["projmatrix"])
    np.testing.assert_allclose(cam["campos"], g["campos"], atol=1e-6)
    assert math.isclose(2 * math.atan(512 / (2 * 1250.0)), float(g["focal2fov"]))
    # the 6-pack order and the two 3-channel colour slices the reference hands to the CUDA extension
    cov = torch.from_numpy(g["cov"])[0]
    np.testing.assert_array_equal(og.pack_cov6(cov).numpy(), g["call0_cov6"])
    feats = torch.cat([torch.from_numpy(g["feats"]), torch.ones(1, g["feats"].shape[1], 1)], -1)
    np.testing.assert_array_equal(feats[0, :, :3].numpy(), g["call0_colors"])
    np.testing.assert_array_equal(torch.cat([feats[0, :, 3:], feats[0, :, :2]], -1).numpy(), g["call1_colors"])


def test_rodrigues_module_matches_reference(golden_dir):
    """oracle.geometry.rodrigues_module against the reference's own RodriguesModule (scripts/make_module_goldens.py): bitwise in fp32
    (the same torch expression), and the fp64 evaluation within fp32 round-off of it."""
    g = _g(golden_dir, "pose_modules.npz")
    rv = torch.from_numpy(g["rod_rvec"]).requires_grad_()
    R = og.rodrigues_module(rv)
    assert np.array_equal(R.detach().numpy(), g["rod_out"])
    gr = torch.autograd.grad((R * torch.arange(1.0, 10.0).view(3, 3)).sum(), rv)[0]
    np.testing.assert_allclose(gr.numpy(), g["rod_grvec"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(og.rodrigues_module(rv.detach().double()).numpy(), g["rod_out"], rtol=0, atol=2e-7)
