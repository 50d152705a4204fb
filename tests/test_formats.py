"""Checkpoint and dataset formats of the reference (gomavatar_amd/formats.py): CPU-only round trips."""
import os
from types import SimpleNamespace as NS

import numpy as np
import pytest
import torch

from gomavatar_amd import formats, synthetic as syn
from gomavatar_amd.model import Model


def _cfg():
    return NS(img_size=(64, 64), canonical_geometry=NS(sigma=1e-3, radius_scale=1.0, deform_so3=True, deform_scale=True), appearance=NS(color_init=0.5),
              normal_renderer=NS(sigma=1e-5, soft_mask=True), shadow_module=NS(name="basic", multires=6, mlp_width=128, mlp_depth=3, skips=(4,)),
              lbs_weights=NS(refine=False))


def test_checkpoint_round_trip_with_subdivision(tmp_path):
    body = syn.icosphere_body(1)
    m = Model(_cfg(), body, device="cpu")
    m.subdivide()
    with torch.no_grad():
        m.appearance.uniform_(0, 1); m.so3.normal_(0, 0.1); m.vertices.add_(0.01)
    opt = torch.optim.Adam(m.get_param_groups(NS(lr=NS(lbs_weights=1e-4, appearance=1e-3, canonical_geometry=1e-3, canonical_geometry_xyz=1e-4, shadow=1e-3))))
    path = os.path.join(tmp_path, "checkpoints", "iter_1000.pt")
    formats.save_checkpoint(path, 1000, m, opt)
    sd = torch.load(path, map_location="cpu")["network"]
    # the reference's key names
    for key in ("faces", "target_edge_length", "lbs_weights", "vertices", "so3", "scale", "appearance_module.appearance", "appearance_module.bg_col",
                "shadow_module.block_mlps.0.weight"):
        assert key in sd, key
    fresh = Model(_cfg(), body, device="cpu")                      # not subdivided yet: resume logic of train.py:282-284
    nxt = formats.load_checkpoint(path, fresh, subdivide_iters=[500])
    assert nxt == 1001 and fresh.faces.shape[0] == m.faces.shape[0]
    for a, b in ((fresh.appearance, m.appearance), (fresh.so3, m.so3), (fresh.vertices, m.vertices), (fresh.lbs_weights, m.lbs_weights)):
        assert torch.equal(a, b)
    assert torch.equal(fresh.target_edge_length, m.target_edge_length)
    auto = Model(_cfg(), body, device="cpu")                       # without the iteration list: subdivide until the face counts match
    formats.load_reference_state_dict(auto, sd)
    assert torch.equal(auto.appearance, m.appearance)


def test_reference_layout_optimizer_state_and_foreign_vertex_numbering_load(tmp_path):
    """A checkpoint as the reference writes it: (i) its Adam state has the `lbs_weights` group first (models/model.py:306-309),
    (ii) its subdivided mesh numbers the midpoints differently from this package's subdivide (trimesh's unique_rows order)."""
    body = syn.icosphere_body(1)
    lr = NS(lr=NS(lbs_weights=1e-4, appearance=1e-3, canonical_geometry=1e-3, canonical_geometry_xyz=1e-4, shadow=1e-3))
    m = Model(_cfg(), body, device="cpu")
    m.subdivide()
    # renumber the vertices (a permutation that keeps the original vertices first, like a different midpoint order)
    N0 = body["canonical_vertex"].shape[0]
    N = m.vertices.shape[1]
    g = torch.Generator().manual_seed(0)
    perm = torch.cat([torch.arange(N0), N0 + torch.randperm(N - N0, generator=g)])      # new index -> old index
    inv = torch.empty_like(perm); inv[perm] = torch.arange(N)
    sd = formats.to_reference_state_dict(m)
    sd["vertices"] = sd["vertices"][:, perm].contiguous()
    sd["lbs_weights"] = sd["lbs_weights"][:, perm].contiguous()
    sd["faces"] = inv[sd["faces"]]
    groups = m.get_param_groups(lr)
    assert [g_["name"] for g_ in groups][:5] == ["lbs_weights", "appearance", "canonical_geometry_xyz", "canonical_geometry", "canonical_geometry"]
    opt = torch.optim.Adam(groups)
    for p_ in m.parameters():
        p_.grad = torch.ones_like(p_)
    opt.step()                                                        # creates Adam state; the lbs_weights buffer gets none (no grad)
    osd = opt.state_dict()
    assert len(osd["param_groups"]) == 6 and osd["param_groups"][0]["name"] == "lbs_weights" and 0 not in osd["state"]
    path = os.path.join(tmp_path, "iter_60000.pt")
    torch.save({"iter": 60000, "network": sd, "optimizer": osd}, path)
    fresh = Model(_cfg(), body, device="cpu")
    opt2 = torch.optim.Adam(fresh.get_param_groups(lr))
    # (the reference rebuilds its optimizer after each subdivision, train.py:341-346; do the same around the load)
    ckpt = torch.load(path, map_location="cpu")
    formats.load_reference_state_dict(fresh, ckpt["network"])
    assert torch.equal(fresh.faces, sd["faces"]) and torch.equal(fresh.vertices, sd["vertices"]) and torch.equal(fresh.lbs_weights, sd["lbs_weights"])
    assert fresh.so3.shape == m.so3.shape and torch.equal(fresh.appearance, sd["appearance_module.appearance"])
    opt2 = torch.optim.Adam(fresh.get_param_groups(lr))
    opt2.load_state_dict(ckpt["optimizer"])                           # same number of groups / parameters per group as the reference's
    assert opt2.state_dict()["param_groups"][1]["lr"] == 1e-3


def test_dataset_directory_round_trip(tmp_path):
    formats.write_synthetic_dataset(str(tmp_path), n_frames=3, img=32, level=1)
    ds = formats.ReferenceDataset(str(tmp_path), bgcolor=[255.0, 0.0, 0.0])
    assert len(ds) == 3
    info = ds.get_canonical_info()
    assert info["canonical_vertex"].shape[1] == 3 and info["canonical_lbs_weights"].shape[1] == 24
    fr = ds[1]
    for key, shape in (("K", (3, 3)), ("E", (4, 4)), ("dst_Rs", (24, 3, 3)), ("dst_Ts", (24, 3)), ("cnl_gtfms", (24, 4, 4)), ("dst_posevec", (69,))):
        assert fr[key].shape == shape, key
    ref = syn.make_frame(1, 32)                                    # same pose / skeleton -> same bone transforms
    pose = syn.random_pose(1)
    Rs, Ts = syn.pose_to_body_RTs(pose, syn.TPOSE_JOINTS)
    np.testing.assert_allclose(fr["dst_Rs"], Rs); np.testing.assert_allclose(fr["dst_Ts"], Ts)
    if "target_rgbs" in fr:
        assert fr["target_rgbs"].shape == (32, 32, 3) and fr["target_masks"].shape == (32, 32)
        assert np.allclose(fr["target_rgbs"][0, 0], [1.0, 0.0, 0.0])          # outside the mask: the background colour
    # Rh / Th are folded into the extrinsics (camera_util.py:111-131)
    E2, g = formats.apply_global_tfm_to_camera(np.eye(4), np.array([0.0, 0.3, 0.0]), np.array([0.1, 0.0, 0.0]))
    np.testing.assert_allclose(E2 @ g, np.eye(4), atol=1e-12)


def test_dataset_reader_undistorts_resizes_and_crops_like_the_reference(tmp_path):
    """dataset/train.py:138-200, 239-256: undistort both images when the camera has `distortions`, composite on the background in
    floating point, LANCZOS4 for the image and LINEAR for the mask (target_size or resize_img_scale), K scaled, crop shifts the
    principal point.  (The image operations themselves: tests/test_imageops.py.)"""
    pytest.importorskip("PIL")
    import pickle, os
    formats.write_synthetic_dataset(str(tmp_path), n_frames=2, img=64, level=1)
    plain = formats.ReferenceDataset(str(tmp_path), bgcolor=[0.0, 255.0, 0.0])[0]
    # zero distortion coefficients: the undistortion branch runs and changes nothing
    with open(os.path.join(tmp_path, "cameras.pkl"), "rb") as f:
        cams = pickle.load(f)
    for c in cams.values():
        c["distortions"] = np.zeros(5)
    with open(os.path.join(tmp_path, "cameras.pkl"), "wb") as f:
        pickle.dump(cams, f)
    same = formats.ReferenceDataset(str(tmp_path), bgcolor=[0.0, 255.0, 0.0])[0]
    assert np.array_equal(same["target_rgbs"], plain["target_rgbs"]) and np.array_equal(same["target_masks"], plain["target_masks"])
    # half size through both spellings: K scales, shapes follow, the mask stays a mask, the background stays the background
    for kw in (dict(target_size=(32, 32)), dict(resize_img_scale=(0.5, 0.5))):
        half = formats.ReferenceDataset(str(tmp_path), bgcolor=[0.0, 255.0, 0.0], **kw)[0]
        assert half["target_rgbs"].shape == (32, 32, 3) and half["target_masks"].shape == (32, 32) and half["target_rgbs"].dtype == np.float32
        np.testing.assert_allclose(half["K"][:2], plain["K"][:2] * 0.5, rtol=1e-6)
        assert half["target_masks"].min() >= 0.0 and half["target_masks"].max() <= 1.0 and half["target_masks"][16, 16] == 1.0
        np.testing.assert_allclose(half["target_rgbs"][0, 0], [0.0, 1.0, 0.0], atol=1e-5)
    # without the pixels the intrinsics still follow resize_img_scale (dataset/train.py:239-244); a size that needs the image is refused
    noimg = formats.ReferenceDataset(str(tmp_path), bgcolor=[0.0, 255.0, 0.0], load_images=False, resize_img_scale=(0.5, 0.5))[0]
    np.testing.assert_allclose(noimg["K"][:2], plain["K"][:2] * 0.5, rtol=1e-6)
    assert "target_rgbs" not in noimg
    for kw in (dict(target_size=(32, 32)), dict(crop_size=(40, 32))):
        with pytest.raises(ValueError):
            formats.ReferenceDataset(str(tmp_path), load_images=False, **kw)
    # a real distortion moves pixels; the crop keeps at least 20 mask units and moves the principal point by the crop offset
    for c in cams.values():
        c["distortions"] = np.array([0.3, 0.0, 0.0, 0.0, 0.0])
    with open(os.path.join(tmp_path, "cameras.pkl"), "wb") as f:
        pickle.dump(cams, f)
    np.random.seed(0)
    warped = formats.ReferenceDataset(str(tmp_path), bgcolor=[0.0, 255.0, 0.0], crop_size=(40, 32))[0]
    assert warped["target_rgbs"].shape == (32, 40, 3) and warped["target_masks"].sum() >= 20
    assert plain["K"][0, 2] - warped["K"][0, 2] >= 0 and plain["K"][1, 2] - warped["K"][1, 2] >= 0
    full = formats.ReferenceDataset(str(tmp_path), bgcolor=[0.0, 255.0, 0.0])[0]
    assert not np.array_equal(full["target_masks"], plain["target_masks"])


def test_shadow_module_and_color_consistency_match_reference_goldens(golden_dir):
    """ShadowModule (positional encoding order, layer layout, state-dict names) and the colour-consistency formula against
    outputs of the reference's own classes (scripts/make_goldens.py)."""
    from gomavatar_amd.model import ShadowModule
    from gomavatar_amd import train_util as tu
    g = np.load(os.path.join(golden_dir, "shadow_color.npz"))
    m = ShadowModule(multires=6, mlp_width=128, mlp_depth=3, skips=(4,))
    sd = {k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w_")}
    m.load_state_dict(sd)                                   # same key names as the reference module
    with torch.no_grad():
        out = m(torch.from_numpy(g["normals"]))
    np.testing.assert_allclose(out.numpy(), g["shadow"], rtol=1e-5, atol=1e-6)
    cc = tu.mesh_color_consistency(torch.from_numpy(g["colors"]), torch.from_numpy(g["pairs"]))
    np.testing.assert_allclose(float(cc), float(g["color_consistency"]), rtol=1e-6)


def test_ndc_T_world_matches_reference_golden(golden_dir):
    from gomavatar_amd.mesh_renderer import ndc_T_world
    from oracle import mesh as om
    g = np.load(os.path.join(golden_dir, "ndc_T_world.npz"))
    pts, K, E = torch.from_numpy(g["pts"]), torch.from_numpy(g["K"])[None], torch.from_numpy(g["E"])[None]
    for (h, w) in ((512, 512), (384, 512), (512, 384)):
        np.testing.assert_allclose(ndc_T_world(pts, K, E, h, w).numpy(), g[f"ndc_{h}x{w}"], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(om.ndc_T_world(pts, K, E, h, w).numpy(), g[f"ndc_{h}x{w}"], rtol=1e-6, atol=1e-6)
