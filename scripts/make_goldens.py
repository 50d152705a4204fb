#!/usr/bin/env python
"""Generates tests/golden/*.npz by IMPORTING the reference in the build
container (it cannot travel to the GPU box).  Inputs come from our own seeded
generators; outputs from the reference's functions:

  utils/body_util.py      get_global_RTs, apply_lbs, body_pose_to_body_RTs,
                          get_canonical_global_tfms, _rvec_to_rmtx
  models/model.py         get_transformation_from_triangle_steiner
  utils/camera_util.py    get_camrot, focal2fov
  models/modules/renderer/gaussian.py   Renderer.forward's camera block is
                          exercised through a recording stub of
                          diff_gaussian_rasterization (captures the settings
                          and tensors the reference would hand to the CUDA ext)

Third-party modules that are absent here (pytorch3d, cv2, seaborn, trimesh,
diff_gaussian_rasterization) are replaced by inert stubs in sys.modules; no
stubbed function contributes to any golden value except where stated.

Run:  python scripts/make_goldens.py   (needs /root/reference)
"""
import os
import sys
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, REPO)
OUT = os.path.join(REPO, "tests", "golden")


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install_stubs(record):
    for n in ["seaborn", "cv2", "trimesh", "trimesh.remesh", "pytorch3d", "pytorch3d.ops", "pytorch3d.structures",
              "pytorch3d.transforms", "pytorch3d.transforms.so3", "pytorch3d.loss", "pytorch3d.loss.chamfer",
              "pytorch3d.ops.knn", "pytorch3d.renderer", "pytorch3d.renderer.mesh", "pytorch3d.renderer.mesh.shader",
              "pytorch3d.renderer.blending", "pytorch3d.renderer.mesh.shading", "pytorch3d.renderer.mesh.rasterizer"]:
        _stub(n)
    sys.modules["trimesh.remesh"].faces_to_edges = None
    sys.modules["trimesh.remesh"].grouping = None
    sys.modules["pytorch3d.structures"].Meshes = object
    sys.modules["pytorch3d.transforms.so3"].so3_exp_map = None
    sys.modules["pytorch3d.transforms.so3"].so3_log_map = None
    sys.modules["pytorch3d.loss.chamfer"].chamfer_distance = None
    sys.modules["pytorch3d.ops.knn"].knn_points = None

    class Settings:  # records the 12 keyword fields (gaussian.py:53-66)
        def __init__(self, **kw):
            self.__dict__.update(kw)
            record["settings"] = kw

    class Rasterizer(torch.nn.Module):
        def __init__(self, s):
            super().__init__()
            self.raster_settings = s

        def forward(self, **kw):
            record.setdefault("calls", []).append({k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in kw.items()})
            rs = self.raster_settings
            return torch.zeros(3, rs.image_height, rs.image_width), None

    _stub("diff_gaussian_rasterization", GaussianRasterizationSettings=Settings, GaussianRasterizer=Rasterizer)


def main():
    os.makedirs(OUT, exist_ok=True)
    record = {}
    install_stubs(record)
    sys.path.insert(0, REF)
    from utils import body_util as ref_bu  # noqa: E402
    from gomavatar_amd import synthetic as syn  # noqa: E402

    # ---------------- FK / LBS ----------------
    body = syn.icosphere_body(2)
    joints = syn.TPOSE_JOINTS
    out = {}
    for frame in (0, 1, 2):
        pose = syn.random_pose(frame)
        Rs_ref, Ts_ref = ref_bu.body_pose_to_body_RTs(pose, joints)
        cnl_ref = ref_bu.get_canonical_global_tfms(joints)
        Rs_syn, Ts_syn = syn.pose_to_body_RTs(pose, joints)
        assert np.allclose(Rs_ref, Rs_syn, atol=1e-6) and np.allclose(Ts_ref, Ts_syn, atol=0)
        assert np.allclose(cnl_ref, syn.canonical_global_tfms(joints), atol=1e-7)
        cnl = torch.from_numpy(cnl_ref)[None]
        dR = torch.from_numpy(Rs_ref)[None]
        dT = torch.from_numpy(Ts_ref)[None]
        gR, gT = ref_bu.get_global_RTs(cnl, dR, dT)
        xyz = torch.from_numpy(body["canonical_vertex"]).T.contiguous()[None]
        w = torch.from_numpy(body["canonical_lbs_weights"]).T
        w25 = torch.cat([w, torch.zeros(1, w.shape[1])], 0).contiguous()
        v_obs = ref_bu.apply_lbs(xyz, gR, gT, w25)
        out.update({f"f{frame}_pose": pose, f"f{frame}_dst_Rs": Rs_ref, f"f{frame}_dst_Ts": Ts_ref, f"f{frame}_cnl_gtfms": cnl_ref,
                    f"f{frame}_global_Rs": gR.numpy(), f"f{frame}_global_Ts": gT.numpy(), f"f{frame}_v_obs": v_obs.numpy()})
    out["xyz"] = body["canonical_vertex"]
    out["faces"] = body["faces"]
    out["lbs_weights25"] = w25.numpy()
    # a non-trivial canonical transform set (rotated joints) so the general 4x4 inverse is exercised
    rng = np.random.default_rng(7)
    cnl_gen = syn.canonical_global_tfms(joints).copy()
    for j in range(24):
        cnl_gen[j, :3, :3] = syn.rodrigues(rng.normal(0, 0.3, 3)).astype(np.float32)
    gR, gT = ref_bu.get_global_RTs(torch.from_numpy(cnl_gen)[None], torch.from_numpy(out["f1_dst_Rs"])[None], torch.from_numpy(out["f1_dst_Ts"])[None])
    out.update(gen_cnl_gtfms=cnl_gen, gen_global_Rs=gR.numpy(), gen_global_Ts=gT.numpy())
    np.savez_compressed(os.path.join(OUT, "geometry_fk_lbs.npz"), **out)

    # ---------------- Steiner frame ----------------
    from models.model import get_transformation_from_triangle_steiner as ref_steiner  # noqa: E402
    v_obs = torch.from_numpy(out["f1_v_obs"][0]).T  # (N,3)
    faces = torch.from_numpy(body["faces"])
    tri = v_obs[faces.reshape(-1)].reshape(-1, 3, 3)
    A = ref_steiner(tri, 1e-3)
    np.savez_compressed(os.path.join(OUT, "geometry_steiner.npz"), tri=tri.numpy(), A=A.numpy(), sigma=np.float32(1e-3))

    # ---------------- camera conventions ----------------
    sys.modules["cv2"].Rodrigues = None
    from utils import camera_util as ref_cu  # noqa: E402
    campos = np.array([0.0, 1.2, 8.0], dtype="float32")
    camrot = ref_cu.get_camrot(campos.copy(), lookat=np.array([0, 1.2, 0.0]), inv_camera=True)
    E = np.eye(4, dtype="float32")
    E[:3, :3] = camrot
    E[:3, 3] = -camrot.dot(campos)
    K = np.eye(3, dtype="float32")
    K[0, 0] = K[1, 1] = 1250.0
    K[:2, 2] = 256.0
    Ks, Es = syn.look_at_camera(512, yaw=0.0, radius=8.0, focal=1250.0, target_y=1.2)
    assert np.allclose(K, Ks) and np.allclose(E, Es, atol=1e-6), (E, Es)

    # gaussian.py::Renderer.forward camera block, through the recording stub
    from models.modules.renderer.gaussian import Renderer as RefRenderer  # noqa: E402
    cfg = types.SimpleNamespace(img_size=(512, 512), feat_dim=4)
    r = RefRenderer(cfg, None)
    Fn = 7
    g = torch.Generator().manual_seed(0)
    xyz = torch.randn(1, 3, Fn, generator=g)
    feats = torch.rand(1, Fn, 3, generator=g)
    opac = torch.ones(1, Fn, 1)
    cov = torch.randn(1, Fn, 3, 3, generator=g)
    cov = cov @ cov.transpose(-1, -2)
    Kt, Et = torch.from_numpy(Ks)[None], torch.from_numpy(Es)[None]
    rgb, mask = r(xyz, feats, opac, Kt, Et, bg_col=torch.zeros(4), skeleton_info={"cov": cov})
    s = record["settings"]
    calls = record["calls"]
    assert len(calls) == 2 and rgb.shape == (1, 512, 512, 3) and mask.shape == (1, 512, 512)
    np.savez_compressed(
        os.path.join(OUT, "renderer_camera.npz"), K=Ks, E=Es, tanfovx=np.float64(s["tanfovx"]), tanfovy=np.float64(s["tanfovy"]),
        viewmatrix=s["viewmatrix"].numpy(), projmatrix=s["projmatrix"].numpy(), campos=s["campos"].numpy(), bg=s["bg"].numpy(),
        image_height=s["image_height"], image_width=s["image_width"], xyz=xyz.numpy(), feats=feats.numpy(), cov=cov.numpy(),
        call0_means3D=calls[0]["means3D"].numpy(), call0_colors=calls[0]["colors_precomp"].numpy(), call0_cov6=calls[0]["cov3D_precomp"].numpy(),
        call1_colors=calls[1]["colors_precomp"].numpy(), call0_opacities=calls[0]["opacities"].numpy(),
        focal2fov=np.float64(ref_cu.focal2fov(1250.0, 512)))
    # ---------------- LPIPS-VGG ----------------
    # lin weights (data): LPIPS v0.1 VGG, 5 x (1,C,1,1) -> product data file
    sd = torch.load(os.path.join(REF, "utils", "lpips", "weights", "v0.1", "vgg.pth"), map_location="cpu")
    lin = {f"lin{k}": sd[f"lin{k}.model.1.weight"].numpy().reshape(-1).astype(np.float32) for k in range(5)}
    os.makedirs(os.path.join(REPO, "gomavatar_amd", "data"), exist_ok=True)
    np.savez(os.path.join(REPO, "gomavatar_amd", "data", "lpips_vgg_lin_v0.1.npz"), **lin)
    # the reference's own LPIPS class with torchvision's ImageNet trunk replaced by our seeded random VGG16
    # (same layer indices as torchvision's vgg16().features, which pretrained_networks.py:99-112 slices by index)
    from gomavatar_amd import lpips as our_lpips  # noqa: E402  (seeded_trunk only: weights are data)
    wb = our_lpips.seeded_trunk(0)
    feats, ci = [], 0
    cfg = [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M"]
    cin = 3
    for v in cfg:
        if v == "M":
            feats.append(torch.nn.MaxPool2d(kernel_size=2, stride=2))
        else:
            conv = torch.nn.Conv2d(cin, v, kernel_size=3, padding=1)
            with torch.no_grad():
                conv.weight.copy_(wb[2 * ci]); conv.bias.copy_(wb[2 * ci + 1])
            feats += [conv, torch.nn.ReLU(inplace=False)]
            cin = v; ci += 1
    tv = _stub("torchvision")
    tvm = _stub("torchvision.models")
    tv.models = tvm
    tvm.vgg16 = lambda pretrained=False: types.SimpleNamespace(features=torch.nn.Sequential(*feats))
    from utils.lpips.lpips import LPIPS as RefLPIPS  # noqa: E402
    ref = RefLPIPS(net="vgg", verbose=False)
    g = torch.Generator().manual_seed(5)
    in0 = torch.rand(2, 3, 64, 48, generator=g) * 2 - 1
    in1 = (in0 + 0.3 * torch.randn(2, 3, 64, 48, generator=g)).clamp(-1, 1)
    with torch.no_grad():
        val, res = ref(in0, in1, retPerLayer=True)
    np.savez_compressed(os.path.join(OUT, "lpips_vgg.npz"), in0=in0.numpy(), in1=in1.numpy(), trunk_seed=np.int64(0),
                        val=val.numpy(), **{f"res{k}": res[k].numpy() for k in range(5)})
    # ---------------- shadow MLP + colour consistency (pure torch modules of the reference) ----------------
    from models.modules.shadow_module import ShadowModule as RefShadow  # noqa: E402
    from utils.network_util import mesh_color_consistency as ref_cc  # noqa: E402
    torch.manual_seed(3)
    scfg = types.SimpleNamespace(condition_code_size=162, mlp_width=128, mlp_depth=3, skips=[4], multires=6, i_embed=0)
    rs = RefShadow(scfg)
    with torch.no_grad():
        rs.block_mlps[-1].weight.normal_(0, 0.3)
    gs = torch.Generator().manual_seed(4)
    nrm = torch.randn(1, 50, 3, generator=gs)
    with torch.no_grad():
        sh_out = rs(nrm)
    col = torch.rand(40, 3, generator=gs)
    conn = torch.randint(0, 40, (60, 2), generator=gs)
    np.savez_compressed(os.path.join(OUT, "shadow_color.npz"), normals=nrm.numpy(), shadow=sh_out.numpy(), colors=col.numpy(), pairs=conn.numpy(),
                        color_consistency=ref_cc(col, conn).numpy(), **{"w_" + k: v.numpy() for k, v in rs.state_dict().items()})
    # ---------------- uniform Laplacian smoothing (utils/network_util.py:669-792, the reference's own copy) ----------------
    # The function only needs verts_packed / faces_packed / laplacian_packed / bookkeeping from its `meshes` argument: a minimal
    # stand-in supplies them.  L = D^-1 A - I is the restated upstream piece (PyTorch3D 0.7.0 Meshes.laplacian_packed, uniform);
    # what the golden pins is the reference's own arithmetic on it: L.mm(V), norm(dim=1) ** 2, un-weighted mean.
    from utils.network_util import mesh_laplacian_smoothing as ref_lap  # noqa: E402
    from oracle import mesh_losses as oml  # noqa: E402  (edge list + dense L only)
    ico = syn.icosphere_body(1)
    gl = torch.Generator().manual_seed(8)
    lv = torch.from_numpy(ico["canonical_vertex"]).double() + 0.01 * torch.randn(ico["canonical_vertex"].shape, generator=gl, dtype=torch.float64)
    lf = torch.from_numpy(ico["faces"]).long()
    ledges, _ = oml.edges_of(lf, lv.shape[0])
    Ld = oml.uniform_laplacian(ledges, lv.shape[0]).to_sparse()

    class MeshesStandIn:
        device = lv.device
        def __init__(self, v): self.v = v
        def isempty(self): return False
        def __len__(self): return 1
        def verts_packed(self): return self.v
        def faces_packed(self): return lf
        def num_verts_per_mesh(self): return torch.tensor([self.v.shape[0]])
        def verts_packed_to_mesh_idx(self): return torch.zeros(self.v.shape[0], dtype=torch.long)
        def laplacian_packed(self): return Ld

    lvg = lv.clone().requires_grad_()
    lval = ref_lap(MeshesStandIn(lvg))
    lval.backward()
    np.savez_compressed(os.path.join(OUT, "mesh_losses.npz"), verts=lv.numpy(), faces=lf.numpy(), edges=ledges.numpy(),
                        laplacian=lval.detach().numpy(), laplacian_grad=lvg.grad.numpy())
    # ---------------- ndc_T_world (utils/pc_util.py:30-46): square and both non-square cases ----------------
    from utils import pc_util as ref_pc  # noqa: E402
    gn = torch.Generator().manual_seed(6)
    pts = torch.randn(1, 3, 40, generator=gn) * 0.5 + torch.tensor([0.0, 1.2, 0.0]).view(1, 3, 1)
    nd = {"pts": pts.numpy(), "K": Ks, "E": Es}
    for (h, w) in ((512, 512), (384, 512), (512, 384)):
        nd[f"ndc_{h}x{w}"] = ref_pc.ndc_T_world(pts, torch.from_numpy(Ks)[None], torch.from_numpy(Es)[None], h, w).numpy()
    np.savez_compressed(os.path.join(OUT, "ndc_T_world.npz"), **nd)
    print("goldens written to", OUT, sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
