"""HIP geometry kernels (FK, LBS, per-face Gaussian frame) against the torch
oracle and the reference-generated goldens; fp32 tolerance 1e-5 relative."""
import os

import numpy as np
import pytest
import torch

from oracle import geometry as og
from helpers import body_scene

pytestmark = pytest.mark.gpu


def _cuda(t):
    return t.detach().clone().cuda()


def test_fk_and_lbs_match_reference_goldens(golden_dir):
    from gomavatar_amd import geometry as G
    g = np.load(os.path.join(golden_dir, "geometry_fk_lbs.npz"))
    xyz = torch.from_numpy(g["xyz"]).T.contiguous()[None].cuda()
    w = torch.from_numpy(g["lbs_weights25"]).cuda()
    for f in (0, 1, 2):
        cnl, dR, dT = (torch.from_numpy(g[f"f{f}_{k}"])[None].cuda() for k in ("cnl_gtfms", "dst_Rs", "dst_Ts"))
        R, T = G.get_global_RTs(cnl, dR, dT)
        assert R.shape == (1, 24, 3, 3) and T.shape == (1, 24, 3)
        np.testing.assert_allclose(R.cpu().numpy(), g[f"f{f}_global_Rs"], atol=2e-6)
        np.testing.assert_allclose(T.cpu().numpy(), g[f"f{f}_global_Ts"], atol=2e-6)
        v = G.apply_lbs(xyz, R, T, w)
        np.testing.assert_allclose(v.cpu().numpy(), g[f"f{f}_v_obs"], atol=2e-6)
    R, T = G.get_global_RTs(torch.from_numpy(g["gen_cnl_gtfms"])[None].cuda(), torch.from_numpy(g["f1_dst_Rs"])[None].cuda(),
                            torch.from_numpy(g["f1_dst_Ts"])[None].cuda())
    np.testing.assert_allclose(R.cpu().numpy(), g["gen_global_Rs"], atol=5e-6)
    np.testing.assert_allclose(T.cpu().numpy(), g["gen_global_Ts"], atol=5e-6)


def test_steiner_cov_matches_reference_golden(golden_dir):
    """cov = A A^T for so3 = 0, scale = 1 (the reference's initial state) with A from the reference's own function."""
    from gomavatar_amd import geometry as G
    g = np.load(os.path.join(golden_dir, "geometry_steiner.npz"))
    fk = np.load(os.path.join(golden_dir, "geometry_fk_lbs.npz"))
    v_obs = torch.from_numpy(fk["f1_v_obs"][0]).cuda()
    faces = torch.from_numpy(fk["faces"])
    topo = G.MeshTopology(faces, v_obs.shape[1], device="cuda")
    F = faces.shape[0]
    xyz, cov6 = G.face_gaussians(v_obs, torch.zeros(3, F).cuda(), torch.ones(3, F).cuda(), topo, 1e-3)
    A = torch.from_numpy(g["A"]).double()
    ref = og.pack_cov6(A @ A.transpose(1, 2)).numpy()
    np.testing.assert_allclose(cov6.cpu().numpy(), ref, rtol=2e-5, atol=2e-6 * np.abs(ref).max())
    np.testing.assert_allclose(xyz.cpu().numpy(), g["tri"].mean(1), atol=1e-6)


@pytest.mark.parametrize("subdiv", [0, 1, 2])     # 13 776 / 55 104 (the metric size) / 220 416 faces
@pytest.mark.parametrize("pose_grad", [False, True])
def test_fused_geometry_forward_backward_vs_oracle(pose_grad, subdiv):
    from gomavatar_amd import geometry as G
    sc = body_scene(subdiv, frame=1, img=128)
    p, fr = sc["params"], sc["frame"]
    F = sc["faces"].shape[0]
    N = p["vertices"].shape[1]
    torch.manual_seed(0)
    wx, wc, wv = torch.randn(F, 3), torch.randn(F, 6) * 1e3, torch.randn(3, N)

    def run(vert, so3, scale, dR, dT, fn_dev):
        return fn_dev(vert, so3, scale, dR, dT)

    # oracle (fp64 for a tight reference)
    o = [p["vertices"].double().requires_grad_(), p["so3"].double().requires_grad_(), p["scale"].double().requires_grad_(),
         fr["dst_Rs"].double().requires_grad_(), fr["dst_Ts"].double().requires_grad_()]
    Rs, Ts = og.fk_global_RTs(fr["cnl_gtfms"].double(), o[3], o[4])
    v_obs = og.lbs(o[0][None], Rs, Ts, sc["lbs_weights"].double())[0]
    xyz, cov = og.face_gaussians(v_obs, sc["faces"], o[1], o[2], 1e-3)
    cov6 = og.pack_cov6(cov)
    ((xyz * wx.double()).sum() + (cov6 * wc.double()).sum() + (v_obs * wv.double()).sum()).backward()

    topo = G.MeshTopology(sc["faces"], N, device="cuda")
    h = [_cuda(p["vertices"]).requires_grad_(), _cuda(p["so3"]).requires_grad_(), _cuda(p["scale"]).requires_grad_(),
         _cuda(fr["dst_Rs"]).requires_grad_(pose_grad), _cuda(fr["dst_Ts"]).requires_grad_(pose_grad)]
    hx, hc, hv = G.posed_face_gaussians(h[0], h[1], h[2], h[3], h[4], fr["cnl_gtfms"].cuda(), sc["lbs_weights"].cuda(), topo, 1e-3)
    np.testing.assert_allclose(hv.detach().cpu().numpy(), v_obs.detach().numpy(), atol=2e-6)
    np.testing.assert_allclose(hx.detach().cpu().numpy(), xyz.detach().numpy(), atol=2e-6)
    ref6 = cov6.detach().numpy()
    np.testing.assert_allclose(hc.detach().cpu().numpy(), ref6, rtol=3e-4, atol=1e-5 * np.abs(ref6).max())
    ((hx * wx.cuda()).sum() + (hc * wc.cuda()).sum() + (hv * wv.cuda()).sum()).backward()
    names = ["vertices", "so3", "scale", "dst_Rs", "dst_Ts"]
    for i in range(5 if pose_grad else 3):
        got, ref = h[i].grad.cpu().numpy().astype(np.float64), o[i].grad.numpy()
        scale = np.abs(ref).max()
        err = np.abs(got - ref)
        assert err.max() <= 2e-3 * scale and np.median(err) <= 2e-5 * scale, (names[i], err.max(), np.median(err), scale)
    if not pose_grad:
        assert h[3].grad is None
    # no float atomics anywhere on this path (the pose gradient included): a second run is bitwise identical
    h2 = [_cuda(p["vertices"]).requires_grad_(), _cuda(p["so3"]).requires_grad_(), _cuda(p["scale"]).requires_grad_(),
          _cuda(fr["dst_Rs"]).requires_grad_(pose_grad), _cuda(fr["dst_Ts"]).requires_grad_(pose_grad)]
    hx2, hc2, hv2 = G.posed_face_gaussians(h2[0], h2[1], h2[2], h2[3], h2[4], fr["cnl_gtfms"].cuda(), sc["lbs_weights"].cuda(), topo, 1e-3)
    ((hx2 * wx.cuda()).sum() + (hc2 * wc.cuda()).sum() + (hv2 * wv.cuda()).sum()).backward()
    for i in range(5 if pose_grad else 3):
        assert torch.equal(h[i].grad, h2[i].grad), names[i]


def test_unfused_mirrors_compose_like_the_reference():
    """get_global_RTs -> apply_lbs -> face_gaussians with autograd through each node."""
    from gomavatar_amd import geometry as G
    sc = body_scene(0, frame=0, img=128)
    p, fr = sc["params"], sc["frame"]
    N = p["vertices"].shape[1]
    topo = G.MeshTopology(sc["faces"], N, device="cuda")
    v = _cuda(p["vertices"]).requires_grad_()
    dR = _cuda(fr["dst_Rs"]).requires_grad_()
    R, T = G.get_global_RTs(fr["cnl_gtfms"].cuda(), dR, fr["dst_Ts"].cuda())
    vo = G.apply_lbs(v[None], R, T, sc["lbs_weights"].cuda())[0]
    xyz, cov6 = G.face_gaussians(vo, _cuda(p["so3"]), _cuda(p["scale"]), topo)
    (xyz.sum() + cov6.sum() * 100).backward()
    v2 = _cuda(p["vertices"]).requires_grad_()
    dR2 = _cuda(fr["dst_Rs"]).requires_grad_()
    x2, c2, _ = G.posed_face_gaussians(v2, _cuda(p["so3"]), _cuda(p["scale"]), dR2, fr["dst_Ts"].cuda(), fr["cnl_gtfms"].cuda(),
                                       sc["lbs_weights"].cuda(), topo)
    (x2.sum() + c2.sum() * 100).backward()
    assert torch.equal(xyz, x2) and torch.equal(cov6, c2)

    def rel(a, b):
        return float((a - b).abs().max()) / float(b.abs().max())
    assert rel(v.grad, v2.grad) <= 1e-5, rel(v.grad, v2.grad)
    assert rel(dR.grad, dR2.grad) <= 1e-4, rel(dR.grad, dR2.grad)   # (the vertex gradient reaches the two paths in different summation orders)
    # the rasterizer's features [appearance.T | 1] out of the face kernel (gaussian.py:49's cat) and their gradient back into the (3, F) parameter: copies, bit for bit
    F = sc["faces"].shape[0]
    app = torch.rand(3, F, device="cuda").requires_grad_()
    v3 = _cuda(p["vertices"]).requires_grad_()
    x3, c3, vo3, feat = G.posed_face_gaussians(v3, _cuda(p["so3"]), _cuda(p["scale"]), _cuda(fr["dst_Rs"]), fr["dst_Ts"].cuda(), fr["cnl_gtfms"].cuda(),
                                               sc["lbs_weights"].cuda(), topo, appearance=app)
    assert torch.equal(x3, x2) and torch.equal(c3, c2) and torch.equal(vo3, vo)
    assert torch.equal(feat, torch.cat([app.detach().T, torch.ones(F, 1, device="cuda")], 1))
    wf = torch.randn(F, 4, device="cuda")
    ((feat * wf).sum() + x3.sum() + c3.sum() * 100).backward()
    assert torch.equal(app.grad, wf[:, :3].T.contiguous()) and torch.equal(v3.grad, v2.grad)
