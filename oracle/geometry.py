"""CPU (PyTorch, fp32/fp64) restatement of the geometry half of the hot path.

TEST INFRASTRUCTURE ONLY: only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module.

Every function cites the reference lines it follows.  The FK / LBS / Steiner
functions are PINNED: scripts/make_goldens.py imports the reference's own
`utils/body_util.py::get_global_RTs, apply_lbs` and
`models/model.py::get_transformation_from_triangle_steiner` in the build
container and the frozen outputs live in tests/golden/geometry_*.npz
(tests/test_oracle_geometry.py).  `so3_exp_map` restates PyTorch3D 0.7.0
(third-party, not vendored in the reference: reference README.md:21,32-33;
call site models/model.py:229) from SURVEY.md Appendix B -- parity unpinned
for that one function (checked against torch.matrix_exp instead).
"""
from __future__ import annotations

import math

import numpy as np
import torch

SMPL_PARENT = (-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21)


def fk_global_RTs(cnl_gtfms: torch.Tensor, dst_Rs: torch.Tensor, dst_Ts: torch.Tensor):
    """utils/body_util.py:612-638 (+ _construct_G_tensor :591-609).
    (B,24,4,4),(B,24,3,3),(B,24,3) -> skinning R (B,24,3,3), T (B,24,3)."""
    B, J = dst_Rs.shape[:2]
    local = torch.zeros(B, J, 4, 4, dtype=dst_Rs.dtype, device=dst_Rs.device)
    local[:, :, :3, :3] = dst_Rs
    local[:, :, :3, 3] = dst_Ts
    local[:, :, 3, 3] = 1.0
    chain = [local[:, 0]]
    for i in range(1, J):
        chain.append(torch.matmul(chain[SMPL_PARENT[i]], local[:, i]))
    G = torch.stack(chain, dim=1)
    f = torch.matmul(G, torch.inverse(cnl_gtfms))
    return f[:, :, :3, :3], f[:, :, :3, 3]


def lbs(xyz: torch.Tensor, Rs: torch.Tensor, Ts: torch.Tensor, lbs_weights: torch.Tensor) -> torch.Tensor:
    """utils/body_util.py:641-644.  xyz (B,3,N), weights (25,N) (last row =
    background, dropped) -> (B,3,N)."""
    moved = torch.einsum("bjik,bkn->bjin", Rs, xyz) + Ts[:, :, :, None]
    return torch.sum(moved * lbs_weights[:-1][None, :, None, :], dim=1)


def so3_exp(v: torch.Tensor, eps: float = 1e-4) -> torch.Tensor:
    """PyTorch3D 0.7.0 so3_exp_map: theta^2 clamped at eps."""
    n2 = (v * v).sum(-1)
    th = torch.clamp(n2, min=eps).sqrt()
    inv = 1.0 / th
    f1 = inv * th.sin()
    f2 = inv * inv * (1.0 - th.cos())
    x, y, z = v.unbind(-1)
    o = torch.zeros_like(x)
    K = torch.stack([o, -z, y, z, o, -x, -y, x, o], dim=-1).reshape(*v.shape[:-1], 3, 3)
    eye = torch.eye(3, dtype=v.dtype, device=v.device)
    return f1[..., None, None] * K + f2[..., None, None] * (K @ K) + eye


def rodrigues_module(rvec: torch.Tensor) -> torch.Tensor:
    """utils/network_util.py:64-92 RodriguesModule, (B,3) -> (B,3,3): theta = sqrt(1e-5 + |r|^2), k = r / theta (shorter than a unit
    vector near zero), R = k k^T + (I - diag(k^2))-style cos terms + sin [k]x exactly as the reference writes them.  Used by
    models/model.py:218-221 (global_R) and train_pose.py.  PINNED: tests/golden/pose_modules.npz rod_* (tests/test_oracle_geometry.py)."""
    theta = torch.sqrt(1e-5 + torch.sum(rvec ** 2, dim=1))
    k = rvec / theta[:, None]
    c, s = torch.cos(theta), torch.sin(theta)
    kx, ky, kz = k[:, 0], k[:, 1], k[:, 2]
    return torch.stack((kx ** 2 + (1. - kx ** 2) * c, kx * ky * (1. - c) - kz * s, kx * kz * (1. - c) + ky * s,
                        kx * ky * (1. - c) + kz * s, ky ** 2 + (1. - ky ** 2) * c, ky * kz * (1. - c) - kx * s,
                        kx * kz * (1. - c) - ky * s, ky * kz * (1. - c) + kx * s, kz ** 2 + (1. - kz ** 2) * c), dim=1).view(-1, 3, 3)


def steiner_frame(tri: torch.Tensor, sigma: float = 1e-3) -> torch.Tensor:
    """models/model.py:27-41.  tri (F,3,3) rows = corners -> A (F,3,3) with
    columns [2 a0 | 2 a1 | sigma n]."""
    c = tri.mean(dim=-2)
    f1 = 0.5 * (tri[..., 2, :] - c)
    f2 = (1.0 / (2.0 * math.sqrt(3.0))) * (tri[..., 1, :] - tri[..., 0, :])
    t0 = torch.atan2((2 * f1 * f2).sum(-1), (f1 * f1).sum(-1) - (f2 * f2).sum(-1)) / 2
    t0 = t0[..., None]
    a0 = f1 * torch.cos(t0) + f2 * torch.sin(t0)
    a1 = f1 * torch.cos(t0 + math.pi / 2) + f2 * torch.sin(t0 + math.pi / 2)
    n = torch.nn.functional.normalize(torch.cross(a0, a1, dim=-1), dim=-1) * sigma
    return torch.stack([a0 * 2, a1 * 2, n], dim=-1)


def face_gaussians(verts_obs: torch.Tensor, faces: torch.Tensor, so3: torch.Tensor, scale: torch.Tensor, sigma: float = 1e-3):
    """models/model.py:225-234.  verts_obs (3,N), faces (F,3) long, so3 (3,F),
    scale (3,F) -> xyz (F,3), cov (F,3,3)."""
    F = faces.shape[0]
    tri = verts_obs.permute(1, 0)[faces.reshape(-1)].reshape(F, 3, 3)
    xyz = tri.mean(dim=1)
    S = torch.diag_embed(scale.permute(1, 0))
    R = so3_exp(so3.permute(1, 0))
    cov_local = R @ S @ S.permute(0, 2, 1) @ R.permute(0, 2, 1)
    A = steiner_frame(tri, sigma)
    return xyz, A @ cov_local @ A.permute(0, 2, 1)


def pack_cov6(cov: torch.Tensor) -> torch.Tensor:
    """models/modules/renderer/gaussian.py:71-75."""
    return torch.stack([cov[:, 0, 0], cov[:, 0, 1], cov[:, 0, 2], cov[:, 1, 1], cov[:, 1, 2], cov[:, 2, 2]], dim=-1)


def camera_from_KE(K, E, w: int, h: int, bg=None) -> dict:
    """models/modules/renderer/gaussian.py:30-47,53-66: K (3,3), E (4,4) ->
    the rasterizer's camera dict (tanfov, viewmatrix = E^T,
    projmatrix = E^T K_ndc^T, campos)."""
    K = np.asarray(K, dtype=np.float32)
    E32 = torch.as_tensor(np.asarray(E, dtype=np.float32))
    fx, fy, px, py = float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2])
    tanfovx = math.tan(2 * math.atan(w / (2 * fx)) * 0.5)
    tanfovy = math.tan(2 * math.atan(h / (2 * fy)) * 0.5)
    znear, zfar = 0.001, 100
    K_ndc = torch.tensor(
        [[2 * fx / w, 0, (2 * px - w) / w, 0],
         [0, 2 * fy / h, (2 * py - h) / h, 0],
         [0, 0, zfar / (zfar - znear), -zfar * znear / (zfar - znear)],
         [0, 0, 1, 0]]).float()
    view = E32.T.contiguous()
    proj = (E32.T @ K_ndc.T).contiguous()
    campos = E32.T.inverse()[3, :3]
    return dict(H=h, W=w, tanfovx=tanfovx, tanfovy=tanfovy, viewmatrix=view.numpy(), projmatrix=proj.numpy(),
                campos=campos.numpy(), bg=np.zeros(4, np.float32) if bg is None else np.asarray(bg, np.float32))


def unpack(rgbs: torch.Tensor, masks: torch.Tensor, bgcolors: torch.Tensor) -> torch.Tensor:
    """train.py:53-55."""
    return rgbs * masks.unsqueeze(-1) + bgcolors[:, None, None, :] * (1 - masks).unsqueeze(-1)


def l1_losses(rgb_pred, mask_pred, rgb_gt, mask_gt):
    """train.py:101-111 (unscaled terms)."""
    return torch.mean(torch.abs(rgb_pred - rgb_gt)), torch.mean(torch.abs(mask_pred - mask_gt))


class _OracleRaster(torch.autograd.Function):
    """Autograd bridge to the C raster oracle so that whole-path gradients of
    the restatement can be taken with torch.autograd on CPU."""

    @staticmethod
    def forward(ctx, cam, means3D, cov6, colors, opacity):
        from . import raster
        dt = np.float64 if means3D.dtype == torch.float64 else np.float32
        fwd = raster.forward(cam, means3D.detach().numpy(), cov6.detach().numpy(), colors.detach().numpy(), opacity.detach().numpy(), dtype=dt)
        ctx.fwd = fwd
        ctx.tdtype = means3D.dtype
        return torch.from_numpy(fwd["color"].copy())

    @staticmethod
    def backward(ctx, g):
        from . import raster
        b = raster.backward(ctx.fwd, g.contiguous().numpy())
        t = lambda a: torch.from_numpy(a).to(ctx.tdtype)
        return None, t(b["dL_dmeans3D"]), t(b["dL_dcov6"]), t(b["dL_dcolors"]), t(b["dL_dopacity"])


def rasterize(cam: dict, means3D, cov6, colors, opacity) -> torch.Tensor:
    """(C,H,W) image through the C oracle, differentiable."""
    return _OracleRaster.apply(cam, means3D, cov6, colors, opacity)


def render_path(params: dict, frame: dict, faces: torch.Tensor, lbs_weights: torch.Tensor, img_size: int, sigma: float = 1e-3,
                global_R=None, global_T=None):
    """The whole restated render path for one frame (models/model.py:213-250 +
    gaussian.py:22-100 fused to one 4-channel pass): returns rgb (1,H,W,3),
    mask (1,H,W) and the intermediates.  params: vertices (3,N), so3 (3,F),
    scale (3,F), appearance (3,F) torch tensors (may require grad).
    global_R (3,) axis-angle / global_T (3,): models/model.py:218-221."""
    Rs, Ts = fk_global_RTs(frame["cnl_gtfms"], frame["dst_Rs"], frame["dst_Ts"])
    v_obs = lbs(params["vertices"].unsqueeze(0), Rs, Ts, lbs_weights)[0]
    if global_R is not None:
        v_obs = rodrigues_module(global_R.unsqueeze(0))[0] @ v_obs + global_T[:, None]
    xyz, cov = face_gaussians(v_obs, faces, params["so3"], params["scale"], sigma)
    cov6 = pack_cov6(cov)
    F = faces.shape[0]
    feat = torch.cat([params["appearance"].permute(1, 0), torch.ones(F, 1, dtype=xyz.dtype)], dim=-1)
    cam = camera_from_KE(frame["K"][0].numpy(), frame["E"][0].numpy(), img_size, img_size)
    opacity = torch.ones(F, dtype=xyz.dtype)
    img = rasterize(cam, xyz, cov6, feat, opacity)  # (4,H,W)
    pred = img.permute(1, 2, 0)[None]
    return pred[..., :3], pred[..., 3], dict(v_obs=v_obs, xyz=xyz, cov6=cov6, cam=cam, img=img, feat=feat)
