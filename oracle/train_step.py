"""TEST INFRASTRUCTURE ONLY (only tests/, smoke() and bench.py's cpu_baseline leg may import oracle/): CPU restatement of
the two harnesses around the hot path that SURVEY.md 8(c) names, composed from the other oracle pieces:

  * the TRAIN STEP, train.py:309-349: zero_grad -> Model.forward (models/model.py:184-303, training mode) -> unpack
    (train.py:53-55) -> compute_loss (train.py:98-163) -> backward -> Adam step -> update_lr (train.py:166-175);
  * the EVAL FRAME, eval.py:336-361: no_grad forward -> unpack on the configured background -> to_8b_image
    (utils/image_util.py:21-22: clip, x255, truncate) -> /255 -> PSNR (eval.py:101-104).

Pieces and what pins them: FK / LBS / Steiner frame (oracle/geometry.py: reference goldens), splat rasterizer
(oracle/raster_oracle.c: parity unpinned, un-vendored CUDA extension), mesh normal map + soft silhouette + vertex normals
(oracle/mesh.py: PyTorch3D restated, unpinned), shadow MLP (below: pinned by tests/golden/shadow_color.npz, recorded from the
reference's ShadowModule), Laplacian / colour consistency (oracle/mesh_losses.py: reference goldens), normal consistency
(PyTorch3D restated, unpinned), LPIPS-VGG head (oracle/lpips.py: reference golden with a seeded trunk).

scripts/make_train_goldens.py runs this in float64 and records tests/golden/train_steps.npz; tests/test_gpu_train_golden.py
replays the same steps through gomavatar_amd.model.Model + train_util.compute_loss on the GPU."""
from __future__ import annotations

import math
from typing import Dict, List

import numpy as np
import torch
import torch.nn.functional as F
from torch.utils.checkpoint import checkpoint

from . import geometry as og, mesh as om, mesh_losses as oml, lpips as olp

LR = dict(lbs_weights=0.0, appearance=0.0005, canonical_geometry=0.0005, canonical_geometry_xyz=0.0005, shadow=0.0005)   # exps/zju-mocap_377.yaml:113-119
LR_DECAY_STEPS = 100000                                                                                              # exps/zju-mocap_377.yaml:122
LOSS = dict(rgb=1.0, mask=5.0, lpips=1.0, laplacian_observation=10.0, normal_mask=1.0, normal_kernel=7, normal_consist=0.1, color_consist=0.05)
# (configs/default.yaml:101-122 + exps/zju-mocap_377.yaml:101-112; laplacian.coeff_canonical = 0)


def shadow_mlp(normals: torch.Tensor, wb: List[torch.Tensor], multires: int = 6) -> torch.Tensor:
    """shadow_module.py:16-64,108-117: PE (input, then sin / cos per frequency 2^0 .. 2^(L-1)) -> (Linear, ReLU) x 3 -> Linear ->
    sigmoid.  wb = [W1, b1, ..., W4, b4] (depth 3, the configured skip index 4 lies beyond it)."""
    out = [normals]
    for k in range(multires):
        f = 2.0 ** k
        out += [torch.sin(normals * f), torch.cos(normals * f)]
    h = torch.cat(out, -1)
    for i in range(3):
        h = F.relu(F.linear(h, wb[2 * i], wb[2 * i + 1]))
    return torch.sigmoid(F.linear(h, wb[6], wb[7]))


def subdivide_midpoint(verts: np.ndarray, faces: np.ndarray, attrs: Dict[str, np.ndarray]):
    """utils/pc_util.py:49-150 (`_subdivide`, adapted there from trimesh; called by models/model.py:145): one midpoint per unique
    edge appended behind the old vertices, every face replaced IN PLACE by its four children [v0 m0 m2] [m0 v1 m1] [m2 m1 v2]
    [m0 m1 m2] (m_k = midpoint of the face's k-th edge (v_k, v_k+1)) -> children of face f are rows 4f .. 4f+3; generic vertex
    attributes are averaged onto the midpoints.  Midpoints are numbered in the lexicographic order of their sorted (lo, hi) vertex
    pair (trimesh numbers them by first occurrence of a row hash instead: a relabelling of the new vertices, results identical up
    to that permutation -- gomavatar_amd.formats adopts a checkpoint's own numbering when it loads one)."""
    e = np.sort(np.stack([faces[:, [0, 1]], faces[:, [1, 2]], faces[:, [2, 0]]], 1).reshape(-1, 2), 1)
    key = e[:, 0].astype(np.int64) * (int(faces.max()) + 1) + e[:, 1]
    uk, first, inv = np.unique(key, return_index=True, return_inverse=True)
    pairs = e[first]
    m = inv.reshape(-1, 3) + len(verts)
    f = np.stack([faces[:, 0], m[:, 0], m[:, 2], m[:, 0], faces[:, 1], m[:, 1], m[:, 2], m[:, 1], faces[:, 2], m[:, 0], m[:, 1], m[:, 2]], 1).reshape(-1, 3)
    new_v = np.concatenate([verts, 0.5 * (verts[pairs[:, 0]] + verts[pairs[:, 1]])], 0)
    new_a = {k: np.concatenate([a, 0.5 * (a[pairs[:, 0]] + a[pairs[:, 1]])], 0) for k, a in attrs.items()}
    return new_v, f.astype(np.int64), new_a


def render_mesh_banded(ndc, faces, vn, H, W, band: int = 16):
    """oracle/mesh.render in row bands, each band re-computed in the backward (dense pixels x faces tensors: 13 776 faces x 128^2
    pixels would otherwise keep tens of GB alive for autograd)."""
    normals, alphas = [], []
    for r in range(0, H, band):
        def fn(ndc_, vn_, r0=r):
            n, a, _ = om.render(ndc_, faces, vn_, H, W, sigma_cfg=1e-5, rows=(r0, min(H, r0 + band)))
            return n, a
        n, a = checkpoint(fn, ndc, vn, use_reentrant=False)
        normals.append(n); alphas.append(a)
    return torch.cat(normals, 0), torch.cat(alphas, 0)


class OracleAvatar:
    """The trainable state of models/model.py::Model on the CPU: vertices (3,N), so3 / scale / appearance (3,F), shadow MLP."""

    def __init__(self, body: Dict[str, np.ndarray], img: int, params: Dict[str, torch.Tensor], shadow_wb: List[torch.Tensor], dtype=torch.float64):
        self.img, self.dtype = img, dtype
        self.faces = torch.from_numpy(body["faces"]).long()
        w = torch.from_numpy(body["canonical_lbs_weights"]).T
        self.w25 = torch.cat([w, torch.zeros(1, w.shape[1])], 0).to(dtype)
        self.p = {k: v.detach().clone().to(dtype).requires_grad_() for k, v in params.items()}
        self.shadow = [t.detach().clone().to(dtype).requires_grad_() for t in shadow_wb]
        self.tiled_mesh = False      # oracle/mesh.py::render_tiled (exact; what a 512 x 512 loop needs)
        self._topology()

    def _topology(self):
        N = self.p["vertices"].shape[1]
        self.edges, _ = oml.edges_of(self.faces, N)
        self.face_connectivity = oml.face_connectivity(self.faces, N)

    def subdivide(self):
        """models/model.py:136-179: midpoint subdivision of the canonical mesh; skinning weights averaged onto the midpoints; so3 /
        scale / appearance of a face inherited by its four children (`x[..., None].repeat(1, 1, 4).reshape(3, -1)`: child 4f + c).
        The caller rebuilds its optimizer (train.py:341-346)."""
        v, f, a = subdivide_midpoint(self.p["vertices"].detach().T.numpy(), self.faces.numpy(), {"w": self.w25.T.numpy()})
        rep = lambda t: t.detach()[..., None].repeat(1, 1, 4).reshape(t.shape[0], -1).clone()
        self.p = dict(vertices=torch.from_numpy(v).to(self.dtype).T.contiguous().requires_grad_(), so3=rep(self.p["so3"]).requires_grad_(),
                      scale=rep(self.p["scale"]).requires_grad_(), appearance=rep(self.p["appearance"]).requires_grad_())
        self.faces = torch.from_numpy(f).long()
        self.w25 = torch.from_numpy(a["w"]).to(self.dtype).T.contiguous()
        self._topology()

    def param_groups(self):
        """models/model.py:305-324 (lbs_weights is a buffer: its group holds no trainable tensor)."""
        return [dict(name="appearance", params=[self.p["appearance"]], lr=LR["appearance"]),
                dict(name="canonical_geometry_xyz", params=[self.p["vertices"]], lr=LR["canonical_geometry_xyz"]),
                dict(name="canonical_geometry", params=[self.p["scale"]], lr=LR["canonical_geometry"]),
                dict(name="canonical_geometry", params=[self.p["so3"]], lr=LR["canonical_geometry"]),
                dict(name="shadow", params=self.shadow, lr=LR["shadow"])]

    def forward(self, fr: Dict[str, torch.Tensor], training: bool = True):
        """models/model.py:184-303 -> rgbs (1,H,W,3), masks (1,H,W), outputs."""
        dt, img = self.dtype, self.img
        fr = {k: (v.to(dt) if v.is_floating_point() else v) for k, v in fr.items()}
        albedo, masks, aux = og.render_path(self.p, fr, self.faces, self.w25, img)
        v_obs = aux["v_obs"]                                                        # (3,N)
        vn = om.vertex_normals(v_obs.T, self.faces)                                 # model.py:271
        vn = (fr["E"][0, :3, :3] @ vn.T).T                                          # model.py:272
        ndc = om.ndc_T_world(v_obs[None], fr["K"], fr["E"], img, img)[0]
        if self.tiled_mesh:
            if training:
                normal, alpha, _ = om.render_tiled(ndc, self.faces, vn, img, img, sigma_cfg=1e-5)
            else:
                with torch.no_grad():
                    normal, alpha, _ = om.render_tiled(ndc, self.faces, vn, img, img, training=False)
        elif training:
            normal, alpha = render_mesh_banded(ndc, self.faces, vn, img, img)
        else:
            with torch.no_grad():
                normal = torch.cat([om.render(ndc, self.faces, vn, img, img, training=False, rows=(r, min(img, r + 16)))[0] for r in range(0, img, 16)], 0)
            alpha = None
        shade = shadow_mlp(normal.reshape(1, -1, 3), self.shadow).reshape(1, img, img, 1) * 2          # model.py:279-283
        rgbs = albedo * shade                                                                           # model.py:287
        out = dict(albedo=albedo[0], normal=normal[None], normal_mask=alpha[None] if alpha is not None else None, shadow=shade, v_obs=v_obs)
        return rgbs, masks, out

    def compute_loss(self, rgb_pred, mask_pred, out, rgb_gt, mask_gt, lpips_trunk=None, lpips_lins=None):
        """train.py:98-163 with the coefficients of exps/zju-mocap_377.yaml."""
        L = {}
        L["rgb"] = torch.mean(torch.abs(rgb_pred - rgb_gt))
        L["mask"] = torch.mean(torch.abs(mask_pred - mask_gt))
        if lpips_trunk is not None:
            L["lpips"] = torch.mean(olp.lpips_vgg(2 * rgb_pred.permute(0, 3, 1, 2) - 1, 2 * rgb_gt.permute(0, 3, 1, 2) - 1, lpips_trunk, lpips_lins))
        L["laplacian_observation"] = oml.laplacian_smoothing(out["v_obs"].T, self.edges)
        k = LOSS["normal_kernel"]
        dil = F.max_pool2d(mask_gt.unsqueeze(1), kernel_size=k, stride=1, padding=k // 2).squeeze(1)
        L["normal_mask"] = torch.mean(torch.abs(out["normal_mask"] - dil))
        L["normal_consist"] = oml.normal_consistency(out["v_obs"].T, self.faces)
        L["color_consist"] = oml.color_consistency(self.p["appearance"].T, self.face_connectivity)
        coeff = dict(rgb=LOSS["rgb"], mask=LOSS["mask"], lpips=LOSS["lpips"], laplacian_observation=LOSS["laplacian_observation"],
                     normal_mask=LOSS["normal_mask"], normal_consist=LOSS["normal_consist"], color_consist=LOSS["color_consist"])
        total = sum(L[k_] * coeff[k_] for k_ in L)
        return total, L


def update_lr(optimizer, iter_step: int):
    """train.py:166-175 (every group's name is in cfg.lr here)."""
    decay = 0.1 ** (iter_step / LR_DECAY_STEPS)
    for g in optimizer.param_groups:
        g["lr"] = LR[g["name"]] * decay


def to_8b(x: torch.Tensor) -> torch.Tensor:
    """utils/image_util.py:21-22."""
    return (255.0 * x.clamp(0.0, 1.0)).to(torch.uint8)


def psnr_8bit(pred: torch.Tensor, gt: torch.Tensor) -> float:
    """eval.py:355-361 + 101-104: both images through the 8-bit round trip, then -10 log10(mse) in float64."""
    p, g = to_8b(pred).double() / 255.0, to_8b(gt).double() / 255.0
    return float(-10.0 * torch.log(torch.mean((p - g) ** 2)) / math.log(10.0))
