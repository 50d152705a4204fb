"""The two optional pose-conditioned MLPs of the reference's `Model` -- the non-rigid vertex offsets
(`models/modules/non_rigid_module.py:73-147`) and the per-bone pose correction (`models/modules/pose_refinement_module.py:10-49`)
-- as plain torch modules with the reference's constructor node, `forward` signatures and state-dict key names
(`block_mlps.<i>.weight / bias`), so that `Model(cfg, info, non_rigid_module=..., pose_refinement_module=...)` holds the very
parameters the reference's optimizer holds (101 123 + 233 029 of the 951 023 floats of the gradient exchange) and a
reference checkpoint loads unchanged.  Both sit in front of the hot path (they produce `vertices_pose` / `dst_Rs`), are small
library GEMMs and only run after their `kick_in_iter` (150 000 / 100 000 of 300 000 on ZJU-MoCap: `exps/zju-mocap_377.yaml:73,85`).

Pinned by `tests/golden/pose_modules.npz` (recorded through the reference's own classes by `scripts/make_module_goldens.py`)."""
from __future__ import annotations

import math

import torch
import torch.nn as nn


def _get(cfg, name, default):
    return getattr(cfg, name) if hasattr(cfg, name) else default


def _relu_stack_init(layers) -> None:
    """network_util.initseq: Xavier-uniform with the ReLU gain for every linear layer followed by a ReLU, gain 1 for the last; zero biases."""
    lin = [m for m in layers if isinstance(m, nn.Linear)]
    for k, m in enumerate(lin):
        nn.init.xavier_uniform_(m.weight, gain=nn.init.calculate_gain("relu") if k + 1 < len(lin) else 1.0)
        nn.init.zeros_(m.bias)


def windowed_posenc(x: torch.Tensor, n_freqs: int, i_iter, kick_in_iter: float, full_band_iter: float) -> torch.Tensor:
    """non_rigid_module.py:15-51: sin / cos of x * 2^k for k < n_freqs, WITHOUT the input itself, band k faded in by a Hann window
    w_k = (1 - cos(pi * clamp(alpha - k, 0, 1))) / 2 with alpha = n_freqs * max(i_iter - kick_in, 0) / (full_band - kick_in).
    Column order: (sin f0, cos f0, sin f1, cos f1, ...), 3 columns each."""
    t = max(float(i_iter) - float(kick_in_iter), 0.0)
    alpha = n_freqs * t / (float(full_band_iter) - float(kick_in_iter))
    cols = []
    for k in range(n_freqs):
        w = (1.0 - math.cos(math.pi * min(max(alpha - k, 0.0), 1.0))) / 2.0
        arg = x * float(2.0 ** k)
        cols += [w * torch.sin(arg), w * torch.cos(arg)]
    return torch.cat(cols, -1)


class NonRigidModule(nn.Module):
    """non_rigid_module.py:73-147 (the HumanNeRF offset MLP): h = [posevec | posenc(xyz)], `mlp_depth` ReLU layers of `mlp_width`
    with the encoding concatenated again in front of the layers in `skips`, a last layer started at U(-1e-5, 1e-5) -> xyz offsets
    (+ scale offsets when the node asks for them; rotation offsets -- unused by every shipped experiment -- raise)."""

    def __init__(self, module_cfg, **kwargs):
        super().__init__()
        self.cfg = module_cfg
        self.update_rot = bool(_get(module_cfg, "update_rot", False))
        self.update_scale = bool(_get(module_cfg, "update_scale", False))
        if self.update_rot:
            raise NotImplementedError("non_rigid.update_rot: no shipped experiment sets it, and Model.forward discards the module's R output (model.py:201-207)")
        self.skips = tuple(module_cfg.skips)
        enc = 6 * int(module_cfg.multires)                         # no identity column (include_input False)
        width, depth = int(module_cfg.mlp_width), int(module_cfg.mlp_depth)
        layers = [nn.Linear(enc + int(module_cfg.condition_code_size), width), nn.ReLU()]
        self.layers_to_cat_inputs = []
        for i in range(1, depth):
            if i in self.skips:
                self.layers_to_cat_inputs.append(len(layers))
            layers += [nn.Linear(width + (enc if i in self.skips else 0), width), nn.ReLU()]
        layers.append(nn.Linear(width, 3 + 3 * self.update_rot + 3 * self.update_scale))
        self.block_mlps = nn.ModuleList(layers)
        _relu_stack_init(self.block_mlps)
        s = float(_get(module_cfg, "init_scale", 1e-5))
        self.block_mlps[-1].weight.data.uniform_(-s, s)
        self.block_mlps[-1].bias.data.zero_()

    def forward(self, xyzs_skeleton, dst_posevec, i_iter, R=None, S=None):
        """xyzs_skeleton (B, 3, N), dst_posevec (B, 69) -> (xyz + offset (B, 3, N), R', S')."""
        xyz = xyzs_skeleton.permute(0, 2, 1)
        n = xyz.shape[1]
        pe = windowed_posenc(xyz, int(self.cfg.multires), i_iter, self.cfg.kick_in_iter, self.cfg.full_band_iter)
        h = torch.cat([dst_posevec[:, None, :].expand(-1, n, -1), pe], -1)
        for i, layer in enumerate(self.block_mlps):
            if i in self.layers_to_cat_inputs:
                h = torch.cat([h, pe], -1)
            h = layer(h)
        out_xyz = xyzs_skeleton + h[..., :3].permute(0, 2, 1)
        S_new = S + h[..., 3:] if self.update_scale else S      # (columns 3.. when no rotation offset is configured)
        return out_xyz, R, S_new


def rodrigues(rvec: torch.Tensor) -> torch.Tensor:
    """network_util.py:66-92 (RodriguesModule): theta = sqrt(1e-5 + |r|^2), R = cos I + (1 - cos) k k^T + sin [k]x, (B, 3) -> (B, 3, 3)."""
    theta = torch.sqrt(1e-5 + (rvec * rvec).sum(1))
    k = rvec / theta[:, None]
    c, s = torch.cos(theta), torch.sin(theta)
    x, y, z = k[:, 0], k[:, 1], k[:, 2]
    v = 1.0 - c
    rows = [x * x + (1.0 - x * x) * c, x * y * v - z * s, x * z * v + y * s,
            x * y * v + z * s, y * y + (1.0 - y * y) * c, y * z * v - x * s,
            x * z * v - y * s, y * z * v + x * s, z * z + (1.0 - z * z) * c]
    return torch.stack(rows, 1).view(-1, 3, 3)


class PoseRefinementModule(nn.Module):
    """pose_refinement_module.py:10-49: posevec (B, 69) -> MLP -> one axis-angle per non-root bone -> rotation matrices, the root's
    correction fixed to the identity -> (B, 24, 3, 3), right-multiplied onto dst_Rs by Model.forward (model.py:193-196)."""

    def __init__(self, module_cfg, **kwargs):
        super().__init__()
        self.cfg = module_cfg
        width, depth = int(module_cfg.mlp_width), int(module_cfg.mlp_depth)
        self.refine_root, self.refine_t = bool(module_cfg.refine_root), bool(module_cfg.refine_t)
        self.total_bones = int(module_cfg.total_bones) - (0 if self.refine_root else 1)
        layers = [nn.Linear(int(module_cfg.embedding_size), width), nn.ReLU()]
        for _ in range(depth - 1):
            layers += [nn.Linear(width, width), nn.ReLU()]
        layers.append(nn.Linear(width, 3 * self.total_bones))
        self.block_mlps = nn.Sequential(*layers)
        _relu_stack_init(self.block_mlps)
        self.block_mlps[-1].weight.data.uniform_(-1e-5, 1e-5)
        self.block_mlps[-1].bias.data.zero_()

    def forward(self, dst_posevec, **kwargs):
        Rs = rodrigues(self.block_mlps(dst_posevec).view(-1, 3)).view(-1, self.total_bones, 3, 3)
        eye = torch.eye(3, device=Rs.device, dtype=Rs.dtype).expand(Rs.shape[0], 1, 3, 3)
        return torch.cat([eye, Rs], 1)
