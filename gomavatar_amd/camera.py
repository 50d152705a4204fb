"""The camera block of the reference's splat adapter (models/modules/renderer/gaussian.py:28-66 + utils/camera_util.py:213-214)
as ONE function, used by every consumer in this package (`pipeline.RenderStep`, `model.Model`, `rasterizer.DeviceCamera`):

    tanfov     = tan(0.5 * focal2fov(f, size)) = tan(atan(size / (2 f)))          (gaussian.py:33-36)
    K_ndc      = [[2fx/w, 0, (2px - w)/w, 0], [0, 2fy/h, (2py - h)/h, 0],
                  [0, 0, zfar/(zfar - znear), -zfar znear/(zfar - znear)], [0, 0, 1, 0]]   (gaussian.py:41-46, znear .001, zfar 100)
    viewmatrix = E^T, projmatrix = E^T K_ndc^T, campos = (E^T)^-1 [3, :3]          (gaussian.py:47,60-62)

It is written with torch ops that run on host or device tensors alike and never reads device data back, so the same
code serves the host-built `GomCamera` struct and the device-resident camera of a captured graph.  Like the reference
(python floats -> float32 tensor / float kernel argument) the scalars are formed in float64 and rounded to float32 once."""
from __future__ import annotations

import torch


def camera_block(K: torch.Tensor, E: torch.Tensor, H: int, W: int, znear: float = 0.001, zfar: float = 100.0, want_campos: bool = False):
    """K (3,3), E (4,4) -> (tanfov (2,) fp32 [x, y], view (4,4) fp32 = E^T, proj (4,4) fp32 = E^T K_ndc^T[, campos (3,)])."""
    Kd = K.detach().to(torch.float64)
    E32 = E.detach().to(torch.float32)
    fx, fy, px, py = Kd[0, 0], Kd[1, 1], Kd[0, 2], Kd[1, 2]
    tanfov = torch.stack([torch.tan(torch.atan(W / (2 * fx))), torch.tan(torch.atan(H / (2 * fy)))]).to(torch.float32)
    zero, one = torch.zeros((), dtype=torch.float64, device=Kd.device), torch.ones((), dtype=torch.float64, device=Kd.device)
    K_ndc = torch.stack([torch.stack([2 * fx / W, zero, (2 * px - W) / W, zero]),
                         torch.stack([zero, 2 * fy / H, (2 * py - H) / H, zero]),
                         torch.stack([zero, zero, one * (zfar / (zfar - znear)), one * (-zfar * znear / (zfar - znear))]),
                         torch.stack([zero, zero, one, zero])]).to(torch.float32)
    view = E32.T.contiguous()
    # proj[i][j] = sum_k view[i][k] K_ndc[j][k], products and sum spelled out (a row of K_ndc has at most two non-zeros, so the
    # result does not depend on the summation order or on FMA contraction of a library GEMM: host and device agree bit for bit)
    proj = (view[:, None, :] * K_ndc[None, :, :]).sum(-1)
    if want_campos:
        return tanfov, view, proj, torch.linalg.inv(view)[3, :3]
    return tanfov, view, proj
