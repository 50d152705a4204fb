"""Host-side mirror of the reference's geometry functions, HIP-backed.

`get_global_RTs` / `apply_lbs` keep the reference names, argument meaning and
shapes (utils/body_util.py:612-644); `posed_face_gaussians` is the fused
FK -> LBS -> per-face Gaussian block of `Model.forward`
(models/model.py:213-234) as ONE autograd node (3 kernels forward, 2-3
backward, no atomics).  Batch size is 1, like the reference
(models/modules/renderer/gaussian.py:24).  No CPU fallback.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import _lib

N_JOINTS = 24


class MeshTopology:
    """Static per-mesh index data on the device: int32 faces and the CSR
    vertex -> (face, corner) adjacency used for the atomic-free vertex gather.
    Rebuild after `subdivide()` (reference models/model.py:136-179)."""

    def __init__(self, faces: torch.Tensor, n_verts: int, device=None):
        faces_cpu = faces.detach().to("cpu", torch.int64).contiguous()
        F = faces_cpu.shape[0]
        flat = faces_cpu.reshape(-1)
        order = torch.argsort(flat, stable=True)
        counts = torch.bincount(flat, minlength=n_verts)
        off = torch.zeros(n_verts + 1, dtype=torch.int64)
        off[1:] = torch.cumsum(counts, 0)
        device = device if device is not None else faces.device
        self.n_verts = int(n_verts)
        self.n_faces = int(F)
        self.faces = faces_cpu.to(torch.int32).to(device).contiguous()
        self.csr_off = off.to(torch.int32).to(device).contiguous()
        self.csr_idx = order.to(torch.int32).to(device).contiguous()  # entries are face*3 + corner


def _check_dev(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("gomavatar_amd.geometry: tensors must be on the HIP device (no CPU fallback)")


class _FK(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cnl_gtfms, dst_Rs, dst_Ts):
        lib = _lib.load()
        cnl = cnl_gtfms.contiguous().float()
        Rs = dst_Rs.contiguous().float()
        Ts = dst_Ts.contiguous().float()
        RT = torch.empty((N_JOINTS, 12), dtype=torch.float32, device=Rs.device)
        save = torch.empty((N_JOINTS, 32), dtype=torch.float32, device=Rs.device)
        _lib.check(lib.gom_fk_forward(_lib.ptr(cnl), _lib.ptr(Rs), _lib.ptr(Ts), _lib.ptr(RT), _lib.ptr(save), _lib.stream_ptr()))
        ctx.save_for_backward(Rs, Ts, save)
        ctx.shapes = (dst_Rs.shape, dst_Ts.shape)
        return RT

    @staticmethod
    def backward(ctx, dRT):
        lib = _lib.load()
        Rs, Ts, save = ctx.saved_tensors
        dRs = torch.empty_like(Rs)
        dTs = torch.empty_like(Ts)
        g = dRT.contiguous()  # keep a reference until the launch is enqueued (see _lib.ptr)
        _lib.check(lib.gom_fk_backward(_lib.ptr(Rs), _lib.ptr(Ts), _lib.ptr(save), _lib.ptr(g), _lib.ptr(dRs),
                                       _lib.ptr(dTs), _lib.stream_ptr()))
        return None, dRs.reshape(ctx.shapes[0]), dTs.reshape(ctx.shapes[1])


def skinning_transforms(cnl_gtfms: torch.Tensor, dst_Rs: torch.Tensor, dst_Ts: torch.Tensor) -> torch.Tensor:
    """(1,24,4,4), (1,24,3,3), (1,24,3) -> RT (24,12): row-major R then T."""
    _check_dev(cnl_gtfms, dst_Rs, dst_Ts)
    assert dst_Rs.shape[-3] == N_JOINTS and (dst_Rs.dim() == 3 or dst_Rs.shape[0] == 1), "batch size must be 1"
    return _FK.apply(cnl_gtfms.reshape(N_JOINTS, 4, 4), dst_Rs.reshape(N_JOINTS, 3, 3), dst_Ts.reshape(N_JOINTS, 3))


def get_global_RTs(cnl_gtfms, dst_Rs, dst_Ts, use_smplx: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
    """Reference utils/body_util.py:612-638: returns (1,24,3,3), (1,24,3)."""
    if use_smplx:
        raise NotImplementedError("SMPL-X skeleton is not on the GoMAvatar hot path")
    RT = skinning_transforms(cnl_gtfms, dst_Rs, dst_Ts)
    return RT[:, :9].reshape(1, N_JOINTS, 3, 3), RT[:, 9:].reshape(1, N_JOINTS, 3)


class _LBS(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz, RT, weights):
        lib = _lib.load()
        xyz_c = xyz.contiguous()
        RT_c = RT.contiguous()
        N = xyz_c.shape[1]
        J = RT_c.shape[0]
        out = torch.empty_like(xyz_c)
        _lib.check(lib.gom_lbs_forward(N, J, _lib.ptr(xyz_c), _lib.ptr(weights), _lib.ptr(RT_c), _lib.ptr(out), _lib.stream_ptr()))
        ctx.save_for_backward(xyz_c, RT_c, weights)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        xyz, RT, weights = ctx.saved_tensors
        N, J = xyz.shape[1], RT.shape[0]
        d_xyz = torch.empty_like(xyz)
        dRT = torch.zeros_like(RT) if ctx.needs_input_grad[1] else None
        g = g.contiguous()
        _lib.check(lib.gom_vertex_backward(N, J, _lib.ptr(xyz), _lib.ptr(weights), _lib.ptr(RT), 0, 0, 0, _lib.ptr(g),
                                           0, _lib.ptr(d_xyz), _lib.ptr(dRT), _lib.stream_ptr()))
        return d_xyz, dRT, None


def apply_lbs(xyzs_canonical, global_Rs, global_Ts, lbs_weights) -> torch.Tensor:
    """Reference utils/body_util.py:641-644: (1,3,N), (1,24,3,3), (1,24,3),
    (25,N) -> (1,3,N).  The last weight row (background) is ignored."""
    _check_dev(xyzs_canonical, global_Rs, global_Ts, lbs_weights)
    assert xyzs_canonical.shape[0] == 1, "batch size must be 1"
    J = global_Rs.shape[1]
    assert lbs_weights.shape[0] == J + 1
    RT = torch.cat([global_Rs.reshape(J, 9), global_Ts.reshape(J, 3)], dim=1)
    if lbs_weights.requires_grad:
        raise NotImplementedError("lbs_weights.refine is off in every reference config (configs/default.yaml:72-73)")
    return _LBS.apply(xyzs_canonical[0], RT, lbs_weights.detach().contiguous().float())[None]


class _FaceGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, verts, so3, scale, topo, sigma):
        lib = _lib.load()
        v, w, s = verts.contiguous(), so3.contiguous(), scale.contiguous()
        N, F = v.shape[1], topo.n_faces
        xyz = torch.empty((F, 3), dtype=torch.float32, device=v.device)
        cov6 = torch.empty((F, 6), dtype=torch.float32, device=v.device)
        _lib.check(lib.gom_face_forward(N, F, _lib.ptr(v), _lib.ptr(topo.faces), _lib.ptr(w), _lib.ptr(s), float(sigma), _lib.ptr(xyz),
                                        _lib.ptr(cov6), 0, 0, _lib.stream_ptr()))
        ctx.save_for_backward(v, w, s)
        ctx.topo, ctx.sigma = topo, float(sigma)
        return xyz, cov6

    @staticmethod
    def backward(ctx, g_xyz, g_cov6):
        lib = _lib.load()
        v, w, s = ctx.saved_tensors
        topo = ctx.topo
        N, F = v.shape[1], topo.n_faces
        d_corner = torch.empty((F, 3, 3), dtype=torch.float32, device=v.device)
        d_so3 = torch.empty_like(w)
        d_scale = torch.empty_like(s)
        g_xyz, g_cov6 = g_xyz.contiguous(), g_cov6.contiguous()
        _lib.check(lib.gom_face_backward(N, F, _lib.ptr(v), _lib.ptr(topo.faces), _lib.ptr(w), _lib.ptr(s), ctx.sigma,
                                         _lib.ptr(g_xyz), _lib.ptr(g_cov6), _lib.ptr(d_corner),
                                         _lib.ptr(d_so3), _lib.ptr(d_scale), 0, 0, _lib.stream_ptr()))
        # CSR gather of the corner gradients onto vertices (deterministic)
        d_verts = _csr_gather(d_corner.reshape(-1, 3), topo, N)
        return d_verts, d_so3, d_scale, None, None


def _csr_gather(corner_grads: torch.Tensor, topo: MeshTopology, N: int) -> torch.Tensor:
    """(3F,3) per-corner gradients -> (3,N): the vertex backward kernel with
    identity skinning (one joint, unit weight)."""
    lib = _lib.load()
    dev = corner_grads.device
    out = torch.empty((3, N), dtype=torch.float32, device=dev)
    ident = torch.tensor([[1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0]], dtype=torch.float32, device=dev)
    ones = torch.ones((3, N), dtype=torch.float32, device=dev)  # read-only: serves as xyz and as the (2,N) weights
    corner_grads = corner_grads.contiguous()
    _lib.check(lib.gom_vertex_backward(N, 1, _lib.ptr(ones), _lib.ptr(ones), _lib.ptr(ident), _lib.ptr(topo.csr_off),
                                       _lib.ptr(topo.csr_idx), _lib.ptr(corner_grads), 0, 0, _lib.ptr(out), 0,
                                       _lib.stream_ptr()))
    return out


def face_gaussians(vertices_observation: torch.Tensor, so3: torch.Tensor, scale: torch.Tensor, topo: MeshTopology,
                   sigma: float = 1e-3) -> Tuple[torch.Tensor, torch.Tensor]:
    """models/model.py:225-234 + gaussian.py:71-75: (3,N), (3,F), (3,F) ->
    centroids (F,3), packed covariances (F,6)."""
    _check_dev(vertices_observation, so3, scale)
    return _FaceGaussians.apply(vertices_observation, so3, scale, topo, sigma)


class _PosedFaceGaussians(torch.autograd.Function):
    """FK -> LBS -> per-face frame as one node: two launches forward (k_fk_lbs_fwd: the 24-joint chain in every workgroup, then k_face_fwd),
    backward = face_bwd + vertex_bwd (CSR gather fused with the LBS transpose) [+ fk_bwd].  With `appearance` (3, F) the face kernel also
    leaves the rasterizer's features [r g b 1] (F, 4) (gaussian.py:49's cat) and its backward the (3, F) colour gradient."""

    @staticmethod
    def forward(ctx, vertices, so3, scale, dst_Rs, dst_Ts, cnl_gtfms, lbs_weights, topo, sigma, appearance):
        lib = _lib.load()
        st = _lib.stream_ptr()
        v, w, s = vertices.contiguous(), so3.contiguous(), scale.contiguous()
        Rs, Ts, cnl = dst_Rs.contiguous().float(), dst_Ts.contiguous().float(), cnl_gtfms.contiguous().float()
        dev = v.device
        N, F = v.shape[1], topo.n_faces
        RT = torch.empty((N_JOINTS, 12), dtype=torch.float32, device=dev)
        save = torch.empty((N_JOINTS, 32), dtype=torch.float32, device=dev)
        v_obs = torch.empty_like(v)
        xyz = torch.empty((F, 3), dtype=torch.float32, device=dev)
        cov6 = torch.empty((F, 6), dtype=torch.float32, device=dev)
        app = feat4 = None
        if appearance is not None:
            app = appearance.detach().contiguous().float()
            assert app.shape == (3, F)
            feat4 = torch.empty((F, 4), dtype=torch.float32, device=dev)
        _lib.check(lib.gom_fk_lbs_forward(N, _lib.ptr(cnl), _lib.ptr(Rs), _lib.ptr(Ts), _lib.ptr(v), _lib.ptr(lbs_weights), _lib.ptr(RT), _lib.ptr(save),
                                          _lib.ptr(v_obs), st))
        _lib.check(lib.gom_face_forward(N, F, _lib.ptr(v_obs), _lib.ptr(topo.faces), _lib.ptr(w), _lib.ptr(s), float(sigma),
                                        _lib.ptr(xyz), _lib.ptr(cov6), _lib.ptr(app), _lib.ptr(feat4), st))
        ctx.save_for_backward(v, w, s, Rs, Ts, RT, save, v_obs, lbs_weights)
        ctx.topo, ctx.sigma = topo, float(sigma)
        ctx.shapes = (dst_Rs.shape, dst_Ts.shape)
        ctx.with_feat = appearance is not None
        if feat4 is None:
            return xyz, cov6, v_obs
        return xyz, cov6, v_obs, feat4

    @staticmethod
    def backward(ctx, g_xyz, g_cov6, g_vobs, g_feat=None):
        lib = _lib.load()
        st = _lib.stream_ptr()
        v, w, s, Rs, Ts, RT, save, v_obs, lbs_weights = ctx.saved_tensors
        topo = ctx.topo
        dev = v.device
        N, F = v.shape[1], topo.n_faces
        d_corner = torch.empty((F, 3, 3), dtype=torch.float32, device=dev)
        d_so3 = torch.empty_like(w)
        d_scale = torch.empty_like(s)
        if g_xyz is None:
            g_xyz = torch.zeros((F, 3), dtype=torch.float32, device=dev)
        if g_cov6 is None:
            g_cov6 = torch.zeros((F, 6), dtype=torch.float32, device=dev)
        g_xyz, g_cov6 = g_xyz.contiguous(), g_cov6.contiguous()
        d_app = None
        if ctx.with_feat and ctx.needs_input_grad[9] and g_feat is not None:
            g_feat = g_feat.contiguous().float()
            d_app = torch.empty((3, F), dtype=torch.float32, device=dev)
        else:
            g_feat = None
        _lib.check(lib.gom_face_backward(N, F, _lib.ptr(v_obs), _lib.ptr(topo.faces), _lib.ptr(w), _lib.ptr(s), ctx.sigma,
                                         _lib.ptr(g_xyz), _lib.ptr(g_cov6), _lib.ptr(d_corner),
                                         _lib.ptr(d_so3), _lib.ptr(d_scale), _lib.ptr(g_feat), _lib.ptr(d_app), st))
        need_pose = ctx.needs_input_grad[3] or ctx.needs_input_grad[4]
        d_v = torch.empty_like(v)
        dRT = torch.zeros_like(RT) if need_pose else None
        extra = g_vobs.contiguous() if g_vobs is not None else None
        _lib.check(lib.gom_vertex_backward(N, N_JOINTS, _lib.ptr(v), _lib.ptr(lbs_weights), _lib.ptr(RT), _lib.ptr(topo.csr_off),
                                           _lib.ptr(topo.csr_idx), _lib.ptr(d_corner), _lib.ptr(extra), 0, _lib.ptr(d_v),
                                           _lib.ptr(dRT), st))
        dRs = dTs = None
        if need_pose:
            dRs = torch.empty_like(Rs)
            dTs = torch.empty_like(Ts)
            _lib.check(lib.gom_fk_backward(_lib.ptr(Rs), _lib.ptr(Ts), _lib.ptr(save), _lib.ptr(dRT), _lib.ptr(dRs), _lib.ptr(dTs), st))
            dRs, dTs = dRs.reshape(ctx.shapes[0]), dTs.reshape(ctx.shapes[1])
        return d_v, d_so3, d_scale, dRs, dTs, None, None, None, None, d_app


def posed_face_gaussians(vertices: torch.Tensor, so3: torch.Tensor, scale: torch.Tensor, dst_Rs: torch.Tensor, dst_Ts: torch.Tensor,
                         cnl_gtfms: torch.Tensor, lbs_weights: torch.Tensor, topo: MeshTopology, sigma: float = 1e-3,
                         appearance: Optional[torch.Tensor] = None):
    """Fused models/model.py:213-234: canonical vertices (3,N) + pose ->
    (centroids (F,3), cov6 (F,6), posed vertices (3,N)); with `appearance` (3,F) also the rasterizer's
    features (F,4) = [appearance.T | 1] (gaussian.py:49), written by the face kernel instead of a cat."""
    _check_dev(vertices, so3, scale, dst_Rs, dst_Ts, cnl_gtfms, lbs_weights, appearance)
    assert lbs_weights.shape[0] == N_JOINTS + 1 and not lbs_weights.requires_grad
    return _PosedFaceGaussians.apply(vertices, so3, scale, dst_Rs.reshape(N_JOINTS, 3, 3), dst_Ts.reshape(N_JOINTS, 3),
                                     cnl_gtfms.reshape(N_JOINTS, 4, 4), lbs_weights.contiguous(), topo, sigma, appearance)
