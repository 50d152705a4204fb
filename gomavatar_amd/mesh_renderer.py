"""Mesh normal map + soft silhouette renderer: mirror of the reference's
`models/modules/renderer/mesh.py::Renderer` (PyTorch3D MeshRasterizer + NormalShader +
SoftSilhouetteShader), used by `Model.forward` at models/model.py:270-273.

    renderer = MeshNormalRenderer(img_size=(W, H), sigma=1e-5)
    normal, mask = renderer(vertices_observation (1,3,N), vertex_normals (1,N,3), K (1,3,3), E (1,4,4), faces (F,3))
    # normal (1,H,W,3), mask (1,H,W,1) in training mode / None in eval mode

Projection (utils/pc_util.py:10-46) and vertex normals (Meshes.verts_normals_padded) are a handful of per-vertex torch
ops; the rasterization -- O(pixels x faces) in the reference -- is the tile-binned HIP kernel pair of csrc/mesh_raster.hip."""
from __future__ import annotations

import math
from typing import Optional

import torch

from . import _lib
from .geometry import MeshTopology
from .rasterizer import RasterState, _StatePool, _StateLease

# Scratch of the mesh rasterizer, leased per forward like the splat rasterizer's: the backward reads what ITS forward left in the
# state (face setup, per-pixel products), so two forwards in flight (gradient accumulation over frames, a no_grad preview between
# a forward and its backward) must not share one.
_MESH_POOL = _StatePool()


def ndc_T_world(xyzs_world: torch.Tensor, K: torch.Tensor, E: torch.Tensor, H: int, W: int) -> torch.Tensor:
    """utils/pc_util.py:30-46.  (B,3,N) world points -> (B,N,3): negated NDC x, y (shorter side in [-1,1]), camera z."""
    if xyzs_world.is_cuda and xyzs_world.shape[0] == 1 and xyzs_world.dtype == torch.float32 and not (K.requires_grad or E.requires_grad):
        return _NdcTWorld.apply(xyzs_world, K, E, H, W)       # one kernel each way (csrc/mesh_raster.hip)
    ones = torch.ones_like(xyzs_world[:, :1])
    cam_ = torch.bmm(E, torch.cat([xyzs_world, ones], 1))
    cam = cam_[:, :3] / cam_[:, 3:]
    p = torch.bmm(K, cam)
    xy = p[:, :2] / p[:, 2:]
    if H < W:
        xs = -((xy[:, 0] / H) * 2.0 - (W / H))
        ys = -((xy[:, 1] / H) * 2.0 - 1.0)
    else:
        xs = -((xy[:, 0] / W) * 2.0 - 1.0)
        ys = -((xy[:, 1] / W) * 2.0 - (H / W))
    return torch.stack([xs, ys, cam[:, 2]], -1)


class _NdcTWorld(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz, K, E, H, W):
        lib = _lib.load()
        v = xyz[0].contiguous()
        k, e = K[0].float().contiguous(), E[0].float().contiguous()
        N = v.shape[1]
        out = torch.empty(1, N, 3, dtype=torch.float32, device=v.device)
        _lib.check(lib.gom_ndc_from_world_forward(N, int(H), int(W), _lib.ptr(v), _lib.ptr(k), _lib.ptr(e), _lib.ptr(out), _lib.stream_ptr()))
        ctx.save_for_backward(v, k, e)
        ctx.hw = (int(H), int(W))
        return out

    @staticmethod
    def backward(ctx, g):
        v, k, e = ctx.saved_tensors
        lib = _lib.load()
        g = g.float().contiguous()
        d = torch.empty(1, 3, v.shape[1], dtype=torch.float32, device=v.device)
        _lib.check(lib.gom_ndc_from_world_backward(v.shape[1], ctx.hw[0], ctx.hw[1], _lib.ptr(v), _lib.ptr(k), _lib.ptr(e), _lib.ptr(g), _lib.ptr(d), _lib.stream_ptr()))
        return d, None, None, None, None


class _VertexNormals(torch.autograd.Function):
    @staticmethod
    def forward(ctx, verts, topo: MeshTopology, rotation):
        lib = _lib.load()
        v = verts.float().contiguous()
        r = rotation.detach().float().contiguous() if rotation is not None else None
        N = v.shape[0]
        sums, normals = torch.empty_like(v), torch.empty_like(v)
        _lib.check(lib.gom_vertex_normals_forward(N, topo.n_faces, _lib.ptr(v), _lib.ptr(topo.faces), _lib.ptr(topo.csr_off), _lib.ptr(topo.csr_idx),
                                                  _lib.ptr(r), _lib.ptr(sums), _lib.ptr(normals), _lib.stream_ptr()))
        ctx.save_for_backward(v, sums, r)
        ctx.topo = topo
        return normals

    @staticmethod
    def backward(ctx, d_normals):
        v, sums, r = ctx.saved_tensors
        topo, lib = ctx.topo, _lib.load()
        dn = d_normals.float().contiguous()
        scratch = torch.empty((topo.n_faces, 9), dtype=torch.float32, device=v.device)
        d_verts = torch.empty_like(v)
        _lib.check(lib.gom_vertex_normals_backward(v.shape[0], topo.n_faces, _lib.ptr(v), _lib.ptr(topo.faces), _lib.ptr(topo.csr_off), _lib.ptr(topo.csr_idx),
                                                   _lib.ptr(r), _lib.ptr(sums), _lib.ptr(dn), _lib.ptr(scratch), _lib.ptr(d_verts), _lib.stream_ptr()))
        return d_verts, None, None


def vertex_normals(verts: torch.Tensor, topo: MeshTopology, rotation: Optional[torch.Tensor] = None) -> torch.Tensor:
    """PyTorch3D `Meshes.verts_normals_packed` (models/model.py:271): area-weighted face normals summed on the corners,
    normalize(eps=1e-6).  verts (N,3) -> (N,3).  CSR gather (deterministic), not index_add.  `rotation` (3,3, no gradient) is
    applied to the unit normals in the same kernel (model.py:272 takes them into the camera frame with E[:3,:3])."""
    return _VertexNormals.apply(verts, topo, rotation)


class _MeshRaster(torch.autograd.Function):
    @staticmethod
    def forward(ctx, verts_ndc, vnormals, topo: MeshTopology, lease, H, W, blur_radius, sigma, want_alpha):
        lib = _lib.load()
        state = lease.st
        v, n = verts_ndc.float().contiguous(), vnormals.float().contiguous()
        N, F = v.shape[0], topo.n_faces
        normal = torch.empty((H, W, 3), dtype=torch.float32, device=v.device)
        alpha = torch.empty((H, W), dtype=torch.float32, device=v.device) if want_alpha else None
        _lib.check(lib.gom_mesh_raster_forward(state.handle, N, F, H, W, _lib.ptr(v), _lib.ptr(topo.faces), _lib.ptr(n), float(blur_radius), float(sigma),
                                               _lib.ptr(normal), _lib.ptr(alpha), _lib.stream_ptr()))
        ctx.topo, ctx.lease, ctx.dims, ctx.want_alpha = topo, lease, (N, F, H, W), want_alpha
        if want_alpha:
            return normal, alpha
        return normal, torch.zeros((), device=v.device)

    @staticmethod
    def backward(ctx, d_normal, d_alpha):
        lib = _lib.load()
        N, F, H, W = ctx.dims
        dn = d_normal.float().contiguous()
        da = d_alpha.float().contiguous() if ctx.want_alpha else None
        d_verts = torch.empty((N, 3), dtype=torch.float32, device=dn.device)
        d_vn = torch.empty((N, 3), dtype=torch.float32, device=dn.device)
        if ctx.lease.done:
            raise RuntimeError("mesh rasterizer: backward called twice on the same forward (its scratch has been released)")
        _lib.check(lib.gom_mesh_raster_backward(ctx.lease.st.handle, N, F, H, W, _lib.ptr(ctx.topo.csr_off), _lib.ptr(ctx.topo.csr_idx), _lib.ptr(dn),
                                                _lib.ptr(da), _lib.ptr(d_verts), _lib.ptr(d_vn), _lib.stream_ptr()))
        ctx.lease.finish()
        return d_verts, d_vn, None, None, None, None, None, None, None


class MeshNormalRenderer(torch.nn.Module):
    """mesh.py:65-128.  `img_size` = (W, H) as in the reference's cfg.img_size; `sigma` = cfg.sigma (1e-4 when absent,
    1e-5 in exps/zju-mocap_377.yaml:89); the silhouette is only rendered in training mode."""

    BLEND_SIGMA = 1e-4   # PyTorch3D BlendParams default, never overridden by the reference

    def __init__(self, img_size=(512, 512), sigma: Optional[float] = None, soft_mask: bool = True):
        super().__init__()
        self.img_size = (int(img_size[0]), int(img_size[1]))
        self.sigma = 1e-4 if sigma is None else float(sigma)
        self.soft_mask = soft_mask
        self.blur_radius = math.log(1.0 / 1e-4 - 1.0) * self.sigma
        self._topo = None
        self._topo_key = None
        self.state = None

    def topology(self, faces: torch.Tensor, n_verts: int) -> MeshTopology:
        key = (faces.data_ptr(), int(faces.shape[0]), n_verts)
        if self._topo_key != key:   # rebuilt after subdivide()
            self._topo, self._topo_key = MeshTopology(faces, n_verts, device=faces.device), key
        return self._topo

    def forward(self, xyzs_observation, vertex_normals_, K, E, faces, **kwargs):
        H, W = self.img_size[1], self.img_size[0]
        xyzs_ndc = ndc_T_world(xyzs_observation, K, E, H, W)      # (1,N,3)
        assert xyzs_ndc.shape[0] == 1, "the reference renders one frame per call (B = 1)"
        topo = self.topology(faces, xyzs_ndc.shape[1])
        lease = _StateLease(_MESH_POOL.acquire(xyzs_ndc.device), pool=_MESH_POOL)      # goes back when the backward has run (or the graph is dropped)
        self.state = lease.st      # scratch of the MOST RECENT forward (pix_to_face export in the tests); may be re-leased once released
        # (squeeze, not [0]: the backward of a select is a zero fill + a copy, two launches per operand; a view's is a view)
        normal, alpha = _MeshRaster.apply(xyzs_ndc.squeeze(0), vertex_normals_.squeeze(0), topo, lease, H, W, self.blur_radius, self.BLEND_SIGMA, self.training)
        if not normal.requires_grad:   # no autograd graph was recorded: nothing will call backward
            lease.finish()
        if not self.training:
            return normal[None], None
        return normal[None], alpha[None, ..., None]
