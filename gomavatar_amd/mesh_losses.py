"""Mesh regularisers of the reference's loss (train.py:123-160) as HIP value+gradient kernels (csrc/mesh_losses.hip):
uniform Laplacian smoothing, normal consistency, colour consistency of edge-adjacent faces."""
from __future__ import annotations

import numpy as np
import torch

from . import _lib


class MeshLossTopology:
    """Static adjacency of one mesh on the device: vertex -> neighbours (CSR over the edge list) and
    face -> (pair, side) (CSR over `face_connectivity`).  Rebuild after subdivide()."""

    def __init__(self, edges: torch.Tensor, face_connectivity: torch.Tensor, n_verts: int, n_faces: int, device, normal_pairs: torch.Tensor = None):
        """face_connectivity: the pairs of the colour consistency (models/model.py:115-125, which skips the last edge id);
        normal_pairs: ALL edge-adjacent face pairs, what PyTorch3D's mesh_normal_consistency(mesh) sums over (train.py:149);
        defaults to face_connectivity."""
        e = edges.detach().cpu().numpy().astype(np.int64)
        src = np.concatenate([e[:, 0], e[:, 1]])
        dst = np.concatenate([e[:, 1], e[:, 0]])
        order = np.lexsort((dst, src))                 # neighbours in ascending order: fixed summation order
        off = np.zeros(n_verts + 1, np.int64)
        np.add.at(off, src + 1, 1)
        self.nbr_off = torch.from_numpy(np.cumsum(off).astype(np.int32)).to(device)
        self.nbr_idx = torch.from_numpy(dst[order].astype(np.int32)).to(device)
        self.n_pairs, self.pairs, self.fp_off, self.fp_idx = self._pair_csr(face_connectivity, n_faces, device)
        if normal_pairs is None:
            self.n_npairs, self.npairs, self.nfp_off, self.nfp_idx = self.n_pairs, self.pairs, self.fp_off, self.fp_idx
        else:
            self.n_npairs, self.npairs, self.nfp_off, self.nfp_idx = self._pair_csr(normal_pairs, n_faces, device)
        self.n_verts, self.n_faces = int(n_verts), int(n_faces)

    @staticmethod
    def _pair_csr(pairs: torch.Tensor, n_faces: int, device):
        p = pairs.detach().cpu().numpy().astype(np.int64).reshape(-1, 2)
        flat_face = p.reshape(-1)                      # entry index = pair*2 + side
        order = np.argsort(flat_face, kind="stable")
        off = np.zeros(n_faces + 1, np.int64)
        np.add.at(off, flat_face + 1, 1)
        return (int(p.shape[0]), torch.from_numpy(p.astype(np.int32)).contiguous().to(device),
                torch.from_numpy(np.cumsum(off).astype(np.int32)).to(device), torch.from_numpy(order.astype(np.int32)).to(device))


class _Laplacian(torch.autograd.Function):
    @staticmethod
    def forward(ctx, verts, lt: MeshLossTopology, reduce=True):
        lib = _lib.load()
        v = verts.float().contiguous()
        dirs = torch.empty_like(v)
        partials = torch.empty(_lib.GOM_LOSS_BLOCKS, dtype=torch.float32, device=v.device)
        _lib.check(lib.gom_mesh_laplacian(v.shape[0], _lib.ptr(v), _lib.ptr(lt.nbr_off), _lib.ptr(lt.nbr_idx), _lib.ptr(dirs), _lib.ptr(partials), _lib.stream_ptr()))
        ctx.save_for_backward(dirs)
        ctx.lt = lt
        return partials.sum() if reduce else partials.view(1, -1)   # (reduce=False: the caller sums -- train_util.compute_loss folds every term in one launch)

    @staticmethod
    def backward(ctx, g):
        (dirs,) = ctx.saved_tensors
        lt, lib = ctx.lt, _lib.load()
        go = g.reshape(-1)[:1].float().contiguous()   # (a scalar, or the expanded gradient of the partial sums: every element is dL/d loss)
        d = torch.empty_like(dirs)
        _lib.check(lib.gom_mesh_laplacian_backward(dirs.shape[0], _lib.ptr(dirs), _lib.ptr(lt.nbr_off), _lib.ptr(lt.nbr_idx), _lib.ptr(go), _lib.ptr(d), _lib.stream_ptr()))
        return d, None, None


class _NormalConsistency(torch.autograd.Function):
    @staticmethod
    def forward(ctx, verts, topo, lt: MeshLossTopology, reduce=True):
        lib = _lib.load()
        v = verts.float().contiguous()
        pg = torch.empty((max(lt.n_npairs, 1), 2, 3), dtype=torch.float32, device=v.device)
        partials = torch.empty(_lib.GOM_LOSS_BLOCKS, dtype=torch.float32, device=v.device)
        _lib.check(lib.gom_mesh_normal_consistency(lt.n_npairs, _lib.ptr(lt.npairs), _lib.ptr(v), _lib.ptr(topo.faces), _lib.ptr(pg), _lib.ptr(partials), _lib.stream_ptr()))
        ctx.save_for_backward(v, pg)
        ctx.topo, ctx.lt = topo, lt
        return partials.sum() if reduce else partials.view(1, -1)

    @staticmethod
    def backward(ctx, g):
        v, pg = ctx.saved_tensors
        topo, lt, lib = ctx.topo, ctx.lt, _lib.load()
        go = g.reshape(-1)[:1].float().contiguous()   # (a scalar, or the expanded gradient of the partial sums: every element is dL/d loss)
        scratch = torch.empty((topo.n_faces, 9), dtype=torch.float32, device=v.device)
        d = torch.empty_like(v)
        _lib.check(lib.gom_mesh_normal_consistency_backward(v.shape[0], topo.n_faces, lt.n_npairs, _lib.ptr(lt.nfp_off), _lib.ptr(lt.nfp_idx), _lib.ptr(pg), _lib.ptr(v),
                                                            _lib.ptr(topo.faces), _lib.ptr(topo.csr_off), _lib.ptr(topo.csr_idx), _lib.ptr(go), _lib.ptr(scratch),
                                                            _lib.ptr(d), _lib.stream_ptr()))
        return d, None, None, None


class _ColorConsistency(torch.autograd.Function):
    @staticmethod
    def forward(ctx, colors_3F, lt: MeshLossTopology, reduce=True):
        lib = _lib.load()
        c = colors_3F.float().contiguous()
        F = c.shape[1]
        sign = torch.empty((max(lt.n_pairs, 1), 3), dtype=torch.float32, device=c.device)
        partials = torch.empty(_lib.GOM_LOSS_BLOCKS, dtype=torch.float32, device=c.device)
        _lib.check(lib.gom_mesh_color_consistency(lt.n_pairs, F, _lib.ptr(lt.pairs), _lib.ptr(c), _lib.ptr(sign), _lib.ptr(partials), _lib.stream_ptr()))
        ctx.save_for_backward(sign)
        ctx.lt, ctx.F = lt, F
        return partials.sum() if reduce else partials.view(1, -1)

    @staticmethod
    def backward(ctx, g):
        (sign,) = ctx.saved_tensors
        lt, lib = ctx.lt, _lib.load()
        go = g.reshape(-1)[:1].float().contiguous()   # (a scalar, or the expanded gradient of the partial sums: every element is dL/d loss)
        d = torch.empty((3, ctx.F), dtype=torch.float32, device=sign.device)
        _lib.check(lib.gom_mesh_color_consistency_backward(ctx.F, lt.n_pairs, _lib.ptr(lt.fp_off), _lib.ptr(lt.fp_idx), _lib.ptr(sign), _lib.ptr(go), _lib.ptr(d),
                                                           _lib.stream_ptr()))
        return d, None, None


def laplacian_smoothing(verts_N3: torch.Tensor, lt: MeshLossTopology, reduce: bool = True) -> torch.Tensor:
    """reduce=False: the (1, GOM_LOSS_BLOCKS) partial sums whose sum is the loss (for losses.loss_tail)."""
    return _Laplacian.apply(verts_N3, lt, reduce)


def normal_consistency(verts_N3: torch.Tensor, topo, lt: MeshLossTopology, reduce: bool = True) -> torch.Tensor:
    return _NormalConsistency.apply(verts_N3, topo, lt, reduce)


def color_consistency(colors_3F: torch.Tensor, lt: MeshLossTopology, reduce: bool = True) -> torch.Tensor:
    """colors in the (3, F) parameter layout (appearance_module.py:14)."""
    return _ColorConsistency.apply(colors_3F, lt, reduce)
