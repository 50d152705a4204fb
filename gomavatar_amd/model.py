"""`Model` -- mirror of the reference's `models/model.py::Model` (constructor arguments, `forward` signature and
outputs, `get_param_groups`, `subdivide`) with the per-frame hot path running through the HIP library:

    FK + LBS + per-face Gaussians   geometry.posed_face_gaussians           (model.py:213-234)
    albedo / mask splat             rasterizer.rasterize, one 4-channel pass (model.py:236-250, gaussian.py:22-100)
    vertex normals, normal map,
    soft silhouette                 mesh_renderer                            (model.py:270-273, mesh.py:65-128)
    pseudo shading                  ShadowModule (plain MLP, torch)          (model.py:279-287, shadow_module.py:66-117)

`model_cfg` is the reference's yacs node or anything with the same attributes; missing attributes fall back to the
values of configs/default.yaml + exps/zju-mocap_377.yaml.  The optional non-rigid and pose-refinement MLPs are taken as
ready-made modules (the reference's own classes are plain torch and can be passed in unchanged)."""
from __future__ import annotations

import os

import math
from types import SimpleNamespace
from typing import Optional

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from .geometry import MeshTopology, apply_lbs, face_gaussians, get_global_RTs, posed_face_gaussians
from .mesh_renderer import MeshNormalRenderer, vertex_normals
from .rasterizer import DeviceCamera, rasterize
from .losses import compose
from . import synthetic as _syn


def _get(cfg, path: str, default):
    cur = cfg
    for part in path.split("."):
        if cur is None or not hasattr(cur, part):
            return default
        cur = getattr(cur, part)
    return cur


class SimpleMesh:
    """What the reference's loss code reads from a PyTorch3D `Meshes` (one mesh): packed verts / faces / edges."""

    def __init__(self, verts: torch.Tensor, faces: torch.Tensor, edges: torch.Tensor, topo=None, loss_topo=None, normal_pairs=None):
        self._v, self._f, self._e = verts, faces, edges
        self._vc = None
        self.topo, self.loss_topo = topo, loss_topo          # device CSR adjacency for the HIP regularisers
        self._normal_pairs = normal_pairs                     # every pair of edge-adjacent faces (mesh_normal_consistency)

    @property
    def normal_pairs(self):
        if self._normal_pairs is None:
            _, f2e = mesh_edges(self._f, self._v.shape[0])
            self._normal_pairs = edge_adjacent_face_pairs(f2e).to(self._f.device)
        return self._normal_pairs

    def verts_packed(self):
        # ONE contiguous (N, 3) copy of the (3, N) parameter view for all the regularisers that read it (each made its own)
        if self._vc is None:
            self._vc = self._v.contiguous()
        return self._vc
    def faces_packed(self): return self._f
    def edges_packed(self): return self._e
    def verts_padded(self): return self.verts_packed()[None]
    def faces_padded(self): return self._f[None]


def mesh_edges(faces: torch.Tensor, n_verts: int):
    """PyTorch3D edge conventions: unique undirected edges sorted by (v0 * V + v1) with v0 < v1, and per face the
    indices of its edges (v1v2, v2v0, v0v1)."""
    f = faces.detach().cpu().numpy().astype(np.int64)
    e = np.concatenate([f[:, [1, 2]], f[:, [2, 0]], f[:, [0, 1]]], 0)
    e.sort(1)
    key = e[:, 0] * n_verts + e[:, 1]
    uniq, inv = np.unique(key, return_inverse=True)
    edges = np.stack([uniq // n_verts, uniq % n_verts], 1)
    F = f.shape[0]
    f2e = np.stack([inv[:F], inv[F:2 * F], inv[2 * F:]], 1)
    return torch.from_numpy(edges), torch.from_numpy(f2e)


def edge_adjacent_face_pairs(f2e: torch.Tensor, skip_last_edge: bool = False) -> torch.Tensor:
    """(P, 2) sorted face pairs sharing an edge.  Default: every pair of every edge with >= 2 faces (what PyTorch3D's
    mesh_normal_consistency sums over).  skip_last_edge: the reference's get_face_connectivity (models/model.py:115-125), which
    loops `for i in range(max_edge_id)` -- the last edge id never gets a pair -- and keeps edges with exactly two faces."""
    f2e_np = f2e.detach().cpu().numpy()
    flat = f2e_np.reshape(-1)
    order = np.argsort(flat, kind="stable")
    eid, fid = flat[order], order // 3
    starts = np.flatnonzero(np.r_[True, eid[1:] != eid[:-1]])
    counts = np.diff(np.r_[starts, len(eid)])
    if skip_last_edge:
        keep = (counts == 2) & (eid[starts] < f2e_np.max())
        pairs = np.stack([fid[starts[keep]], fid[starts[keep] + 1]], 1)
    else:
        two = counts == 2
        pairs = np.stack([fid[starts[two]], fid[starts[two] + 1]], 1)
        extra = [(fid[i], fid[j]) for s_, c_ in zip(starts[counts > 2], counts[counts > 2]) for i in range(s_, s_ + c_) for j in range(i + 1, s_ + c_)]
        if extra:    # non-manifold edges: all pairs, kept in edge order
            pairs = np.concatenate([pairs, np.asarray(extra, np.int64)], 0)
    pairs = np.sort(pairs.astype(np.int64).reshape(-1, 2), 1)
    return torch.from_numpy(pairs)


class _PosEnc(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, L):
        lib = _lib.load()
        x32 = x.detach().float().contiguous()
        n = x32.numel() // 3
        out = torch.empty(x32.shape[:-1] + (3 + 6 * L,), dtype=torch.float32, device=x.device)
        _lib.check(lib.gom_posenc_forward(n, L, _lib.ptr(x32), _lib.ptr(out), _lib.stream_ptr()))
        ctx.save_for_backward(x32)
        ctx.L, ctx.dtype = L, x.dtype
        return out.to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        (x32,) = ctx.saved_tensors
        lib = _lib.load()
        g32 = g.float().contiguous()
        dx = torch.empty_like(x32)
        _lib.check(lib.gom_posenc_backward(x32.numel() // 3, ctx.L, _lib.ptr(x32), _lib.ptr(g32), _lib.ptr(dx), _lib.stream_ptr()))
        return dx.to(ctx.dtype), None


class _LinearLongBatch(torch.autograd.Function):
    """F.linear whose weight / bias gradients come from csrc/mlp.hip (rows split over the workgroups); the forward and the input
    gradient are ordinary library GEMMs (their long dimension is M, which the library tiles well)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return torch.nn.functional.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        lib = _lib.load()
        out_dim, in_dim = weight.shape
        g2 = g.reshape(-1, out_dim).float().contiguous()
        x2 = x.reshape(-1, in_dim).float().contiguous()
        dW = torch.empty_like(weight, dtype=torch.float32)
        db = torch.empty(out_dim, dtype=torch.float32, device=weight.device) if ctx.has_bias else None
        ws = torch.empty(lib.gom_linear_wgrad_slices() * 129 * 128, dtype=torch.float32, device=weight.device)
        _lib.check(lib.gom_linear_wgrad(x2.shape[0], in_dim, out_dim, _lib.ptr(x2), _lib.ptr(g2), _lib.ptr(dW), _lib.ptr(db), _lib.ptr(ws), _lib.stream_ptr()))
        dx = (g2 @ weight.float()).reshape(x.shape).to(x.dtype) if ctx.needs_input_grad[0] else None
        return dx, dW.to(weight.dtype), (db.to(weight.dtype) if ctx.has_bias else None)


class _ShadowMLP3(torch.autograd.Function):
    """Linear-ReLU x 3 -> Linear -> sigmoid as one HIP kernel forward and one backward (csrc/mlp.hip: gom_mlp3_forward / _backward),
    weight gradients of the four layers through gom_mlp3_wgrad (two launches).  Same values as the nn.Sequential up to fp32
    summation order."""

    @staticmethod
    def forward(ctx, x, W1, b1, W2, b2, W3, b3, W4, b4):
        lib = _lib.load()
        x2 = x.reshape(-1, x.shape[-1]).float().contiguous()
        n, D0 = x2.shape
        H = W1.shape[0]
        ps = [t.detach().float().contiguous() for t in (W1, b1, W2, b2, W3, b3, W4, b4)]
        hs = torch.empty(3, n, H, dtype=torch.float32, device=x.device)
        out = torch.empty(n, dtype=torch.float32, device=x.device)
        _lib.check(lib.gom_mlp3_forward(n, D0, H, _lib.ptr(x2), *[_lib.ptr(t) for t in ps], _lib.ptr(hs[0]), _lib.ptr(hs[1]), _lib.ptr(hs[2]),
                                        _lib.ptr(out), _lib.stream_ptr()))
        ctx.save_for_backward(x2, hs, out, *ps)
        ctx.shape, ctx.dtype = x.shape, x.dtype
        return out.reshape(*x.shape[:-1], 1).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        x2, hs, out, W1, b1, W2, b2, W3, b3, W4, b4 = ctx.saved_tensors
        lib = _lib.load()
        n, D0 = x2.shape
        H = W1.shape[0]
        dev = x2.device
        g2 = g.reshape(-1).float().contiguous()
        dz = torch.empty(3, n, H, dtype=torch.float32, device=dev)     # dz1, dz2, dz3
        dz4 = torch.empty(n, dtype=torch.float32, device=dev)
        dx = torch.empty(n, D0, dtype=torch.float32, device=dev)
        st = _lib.stream_ptr()
        _lib.check(lib.gom_mlp3_backward(n, D0, H, _lib.ptr(g2), _lib.ptr(out), _lib.ptr(hs[0]), _lib.ptr(hs[1]), _lib.ptr(hs[2]), _lib.ptr(W1),
                                         _lib.ptr(W2), _lib.ptr(W3), _lib.ptr(W4), _lib.ptr(dz4), _lib.ptr(dz[2]), _lib.ptr(dz[1]), _lib.ptr(dz[0]),
                                         _lib.ptr(dx), st))
        ws = torch.empty(4 * lib.gom_linear_wgrad_slices() * 129 * 128, dtype=torch.float32, device=dev)
        grads = [torch.empty_like(p) for p in (W1, b1, W2, b2, W3, b3, W4, b4)]
        _lib.check(lib.gom_mlp3_wgrad(n, D0, H, _lib.ptr(x2), _lib.ptr(hs[0]), _lib.ptr(hs[1]), _lib.ptr(hs[2]), _lib.ptr(dz[0]), _lib.ptr(dz[1]), _lib.ptr(dz[2]),
                                      _lib.ptr(dz4), *[_lib.ptr(g_) for g_ in grads], _lib.ptr(ws), st))
        return (dx.reshape(ctx.shape).to(ctx.dtype) if ctx.needs_input_grad[0] else None, *grads)


class _ShadeUnderMesh(torch.autograd.Function):
    """model.py:279-283 for the default shadow MLP, fused natively (csrc/mlp.hip, gom_shade_*): shading (HW, 1) = 2 * MLP(embed(normal)) with the MLP
    evaluated on the pixels under the mesh only (normal != 0) and once for the background -- selection, embedding, scatter and their backward
    without torch glue, without a host synchronisation (the row count stays on the device), ~10 launches instead of ~35.
    The workspace (block counts, gather partials, the ROW COUNT, a counter) is allocated per forward call and saved for that call's backward:
    a second forward before the first backward (gradient accumulation over frames, a second Model, a render in between, another stream)
    must not overwrite the row count the first backward reads."""
    # GOM_MLP_MATRIX_CORES=1: the layers on the bf16 matrix cores (csrc/mlp_mc.hip: hi / lo planes, three MFMA passes).  Measured: forward 49 -> 22 us,
    # backward 63 -> 37 us (+ 8 us of weight packing) per frame (Model iteration 377 -> 383 it/s), the shading moves by 6e-6 relative.  OPT-IN, for a reason that is
    # not in this path's own results (bitwise repeatable, every buffer): while its waves are resident, a `v_pk_fma_f32 ... op_sel:[0,1,0]` of ANY OTHER kernel on the
    # same SIMD -- another stream, another process on the device -- can lose a term in lanes 48..63 (LABBOOK R6.8, scripts/ubench/pkfma_beside_mfma.hip: MFMAs fed
    # straight from LDS reads beside packed fp32; profiles/r06_coresidency/).  Made the default for half a day in round 6, it turned the eight-process bitwise test
    # intermittent through the fp32 weight-gradient kernel next door.  The trigger is an MFMA consuming a register an LDS read has only just written; the kernels now
    # request their fragments a k-step ahead like the LPIPS trunk's (which were measured NOT to do this) and the fault is down from 1e8 to 0-96 wrong results per loop of
    # the strongest victim, the neighbouring kernels of the library bitwise again -- but not to zero, so: still opt-in, still not beside other work.
    matrix_cores = os.environ.get("GOM_MLP_MATRIX_CORES", "0") != "0"

    @staticmethod
    def forward(ctx, flat, L, W1, b1, W2, b2, W3, b3, W4, b4):
        lib = _lib.load()
        x = flat.detach().float().contiguous()
        HW, dev = x.shape[0], x.device
        D0, H = 3 + 6 * L, W1.shape[0]
        ws = torch.zeros(lib.gom_shade_workspace_ints(HW), dtype=torch.int32, device=dev)      # (per call: see the class docstring)
        ps = [t.detach().float().contiguous() for t in (W1, b1, W2, b2, W3, b3, W4, b4)]
        pos = torch.empty(HW, dtype=torch.int32, device=dev)
        pe = torch.empty(HW + 1, D0, dtype=torch.float32, device=dev)
        hs = torch.empty(3, HW + 1, H, dtype=torch.float32, device=dev)
        out = torch.empty(HW + 1, dtype=torch.float32, device=dev)
        shading = torch.empty(HW, 1, dtype=torch.float32, device=dev)
        st, P = _lib.stream_ptr(), _lib.ptr
        _lib.check(lib.gom_shade_select(HW, L, P(x), P(pos), P(pe), P(ws), st))
        pack = torch.empty(lib.gom_mlp3_pack_elems(), dtype=torch.int16, device=dev) if _ShadeUnderMesh.matrix_cores else None
        _lib.check(lib.gom_mlp3_forward_rows(HW, P(ws), D0, H, P(pe), *[P(t) for t in ps], P(hs[0]), P(hs[1]), P(hs[2]), P(out), P(pack), st))
        _lib.check(lib.gom_shade_scatter(HW, P(pos), P(out), P(ws), 2.0, P(shading), st))
        ctx.save_for_backward(x, pos, pe, hs, out, ws, *ps)
        ctx.L, ctx.dtype = L, flat.dtype
        return shading.to(flat.dtype)

    @staticmethod
    def backward(ctx, g):
        x, pos, pe, hs, out, ws, W1, b1, W2, b2, W3, b3, W4, b4 = ctx.saved_tensors
        lib = _lib.load()
        HW, dev, L = x.shape[0], x.device, ctx.L
        D0, H = pe.shape[1], W1.shape[0]
        st, P = _lib.stream_ptr(), _lib.ptr
        g2 = g.reshape(-1).float().contiguous()
        g_rows = torch.empty(HW + 1, dtype=torch.float32, device=dev)
        pack = torch.empty(lib.gom_mlp3_pack_elems(), dtype=torch.int16, device=dev) if _ShadeUnderMesh.matrix_cores else None
        dz = torch.empty(3, HW + 1, H, dtype=torch.float32, device=dev)
        dz4 = torch.empty(HW + 1, dtype=torch.float32, device=dev)
        dpe = torch.empty(HW + 1, D0, dtype=torch.float32, device=dev)
        _lib.check(lib.gom_shade_backward_gather(HW, P(pos), P(g2), P(ws), 2.0, P(g_rows), st))
        _lib.check(lib.gom_mlp3_backward_rows(HW, P(ws), D0, H, P(g_rows), P(out), P(hs[0]), P(hs[1]), P(hs[2]), P(W1), P(W2), P(W3), P(W4), P(dz4), P(dz[2]),
                                              P(dz[1]), P(dz[0]), P(dpe), P(pack), st))
        wws = torch.empty(4 * lib.gom_linear_wgrad_slices() * 129 * 128, dtype=torch.float32, device=dev)
        grads = [torch.empty_like(p) for p in (W1, b1, W2, b2, W3, b3, W4, b4)]
        _lib.check(lib.gom_mlp3_wgrad_rows(HW, P(ws), D0, H, P(pe), P(hs[0]), P(hs[1]), P(hs[2]), P(dz[0]), P(dz[1]), P(dz[2]), P(dz4), *[P(t) for t in grads], P(wws), st))
        d_flat = None
        if ctx.needs_input_grad[0]:
            d_flat = torch.empty_like(x)
            _lib.check(lib.gom_shade_backward_scatter(HW, L, P(pos), P(x), P(dpe), P(d_flat), st))
            d_flat = d_flat.to(ctx.dtype)
        return (d_flat, None, *grads)


class ShadowModule(nn.Module):
    """shadow_module.py:66-117: positional encoding of the normal (multires frequencies, sin/cos, input included) ->
    MLP (width, depth, optional skip) -> sigmoid.  The last layer starts at U(-1e-5, 1e-5) / zero bias."""

    def __init__(self, multires: int = 6, mlp_width: int = 128, mlp_depth: int = 3, skips=(4,), init_scale: float = 1e-5):
        super().__init__()
        self.multires, self.skips = int(multires), tuple(skips)
        embed = 3 + 3 * 2 * self.multires
        layers = [nn.Linear(embed, mlp_width), nn.ReLU()]
        self.layers_to_cat_inputs = []
        for i in range(1, mlp_depth):
            if i in self.skips:
                self.layers_to_cat_inputs.append(len(layers))
                layers += [nn.Linear(mlp_width + embed, mlp_width), nn.ReLU()]
            else:
                layers += [nn.Linear(mlp_width, mlp_width), nn.ReLU()]
        layers += [nn.Linear(mlp_width, 1)]
        self.block_mlps = nn.ModuleList(layers)
        for m in self.block_mlps:   # initseq (network_util): xavier weights, zero bias
            if isinstance(m, nn.Linear):
                nn.init.xavier_uniform_(m.weight)
                nn.init.zeros_(m.bias)
        last = self.block_mlps[-1]
        last.weight.data.uniform_(-init_scale, init_scale)
        last.bias.data.zero_()

    def embed(self, x):
        if x.is_cuda:   # one HIP kernel forward, one backward (csrc/posenc.hip); the torch formula below serves host tensors
            return _PosEnc.apply(x, self.multires)
        freqs = 2.0 ** torch.linspace(0.0, self.multires - 1, self.multires, device=x.device, dtype=x.dtype)
        out = [x]
        for f in freqs:
            out += [torch.sin(x * f), torch.cos(x * f)]
        return torch.cat(out, -1)

    def forward(self, normals, **kwargs):
        pe = self.embed(normals)
        lin = [m for m in self.block_mlps if isinstance(m, nn.Linear)]
        if (pe.is_cuda and len(lin) == 4 and not self.layers_to_cat_inputs and lin[0].out_features <= 128 and lin[0].out_features % 4 == 0 and lin[0].in_features <= 128
                and pe.dtype == torch.float32):
            # the default shape (depth 3, the configured skip index lies beyond it): the whole MLP is one kernel each way
            return _ShadowMLP3.apply(pe, *[p for m in lin for p in (m.weight, m.bias)])
        h = pe
        for i, layer in enumerate(self.block_mlps):
            if i in self.layers_to_cat_inputs:
                h = torch.cat([h, pe], -1)
            if isinstance(layer, nn.Linear) and h.is_cuda and layer.in_features <= 128 and layer.out_features <= 128 and torch.is_grad_enabled():
                h = _LinearLongBatch.apply(h, layer.weight, layer.bias)      # same values; the weight gradient takes the HIP kernel
            else:
                h = layer(h)
        return torch.sigmoid(h)


class Model(nn.Module):
    def __init__(self, model_cfg, canonical_info, non_rigid_module: Optional[nn.Module] = None,
                 pose_refinement_module: Optional[nn.Module] = None, device=None):
        super().__init__()
        self.cfg = model_cfg
        dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.img_size = tuple(_get(model_cfg, "img_size", (512, 512)))     # (W, H) like the reference
        faces = torch.as_tensor(np.asarray(canonical_info["faces"]).astype(np.int64))
        verts = torch.as_tensor(np.asarray(canonical_info["canonical_vertex"])).float()
        self.register_buffer("faces", faces.to(dev))
        lbs = torch.as_tensor(np.asarray(canonical_info["canonical_lbs_weights"])).float().T   # (24, N)
        if _get(model_cfg, "lbs_weights.refine", False):
            raise NotImplementedError("lbs_weights.refine: the HIP LBS kernel treats the weights as constants")
        self.register_buffer("lbs_weights", torch.cat([lbs, torch.zeros_like(lbs[:1])], 0).contiguous().to(dev))
        self.vertices = nn.Parameter(verts.T.contiguous().to(dev))
        F = faces.shape[0]
        radius_scale = float(_get(model_cfg, "canonical_geometry.radius_scale", 1.0))
        self.so3 = nn.Parameter(torch.zeros(3, F, device=dev), requires_grad=bool(_get(model_cfg, "canonical_geometry.deform_so3", True)))
        self.scale = nn.Parameter(torch.ones(3, F, device=dev) * radius_scale, requires_grad=bool(_get(model_cfg, "canonical_geometry.deform_scale", True)))
        self.sigma = float(_get(model_cfg, "canonical_geometry.sigma", 1e-3))
        self.appearance = nn.Parameter(torch.ones(3, F, device=dev) * float(_get(model_cfg, "appearance.color_init", 0.5)))   # AppearanceModule
        self.register_buffer("bg_col", torch.zeros(3, device=dev))
        # capture_safe: no host read of device data anywhere in forward() (camera kept in device memory, shadow MLP on a
        # fixed-capacity pixel list), so that a whole training iteration can be captured in one HIP graph and replayed for
        # other frames (train_util.GraphedTrainStep).  Off: the reference's own behaviour (host camera, exact pixel list).
        self.capture_safe = False
        self.shadow_capacity = None          # pixels the shadow MLP is evaluated on in capture_safe mode (default H*W/3)
        self._dcam = None
        # the camera block on the device (one launch, no host read of K / E) whenever K and E are fp32 device tensors; GOM_DEVICE_CAMERA=0: the host
        # camera (the reference's .item() reads).  The same bits either way (tests/test_camera.py).
        self.device_camera = os.environ.get("GOM_DEVICE_CAMERA", "1") != "0"
        self._bg4 = None
        # rendering (no autograd) only: mesh branch and splat rasterizer on two streams.  Shortens a frame's latency inside a
        # captured graph (0.71 -> 0.62 ms at 55k faces, 1.39 -> 1.09 ms at 220k); costs host time when launched eagerly: off by default
        self.overlap_branches = False
        self.overlap_branches_train = os.environ.get("GOM_OVERLAP_BRANCHES_TRAIN", "0") == "1"   # (experiment) the same under autograd: the backward follows the forward's streams
        self._ones = None
        self._side_stream = None
        # the shading of the pixels under the mesh as one native op (csrc/mlp.hip: gom_shade_*); False: the torch selection around the MLP kernels
        self.fused_shading = os.environ.get("GOM_FUSED_SHADING", "1") != "0"
        self.non_rigid_module, self.pose_refinement_module = non_rigid_module, pose_refinement_module
        self.normal_renderer = MeshNormalRenderer(self.img_size, sigma=_get(model_cfg, "normal_renderer.sigma", None),
                                                  soft_mask=_get(model_cfg, "normal_renderer.soft_mask", True))
        if _get(model_cfg, "shadow_module.name", "basic") != "none":
            self.shadow_module = ShadowModule(_get(model_cfg, "shadow_module.multires", 6), _get(model_cfg, "shadow_module.mlp_width", 128),
                                              _get(model_cfg, "shadow_module.mlp_depth", 3), tuple(_get(model_cfg, "shadow_module.skips", (4,)))).to(dev)
        else:
            self.shadow_module = None
        self._rebuild_topology()

    # -- topology-dependent, rebuilt by subdivide() ---------------------------------------------------------------------
    def _rebuild_topology(self):
        N = self.vertices.shape[1]
        self.topo = MeshTopology(self.faces, N, device=self.vertices.device)
        edges, f2e = mesh_edges(self.faces, N)
        self.edges = edges.to(self.vertices.device)
        # get_face_connectivity (model.py:115-125): faces sharing edge i, for i in range(max_edge_id) -- the last edge id is
        # skipped by the reference's loop bound and therefore here.  Used by the colour consistency only (train.py:153-158);
        # the normal consistency is PyTorch3D's own function on the mesh (train.py:149): every edge-adjacent pair.
        self.face_connectivity = edge_adjacent_face_pairs(f2e, skip_last_edge=True).to(self.vertices.device)
        self.normal_pairs = edge_adjacent_face_pairs(f2e).to(self.vertices.device)
        from .mesh_losses import MeshLossTopology
        self.loss_topo = MeshLossTopology(self.edges, self.face_connectivity, N, self.faces.shape[0], self.vertices.device, normal_pairs=self.normal_pairs)
        v = self.vertices.detach().T
        self.target_edge_length = (v[self.edges[:, 0]] - v[self.edges[:, 1]]).norm(dim=1)     # get_init_edge_length (model.py:127-134)

    def subdivide(self, need_face_connectivity: bool = True):
        """model.py:136-179: midpoint subdivision; per-face parameters are inherited by the four children."""
        v, f, attrs = _syn.subdivide(self.vertices.detach().T.cpu().numpy(), self.faces.cpu().numpy(),
                                     {"weights": self.lbs_weights.T.detach().cpu().numpy()})
        dev = self.vertices.device
        rep = lambda t: t.detach()[..., None].repeat(1, 1, 4).reshape(t.shape[0], -1).contiguous()
        self.vertices = nn.Parameter(torch.from_numpy(v).float().T.contiguous().to(dev))
        self.faces = torch.from_numpy(f).to(dev)
        self.lbs_weights = torch.from_numpy(attrs["weights"]).float().T.contiguous().to(dev)
        self.appearance = nn.Parameter(rep(self.appearance))
        self.so3 = nn.Parameter(rep(self.so3), requires_grad=self.so3.requires_grad)
        self.scale = nn.Parameter(rep(self.scale), requires_grad=self.scale.requires_grad)
        self._rebuild_topology()

    def get_param_groups(self, cfg):
        """model.py:305-327 (lr names of configs/default.yaml)."""
        lr = cfg.lr
        # The reference's FIRST group is the skinning-weight tensor -- a buffer unless lbs_weights.refine (model.py:306-309,64-69):
        # it receives no gradient and Adam skips it, but it occupies param_groups[0] / state index 0 of every optimizer state the
        # reference saves, so it is kept for `optimizer.load_state_dict` of a reference checkpoint (formats.load_checkpoint).
        groups = [{"name": "lbs_weights", "params": [self.lbs_weights], "lr": _get(lr, "lbs_weights", 0.0)},
                  {"name": "appearance", "params": [self.appearance], "lr": lr.appearance},
                  {"name": "canonical_geometry_xyz", "params": [self.vertices], "lr": lr.canonical_geometry_xyz},
                  {"name": "canonical_geometry", "params": [self.scale], "lr": lr.canonical_geometry},
                  {"name": "canonical_geometry", "params": [self.so3], "lr": lr.canonical_geometry}]
        if self.non_rigid_module is not None:
            groups.append({"name": "non_rigid", "params": self.non_rigid_module.parameters(), "lr": lr.non_rigid})
        if self.pose_refinement_module is not None:
            groups.append({"name": "pose_refinement", "params": self.pose_refinement_module.parameters(), "lr": lr.pose_refinement})
        if self.shadow_module is not None:
            groups.append({"name": "shadow", "params": self.shadow_module.parameters(), "lr": lr.shadow})
        return groups

    # -- the per-frame forward (model.py:184-303) -------------------------------------------------------------------------
    def _camera(self, K, E, bg4):
        from .camera import camera_block
        W, H = self.img_size
        tanfov, view, proj = camera_block(K[0].detach().cpu(), E[0].detach().cpu(), H, W)      # host camera: the reference's own .item() reads
        return _lib.make_camera(H, W, float(tanfov[0]), float(tanfov[1]), view.reshape(-1).numpy(), proj.reshape(-1).numpy(), list(bg4))

    def _mesh_branch(self, vertices_observation, K, E, vo_T=None):
        """model.py:270-282: camera-space vertex normals, normal map + soft silhouette, shading = 2 * shadow_module(normal).
        vo_T: the (N, 3) contiguous copy of the posed vertices when the caller has one (training: the mesh regularisers read the same copy)."""
        # normals, normal map, silhouette (model.py:270-273)
        vn = vertex_normals(vertices_observation.T if vo_T is None else vo_T, self.topo, rotation=E[0, :3, :3])      # (E[:3,:3] @ vn.T).T inside the kernel
        normal, normal_mask = self.normal_renderer(vertices_observation.unsqueeze(0), vn[None], K, E, faces=self.faces)
        if self.shadow_module is not None:
            Bn, H, W, _ = normal.shape
            # shadow_module(normal) for every pixel (model.py:279-282).  The normal map is exactly 0 outside the mesh, where
            # the MLP output is one constant: evaluate it once there and per pixel only under the mesh (~15 % of the image).
            flat = normal.reshape(-1, 3)
            # The all-zero background normal rides along as one extra row of the same MLP call (a second call for that single
            # row would double the module's launches, forward and backward).
            lin = [m for m in self.shadow_module.block_mlps if isinstance(m, nn.Linear)]
            if (self.fused_shading and flat.is_cuda and flat.dtype == torch.float32 and len(lin) == 4 and not self.shadow_module.layers_to_cat_inputs
                    and lin[0].out_features <= 128 and lin[0].out_features % 4 == 0 and lin[0].in_features <= 128 and lin[3].out_features == 1):
                # the default module: selection, embedding, MLP, scatter and their backward natively, row count on the device (csrc/mlp.hip: gom_shade_*)
                s_all = _ShadeUnderMesh.apply(flat, self.shadow_module.multires, *[p for m in lin for p in (m.weight, m.bias)])
                return normal, normal_mask, s_all.reshape(Bn, H, W, 1)
            if self.capture_safe:
                # same result with static shapes: the pixels under the mesh are compacted into a list of fixed capacity
                # (prefix sum, no nonzero()); every unused slot points at its own dummy row (duplicate indices would send
                # the gather's backward down a serial path); more mesh pixels than slots -> NaN, loudly
                n = flat.shape[0]
                cap = int(self.shadow_capacity or n // 3)
                under = (flat != 0).any(-1)
                pos = torch.cumsum(under, 0) - 1
                slot = torch.where(under & (pos < cap), pos, torch.full_like(pos, cap))
                idx = torch.cat([n + torch.arange(cap, device=flat.device), pos[:1]]).scatter(0, slot, torch.arange(n, device=flat.device))[:cap]
                flat1 = torch.cat([flat, flat.new_zeros(cap + 1, 3)], 0)                       # row n + cap: the background normal
                idx1 = torch.cat([idx, torch.full_like(idx[:1], n + cap)])
                s_sel = self.shadow_module(flat1[idx1][None]).reshape(-1, 1)
                s_all = s_sel[-1:].expand(n + cap, 1).clone().index_put((idx,), s_sel[:-1])[:n]
                s_all = torch.where(pos[-1] >= cap, torch.full_like(s_all, float("nan")), s_all)
            else:
                idx = (flat != 0).any(-1).nonzero(as_tuple=True)[0]
                # (index_select: its backward is one index_add; `flat[idx]` goes through a sort-based accumulate, ~12 launches)
                s_sel = self.shadow_module(torch.cat([flat.index_select(0, idx), flat.new_zeros(1, 3)], 0)[None]).reshape(-1, 1)
                body, bg_val = torch.split(s_sel, [s_sel.shape[0] - 1, 1])        # (one cat in the backward instead of two padded slices)
                s_all = bg_val.expand(flat.shape[0], 1).clone().index_put((idx,), body)
            return normal, normal_mask, s_all.reshape(Bn, H, W, 1) * 2
        return normal, normal_mask, None

    def forward(self, K, E, cnl_gtfms, dst_Rs, dst_Ts, dst_posevec=None, canonical_joints=None, i_iter=1e7, bgcolor=None,
                global_R=None, global_T=None, tb=None):
        B = dst_Rs.shape[0]
        assert B == 1, "batch size 1, like the reference renderer (gaussian.py:24)"
        F = self.faces.shape[0]
        if self.pose_refinement_module is not None and i_iter >= _get(self.cfg, "pose_refinement.kick_in_iter", 0):
            delta_Rs = self.pose_refinement_module(dst_posevec)
            nb = dst_Rs.shape[1]
            dst_Rs = torch.matmul(dst_Rs.reshape(B * nb, 3, 3), delta_Rs.reshape(B * nb, 3, 3)).reshape(B, nb, 3, 3)
        vertices_canonical = self.vertices
        if self.non_rigid_module is not None and i_iter >= _get(self.cfg, "non_rigid.kick_in_iter", 0):
            vertices_pose = self.non_rigid_module(vertices_canonical.unsqueeze(0), dst_posevec, i_iter, R=None, S=None)[0][0]
        else:
            vertices_pose = vertices_canonical
        feat = None
        if global_R is None:
            # (the rasterizer's features [appearance.T | 1] come out of the face kernel: no cat forward, no slice + transpose backward)
            xyz, cov6, vertices_observation, feat = posed_face_gaussians(vertices_pose, self.so3, self.scale, dst_Rs[0], dst_Ts[0], cnl_gtfms[0],
                                                                         self.lbs_weights, self.topo, self.sigma, appearance=self.appearance)
        else:   # PeopleSnapshot test-time pose optimisation (model.py:218-221): a rigid transform after the skinning
            from .modules import rodrigues
            vertices_observation = apply_lbs(vertices_pose.unsqueeze(0), *get_global_RTs(cnl_gtfms, dst_Rs, dst_Ts), self.lbs_weights)[0]
            # RodriguesModule's own arithmetic (utils/network_util.py:64-92): theta = sqrt(1e-5 + |r|^2), "axis" r / theta (NOT a unit vector)
            Rg = rodrigues(global_R.unsqueeze(0))[0]
            vertices_observation = Rg @ vertices_observation + global_T[:, None]
            xyz, cov6 = face_gaussians(vertices_observation, self.so3, self.scale, self.topo, self.sigma)
        # pseudo albedo + mask: one 4-channel pass (the reference pads to 6 channels and rasterizes twice)
        if self._ones is None or self._ones.shape[0] != F or self._ones.device != xyz.device:
            self._ones = torch.ones(F, 1, device=xyz.device)          # (constants of the topology: not two fill launches per frame)
        if feat is None:
            feat = torch.cat([self.appearance.T, self._ones], 1)
        opacity = self._ones[:, 0]
        if self.capture_safe:
            if self._dcam is None or (self._dcam.H, self._dcam.W) != (self.img_size[1], self.img_size[0]):
                self._dcam = DeviceCamera(self.img_size[1], self.img_size[0], xyz.device)
            cam = self._dcam.update(K[0], E[0])               # 160 bytes rewritten on the device
        elif (self.device_camera and K.is_cuda and K.dtype == torch.float32 and E.dtype == torch.float32 and K[0].is_contiguous() and E[0].is_contiguous()):
            # the camera block formed ON the device by one launch, into 160 bytes of this call's own: the reference reads K and E back with .item()
            # (gaussian.py:30-31) -- a stream synchronisation per frame, behind which the device waits for the host to enqueue the frame
            if self._bg4 is None or self._bg4.device != K.device:
                self._bg4 = torch.zeros(4, device=K.device)   # bg_col = [bg_feat (zeros), 0] (model.py:243)
            cam = DeviceCamera.fresh(self.img_size[1], self.img_size[0], K[0], E[0], self._bg4)
        else:
            cam = self._camera(K, E, (0.0, 0.0, 0.0, 0.0))    # bg_col = [bg_feat (zeros), 0] (model.py:243)
        # The mesh branch (vertex normals -> normal map + silhouette -> shadow MLP) and the splat rasterizer only share their
        # input; without autograd (rendering) they are issued on two streams: both are chains of single-frame kernels that leave
        # most of the GPU idle, and side by side the frame costs the longer chain instead of the sum.
        # ONE contiguous (N, 3) copy of the posed (3, N) vertices for the vertex normals AND the mesh regularisers of compute_loss (each made its own)
        vo_T = vertices_observation.T.contiguous() if self.training else None
        overlap = xyz.is_cuda and (self.overlap_branches_train if torch.is_grad_enabled() else self.overlap_branches)
        if _ShadeUnderMesh.matrix_cores:
            overlap = False     # the matrix-core shading kernels never run beside the rasterizer's (LABBOOK R6.8: their waves can corrupt a neighbour's packed-fp32 FMA)
        if overlap:
            cur = torch.cuda.current_stream()
            if self._side_stream is None:
                self._side_stream = torch.cuda.Stream()
            self._side_stream.wait_stream(cur)
            with torch.cuda.stream(self._side_stream):
                normal, normal_mask, shadings = self._mesh_branch(vertices_observation, K, E, vo_T)
        img, _ = rasterize(xyz, cov6, feat, opacity, cam)
        if overlap:
            cur.wait_stream(self._side_stream)
            for t_ in (normal, normal_mask, shadings):
                if t_ is not None:
                    t_.record_stream(cur)
        else:
            normal, normal_mask, shadings = self._mesh_branch(vertices_observation, K, E, vo_T)
        # albedos = img[:3] as (1,H,W,3), masks = img[3], rgbs = albedos * shadings: one kernel each way (csrc/loss.hip)
        albedos, masks, rgbs = compose(img, shadings)
        outputs = {}
        if self.training:
            vo, vc = vo_T, vertices_canonical.T
            outputs.update(colors=self.appearance.T, face_connectivity=self.face_connectivity,
                           mesh=SimpleMesh(vo, self.faces, self.edges, self.topo, self.loss_topo, self.normal_pairs),
                           mesh_canonical=SimpleMesh(vc, self.faces, self.edges, self.topo, self.loss_topo, self.normal_pairs),
                           target_edge_length=self.target_edge_length,
                           albedo=albedos[0],
                           normal=normal, normal_mask=normal_mask.squeeze(-1) if normal_mask is not None else None, shadow=shadings)   # (squeeze: a view's backward is a view; a select's is a zero fill + a copy)
        return rgbs, masks, outputs
