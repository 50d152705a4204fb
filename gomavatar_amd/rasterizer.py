"""Drop-in `diff_gaussian_rasterization` call surface on top of libgom_hip.so.

Mirrors the public Python API the reference uses
(models/modules/renderer/gaussian.py:9,20,53-67,83-91): a 12-field
`GaussianRasterizationSettings` NamedTuple constructed by keyword and a
`GaussianRasterizer(nn.Module)` whose `raster_settings` attribute is
re-assigned per frame and whose `forward(...)` returns the 2-tuple
`(color (C,H,W), radii (P,) int32)`.  Extras over the CUDA extension:

* 3- or 4-channel `colors_precomp` (the reference pads RGB+alpha to 6 channels
  and rasterizes twice; one 4-channel pass is exact and half the work);
* a second call on identical geometry/camera (the reference's pattern) re-uses
  the first call's projection, binning and sort;
* no device->host read of the pair count; gradients are bitwise reproducible.

There is no CPU fallback: tensors must live on the HIP device.
"""
from __future__ import annotations

import ctypes
import threading
from typing import NamedTuple, Optional

import torch
import torch.nn as nn

from . import _lib


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


class RasterState:
    """One GomState (scratch of one in-flight frame).  States created directly
    by the user are pinned; those handed out by the pool go back to it when the
    last holder (autograd context or rasterizer cache) lets go."""

    def __init__(self):
        lib = _lib.load()
        self.handle = lib.gom_state_create()
        if not self.handle:
            raise RuntimeError("gom_state_create failed: " + lib.gom_last_error().decode())
        self.users = 0
        self.pool_key = None  # set by the pool for pooled states
        self.owner = 0        # id of the forward whose colour-dependent checkpoints the state currently holds

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                _lib.load().gom_state_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def set_option(self, opt: int, value: int) -> None:
        _lib.check(_lib.load().gom_state_set_option(self.handle, opt, int(value)))

    def poll(self):
        """(num_pairs, overflowed) of the last forward; synchronises the stream."""
        n = ctypes.c_int64(0)
        ov = ctypes.c_int32(0)
        _lib.check(_lib.load().gom_state_poll(self.handle, ctypes.byref(n), ctypes.byref(ov), _lib.stream_ptr()))
        return int(n.value), bool(ov.value)

    def poll_flags(self) -> int:
        """The raw overflow word of the last forward: bit 0 = pair buffers (image poisoned), bit 1 = the record buffer of GOM_OPT_BWD_MODE 3
        (gradients poisoned); synchronises the stream."""
        n = ctypes.c_int64(0)
        ov = ctypes.c_int32(0)
        _lib.check(_lib.load().gom_state_poll(self.handle, ctypes.byref(n), ctypes.byref(ov), _lib.stream_ptr()))
        return int(ov.value)

    def kernel_times_ms(self) -> dict:
        """Per-kernel duration (ms) of the most recent launches; needs set_option(OPT_PROFILE, 1)."""
        arr = (ctypes.c_float * len(_lib.KERNEL_NAMES))()
        _lib.check(_lib.load().gom_state_kernel_times(self.handle, arr))
        return {n: float(arr[i]) for i, n in enumerate(_lib.KERNEL_NAMES)}

    def export(self, buf_id: int, out: torch.Tensor) -> torch.Tensor:
        _lib.check(_lib.load().gom_state_export(self.handle, buf_id, _lib.ptr(out), out.numel() * out.element_size(), _lib.stream_ptr()))
        return out


class _StatePool:
    """Any number of frames may be in flight: a state is leased to a forward and
    returns when its backward has run (or its autograd context died)."""

    def __init__(self):
        self._free = {}
        self._lock = threading.Lock()

    def acquire(self, device: torch.device) -> RasterState:
        idx = device.index if device.index is not None else torch.cuda.current_device()
        with self._lock:
            lst = self._free.setdefault(idx, [])
            st = lst.pop() if lst else None
        if st is None:
            with torch.cuda.device(idx):
                st = RasterState()
            st.pool_key = idx
        st.users = 1
        return st

    def retain(self, st: RasterState) -> None:
        with self._lock:
            st.users += 1

    def release(self, st: RasterState) -> None:
        with self._lock:
            st.users -= 1
            if st.users <= 0 and st.pool_key is not None:
                st.users = 0
                self._free.setdefault(st.pool_key, []).append(st)


_POOL = _StatePool()


class _StateLease:
    """Lets go of the state when finished or when the autograd graph drops it."""

    def __init__(self, st: RasterState, pool: "_StatePool" = None):
        self.st = st
        self.done = False
        self.pool = pool if pool is not None else _POOL

    def finish(self):
        if not self.done:
            self.done = True
            self.pool.release(self.st)

    def __del__(self):
        try:
            self.finish()
        except Exception:
            pass


def camera_from_settings(rs: GaussianRasterizationSettings) -> _lib.GomCamera:
    """Host-side camera struct.  viewmatrix/projmatrix/bg are read back from the
    device once per distinct settings object (the reference itself does four
    `.item()` syncs per frame, gaussian.py:30-31)."""
    view = rs.viewmatrix.detach().to("cpu", torch.float32).contiguous().reshape(-1).tolist()
    proj = rs.projmatrix.detach().to("cpu", torch.float32).contiguous().reshape(-1).tolist()
    bg = rs.bg.detach().to("cpu", torch.float32).reshape(-1).tolist()
    return _lib.make_camera(rs.image_height, rs.image_width, rs.tanfovx, rs.tanfovy, view, proj, bg)


class DeviceCamera:
    """A GomCamera that lives in device memory (`data`: 40 x 4 bytes in the struct's layout) and is read by the kernels
    themselves.  Nothing of it is baked into the launches, so a rasterizer call made with it can be part of a captured
    HIP graph that is replayed for other frames: `update(K, E)` rewrites the same 160 bytes with device-side tensor
    ops (no host read of K / E, unlike gaussian.py:30-31)."""

    def __init__(self, H: int, W: int, device):
        self.H, self.W = int(H), int(W)
        self.data = torch.zeros(40, dtype=torch.float32, device=device)
        self.data[:2] = torch.tensor([self.H, self.W], dtype=torch.int32).view(torch.float32).to(device)
        self._bg_host = (0.0, 0.0, 0.0, 0.0)     # what data[36:40] currently holds, when it came from a host tuple

    @classmethod
    def fresh(cls, H: int, W: int, K: torch.Tensor, E: torch.Tensor, bg4: torch.Tensor, znear: float = 0.001, zfar: float = 100.0) -> "DeviceCamera":
        """A camera of its own for ONE forward / backward pair (a later forward cannot overwrite it under an earlier backward): 160 uninitialised bytes
        that one launch fills completely (H, W, tanfov, view, proj, bg).  K (3,3), E (4,4), bg4 (4,): contiguous fp32 device tensors."""
        self = cls.__new__(cls)
        self.H, self.W = int(H), int(W)
        self.data = torch.empty(40, dtype=torch.float32, device=K.device)
        self._bg_host = None
        _lib.check(_lib.load().gom_camera_update_device(_lib.ptr(K.detach()), _lib.ptr(E.detach()), self.H, self.W, float(znear), float(zfar), _lib.ptr(bg4),
                                                        _lib.ptr(self.data), _lib.stream_ptr()))
        return self

    def update(self, K: torch.Tensor, E: torch.Tensor, bg=(0.0, 0.0, 0.0, 0.0), znear: float = 0.001, zfar: float = 100.0) -> "DeviceCamera":
        """K (3,3), E (4,4) device tensors -> tanfov, viewmatrix = E^T, projmatrix = E^T K_ndc^T (gaussian.py:28-51)."""
        if (K.is_cuda and K.dtype == torch.float32 and E.dtype == torch.float32 and K.is_contiguous() and E.is_contiguous() and tuple(K.shape) == (3, 3)
                and tuple(E.shape) == (4, 4) and (torch.is_tensor(bg) or tuple(float(b) for b in bg) == self._bg_host)):
            # ONE launch (csrc/gom_api.hip k_camera_update: camera_block's arithmetic in its order and precision) instead of ~40 small tensor launches
            bgt = bg.detach().float().reshape(-1).contiguous() if torch.is_tensor(bg) else None
            if bgt is not None:
                self._bg_host = None
            _lib.check(_lib.load().gom_camera_update_device(_lib.ptr(K.detach()), _lib.ptr(E.detach()), self.H, self.W, float(znear), float(zfar), _lib.ptr(bgt),
                                                            _lib.ptr(self.data), _lib.stream_ptr()))
            return self
        from .camera import camera_block
        tanfov, view, proj = camera_block(K, E, self.H, self.W, znear, zfar)      # the one camera function of the package, on the device
        self.data[2:4] = tanfov
        self.data[4:20] = view.reshape(-1)
        self.data[20:36] = proj.reshape(-1)
        if torch.is_tensor(bg):
            self.data[36:40] = bg.detach().float().reshape(-1)
            self._bg_host = None
        elif tuple(float(b) for b in bg) != self._bg_host:   # host -> device copy: only when the value really changes (never inside a capture)
            self._bg_host = tuple(float(b) for b in bg)
            self.data[36:40] = torch.tensor(self._bg_host, dtype=torch.float32).to(K.device)
        return self


class _Rasterize(torch.autograd.Function):
    _next_id = 0

    @staticmethod
    def forward(ctx, means3D, means2D, colors, opacities, cov6, cam, lease, reuse):
        lib = _lib.load()
        P, C = colors.shape
        H, W = cam.H, cam.W
        means3D_c = means3D.contiguous()
        colors_c = colors.contiguous()
        opac_c = opacities.contiguous().reshape(-1)
        cov_c = cov6.contiguous()
        out = torch.empty((C, H, W), dtype=torch.float32, device=means3D.device)
        radii = torch.empty((P,), dtype=torch.int32, device=means3D.device)
        if isinstance(cam, DeviceCamera):
            _lib.check(lib.gom_raster_forward_dcam(lease.st.handle, H, W, _lib.ptr(cam.data), P, C, _lib.ptr(means3D_c), _lib.ptr(cov_c),
                                                   _lib.ptr(colors_c), _lib.ptr(opac_c), _lib.ptr(out), _lib.ptr(radii),
                                                   _lib.GOM_FWD_REUSE_BINNING if reuse else 0, _lib.stream_ptr()))
        else:
            _lib.check(lib.gom_raster_forward(lease.st.handle, ctypes.byref(cam), P, C, _lib.ptr(means3D_c), _lib.ptr(cov_c),
                                              _lib.ptr(colors_c), _lib.ptr(opac_c), _lib.ptr(out), _lib.ptr(radii),
                                              _lib.GOM_FWD_REUSE_BINNING if reuse else 0, _lib.stream_ptr()))
        _Rasterize._next_id += 1
        ctx.fid = _Rasterize._next_id
        lease.st.owner = ctx.fid
        ctx.cam = cam
        ctx.lease = lease
        ctx.opac_shape = opacities.shape
        ctx.save_for_backward(means3D_c, colors_c, opac_c, cov_c)
        ctx.mark_non_differentiable(radii)
        ctx.set_materialize_grads(False)    # (no zero tensor filled for the radii's "gradient")
        return out, radii

    @staticmethod
    def backward(ctx, g_out, _g_radii):
        lib = _lib.load()
        means3D, colors, opac, cov6 = ctx.saved_tensors
        P, C = colors.shape
        if g_out is None:
            g_out = torch.zeros((C, ctx.cam.H, ctx.cam.W), dtype=torch.float32, device=means3D.device)
        g = g_out.contiguous()
        d_means = torch.empty_like(means3D)
        d_cov = torch.empty_like(cov6)
        d_col = torch.empty_like(colors)
        d_op = torch.empty_like(opac)
        d_m2d = torch.empty((P, 3), dtype=torch.float32, device=means3D.device)
        flags = _lib.GOM_BWD_RECOMPUTE_FORWARD if ctx.lease.st.owner != ctx.fid else 0
        if isinstance(ctx.cam, DeviceCamera):
            _lib.check(lib.gom_raster_backward_dcam(ctx.lease.st.handle, ctx.cam.H, ctx.cam.W, _lib.ptr(ctx.cam.data), P, C, _lib.ptr(means3D),
                                                    _lib.ptr(cov6), _lib.ptr(colors), _lib.ptr(opac), _lib.ptr(g), _lib.ptr(d_means), _lib.ptr(d_cov),
                                                    _lib.ptr(d_col), _lib.ptr(d_op), _lib.ptr(d_m2d), flags, _lib.stream_ptr()))
        else:
            _lib.check(lib.gom_raster_backward(ctx.lease.st.handle, ctypes.byref(ctx.cam), P, C, _lib.ptr(means3D), _lib.ptr(cov6),
                                               _lib.ptr(colors), _lib.ptr(opac), _lib.ptr(g), _lib.ptr(d_means), _lib.ptr(d_cov),
                                               _lib.ptr(d_col), _lib.ptr(d_op), _lib.ptr(d_m2d), flags, _lib.stream_ptr()))
        ctx.lease.st.owner = ctx.fid
        ctx.lease.finish()
        return d_means, d_m2d, d_col, d_op.reshape(ctx.opac_shape), d_cov, None, None, None


def rasterize(means3D: torch.Tensor, cov6: torch.Tensor, colors: torch.Tensor, opacities: torch.Tensor, cam: _lib.GomCamera,
              means2D: Optional[torch.Tensor] = None, state: Optional[RasterState] = None, reuse_binning: bool = False):
    """Functional entry: (P,3), (P,6), (P,C in {3,4}), (P,) or (P,1) -> (C,H,W), radii.
    `state` pins the scratch (needed for `reuse_binning` and for export-based tests)."""
    if not means3D.is_cuda:
        raise RuntimeError("gomavatar_amd.rasterize: tensors must be on the HIP device (no CPU fallback)")
    if colors.shape[-1] not in (3, 4):
        raise ValueError("colors must have 3 or 4 channels")
    if means2D is None:   # (a carrier for dL/dmeans2D when the caller wants it; its values are never read: not filled)
        means2D = torch.empty((means3D.shape[0], 3), dtype=means3D.dtype, device=means3D.device)
    if state is None:
        st = _POOL.acquire(means3D.device)
    else:
        st = state
        _POOL.retain(st)
    lease = _StateLease(st)
    out, radii = _Rasterize.apply(means3D, means2D, colors, opacities, cov6, cam, lease, reuse_binning)
    if not out.requires_grad:  # no autograd graph was recorded: nothing will call backward
        lease.finish()
    return out, radii


class GaussianRasterizer(nn.Module):
    """diff_gaussian_rasterization.GaussianRasterizer, HIP-backed."""

    def __init__(self, raster_settings: Optional[GaussianRasterizationSettings] = None):
        super().__init__()
        self.raster_settings = raster_settings
        self._cache = None  # (settings, inputs kept alive, versions, state, camera)

    def markVisible(self, positions: torch.Tensor) -> torch.Tensor:
        with torch.no_grad():
            v = self.raster_settings.viewmatrix
            z = positions @ v[:3, 2] + v[3, 2]
            return z > 0.2

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None):
        rs = self.raster_settings
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception("Please provide excatly one of either SHs or precomputed colors!")
        if ((scales is None or rotations is None) and cov3D_precomp is None) or (
                (scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
        if shs is not None:
            raise NotImplementedError("spherical-harmonics colours are not on the GoMAvatar path (sh_degree=0, shs=None)")
        if cov3D_precomp is None:
            cov3D_precomp = _cov6_from_scale_rotation(scales, rotations, rs.scale_modifier)

        # The reference calls the rasterizer twice per frame on identical geometry
        # (gaussian.py:82-92).  Detect that and re-use binning/sort.
        key_tensors = (means3D, cov3D_precomp, opacities)
        sig = tuple((t.untyped_storage().data_ptr(), t.data_ptr(), t._version, tuple(t.shape), tuple(t.stride())) for t in key_tensors)
        c = self._cache
        if c is not None and c["rs"] is rs and c["sig"] == sig:
            return rasterize(means3D, cov3D_precomp, colors_precomp, opacities, c["cam"], means2D=means2D,
                             state=c["state"], reuse_binning=True)
        self._drop_cache()
        cam = camera_from_settings(rs)
        st = _POOL.acquire(means3D.device)  # this hold belongs to the cache
        self._cache = dict(rs=rs, sig=sig, keep=key_tensors, state=st, cam=cam)
        return rasterize(means3D, cov3D_precomp, colors_precomp, opacities, cam, means2D=means2D, state=st)

    def _drop_cache(self):
        c, self._cache = self._cache, None
        if c is not None:
            _POOL.release(c["state"])

    def __del__(self):
        try:
            self._drop_cache()
        except Exception:
            pass


def _cov6_from_scale_rotation(scales, rotations, scale_modifier: float) -> torch.Tensor:
    """computeCov3D of the CUDA extension, in plain torch (not used by GoMAvatar,
    kept so the API surface is whole): Sigma = R S S^T R^T, R from (w,x,y,z)."""
    q = rotations / rotations.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    r, x, y, z = q.unbind(-1)
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                     2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                     2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=-1).reshape(-1, 3, 3)
    M = R * (scales * scale_modifier)[:, None, :]
    S = M @ M.transpose(1, 2)
    return torch.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], dim=-1)
