"""The per-frame hot path as one pre-planned launch sequence (no autograd, no
allocation, no host sync): the production fast path used by bench.py, by the
frame-parallel trainer and by smoke().

    FK -> LBS -> per-face Gaussians -> splat forward (one 4-channel pass)
       -> fused unpack + L1(rgb) + L1(mask) forward/backward
       -> splat backward -> face backward -> vertex gather + LBS backward

i.e. reference models/model.py:213-250 + models/modules/renderer/gaussian.py:22-100
+ train.py:53-55,101-111 and the autograd backward of all of it, as 12 kernel
launches on one stream (13 for a batch: + the frame sum).  Gradients land in `self.grads` (vertices (3,N), so3
(3,F), scale (3,F), appearance (3,F)) and are bitwise reproducible.

`batch=B > 1` runs B frames through the SAME launches (every kernel covers
all B frames; `gom_batch_forward_backward`): per-frame tensors get a leading B
dimension, the gradients are the sum over the B frames.
"""
from __future__ import annotations

import ctypes
import os
from typing import Dict, Optional

import torch

from . import _lib
from .geometry import MeshTopology, N_JOINTS
from .rasterizer import RasterState


class RenderStep:
    def __init__(self, faces: torch.Tensor, n_verts: int, img_hw, lbs_weights: torch.Tensor, sigma: float = 1e-3,
                 c_rgb: float = 1.0, c_mask: float = 5.0, device: Optional[torch.device] = None, batch: int = 1):
        self.lib = _lib.load()
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        dev = self.device
        self.topo = MeshTopology(faces, n_verts, device=dev)
        self.N, self.F = int(n_verts), self.topo.n_faces
        self.H, self.W = int(img_hw[0]), int(img_hw[1])
        self.sigma, self.c_rgb, self.c_mask = float(sigma), float(c_rgb), float(c_mask)
        self.lbs_weights = lbs_weights.to(dev, torch.float32).contiguous()
        assert self.lbs_weights.shape == (N_JOINTS + 1, self.N)
        f32 = dict(dtype=torch.float32, device=dev)
        N, F, H, W = self.N, self.F, self.H, self.W
        self.B = int(batch)
        assert self.B >= 1
        lead = () if self.B == 1 else (self.B,)   # per-frame tensors: leading batch dimension when B > 1
        self.state = RasterState()
        # forward intermediates
        self.RT = torch.empty(lead + (N_JOINTS, 12), **f32)
        self.fk_save = torch.empty(lead + (N_JOINTS, 32), **f32)
        self.v_obs = torch.empty(lead + (3, N), **f32)
        self.xyz = torch.empty(lead + (F, 3), **f32)
        self.cov6 = torch.empty(lead + (F, 6), **f32)
        self.feat = torch.ones(lead + (F, 4), **f32)       # rgb + constant 1 (alpha channel), gaussian.py:49
        self.opacity = torch.ones(lead + (F,), **f32)      # model.py:242
        self.image = torch.empty(lead + (4, H, W), **f32)  # albedo rgb + mask, CHW
        self.radii = torch.empty(lead + (F,), dtype=torch.int32, device=dev)
        # one slot per 16x16 tile (at least GOM_LOSS_BLOCKS): with GOM_OPT_FUSE_LOSS the loss rides in the forward and tile t fills slot t
        self.loss_partials = torch.zeros(lead + (_lib.load().gom_frame_loss_slots(H, W), 2), **f32)
        # backward intermediates
        self.d_image = torch.zeros(lead + (4, H, W), **f32)   # (the loss kernel leaves the pixels of empty tiles alone: finite from the start)
        self.d_xyz = torch.empty(lead + (F, 3), **f32)
        self.d_cov6 = torch.empty(lead + (F, 6), **f32)
        self.d_feat = torch.empty(lead + (F, 4), **f32)
        self.d_opacity = torch.empty(lead + (F,), **f32)
        self.d_corner = torch.empty(lead + (F, 3, 3), **f32)
        # device-resident cameras of a batched call (one GomCamera per frame) + pinned staging
        self.cams_dev = torch.zeros((self.B, ctypes.sizeof(_lib.GomCamera)), dtype=torch.uint8, device=dev)
        self._cams_host = torch.zeros((self.B, ctypes.sizeof(_lib.GomCamera)), dtype=torch.uint8).pin_memory() \
            if dev.type == "cuda" else None
        self.grads: Dict[str, torch.Tensor] = {
            "vertices": torch.empty((3, N), **f32), "so3": torch.empty((3, F), **f32), "scale": torch.empty((3, F), **f32),
            "appearance": torch.empty((3, F), **f32)}
        self.cam = None
        self._frame = None
        self._cams_copied = None
        if os.environ.get("GOM_DEBUG_ADDRS", "0") != "0":    # (development: which buffer does a faulting address belong to)
            import sys
            for nm in ("RT", "fk_save", "v_obs", "xyz", "cov6", "feat", "opacity", "image", "radii", "loss_partials", "d_image", "d_xyz", "d_cov6", "d_feat", "d_opacity",
                       "d_corner", "cams_dev", "lbs_weights"):
                t = getattr(self, nm)
                print(f"[gom torch pid {os.getpid()}] RenderStep.{nm} {t.data_ptr():#x} .. {t.data_ptr() + t.numel() * t.element_size():#x}", file=sys.stderr)
            for nm, t in list(self.grads.items()) + [("topo.faces", self.topo.faces), ("topo.csr_off", self.topo.csr_off), ("topo.csr_idx", self.topo.csr_idx)]:
                print(f"[gom torch pid {os.getpid()}] RenderStep.{nm} {t.data_ptr():#x} .. {t.data_ptr() + t.numel() * t.element_size():#x}", file=sys.stderr)
        if os.environ.get("GOM_DEBUG_POISON", "0") != "0":   # (development, with the library's switch of the same name: every `torch.empty` above starts as 0xA5 bytes)
            for t in (self.RT, self.fk_save, self.v_obs, self.xyz, self.cov6, self.image, self.radii, self.d_xyz, self.d_cov6, self.d_feat, self.d_opacity,
                      self.d_corner, *self.grads.values()):
                t.view(torch.uint8).fill_(0xA5)

    # -- inputs ---------------------------------------------------------------
    def set_cameras(self, Ks, Es, bg4=(0.0, 0.0, 0.0, 0.0)) -> None:
        """One (K, E) per frame of the batch -> device camera array (async copy on the current stream)."""
        assert len(Ks) == self.B and len(Es) == self.B
        if self._cams_copied is not None:
            self._cams_copied.synchronize()   # the previous async copy still reads the pinned staging buffer
        for b in range(self.B):
            cam = self._make_camera(Ks[b], Es[b], bg4)
            self._cams_host[b] = torch.frombuffer(bytearray(bytes(cam)), dtype=torch.uint8)
            if b == 0:
                self.cam = cam
        self.cams_dev.copy_(self._cams_host, non_blocking=True)
        self._cams_copied = torch.cuda.Event()
        self._cams_copied.record()

    def set_camera(self, K, E, bg4=(0.0, 0.0, 0.0, 0.0)) -> None:
        """K (3,3), E (4,4) host arrays/tensors -> rasterizer camera, as
        gaussian.py:30-47,53-66 (znear 0.001, zfar 100)."""
        if self.B > 1:
            return self.set_cameras([K] * self.B, [E] * self.B, bg4)
        self.cam = self._make_camera(K, E, bg4)

    def _make_camera(self, K, E, bg4):
        from .camera import camera_block
        K = torch.as_tensor(K).detach().cpu().reshape(3, 3)
        E = torch.as_tensor(E).detach().cpu().reshape(4, 4)
        tanfov, view, proj = camera_block(K, E, self.H, self.W)
        return _lib.make_camera(self.H, self.W, float(tanfov[0]), float(tanfov[1]), view.reshape(-1).numpy(), proj.reshape(-1).numpy(), list(bg4))

    # -- one frame --------------------------------------------------------------
    def _frame_struct(self) -> "_lib.GomFrame":
        P = _lib.ptr
        f = _lib.GomFrame()
        f.N, f.F, f.H, f.W = self.N, self.F, self.H, self.W
        f.sigma, f.c_rgb, f.c_mask = self.sigma, self.c_rgb, self.c_mask
        f.faces, f.csr_off, f.csr_idx, f.lbs_weights = P(self.topo.faces), P(self.topo.csr_off), P(self.topo.csr_idx), P(self.lbs_weights)
        f.image, f.loss_partials = P(self.image), P(self.loss_partials)
        f.work_RT, f.work_fk, f.work_vobs = P(self.RT), P(self.fk_save), P(self.v_obs)
        f.work_xyz, f.work_cov6, f.work_feat, f.work_opacity = P(self.xyz), P(self.cov6), P(self.feat), P(self.opacity)
        f.work_dimage, f.work_dxyz, f.work_dcov6 = P(self.d_image), P(self.d_xyz), P(self.d_cov6)
        f.work_dfeat, f.work_dopacity, f.work_dcorner, f.work_radii = P(self.d_feat), P(self.d_opacity), P(self.d_corner), P(self.radii)
        return f

    def _fill_frame(self, params, frame, target_rgb, target_mask, bgcolor) -> "_lib.GomFrame":
        """The call's GomFrame descriptor (the cached struct with this call's parameter / pose / target / gradient pointers)."""
        P = _lib.ptr
        f = self._frame
        if f is None:
            f = self._frame = self._frame_struct()
        f.cam = self.cam
        f.vertices, f.so3, f.scale, f.appearance = P(params["vertices"]), P(params["so3"]), P(params["scale"]), P(params["appearance"])
        f.cnl_gtfms, f.dst_Rs, f.dst_Ts = P(frame["cnl_gtfms"]), P(frame["dst_Rs"]), P(frame["dst_Ts"])
        f.gt_rgb, f.gt_mask, f.bgcolor = P(target_rgb), P(target_mask), P(bgcolor)
        g = self.grads
        f.g_vertices, f.g_so3, f.g_scale, f.g_appearance = P(g["vertices"]), P(g["so3"]), P(g["scale"]), P(g["appearance"])
        return f

    def forward_backward(self, params: Dict[str, torch.Tensor], frame: Dict[str, torch.Tensor], target_rgb: torch.Tensor,
                         target_mask: torch.Tensor, bgcolor: torch.Tensor, backward: bool = True, graph: bool = False,
                         image_grad_hook=None) -> None:
        """One native call (`gom_frame_forward_backward` / `gom_batch_forward_backward`) that enqueues the 12 kernels (13 for a batch).
        params: vertices (3,N), so3 (3,F), scale (3,F), appearance (3,F) device tensors.
        frame: cnl_gtfms (24,4,4), dst_Rs (24,3,3), dst_Ts (24,3) device tensors (contiguous fp32).
        target_rgb (H,W,3), target_mask (H,W), bgcolor (3,) device tensors.
        With batch=B > 1 every frame/target tensor has a leading B dimension.
        `image_grad_hook(step)`: called between the forward half (image + d(L1)/d(image) in `self.d_image`) and the backward
        half of a split call; it may ADD any other image-space gradient to `self.d_image` ((B,)4,H,W) -- see `lpips_hook`."""
        P = _lib.ptr
        f = self._fill_frame(params, frame, target_rgb, target_mask, bgcolor)
        gflag = _lib.GOM_FRAME_USE_GRAPH if graph else 0

        def call(flags):
            if self.B == 1:
                _lib.check(self.lib.gom_frame_forward_backward(self.state.handle, ctypes.byref(f), flags | gflag, _lib.stream_ptr()))
            else:
                _lib.check(self.lib.gom_batch_forward_backward(self.state.handle, ctypes.byref(f), self.B, P(self.cams_dev), flags | gflag,
                                                               _lib.stream_ptr()))
        if image_grad_hook is None or not backward:
            call(0 if backward else _lib.GOM_FRAME_FORWARD_ONLY)
        else:
            call(_lib.GOM_FRAME_FORWARD_ONLY)
            image_grad_hook(self)
            call(_lib.GOM_FRAME_BACKWARD_ONLY)

    def lpips_hook(self, lpips_model, target_rgb: torch.Tensor, bgcolor: torch.Tensor, coeff: float = 1.0):
        """image_grad_hook adding coeff * sum_b LPIPS(unpack(render_b), target_b) (train.py:53-55, 113-121) to the step: LPIPS value +
        image gradient from `LPIPSMatrixCore.value_and_grad`, chained through `unpack` into the 4-channel image gradient.
        The value of the last call is kept in `self.lpips_value` (mean over the batch)."""
        def hook(step):
            img = step.image if step.B > 1 else step.image[None]                    # (B,4,H,W)
            tgt = target_rgb if step.B > 1 else target_rgb[None]
            bg = (bgcolor if step.B > 1 else bgcolor[None])[:, :, None, None]      # (B,3,1,1)
            rgb, mask = img[:, :3], img[:, 3:4]
            unpacked = (rgb * mask + bg * (1 - mask)).permute(0, 2, 3, 1).contiguous()
            step.lpips_value, d_pred = lpips_model.value_and_grad(unpacked, tgt)    # d(mean_b)/d(pred): (B,H,W,3)
            d = d_pred.permute(0, 3, 1, 2) * (coeff * step.B)                        # gradients of a batch are SUMMED over its frames
            di = step.d_image if step.B > 1 else step.d_image[None]
            di[:, :3] += d * mask
            di[:, 3:4] += (d * (rgb - bg)).sum(1, keepdim=True)
        return hook

    def losses(self):
        """(L_rgb, L_mask) of the last frame as 0-d device tensors ((B,) tensors when batched)."""
        s = self.loss_partials.sum(-2)
        return s[..., 0] / (3.0 * self.H * self.W), s[..., 1] / float(self.H * self.W)

    def rgb_mask(self):
        """Last rendered (B,H,W,3) albedo and (B,H,W) mask, the layout
        gaussian.py:93-100 returns (B = 1 unless batched)."""
        img = self.image if self.B > 1 else self.image[None]
        pred = img.permute(0, 2, 3, 1)
        return pred[..., :3], pred[..., 3]


class _SplitState:
    """`RenderStep.state`'s interface over the branches' states (options go to all of them, pair counts and kernel times add up)."""

    def __init__(self, states):
        self.states = list(states)

    def set_option(self, opt: int, value: int) -> None:
        for st in self.states:
            st.set_option(opt, value)

    def poll(self):
        got = [st.poll() for st in self.states]
        return sum(g[0] for g in got), any(g[1] for g in got)

    def kernel_times_ms(self) -> dict:
        """Per kernel: the SUM over the branches' launches (profiling enqueues the branches kernel by kernel; they still share the chip)."""
        out = {}
        for st in self.states:
            for k, v in st.kernel_times_ms().items():
                if v >= 0.0:
                    out[k] = out.get(k, 0.0) + v
                else:
                    out.setdefault(k, -1.0)
        return out


class SplitRenderStep:
    """ONE step of `batch` frames as `split` CONCURRENT launch sequences (`gom_split_forward_backward`): `split` RenderSteps of
    batch / split frames each, every one with its own scratch state, forked and joined inside one native call (one hipGraph) and closed by
    one frame sum over all `batch` frames in frame order.  Why: the resident-grid segment kernels leave the chip idle behind their last
    workgroups at the end of each of a step's ~12 launches; two sequences side by side fill each other's tails.  Same interface and the same
    bits as RenderStep(batch=batch): `image`, `radii`, `loss_partials`, `d_image`, `xyz`, `cov6`, `v_obs` are whole-batch tensors whose
    slices the branches write; `grads` is shared by the branches (the frame sum writes it).
    (bench.py releases its split runners before it measures eager torch loops: with the round's first version of the library -- three side streams per lead state -- such a loop
    read 7 % slower next to a live split runner; not reproduced with the one stream a two-way split creates now.  LABBOOK R6.5.)"""

    _PER_FRAME = ("RT", "fk_save", "v_obs", "xyz", "cov6", "feat", "opacity", "image", "radii", "loss_partials", "d_image", "d_xyz", "d_cov6", "d_feat",
                  "d_opacity", "d_corner", "cams_dev")

    def __init__(self, faces: torch.Tensor, n_verts: int, img_hw, lbs_weights: torch.Tensor, sigma: float = 1e-3, c_rgb: float = 1.0,
                 c_mask: float = 5.0, device: Optional[torch.device] = None, batch: int = 8, split: int = 2):
        # split: the number of equal sequences, or the list of their sizes (e.g. (3, 3, 2))
        sizes = [int(batch) // int(split)] * int(split) if isinstance(split, int) else [int(x) for x in split]
        assert len(sizes) >= 1 and all(x >= 1 for x in sizes) and sum(sizes) == int(batch), "the sequences' sizes must add up to the batch"
        self.B, self.K, self.sizes = int(batch), len(sizes), sizes
        self.offs = [sum(sizes[:k]) for k in range(self.K + 1)]
        self.b = sizes[0]
        self.parts = [RenderStep(faces, n_verts, img_hw, lbs_weights, sigma, c_rgb, c_mask, device, batch=x) for x in sizes]
        p0 = self.parts[0]
        self.lib, self.device, self.topo, self.N, self.F, self.H, self.W = p0.lib, p0.device, p0.topo, p0.N, p0.F, p0.H, p0.W
        for name in self._PER_FRAME:            # whole-batch tensors; every branch's tensor becomes a view of its frames
            t0 = getattr(p0, name)
            per = t0.shape if sizes[0] == 1 and name != "cams_dev" else t0.shape[1:]
            whole = torch.zeros((self.B,) + tuple(per), dtype=t0.dtype, device=t0.device)
            if name in ("feat", "opacity"):
                whole.fill_(1.0)
            for k, p in enumerate(self.parts):
                setattr(p, name, whole[self.offs[k]:self.offs[k + 1]].view(getattr(p, name).shape))
            setattr(self, "_whole_" + name if name == "cams_dev" else name, whole)
        self.grads = p0.grads
        for p in self.parts[1:]:
            p.grads = self.grads                  # ONE dict: the caller may re-seat its tensors (bench.py: views of the flat exchange buffer)
        self.state = _SplitState([p.state for p in self.parts])
        self.cam = None

    # -- cameras: one device array of B cameras, branch k reads rows [k b, (k + 1) b) ------------------------------------------------
    @property
    def cams_dev(self) -> torch.Tensor:
        return self._whole_cams_dev

    @cams_dev.setter
    def cams_dev(self, t: torch.Tensor) -> None:
        assert t.shape == self._whole_cams_dev.shape and t.dtype == torch.uint8
        self._whole_cams_dev = t
        for k, p in enumerate(self.parts):
            p.cams_dev = t[self.offs[k]:self.offs[k + 1]]

    def set_cameras(self, Ks, Es, bg4=(0.0, 0.0, 0.0, 0.0)) -> None:
        assert len(Ks) == self.B and len(Es) == self.B
        for k, p in enumerate(self.parts):
            p.set_cameras(Ks[self.offs[k]:self.offs[k + 1]], Es[self.offs[k]:self.offs[k + 1]], bg4)
        self.cam = self.parts[0].cam

    def set_camera(self, K, E, bg4=(0.0, 0.0, 0.0, 0.0)) -> None:
        self.set_cameras([K] * self.B, [E] * self.B, bg4)

    def forward_backward(self, params, frame, target_rgb, target_mask, bgcolor, backward: bool = True, graph: bool = False, image_grad_hook=None) -> None:
        """As RenderStep.forward_backward with a leading dimension `batch` on every per-frame tensor; whole steps only."""
        if not backward or image_grad_hook is not None:
            raise NotImplementedError("SplitRenderStep runs whole steps (forward + backward, no hook): use RenderStep for split calls")
        K = self.K
        frames = (_lib.GomFrame * K)()
        for k, p in enumerate(self.parts):
            sl = slice(self.offs[k], self.offs[k + 1])
            if self.cam is not None:
                p.cam = self.cam                      # (a batched call reads only H and W from it: the per-frame cameras are the device array)
            cut = (lambda t, sl=sl: t[sl]) if self.sizes[k] > 1 else (lambda t, i=self.offs[k]: t[i])
            f = p._fill_frame(params, {n: cut(frame[n]) for n in ("cnl_gtfms", "dst_Rs", "dst_Ts")}, cut(target_rgb), cut(target_mask), cut(bgcolor))
            ctypes.memmove(ctypes.byref(frames[k]), ctypes.byref(f), ctypes.sizeof(_lib.GomFrame))
        states = (ctypes.c_void_p * K)(*[p.state.handle for p in self.parts])
        cams = (ctypes.c_void_p * K)(*[_lib.ptr(p.cams_dev) for p in self.parts])
        Bs = (ctypes.c_int32 * K)(*self.sizes)
        _lib.check(self.lib.gom_split_forward_backward(states, frames, K, Bs, cams, _lib.GOM_FRAME_USE_GRAPH if graph else 0, _lib.stream_ptr()))

    def losses(self):
        s = self.loss_partials.sum(-2)
        return s[..., 0] / (3.0 * self.H * self.W), s[..., 1] / float(self.H * self.W)

    def rgb_mask(self):
        pred = self.image.permute(0, 2, 3, 1)
        return pred[..., :3], pred[..., 3]
