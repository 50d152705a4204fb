"""LPIPS-VGG perceptual loss (reference utils/lpips/lpips.py:23-123, called at
train.py:113-121 as `LPIPS(net='vgg')(2*pred-1, 2*gt-1)`).

    in0, in1 (B,3,H,W) in [-1,1] -> ScalingLayer -> VGG16 conv trunk -> 5 taps
    (relu1_2, relu2_2, relu3_3, relu4_3, relu5_3) -> per tap: channel-normalise,
    squared difference, 1x1 "lin" layer, spatial mean -> sum over taps -> (B,1,1,1)

Two trunks for MI355X.  `LPIPSMatrixCore` (the training path): the 13 3x3 convolutions, pools, heads and their
backward as hand-written implicit-GEMM kernels on the bf16 matrix cores (`csrc/vgg_bf16.hip`, `lpips_vgg_api.hip`;
`v_mfma_f32_16x16x32_bf16`, NHWC, LDS-DMA staging) -- precision "bf16x3" (DEFAULT: hi / lo bf16 planes, three MFMA
passes = the reference's fp32 precision, value within 1.4e-6 of fp64) or "bf16" (one pass, 3 % on the value, opt-in).
`LPIPS` (the fp32 yardstick of the tests and of bench.py's `full_step_lpips_fp32` row): the trunk through the library
convolutions (torch -> MIOpen), the LPIPS-specific head as a fused HIP kernel pair per tap (`csrc/lpips.hip`) instead
of ~12 element-wise / reduction launches with full-size temporaries.

Weights: the five `lin` vectors are the LPIPS v0.1 VGG weights
(`data/lpips_vgg_lin_v0.1.npz`, 1 472 floats, BSD-licensed, converted by
scripts/make_goldens.py).  The ImageNet VGG16 trunk cannot be fetched offline:
the trunk is randomly initialised from a seed unless `load_trunk_state_dict` is
given torchvision's `vgg16().features` state dict.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

from . import _lib

# torchvision vgg16().features: conv indices 0,2 | 5,7 | 10,12,14 | 17,19,21 | 24,26,28 ; max-pools at 4, 9, 16, 23
VGG16_CONV_INDEX = (0, 2, 5, 7, 10, 12, 14, 17, 19, 21, 24, 26, 28)
VGG16_CHANNELS = (64, 64, 128, 128, 256, 256, 256, 512, 512, 512, 512, 512, 512)
TAP_AFTER_CONV = (1, 3, 6, 9, 12)          # relu1_2, relu2_2, relu3_3, relu4_3, relu5_3 (pretrained_networks.py:103-112)
POOL_BEFORE_CONV = (2, 4, 7, 10)           # a 2x2 max-pool precedes these convs
LIN_CHANNELS = (64, 128, 256, 512, 512)
SHIFT = (-0.030, -0.088, -0.188)           # ScalingLayer, lpips.py:126-133
SCALE = (0.458, 0.448, 0.450)

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "lpips_vgg_lin_v0.1.npz")


def seeded_trunk(seed: int = 0) -> List[torch.Tensor]:
    """He-initialised VGG16 conv weights/biases [w0, b0, w1, b1, ...] from a CPU generator (deterministic)."""
    g = torch.Generator().manual_seed(int(seed))
    out, cin = [], 3
    for cout in VGG16_CHANNELS:
        out.append(torch.randn(cout, cin, 3, 3, generator=g) * float(np.sqrt(2.0 / (cin * 9))))
        out.append(torch.randn(cout, generator=g) * 0.01)
        cin = cout
    return out


def trunk_features(x: torch.Tensor, wb: Sequence[torch.Tensor]) -> List[torch.Tensor]:
    """The 5 taps of the VGG16 trunk (pretrained_networks.py:96-134)."""
    taps, h = [], x
    for i in range(13):
        if i in POOL_BEFORE_CONV:
            h = F.max_pool2d(h, 2, 2)
        h = F.relu(F.conv2d(h, wb[2 * i].to(h.dtype), wb[2 * i + 1].to(h.dtype), padding=1))
        if i in TAP_AFTER_CONV:
            taps.append(h)
    return taps


class _LpipsHead(torch.autograd.Function):
    """One tap: (f0, f1, w) -> (B,) values, through the HIP kernels."""

    @staticmethod
    def forward(ctx, f0, f1, w):
        lib = _lib.load()
        f0c, f1c = f0.float().contiguous(), f1.float().contiguous()
        B, C, H, W = f0c.shape
        partials = torch.empty((B, _lib.GOM_LOSS_BLOCKS), dtype=torch.float32, device=f0c.device)
        _lib.check(lib.gom_lpips_layer_forward(B, C, H * W, _lib.ptr(f0c), _lib.ptr(f1c), _lib.ptr(w), _lib.ptr(partials), _lib.stream_ptr()))
        ctx.save_for_backward(f0c, f1c, w)
        ctx.in_dtype = f0.dtype
        return partials.sum(1)

    @staticmethod
    def backward(ctx, grad_out):
        f0c, f1c, w = ctx.saved_tensors
        lib = _lib.load()
        B, C, H, W = f0c.shape
        go = grad_out.float().contiguous()
        d_f0 = torch.empty_like(f0c)
        _lib.check(lib.gom_lpips_layer_backward(B, C, H * W, _lib.ptr(f0c), _lib.ptr(f1c), _lib.ptr(w), _lib.ptr(go), _lib.ptr(d_f0),
                                                _lib.stream_ptr()))
        return d_f0.to(ctx.in_dtype), None, None   # the target branch carries no gradient (train.py passes the ground truth)


class LPIPS(torch.nn.Module):
    """Mirror of the reference's `LPIPS(net='vgg')` (lpips.py:23-79 defaults: pretrained lin layers, version 0.1,
    eval mode, non-spatial): `forward(in0, in1, retPerLayer=False, normalize=False)` -> (B,1,1,1)."""

    def __init__(self, net: str = "vgg", trunk_seed: int = 0, trunk_dtype: torch.dtype = torch.float32, device=None):
        super().__init__()
        if net not in ("vgg", "vgg16"):
            raise NotImplementedError("only the VGG variant is on GoMAvatar's path (train.py:299-303)")
        dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        lin = np.load(_DATA)
        self.lins = [torch.from_numpy(lin[f"lin{k}"].astype(np.float32).reshape(-1)).to(dev).contiguous() for k in range(5)]
        self.trunk = [t.to(dev) for t in seeded_trunk(trunk_seed)]
        self.trunk_dtype = trunk_dtype
        self.shift = torch.tensor(SHIFT, device=dev).view(1, 3, 1, 1)
        self.scale = torch.tensor(SCALE, device=dev).view(1, 3, 1, 1)
        self.eval()

    def load_trunk_state_dict(self, sd: Dict[str, torch.Tensor]) -> None:
        """torchvision `vgg16().features` (or `vgg16()`) state dict -> trunk weights."""
        dev = self.trunk[0].device
        for i, idx in enumerate(VGG16_CONV_INDEX):
            wk = f"features.{idx}.weight" if f"features.{idx}.weight" in sd else f"{idx}.weight"
            bk = wk.replace("weight", "bias")
            assert tuple(sd[wk].shape) == tuple(self.trunk[2 * i].shape), (wk, sd[wk].shape)
            self.trunk[2 * i] = sd[wk].detach().to(dev, torch.float32).contiguous()
            self.trunk[2 * i + 1] = sd[bk].detach().to(dev, torch.float32).contiguous()

    def features(self, x: torch.Tensor) -> List[torch.Tensor]:
        x = ((x - self.shift) / self.scale).to(self.trunk_dtype)
        return trunk_features(x, self.trunk)

    def forward(self, in0: torch.Tensor, in1: torch.Tensor, retPerLayer: bool = False, normalize: bool = False):
        if normalize:   # inputs in [0,1] (lpips.py:82-84)
            in0, in1 = 2 * in0 - 1, 2 * in1 - 1
        with torch.no_grad():
            f1 = self.features(in1)
        f0 = self.features(in0)
        res = [_LpipsHead.apply(f0[k], f1[k], self.lins[k]).view(-1, 1, 1, 1) for k in range(5)]
        val = res[0]
        for k in range(1, 5):
            val = val + res[k]
        return (val, res) if retPerLayer else val


def lpips_loss(lpips: LPIPS, rgb_pred: torch.Tensor, rgb_gt: torch.Tensor) -> torch.Tensor:
    """train.py:113-117: mean over the batch of LPIPS(2*pred-1, 2*gt-1) for (B,H,W,3) images in [0,1]."""
    return lpips(2 * rgb_pred.permute(0, 3, 1, 2) - 1, 2 * rgb_gt.permute(0, 3, 1, 2) - 1).mean()


# ---------------------------------------------------------------------------------------------------------------------
# bf16 trunk on the matrix cores (csrc/vgg_bf16.hip): hand-written implicit-GEMM 3x3 convolutions instead of the library's
# ---------------------------------------------------------------------------------------------------------------------
def _pad_to(n: int, m: int) -> int:
    return (n + m - 1) // m * m


def pack_conv_weight(w: torch.Tensor) -> torch.Tensor:
    """(Cout, Cin, 3, 3) fp32 -> [Cin_p/32][9][Cout_p][32] bf16 (Cin padded to 32, Cout to 64 with zeros)."""
    co, ci = w.shape[0], w.shape[1]
    cop, cip = _pad_to(co, 64), _pad_to(ci, 32)
    wp = torch.zeros(cop, cip, 3, 3, dtype=torch.float32, device=w.device)
    wp[:co, :ci] = w
    return wp.permute(2, 3, 0, 1).reshape(9, cop, cip // 32, 32).permute(2, 0, 1, 3).contiguous().to(torch.bfloat16)


def pack_conv_weight_x3(w: torch.Tensor) -> torch.Tensor:
    """bf16x3 (csrc/vgg_bf16.hip): w = w_hi + w_lo as two bf16 tensors, laid out as THREE chunks per 32 input channels --
    (w_hi, w_hi, w_lo), the partners of the activation chunks (x_hi, x_lo, x_hi) -> [3 Cin_p/32][9][Cout_p][32] bf16."""
    co, ci = w.shape[0], w.shape[1]
    cop, cip = _pad_to(co, 64), _pad_to(ci, 32)
    wp = torch.zeros(cop, cip, 3, 3, dtype=torch.float32, device=w.device)
    wp[:co, :ci] = w
    hi = wp.to(torch.bfloat16)
    lo = (wp - hi.float()).to(torch.bfloat16)
    chunk = lambda t: t.permute(2, 3, 0, 1).reshape(9, cop, cip // 32, 32).permute(2, 0, 1, 3)      # [Cin_p/32][9][Cout_p][32]
    return torch.stack([chunk(hi), chunk(hi), chunk(lo)], 1).reshape(3 * (cip // 32), 9, cop, 32).contiguous()


def pack_conv_weight_backward(w: torch.Tensor) -> torch.Tensor:
    """Weights of the backward-data convolution: dX = conv3x3(dY, rot180(W) with in/out channels swapped)."""
    return pack_conv_weight(w.flip(2, 3).permute(1, 0, 2, 3).contiguous())


def pack_first_layer(w: torch.Tensor, x3: bool):
    """conv1_1 (Cout, 3, 3, 3) as 1 x 1 convolutions over im2col rows (csrc/vgg_bf16.hip: k_lpips_prepare_im2col): column
    k = 3 (3 ky + kx) + c.  -> (forward [chunks][Cout][32], backward-data [chunks per 32 of Cout][32][32]) bf16; bf16x3: every chunk three
    times (hi, hi, lo), the partners of (x_hi, x_lo, x_hi)."""
    co = w.shape[0]
    assert tuple(w.shape[1:]) == (3, 3, 3) and co % 32 == 0
    w1 = torch.zeros(co, 32, dtype=torch.float32, device=w.device)
    w1[:, :27] = w.permute(0, 2, 3, 1).reshape(co, 27)

    def planes(t):          # t: [chunks][rows][32] fp32
        hi = t.to(torch.bfloat16)
        if not x3:
            return hi.contiguous()
        lo = (t - hi.float()).to(torch.bfloat16)
        return torch.stack([hi, hi, lo], 1).reshape(3 * t.shape[0], t.shape[1], 32).contiguous()
    fwd = planes(w1[None])                                                   # one source chunk: the 32 im2col columns
    bwd = planes(w1.T.reshape(32, co // 32, 32).permute(1, 0, 2))            # [co / 32][k = 32][co_local = 32]
    return fwd, bwd


class LPIPSMatrixCore:
    supports_partial_sums = True       # loss(..., reduce=False) -> partial sums for train_util.compute_loss's fused tail
    """LPIPS-VGG value AND gradient w.r.t. the predicted image in one pass (train.py:113-121 semantics:
    `mean_b LPIPS(2*pred-1, 2*gt-1)`), bf16 activations / fp32 accumulation on MFMA.

        value, d_pred = m.value_and_grad(pred (B,H,W,3) in [0,1], gt (B,H,W,3))      # d_pred = d value / d pred

    H and W must be multiples of 16 (four 2x2 pools)."""

    def __init__(self, trunk_seed: int = 0, device=None, precision: str = "bf16x3"):
        """precision: "bf16x3" (default: two bf16 planes per tensor, three MFMA passes = the reference's fp32 precision on the matrix cores)
        or "bf16" (one pass, bf16 activations: 3 % on the value -- BELOW the reference's precision, hence opt-in)."""
        assert precision in ("bf16", "bf16x3")
        self.precision = precision
        self.lib = _lib.load()
        dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.device = dev
        lin = np.load(_DATA)
        self.lins = [torch.from_numpy(lin[f"lin{k}"].astype(np.float32).reshape(-1)).to(dev).contiguous() for k in range(5)]
        self._h = None
        self._side, self._target = None, None          # prefetch_target: its stream, (key, event, the fp32 copy kept alive)
        self.prefetch_enabled = os.environ.get("GOM_LPIPS_PREFETCH", "1") != "0"   # (development switch: 0 = both images as one batch at the loss)
        self.first_layer_im2col = os.environ.get("GOM_LPIPS_FIRST_LAYER_IM2COL", "1") != "0"   # (development switch: 0 = the padded 3 x 3 kernels for conv1_1 too)
        self.set_trunk([t.to(dev) for t in seeded_trunk(trunk_seed)])

    def set_trunk(self, wb: Sequence[torch.Tensor]) -> None:
        if getattr(self, "_h", None):
            self.lib.gom_lpips_vgg_destroy(self._h)
            self._h = None
        if self.precision == "bf16x3":
            self.w_fwd = [pack_conv_weight_x3(wb[2 * i]) for i in range(13)]
            self.w_bwd = [pack_conv_weight_x3(wb[2 * i].flip(2, 3).permute(1, 0, 2, 3).contiguous()) for i in range(13)]
        else:
            self.w_fwd = [pack_conv_weight(wb[2 * i]) for i in range(13)]
            self.w_bwd = [pack_conv_weight_backward(wb[2 * i]) for i in range(13)]
        self.bias = [torch.cat([wb[2 * i + 1].float(), torch.zeros(_pad_to(wb[2 * i + 1].numel(), 64) - wb[2 * i + 1].numel(), device=self.device)]).contiguous()
                     for i in range(13)]
        self.cin = [_pad_to(wb[2 * i].shape[1], 32) for i in range(13)]
        self.cout = [_pad_to(wb[2 * i].shape[0], 64) for i in range(13)]
        # the first layer without its channel padding (3 of 32 input channels are real): 1 x 1 convolutions over im2col rows
        self.w1 = pack_first_layer(wb[0], self.precision == "bf16x3") if (self.first_layer_im2col and tuple(wb[0].shape) == (64, 3, 3, 3)) else None

    def load_trunk_state_dict(self, sd: Dict[str, torch.Tensor]) -> None:
        wb = []
        for idx in VGG16_CONV_INDEX:
            wk = f"features.{idx}.weight" if f"features.{idx}.weight" in sd else f"{idx}.weight"
            wb += [sd[wk].detach().to(self.device, torch.float32), sd[wk.replace("weight", "bias")].detach().to(self.device, torch.float32)]
        self.set_trunk(wb)

    # -- one native call ---------------------------------------------------------------------------------------------
    def _handle(self):
        if self._h is None:
            import ctypes
            arr = lambda ts: (ctypes.c_void_p * len(ts))(*[_lib.ptr(t) for t in ts])
            ints = lambda v: (ctypes.c_int32 * len(v))(*v)
            self._keep = (arr(self.w_fwd), arr(self.w_bwd), arr(self.bias), arr(self.lins), ints(self.cin), ints(self.cout))
            self._h = self.lib.gom_lpips_vgg_create(*self._keep)
            if not self._h:
                _lib.check(-1)
            _lib.check(self.lib.gom_lpips_vgg_set_precision(self._h, 1 if self.precision == "bf16x3" else 0))
            if self.w1 is not None:
                _lib.check(self.lib.gom_lpips_vgg_set_first_layer(self._h, _lib.ptr(self.w1[0]), _lib.ptr(self.w1[1])))
        return self._h

    def __del__(self):
        if getattr(self, "_h", None):
            self.lib.gom_lpips_vgg_destroy(self._h)
            self._h = None

    def prefetch_target(self, gt: torch.Tensor) -> None:
        """The target image's half of the trunk forward on a second stream, to be called BEFORE the frame's forward is enqueued: it depends
        on nothing the model computes, and the geometry / raster launches of a frame leave most of the chip idle (train_util.train_iteration
        does this).  The next `value_and_grad` / `loss` with the same `gt` tensor then walks the trunk with the prediction alone."""
        assert gt.is_cuda and gt.dim() == 4 and gt.shape[-1] == 3
        if not self.prefetch_enabled:
            return
        B, H, W, _ = gt.shape
        g32 = gt.detach().float().contiguous()
        main = torch.cuda.current_stream(gt.device)
        if self._side is None:
            self._side = torch.cuda.Stream(device=gt.device)
        self._side.wait_stream(main)           # after the target tensor is written AND after the previous value_and_grad read the handle's buffers
        with torch.cuda.stream(self._side):
            _lib.check(self.lib.gom_lpips_vgg_target_features(self._handle(), B, H, W, _lib.ptr(g32), _lib.stream_ptr()))
            ev = torch.cuda.Event()
            ev.record(self._side)
        self._target = ((gt.data_ptr(), gt._version, tuple(gt.shape)), ev, g32)

    def _target_ready(self, gt: torch.Tensor) -> bool:
        t, self._target = self._target, None
        if t is None or t[0] != (gt.data_ptr(), gt._version, tuple(gt.shape)):
            if t is not None:   # another target after all: the side stream's writes must still land before this call's
                torch.cuda.current_stream(gt.device).wait_event(t[1])
            return False
        torch.cuda.current_stream(gt.device).wait_event(t[1])
        return True

    def value_and_grad(self, pred: torch.Tensor, gt: torch.Tensor, want_grad: bool = True, out=None, reduce: bool = True):
        """(mean_b LPIPS_b, d/d pred of it): ~75 kernel launches from one `gom_lpips_vgg_value_and_grad` call.
        `out=(partials, d_pred)` persistent buffers + contiguous fp32 `pred`/`gt` at stable addresses on a non-default
        stream make the call replay a captured hipGraph."""
        assert pred.is_cuda and pred.dim() == 4 and pred.shape[-1] == 3 and pred.shape == gt.shape
        B, H, W, _ = pred.shape
        p32, g32 = pred.detach().float().contiguous(), gt.detach().float().contiguous()
        if out is None:
            partials = torch.empty((5, B, _lib.GOM_LOSS_BLOCKS), dtype=torch.float32, device=pred.device)
            d_pred = torch.empty((B, H, W, 3), dtype=torch.float32, device=pred.device) if want_grad else None
        else:
            partials, d_pred = out
        flags = (1 if (out is not None and _lib.stream_ptr() != 0) else 0) | (2 if self._target_ready(gt) else 0)
        _lib.check(self.lib.gom_lpips_vgg_value_and_grad(self._handle(), B, H, W, _lib.ptr(p32), _lib.ptr(g32), _lib.ptr(partials), 1.0 / B,
                                                         _lib.ptr(d_pred), flags, _lib.stream_ptr()))
        if not reduce:                                                                 # (the caller sums: value = sum(partials) / B)
            return partials.view(1, -1), d_pred
        return (partials.sum() if B == 1 else partials.sum() * (1.0 / B)), d_pred      # mean_b of (sum over layers and blocks): one reduction

    def loss(self, rgb_pred: torch.Tensor, rgb_gt: torch.Tensor, reduce: bool = True) -> torch.Tensor:
        """Differentiable `lpips_loss` (train.py:113-117) for autograd callers.  reduce=False: the (1, 5 B GOM_LOSS_BLOCKS) partial sums,
        value = their sum / B (train_util.compute_loss folds every term's partial sums in one launch: losses.loss_tail)."""
        return _LpipsMC.apply(rgb_pred, rgb_gt, self, reduce)


class _LpipsMC(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, gt, model, reduce=True):
        value, d_pred = model.value_and_grad(pred, gt, want_grad=pred.requires_grad, reduce=reduce)
        ctx.save_for_backward(d_pred)
        ctx.dtype, ctx.scale = pred.dtype, (1.0 if reduce else float(pred.shape[0]))
        return value

    @staticmethod
    def backward(ctx, g):
        (d_pred,) = ctx.saved_tensors
        g0 = g.reshape(-1)[0]      # (a scalar, or the expanded gradient of the partial sums: every element is dL/d sum = dL/d value / B)
        if ctx.scale != 1.0:       # d_pred is d value / d pred
            g0 = g0 * ctx.scale
        return (d_pred * g0).to(ctx.dtype), None, None, None
