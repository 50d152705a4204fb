"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU restatement of LPIPS-VGG
(reference utils/lpips/lpips.py:81-123, utils/lpips/__init__.py:40-42,
utils/lpips/pretrained_networks.py:96-134) in plain torch ops.

Pinned by tests/golden/lpips_vgg.npz, which scripts/make_goldens.py records from
the reference's own `LPIPS(net='vgg')` class with its torchvision trunk replaced by
a seeded random VGG16 (the ImageNet weights are not obtainable offline -> absolute
LPIPS values against the real trunk: parity unpinned)."""
import torch
import torch.nn.functional as F

SHIFT = (-0.030, -0.088, -0.188)
SCALE = (0.458, 0.448, 0.450)
_TAPS = (1, 3, 6, 9, 12)
_POOLS = (2, 4, 7, 10)


def normalize_tensor(f, eps=1e-10):          # utils/lpips/__init__.py:40-42
    n = torch.sqrt(torch.sum(f ** 2, dim=1, keepdim=True) + eps)
    return f / (n + eps)


def vgg_taps(x, wb):                          # pretrained_networks.py:96-134
    taps, h = [], x
    for i in range(13):
        if i in _POOLS:
            h = F.max_pool2d(h, kernel_size=2, stride=2)
        h = F.relu(F.conv2d(h, wb[2 * i], wb[2 * i + 1], padding=1))
        if i in _TAPS:
            taps.append(h)
    return taps


def lpips_vgg(in0, in1, trunk_wb, lins, per_layer=False):
    """in0, in1 (B,3,H,W) in [-1,1]; trunk_wb = [w0,b0,...]; lins = 5 tensors (C,). Returns (B,1,1,1)."""
    dt = in0.dtype
    shift = torch.tensor(SHIFT, dtype=dt).view(1, 3, 1, 1)
    scale = torch.tensor(SCALE, dtype=dt).view(1, 3, 1, 1)
    wb = [t.to(dt) for t in trunk_wb]
    o0, o1 = vgg_taps((in0 - shift) / scale, wb), vgg_taps((in1 - shift) / scale, wb)
    res = []
    for k in range(5):
        d = (normalize_tensor(o0[k]) - normalize_tensor(o1[k])) ** 2          # lpips.py:104-106
        lin = (d * lins[k].to(dt).view(1, -1, 1, 1)).sum(1, keepdim=True)     # NetLinLayer: 1x1 conv, no bias
        res.append(lin.mean([2, 3], keepdim=True))                            # spatial_average
    val = res[0]
    for k in range(1, 5):
        val = val + res[k]
    return (val, res) if per_layer else val
