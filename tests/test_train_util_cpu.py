"""Host-tensor paths of train_util (`unpack`, `compute_loss` with the torch formulas of the three mesh regularisers): no GPU, no HIP
library call.  The device paths of the same functions are compared with these formulas in tests/test_gpu_loss.py / test_gpu_model.py."""
from types import SimpleNamespace as NS

import torch
import torch.nn.functional as F

from gomavatar_amd import synthetic as syn
from gomavatar_amd.model import SimpleMesh, mesh_edges
from gomavatar_amd.train_util import compute_loss, unpack


def test_unpack_is_the_reference_formula():
    g = torch.Generator().manual_seed(0)
    rgbs, masks, bg = torch.rand(2, 5, 7, 3, generator=g), torch.rand(2, 5, 7, generator=g), torch.rand(2, 3, generator=g)
    out = unpack(rgbs, masks, bg)                                 # train.py:53-55
    ref = rgbs * masks[..., None] + bg[:, None, None, :] * (1 - masks[..., None])
    assert torch.equal(out, ref)


def test_compute_loss_keys_order_total_and_gradients_on_host_tensors():
    body = syn.icosphere_body(1)
    faces = torch.as_tensor(body["faces"].astype("int64"))
    verts = torch.as_tensor(body["canonical_vertex"]).float().requires_grad_()
    edges, _ = mesh_edges(faces, verts.shape[0])
    # faces sharing an edge (any two faces that have two vertices in common)
    fs = [set(f.tolist()) for f in faces]
    pairs = torch.tensor([(i, j) for i in range(len(fs)) for j in range(i + 1, len(fs)) if len(fs[i] & fs[j]) == 2])
    g = torch.Generator().manual_seed(1)
    H = W = 12
    rgb, mask, nm = (torch.rand(1, H, W, 3, generator=g).requires_grad_(), torch.rand(1, H, W, generator=g).requires_grad_(),
                     torch.rand(1, H, W, generator=g).requires_grad_())
    rgb_gt, mask_gt = torch.rand(1, H, W, 3, generator=g), (torch.rand(1, H, W, generator=g) > 0.5).float()
    colors = torch.rand(faces.shape[0], 3, generator=g).requires_grad_()
    mesh = SimpleMesh(verts, faces, edges)
    outputs = {"mesh": mesh, "mesh_canonical": mesh, "normal_mask": nm, "face_connectivity": pairs, "colors": colors}
    cfg = NS(rgb=NS(coeff=1.0), mask=NS(coeff=5.0), lpips=NS(coeff=0.0), laplacian=NS(coeff_canonical=2.0, coeff_observation=10.0),
             normal=NS(coeff_mask=1.0, kernel_size=5, coeff_consist=0.1), color_consist=NS(coeff=0.05))
    total, losses = compute_loss(rgb, mask, outputs, rgb_gt, mask_gt, cfg)
    # the reference's dict order (train.py:98-163; `laplacian_canoincal` is its spelling)
    assert list(losses) == ["rgb", "mask", "laplacian_canoincal", "laplacian_observation", "normal_mask", "normal_consist", "color_consist"]
    assert torch.allclose(total, sum(v["scaled"] for v in losses.values()))
    assert torch.allclose(losses["rgb"]["unscaled"], (rgb - rgb_gt).abs().mean())
    assert torch.allclose(losses["mask"]["scaled"], 5.0 * (mask - mask_gt).abs().mean())
    dil = F.max_pool2d(mask_gt.unsqueeze(1), kernel_size=5, stride=1, padding=2).squeeze(1)
    assert torch.allclose(losses["normal_mask"]["unscaled"], (nm - dil).abs().mean())
    assert torch.allclose(losses["color_consist"]["unscaled"], (colors[pairs[:, 0]] - colors[pairs[:, 1]]).abs().mean())
    # a closed, nearly uniform sphere: small but non-zero Laplacian and normal-consistency terms
    assert 0.0 < float(losses["laplacian_observation"]["unscaled"].detach()) < 0.2 and 0.0 < float(losses["normal_consist"]["unscaled"].detach()) < 0.5
    total.backward()
    for t in (rgb, mask, nm, colors, verts):
        assert t.grad is not None and torch.isfinite(t.grad).all() and float(t.grad.abs().sum()) > 0


def test_compute_loss_accepts_a_callers_lpips_object_with_the_plain_signature():
    """Round-5 advisor finding: compute_loss passed `reduce=` to ANY object with a `.loss` attribute; only LPIPSMatrixCore takes it.  A caller's own
    object with loss(pred, gt) -- and a plain callable (the reference's LPIPS module, train.py:113-117) -- both work; the term lands where the reference puts it."""
    g = torch.Generator().manual_seed(2)
    H = W = 8
    rgb, mask = torch.rand(1, H, W, 3, generator=g).requires_grad_(), torch.rand(1, H, W, generator=g)
    rgb_gt, mask_gt = torch.rand(1, H, W, 3, generator=g), torch.rand(1, H, W, generator=g)
    cfg = NS(rgb=NS(coeff=1.0), mask=NS(coeff=5.0), lpips=NS(coeff=2.0))

    class Plain:
        def loss(self, pred, gt):
            return ((pred - gt) ** 2).mean()
    total, losses = compute_loss(rgb, mask, {}, rgb_gt, mask_gt, cfg, lpips_func=Plain())
    assert torch.allclose(losses["lpips"]["unscaled"], ((rgb - rgb_gt) ** 2).mean()) and torch.allclose(losses["lpips"]["scaled"], 2.0 * ((rgb - rgb_gt) ** 2).mean())
    fn = lambda a, b: ((a - b) ** 2).mean((1, 2, 3))         # called on 2 x - 1 in NCHW, like the reference's module
    total2, losses2 = compute_loss(rgb, mask, {}, rgb_gt, mask_gt, cfg, lpips_func=fn)
    assert torch.allclose(losses2["lpips"]["unscaled"], 4.0 * ((rgb - rgb_gt) ** 2).mean())
    total.backward()
    assert torch.isfinite(rgb.grad).all()
