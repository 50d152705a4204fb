"""CPU: the tile-culled evaluation of the mesh oracle (oracle/mesh.py::render_tiled, what lets the float64 oracle afford a 512 x 512
training loop) IS the dense O(pixels x faces) restatement: same normal map, silhouette, pix_to_face and gradients, bit for bit."""
import numpy as np
import torch

from gomavatar_amd import synthetic as syn
from oracle import mesh as om


def _scene(img, level=2, seed=0):
    body = syn.icosphere_body(level)
    fr = {k: torch.from_numpy(v).double() for k, v in syn.make_frame(1, img).items() if v.dtype.kind == "f"}
    g = torch.Generator().manual_seed(seed)
    v = torch.from_numpy(body["canonical_vertex"]).double()
    v = v + 0.01 * torch.randn(v.shape, generator=g, dtype=torch.float64)
    faces = torch.from_numpy(body["faces"]).long()
    return v, faces, fr


def test_tiled_mesh_oracle_equals_dense_bitwise():
    img = 48
    v, faces, fr = _scene(img)
    outs = []
    for tiled in (False, True):
        vv = v.clone().requires_grad_()
        ndc = om.ndc_T_world(vv.T[None], fr["K"], fr["E"], img, img)[0]
        vn = om.vertex_normals(vv, faces)
        if tiled:
            n, a, t = om.render_tiled(ndc, faces, vn, img, img, sigma_cfg=1e-5, tile=16)
        else:
            n, a, t = om.render(ndc, faces, vn, img, img, sigma_cfg=1e-5)
        w = torch.linspace(0.5, 1.5, img * img, dtype=torch.float64).reshape(img, img)
        ((n * w[..., None]).sum() + (a * w).sum()).backward()
        outs.append((n.detach(), a.detach(), t, vv.grad.clone()))
    (n0, a0, t0, g0), (n1, a1, t1, g1) = outs
    assert int((t0 >= 0).sum()) > 100 and float(a0.sum()) > 100          # the body is in the picture
    assert torch.equal(n0, n1) and torch.equal(a0, a1) and torch.equal(t0, t1)
    # gradients: same terms; summed over tiles in a different order than the dense index_add
    assert float((g0 - g1).abs().max()) <= 1e-12 * float(g0.abs().max())


def test_tiled_mesh_oracle_eval_mode_and_ragged_tiles():
    img = 40          # not a multiple of the tile
    v, faces, fr = _scene(img, seed=3)
    ndc = om.ndc_T_world(v.T[None], fr["K"], fr["E"], img, img)[0]
    vn = om.vertex_normals(v, faces)
    n0, a0, t0 = om.render(ndc, faces, vn, img, img, training=False)
    n1, a1, t1 = om.render_tiled(ndc, faces, vn, img, img, training=False, tile=16)
    assert a0 is None and a1 is None and torch.equal(n0, n1) and torch.equal(t0, t1)
