"""Mesh normal map / soft silhouette rasterizer (csrc/mesh_raster.hip) against the brute-force torch restatement of
the reference's PyTorch3D renderer (oracle/mesh.py)."""
import numpy as np
import pytest
import torch

from gomavatar_amd import synthetic as syn
from oracle import mesh as om

pytestmark = pytest.mark.gpu


def _scene(img, body=None, frame=1):
    body = body or syn.icosphere_body(3)
    fr = syn.make_frame(frame, img)
    v = torch.from_numpy(body["canonical_vertex"]).T.contiguous()[None]
    faces = torch.from_numpy(body["faces"]).long()
    return v, faces, torch.from_numpy(fr["K"]), torch.from_numpy(fr["E"])


@pytest.mark.parametrize("img,zoom", [(64, 1.0), (96, 3.0)])
def test_forward_and_backward_match_oracle(img, zoom):
    from gomavatar_amd.mesh_renderer import MeshNormalRenderer, ndc_T_world, vertex_normals
    v, faces, K, E = _scene(img)
    K = K.clone(); K[:, 0, 0] *= zoom; K[:, 1, 1] *= zoom      # zoom: faces of several pixels (real footprints, blur band < face)
    # --- oracle (fp64) ---
    vo = v.double().requires_grad_()
    ndc_o = om.ndc_T_world(vo, K.double(), E.double(), img, img)[0]
    vn_o = om.vertex_normals(vo[0].T, faces)
    n_o, a_o, p2f_o = om.render(ndc_o, faces, vn_o, img, img, sigma_cfg=1e-5)
    g = torch.Generator().manual_seed(0)
    wn, wa = torch.randn(img, img, 3, generator=g).double(), torch.randn(img, img, generator=g).double()
    ((n_o * wn).sum() + (a_o * wa).sum()).backward()
    # --- HIP ---
    r = MeshNormalRenderer(img_size=(img, img), sigma=1e-5).cuda().train()
    vh = v.cuda().requires_grad_()
    fc = faces.cuda()
    vn_h = vertex_normals(vh[0].T, r.topology(fc, vh.shape[2]))
    assert torch.allclose(vn_h.detach().cpu().double(), vn_o.detach(), atol=2e-6)
    normal, mask = r(vh, vn_h[None], K.cuda(), E.cuda(), fc)
    assert normal.shape == (1, img, img, 3) and mask.shape == (1, img, img, 1)
    ((normal[0] * wn.float().cuda()).sum() + (mask[0, ..., 0] * wa.float().cuda()).sum()).backward()
    p2f = torch.empty(img, img, dtype=torch.int32, device="cuda")
    from gomavatar_amd import _lib
    _lib.check(_lib.load().gom_mesh_pix_to_face(r.state.handle, _lib.ptr(p2f), _lib.stream_ptr()))
    same = (p2f.cpu().long() == p2f_o)
    assert same.float().mean() > 0.999, same.float().mean()                # pixels exactly on an edge may pick the neighbour
    assert (p2f_o >= 0).float().mean() > 0.03
    dn = (normal[0].detach().cpu().double() - n_o.detach()).abs().amax(-1)
    assert float(dn[same].max()) < 1e-5
    da = (mask[0, ..., 0].detach().cpu().double() - a_o.detach()).abs()
    # The reference's silhouette is discontinuous at the rim of a face's blur band: a face enters the product at squared distance
    # blur_radius = 9.2 sigma with probability sigmoid(-9.2 sigma / 1e-4) = 0.285 (sigma = 1e-5, BlendParams sigma 1e-4), so a
    # pixel whose distance to some edge equals blur_radius to the last bit carries (1 - alpha) x 0.715 or not.  The soak met that
    # at one pixel in ~3 % of random scenes: a flip count like the splat rasterizer's, bounded by the size of the jump.
    assert int((da > 2e-5).sum()) <= max(1, int(2e-4 * da.numel())), (int((da > 2e-5).sum()), float(da.max()))
    # (two faces at their rim in ONE pixel: 1 - 0.715^2 = 0.489 -- soak of round 3: once in ~650 random scenes, (img, zoom) = (64, 1.6486379177940387))
    assert float(da.max()) <= 0.49, float(da.max())
    # ndc_T_world mirror
    assert torch.allclose(ndc_T_world(v, K, E, img, img), om.ndc_T_world(v, K, E, img, img), atol=1e-6)
    gr, gg = vo.grad[0].numpy(), vh.grad[0].cpu().numpy().astype(np.float64)
    scale = np.abs(gr).max()
    err = np.abs(gg - gr)
    # (the 0.99 quantile is the vertices of the faces whose rim pixels flipped: 2.04e-3 of the scale once in ~650 random scenes of the round-3 soak,
    #  (img, zoom) = (111, 2.473560405789459); the median is what holds the arithmetic)
    assert np.quantile(err, 0.99) <= 3e-3 * scale and np.median(err) <= 1e-5 * scale, (np.quantile(err, 0.99), np.median(err), scale)


def test_eval_mode_and_reproducibility():
    from gomavatar_amd.mesh_renderer import MeshNormalRenderer, vertex_normals
    img = 128
    v, faces, K, E = _scene(img, body=syn.make_body(0), frame=2)
    r = MeshNormalRenderer(img_size=(img, img), sigma=1e-5).cuda()
    vc, fc = v.cuda(), faces.cuda()
    topo = r.topology(fc, vc.shape[2])
    vn = vertex_normals(vc[0].T, topo)
    r.eval()
    n_eval, m_eval = r(vc, vn[None], K.cuda(), E.cuda(), fc)
    assert m_eval is None
    r.train()
    outs = []
    for _ in range(2):
        x = vc.clone().requires_grad_()
        n, m = r(x, vertex_normals(x[0].T, topo)[None], K.cuda(), E.cuda(), fc)
        (n.square().sum() + m.sum()).backward()
        outs.append((n.clone(), m.clone(), x.grad.clone()))
    assert torch.equal(outs[0][0], n_eval) and all(torch.equal(a, b) for a, b in zip(outs[0], outs[1]))
    assert float(outs[0][1].max()) > 0.99 and float(outs[0][1].min()) == 0.0


def test_non_square_image_ndc_convention():
    """H != W: the shorter side spans [-1, 1] (pc_util.py:38-43 and PyTorch3D's pix_to_non_square_ndc)."""
    from gomavatar_amd.mesh_renderer import MeshNormalRenderer, vertex_normals
    H, W = 64, 96
    body = syn.icosphere_body(2)
    fr = syn.make_frame(0, 64)
    K = torch.from_numpy(fr["K"]).clone(); K[:, 0, 2] = W / 2; K[:, 0, 0] *= 2.0; K[:, 1, 1] *= 2.0
    E = torch.from_numpy(fr["E"])
    v = torch.from_numpy(body["canonical_vertex"]).T.contiguous()[None]
    faces = torch.from_numpy(body["faces"]).long()
    ndc_o = om.ndc_T_world(v.double(), K.double(), E.double(), H, W)[0]
    vn_o = om.vertex_normals(v[0].T.double(), faces)
    n_o, a_o, p2f_o = om.render(ndc_o, faces, vn_o, H, W, sigma_cfg=1e-5)
    r = MeshNormalRenderer(img_size=(W, H), sigma=1e-5).cuda().train()
    vc, fc = v.cuda(), faces.cuda()
    normal, mask = r(vc, vertex_normals(vc[0].T, r.topology(fc, vc.shape[2]))[None], K.cuda(), E.cuda(), fc)
    assert normal.shape == (1, H, W, 3) and mask.shape == (1, H, W, 1)
    assert (p2f_o >= 0).float().mean() > 0.02
    assert float((mask[0, ..., 0].cpu().double() - a_o).abs().max()) < 2e-5
    d = (normal[0].cpu().double() - n_o).abs().amax(-1)
    assert float((d > 1e-5).float().mean()) < 1e-3


def test_degenerate_and_offscreen_faces_are_harmless():
    """Zero-area faces are skipped (PyTorch3D kEpsilon rule), faces outside the image or behind the camera touch nothing,
    and gradients stay finite."""
    from gomavatar_amd.mesh_renderer import _MeshRaster, vertex_normals
    from gomavatar_amd.geometry import MeshTopology
    from gomavatar_amd.rasterizer import RasterState, _StateLease
    H = W = 48
    verts = torch.tensor([[0.3, 0.3, 5.0], [-0.3, 0.3, 5.0], [0.0, -0.3, 5.0],      # a visible triangle
                          [0.1, 0.1, 4.0], [0.1, 0.1, 4.0], [0.1, 0.1, 4.0],        # three coincident points: zero area
                          [5.0, 5.0, 5.0], [5.2, 5.0, 5.0], [5.0, 5.2, 5.0],        # far outside the image
                          [0.2, 0.0, -1.0], [-0.2, 0.0, -1.0], [0.0, 0.2, -1.0]],   # behind the camera
                         dtype=torch.float32)
    faces = torch.arange(12).reshape(4, 3)
    v = verts.cuda().requires_grad_()
    topo = MeshTopology(faces.cuda(), 12)
    vn = vertex_normals(v, topo)
    normal, alpha = _MeshRaster.apply(v, vn, topo, _StateLease(RasterState()), H, W, 9.21e-5, 1e-4, True)      # a pinned (un-pooled) state
    (normal.sum() + alpha.sum()).backward()
    assert torch.isfinite(normal).all() and torch.isfinite(alpha).all() and torch.isfinite(v.grad).all()
    n_o, a_o, p2f = om.render(verts.double(), faces, om.vertex_normals(verts.double(), faces), H, W, sigma_cfg=1e-5)
    assert set(p2f.unique().tolist()) <= {-1, 0}                       # only the first face is ever on top
    assert float((alpha.detach().cpu().double() - a_o).abs().max()) < 2e-5
    assert float(v.grad[3:].abs().max()) == 0.0                        # nothing reaches the other three faces' vertices


@pytest.mark.parametrize("H,W", [(96, 96), (64, 112), (120, 72)])
def test_ndc_from_world_kernel_matches_the_torch_formula(H, W):
    """gom_ndc_from_world_forward / _backward (utils/pc_util.py:30-46) against the same formula in torch on the CPU, square and both
    non-square orientations, values and vertex gradient."""
    from gomavatar_amd.mesh_renderer import ndc_T_world
    v, faces, K, E = _scene(96)
    g = torch.Generator().manual_seed(H + W)
    wgt = torch.randn(1, v.shape[2], 3, generator=g)
    vc = v.clone().requires_grad_()
    ref = ndc_T_world(vc, K, E, H, W)                     # host tensors: the torch formula
    (ref * wgt).sum().backward()
    vg = v.cuda().requires_grad_()
    out = ndc_T_world(vg, K.cuda(), E.cuda(), H, W)       # device tensors: the kernel
    (out * wgt.cuda()).sum().backward()
    assert out.shape == ref.shape
    # (a few ulp of the largest coordinate: with a very narrow image the NDC range of the long side reaches +-10 and more)
    assert torch.allclose(out.detach().cpu(), ref.detach(), rtol=1e-5, atol=max(1e-6, 4e-7 * float(ref.abs().max())))
    assert float((vg.grad.cpu() - vc.grad).abs().max()) <= 1e-5 * float(vc.grad.abs().max())


def test_vertex_normals_rotation_option_equals_the_matrix_product():
    """vertex_normals(..., rotation=R) = (R @ vertex_normals(...).T).T (models/model.py:271-272), values and vertex gradient."""
    from gomavatar_amd.mesh_renderer import MeshNormalRenderer, vertex_normals
    v, faces, K, E = _scene(64)
    r = MeshNormalRenderer(img_size=(64, 64)).cuda()
    fc = faces.cuda()
    topo = r.topology(fc, v.shape[2])
    R = E[0, :3, :3].cuda()
    w = torch.randn(v.shape[2], 3, generator=torch.Generator().manual_seed(1)).cuda()
    a = v[0].T.contiguous().cuda().requires_grad_()
    n_a = (R @ vertex_normals(a, topo).T).T
    (n_a * w).sum().backward()
    b = v[0].T.contiguous().cuda().requires_grad_()
    n_b = vertex_normals(b, topo, rotation=R)
    (n_b * w).sum().backward()
    assert torch.allclose(n_a.detach(), n_b.detach(), rtol=1e-5, atol=1e-6)
    assert float((a.grad - b.grad).abs().max()) <= 1e-5 * float(a.grad.abs().max())


def test_two_forwards_in_flight_keep_their_own_scratch():
    """Two Model-style forwards before either backward (gradient accumulation over frames) and a no_grad preview render in
    between: every backward must see ITS forward's scratch (the first version kept one state per renderer)."""
    from gomavatar_amd import synthetic as syn
    from gomavatar_amd.mesh_renderer import MeshNormalRenderer, vertex_normals
    body = syn.icosphere_body(3)
    img = 96
    faces = torch.from_numpy(body["faces"]).long().cuda()
    r = MeshNormalRenderer(img_size=(img, img), sigma=1e-5).cuda().train()
    g = torch.Generator().manual_seed(0)
    wn, wa = torch.randn(img, img, 3, generator=g).cuda(), torch.randn(img, img, generator=g).cuda()

    def frame(i):
        fr = syn.make_frame(i, img)
        K, E = torch.from_numpy(fr["K"]).cuda(), torch.from_numpy(fr["E"]).cuda()
        K = K.clone(); K[:, 0, 0] *= 4; K[:, 1, 1] *= 4
        v = (torch.from_numpy(body["canonical_vertex"]).T[None] * 0.5 + torch.tensor([0.0, 1.2, 0.0]).view(1, 3, 1)).cuda().requires_grad_()
        vn = vertex_normals(v[0].T, r.topology(faces, v.shape[2]))
        return v, vn, K, E

    def loss(n, a):
        return (n[0] * wn).sum() + (a[0, ..., 0] * wa).sum()

    # reference gradients: one forward/backward at a time
    refs = []
    for i in (0, 3):
        v, vn, K, E = frame(i)
        loss(*r(v, vn[None], K, E, faces)).backward()
        refs.append(v.grad.clone())
    # interleaved: forward 0, forward 3, a no_grad preview, then the two backwards in the "wrong" order
    v0, vn0, K0, E0 = frame(0)
    out0 = r(v0, vn0[None], K0, E0, faces)
    v3, vn3, K3, E3 = frame(3)
    out3 = r(v3, vn3[None], K3, E3, faces)
    with torch.no_grad():
        vp, vnp, Kp, Ep = frame(5)
        r(vp.detach(), vnp[None], Kp, Ep, faces)
    loss(*out0).backward()
    loss(*out3).backward()
    assert torch.equal(v0.grad, refs[0]) and torch.equal(v3.grad, refs[1])


def test_silhouette_backward_skips_zero_gradient_pixels_exactly():
    """k_mesh_backward_entries evaluates a (face, pixel) pair only where dL/d(alpha) != 0 (the band around the outline in training: under the body
    alpha rounds to 1.0f and |alpha - target|'s gradient is sign(0) = 0, train.py:142) and skips whole 8 x 8 tiles without such a pixel.
    A zero cotangent adds +-0; standing an infinitesimal in for every zero forces the full evaluation and must give the same gradient:
    bit for bit wherever the gradient is not itself infinitesimal."""
    from gomavatar_amd.mesh_renderer import MeshNormalRenderer, vertex_normals
    img = 128
    v, faces, K, E = _scene(img, body=syn.make_body(0), frame=2)
    r = MeshNormalRenderer(img_size=(img, img), sigma=1e-5).cuda().train()
    vc, fc = v.cuda(), faces.cuda()
    topo = r.topology(fc, vc.shape[2])
    g = torch.Generator().manual_seed(3)
    w = torch.randn(img, img, generator=g)
    blocks = (torch.rand(img // 8, img // 8, generator=g) < 0.5).repeat_interleave(8, 0).repeat_interleave(8, 1)    # whole tiles without a gradient
    keep = blocks & (torch.rand(img, img, generator=g) < 0.4)                                                     # and most pixels of the others
    grads = []
    for fill in (0.0, 1e-30):
        x = vc.clone().requires_grad_()
        n, m = r(x, vertex_normals(x[0].T, topo)[None], K.cuda(), E.cuda(), fc)
        cot = torch.where(keep, w, torch.full_like(w, fill)).cuda()
        (m[0, ..., 0] * cot).sum().backward()
        grads.append(x.grad[0].cpu())
    a, b = grads
    assert float(a.abs().max()) > 1e-3 and torch.isfinite(a).all()
    assert float((a - b).abs().max()) <= 1e-20, float((a - b).abs().max())
    big = b.abs() > 1e-12
    assert big.float().mean() > 0.01 and torch.equal(a[big], b[big])
    # and the masked cotangent against the float64 oracle (the gate itself: a tile wrongly taken for gradient-free would lose its faces' terms)
    vo = v.double().requires_grad_()
    ndc_o = om.ndc_T_world(vo, K.double(), E.double(), img, img)[0]
    _, a_o, _ = om.render_tiled(ndc_o, faces, om.vertex_normals(vo[0].T, faces), img, img, sigma_cfg=1e-5, tile=16)
    (a_o * torch.where(keep, w, torch.zeros_like(w)).double()).sum().backward()
    gr = vo.grad[0].numpy()
    err = np.abs(a.numpy().astype(np.float64) - gr)
    scale = np.abs(gr).max()
    assert np.quantile(err, 0.99) <= 3e-3 * scale and np.median(err) <= 1e-5 * scale, (np.quantile(err, 0.99), np.median(err), scale)
