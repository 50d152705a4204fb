"""TEST INFRASTRUCTURE ONLY: brute-force torch restatement of the reference's mesh normal / soft-silhouette renderer
(models/modules/renderer/mesh.py:65-128, utils/pc_util.py:10-46) and of the PyTorch3D 0.7 pieces it calls
(MeshRasterizer naive path, hard_rgb_blend, SoftSilhouetteShader / sigmoid_alpha_blend, Meshes.verts_normals_padded).
PyTorch3D is absent from this image and from /root/reference, so these semantics are restated from the published
implementation -> parity unpinned.  Everything is differentiable torch, O(pixels x faces)."""
import math

import torch

K_EPS = 1e-8          # PyTorch3D kEpsilon
BLEND_SIGMA = 1e-4    # BlendParams default sigma (mesh.py never overrides it)


def ndc_T_world(xyz_world, K, E, H, W):
    """pc_util.py:30-46.  xyz_world (B,3,N), K (B,3,3), E (B,4,4) -> (B,N,3): x, y negated NDC (shorter side in
    [-1,1]), z = camera depth."""
    ones = torch.ones_like(xyz_world[:, :1])
    cam_ = torch.bmm(E, torch.cat([xyz_world, ones], 1))
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


def vertex_normals(verts, faces):
    """Meshes.verts_normals_packed: area-weighted face normals accumulated on the three corners, then
    normalize(eps=1e-6).  verts (N,3), faces (F,3)."""
    v0, v1, v2 = verts[faces[:, 0]], verts[faces[:, 1]], verts[faces[:, 2]]
    n = torch.zeros_like(verts)
    n = n.index_add(0, faces[:, 1], torch.cross(v2 - v1, v0 - v1, dim=1))
    n = n.index_add(0, faces[:, 2], torch.cross(v0 - v2, v1 - v2, dim=1))
    n = n.index_add(0, faces[:, 0], torch.cross(v1 - v0, v2 - v0, dim=1))
    return torch.nn.functional.normalize(n, eps=1e-6, dim=1)


def pixel_centers(H, W, dtype):
    """NDC of pixel centres, +X left, +Y up (PyTorch3D pix_to_non_square_ndc + the (S-1-i) flip)."""
    def axis(S1, S2):
        rng = 2.0 * S1 / S2 if S1 > S2 else 2.0
        off = rng / 2.0
        i = torch.arange(S1 - 1, -1, -1, dtype=dtype)
        return -off + (rng * i + off) / S1
    return axis(W, H), axis(H, W)      # xf[W], yf[H]


def _edge(px, py, ax, ay, bx, by):
    return (px - ax) * (by - ay) - (py - ay) * (bx - ax)


def _seg_dist2(px, py, ax, ay, bx, by):
    dx, dy = bx - ax, by - ay
    l2 = dx * dx + dy * dy
    t = ((px - ax) * dx + (py - ay) * dy) / torch.where(l2 > K_EPS, l2, torch.ones_like(l2))
    t = t.clamp(0.0, 1.0)
    qx, qy = ax + t * dx, ay + t * dy
    d = (px - qx) ** 2 + (py - qy) ** 2
    return torch.where(l2 > K_EPS, d, (px - bx) ** 2 + (py - by) ** 2)


def rasterize(verts_ndc, faces, H, W, blur_radius, rows=None, cols=None):
    """Per pixel and face: qualifies (bool), signed squared distance, interpolated depth.  Returns dense (HW, F); `rows` = (r0, r1)
    restricts the pixels to image rows r0 .. r1-1, `cols` = (c0, c1) to columns c0 .. c1-1 (pixels are independent: large meshes are
    rendered in row bands or tiles)."""
    dt = verts_ndc.dtype
    xf, yf = pixel_centers(H, W, dt)
    r0, r1 = rows if rows is not None else (0, H)
    c0, c1 = cols if cols is not None else (0, W)
    px = xf[None, c0:c1].expand(r1 - r0, c1 - c0).reshape(-1, 1)
    py = yf[r0:r1, None].expand(r1 - r0, c1 - c0).reshape(-1, 1)
    v = verts_ndc[faces]                                  # (F,3,3)
    x0, y0, z0 = v[:, 0, 0][None], v[:, 0, 1][None], v[:, 0, 2][None]
    x1, y1, z1 = v[:, 1, 0][None], v[:, 1, 1][None], v[:, 1, 2][None]
    x2, y2, z2 = v[:, 2, 0][None], v[:, 2, 1][None], v[:, 2, 2][None]
    area = _edge(x2, y2, x0, y0, x1, y1)
    ok_face = area.abs() > K_EPS
    blur = math.sqrt(blur_radius)
    xmin, xmax = torch.minimum(torch.minimum(x0, x1), x2), torch.maximum(torch.maximum(x0, x1), x2)
    ymin, ymax = torch.minimum(torch.minimum(y0, y1), y2), torch.maximum(torch.maximum(y0, y1), y2)
    in_box = (px <= xmax + blur) & (px >= xmin - blur) & (py <= ymax + blur) & (py >= ymin - blur)
    den = area + K_EPS
    w0, w1, w2 = _edge(px, py, x1, y1, x2, y2) / den, _edge(px, py, x2, y2, x0, y0) / den, _edge(px, py, x0, y0, x1, y1) / den
    inside = (w0 > 0) & (w1 > 0) & (w2 > 0)
    if blur_radius > 0:                                    # clip_barycentric_coords defaults to blur_radius > 0
        c0, c1, c2 = w0.clamp(0, 1), w1.clamp(0, 1), w2.clamp(0, 1)
        s = (c0 + c1 + c2).clamp_min(1e-5)
        b0, b1, b2 = c0 / s, c1 / s, c2 / s
    else:
        b0, b1, b2 = w0, w1, w2
    pz = b0 * z0 + b1 * z1 + b2 * z2
    d = torch.minimum(torch.minimum(_seg_dist2(px, py, x0, y0, x1, y1), _seg_dist2(px, py, x1, y1, x2, y2)), _seg_dist2(px, py, x2, y2, x0, y0))
    qual = ok_face & in_box & (pz >= 0) & (inside | (d < blur_radius))
    return qual, torch.where(inside, -d, d), pz


def render(verts_ndc, faces, vnormals, H, W, sigma_cfg=1e-5, faces_per_pixel=50, training=True, rows=None, cols=None):
    """mesh.py:114-128.  Returns (normal (H,W,3), mask (H,W) or None, pix_to_face (H,W)) -- of the row band `rows` (and the columns
    `cols`) when given."""
    Hfull, Wfull = H, W
    qual, _, pz = rasterize(verts_ndc, faces, H, W, 0.0, rows, cols)
    if rows is not None:
        H = rows[1] - rows[0]
    if cols is not None:
        W = cols[1] - cols[0]
    zsel = torch.where(qual, pz, torch.full_like(pz, float("inf")))
    zmin, top = zsel.min(1)
    hit = torch.isfinite(zmin)
    fn = vnormals[faces].sum(1)                            # interpolate_face_attributes with bary = ones: n0 + n1 + n2
    normal = torch.where(hit[:, None], fn[top], torch.zeros_like(fn[top]))       # hard_rgb_blend, then x alpha
    pix_to_face = torch.where(hit, top, torch.full_like(top, -1))
    if not training:
        return normal.reshape(H, W, 3), None, pix_to_face.reshape(H, W)
    blur_radius = math.log(1.0 / 1e-4 - 1.0) * sigma_cfg
    qual, sd, pz = rasterize(verts_ndc, faces, Hfull, Wfull, blur_radius, rows, cols)
    zsel = torch.where(qual, pz, torch.full_like(pz, float("inf")))
    k = min(faces_per_pixel, zsel.shape[1])
    zk, idx = torch.topk(zsel, k, dim=1, largest=False)
    valid = torch.isfinite(zk)
    prob = torch.sigmoid(-torch.gather(sd, 1, idx) / BLEND_SIGMA) * valid
    alpha = 1.0 - torch.prod(1.0 - prob, 1)
    return normal.reshape(H, W, 3), alpha.reshape(H, W), pix_to_face.reshape(H, W)


def render_tiled(verts_ndc, faces, vnormals, H, W, sigma_cfg=1e-5, faces_per_pixel=50, training=True, tile=16, use_checkpoint=True):
    """`render` tile by tile, each tile over only the faces whose blur-expanded bounding box reaches it.  EXACT, not an approximation:
    `rasterize` requires `in_box` of every qualifying (pixel, face) pair, a face outside every pixel's box of a tile qualifies nowhere
    in it, and a non-qualifying face contributes nothing to the top-1 / product and receives no gradient -- so the result and its
    gradients are those of the dense O(pixels x faces) evaluation (tests/test_oracle_mesh_tiled.py checks bitwise equality).  It
    is what makes a 512 x 512 / 55 104-face training loop affordable for the float64 oracle (scripts/make_train_loop_goldens.py).
    Tiles are re-computed in the backward (torch checkpoint) so that only one tile's dense tensors are alive at a time."""
    from torch.utils.checkpoint import checkpoint
    dt = verts_ndc.dtype
    xf, yf = pixel_centers(H, W, dt)
    blur = math.sqrt(math.log(1.0 / 1e-4 - 1.0) * sigma_cfg) if training else 0.0
    with torch.no_grad():
        v = verts_ndc[faces]
        xmin, xmax = v[:, :, 0].min(1).values - blur, v[:, :, 0].max(1).values + blur
        ymin, ymax = v[:, :, 1].min(1).values - blur, v[:, :, 1].max(1).values + blur
    normal = torch.zeros(H, W, 3, dtype=dt)
    alpha = torch.zeros(H, W, dtype=dt) if training else None
    p2f = torch.full((H, W), -1, dtype=torch.long)
    n_rows, a_rows = [], []
    for r0 in range(0, H, tile):
        r1 = min(H, r0 + tile)
        ylo, yhi = min(float(yf[r0]), float(yf[r1 - 1])), max(float(yf[r0]), float(yf[r1 - 1]))
        row_sel = (ymin <= yhi) & (ymax >= ylo)
        n_cols, a_cols = [], []
        for c0 in range(0, W, tile):
            c1 = min(W, c0 + tile)
            xlo, xhi = min(float(xf[c0]), float(xf[c1 - 1])), max(float(xf[c0]), float(xf[c1 - 1]))
            idx = torch.nonzero(row_sel & (xmin <= xhi) & (xmax >= xlo)).squeeze(1)
            if idx.numel() == 0:
                n_cols.append(torch.zeros(r1 - r0, c1 - c0, 3, dtype=dt))
                a_cols.append(torch.zeros(r1 - r0, c1 - c0, dtype=dt))
                continue
            sub = faces[idx]

            def fn(ndc_, vn_, sub=sub, r0=r0, r1=r1, c0=c0, c1=c1):
                n, a, t = render(ndc_, sub, vn_, H, W, sigma_cfg, faces_per_pixel, training, rows=(r0, r1), cols=(c0, c1))
                return n, (a if a is not None else torch.zeros(r1 - r0, c1 - c0, dtype=dt)), t
            if use_checkpoint and torch.is_grad_enabled() and (verts_ndc.requires_grad or vnormals.requires_grad):
                n, a, t = checkpoint(fn, verts_ndc, vnormals, use_reentrant=False)
            else:
                n, a, t = fn(verts_ndc, vnormals)
            n_cols.append(n); a_cols.append(a)
            p2f[r0:r1, c0:c1] = torch.where(t >= 0, idx[t.clamp_min(0)], t)
        n_rows.append(torch.cat(n_cols, 1)); a_rows.append(torch.cat(a_cols, 1))
    normal = torch.cat(n_rows, 0)
    alpha = torch.cat(a_rows, 0) if training else None
    return normal, alpha, p2f
