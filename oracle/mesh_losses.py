"""TEST INFRASTRUCTURE ONLY: torch restatements of the three mesh regularisers of the reference's training loss
(train.py:123-160).  Only tests/, smoke() and bench.py's cpu_baseline leg may import this.

* `laplacian_smoothing`  -- the reference's OWN copy of the uniform Laplacian loss, utils/network_util.py:669-792:
  `loss = L.mm(verts)` (:782), `loss = loss.norm(dim=1) ** 2` (:789), un-weighted `loss.mean()` (:791-792), with
  L = `Meshes.laplacian_packed()` = D^-1 A - I [UPSTREAM PyTorch3D 0.7.0, SURVEY.md App. B], no gradient through L
  (:763-766).  PINNED by tests/golden/mesh_losses.npz, recorded by importing the reference's function with a
  minimal `Meshes` stand-in (scripts/make_goldens.py); the stand-in's L is the restated upstream piece.
* `normal_consistency`   -- PyTorch3D 0.7.0 `mesh_normal_consistency(meshes)` as called at train.py:149 [UPSTREAM, not in
  /root/reference -> unpinned]: for EVERY pair of faces sharing an edge (all pairs when more than two faces meet),
  n0 = (v1 - v0) x (a - v0), n1 = (v1 - v0) x (b - v0) with (v0, v1) the shared edge and a, b the remaining
  vertices, loss = 1 - cosine_similarity(n0, -n1) (eps 1e-8 on the product of the norms), averaged over the pairs.
* `color_consistency`    -- utils/network_util.py:795-799, over the reference's `face_connectivity`
  (models/model.py:115-125: edges `range(max_edge_id)`, i.e. without the last edge id).  PINNED by
  tests/golden/shadow_color.npz.
"""
import numpy as np
import torch


def edges_of(faces: torch.Tensor, n_verts: int):
    """PyTorch3D edge conventions [UPSTREAM]: unique undirected edges (v0 < v1) ordered by v0 * V + v1, and per face the
    ids of its edges (v1v2, v2v0, v0v1)."""
    f = faces.detach().cpu().numpy().astype(np.int64)
    e = np.concatenate([f[:, [1, 2]], f[:, [2, 0]], f[:, [0, 1]]], 0)
    e.sort(1)
    uniq, inv = np.unique(e[:, 0] * n_verts + e[:, 1], return_inverse=True)
    F = f.shape[0]
    return torch.from_numpy(np.stack([uniq // n_verts, uniq % n_verts], 1)), torch.from_numpy(np.stack([inv[:F], inv[F:2 * F], inv[2 * F:]], 1))


def uniform_laplacian(edges: torch.Tensor, n_verts: int, dtype=torch.float64) -> torch.Tensor:
    """Dense D^-1 A - I (tests use small meshes)."""
    A = torch.zeros(n_verts, n_verts, dtype=dtype)
    A[edges[:, 0], edges[:, 1]] = 1
    A[edges[:, 1], edges[:, 0]] = 1
    deg = A.sum(1)
    L = A / deg.clamp_min(1)[:, None]
    L[deg == 0] = 0
    return L - torch.eye(n_verts, dtype=dtype)


def laplacian_smoothing(verts: torch.Tensor, edges: torch.Tensor) -> torch.Tensor:
    """network_util.py:782,789,792.  Small meshes multiply by the dense L; from 4 096 vertices on (a dense float64 L of the
    27 554-vertex body is 6 GB) the same rows are applied edge by edge: (L v)_i = (sum of the neighbours of i) / deg_i - v_i."""
    N = verts.shape[0]
    if N <= 4096:
        with torch.no_grad():
            L = uniform_laplacian(edges, N, verts.dtype)
        return ((L @ verts).norm(dim=1) ** 2).mean()
    e0, e1 = edges[:, 0], edges[:, 1]
    one = torch.ones(e0.shape[0], dtype=verts.dtype)
    deg = torch.zeros(N, dtype=verts.dtype).index_add_(0, e0, one).index_add_(0, e1, one)
    nb = torch.zeros_like(verts).index_add(0, e0, verts[e1]).index_add(0, e1, verts[e0])
    Lv = torch.where(deg[:, None] > 0, nb / deg.clamp_min(1)[:, None], torch.zeros_like(nb)) - verts
    return (Lv.norm(dim=1) ** 2).mean()


def edge_face_pairs(faces: torch.Tensor, n_verts: int):
    """All pairs of faces sharing an edge: (pairs (P,2) face ids, edge (P,2) vertex ids)."""
    edges, f2e = edges_of(faces, n_verts)
    flat = f2e.reshape(-1).numpy()
    order = np.argsort(flat, kind="stable")
    eid, fid = flat[order], order // 3
    starts = np.flatnonzero(np.r_[True, eid[1:] != eid[:-1]])
    ends = np.r_[starts[1:], len(eid)]
    n = ends - starts
    two = n == 2                                     # a manifold edge: one pair (all of them on a closed mesh), taken without a Python loop
    pf = [np.stack([fid[starts[two]], fid[starts[two] + 1]], 1)]
    pe = [eid[starts[two]]]
    for s, e in zip(starts[n > 2], ends[n > 2]):     # more than two faces on an edge: all pairs
        for i in range(s, e):
            for j in range(i + 1, e):
                pf.append(np.asarray([[fid[i], fid[j]]])); pe.append(np.asarray([eid[s]]))
    pf = np.concatenate(pf, 0).astype(np.int64).reshape(-1, 2)
    pe = np.concatenate(pe, 0).astype(np.int64)
    order2 = np.argsort(pe, kind="stable")           # pairs in edge order, as the loop over edges produced them
    return torch.from_numpy(pf[order2]), edges[torch.from_numpy(pe[order2])]


def normal_consistency(verts: torch.Tensor, faces: torch.Tensor) -> torch.Tensor:
    pairs, ev = edge_face_pairs(faces, verts.shape[0])
    if pairs.shape[0] == 0:
        return verts.sum() * 0
    f0, f1 = faces[pairs[:, 0]], faces[pairs[:, 1]]
    v0i, v1i = ev[:, 0], ev[:, 1]
    other = lambda f: (f.sum(1) - v0i - v1i)            # the vertex of the face that is not on the shared edge
    v0, v1, a, b = verts[v0i], verts[v1i], verts[other(f0)], verts[other(f1)]
    n0 = torch.cross(v1 - v0, a - v0, dim=1)
    n1 = torch.cross(v1 - v0, b - v0, dim=1)
    w = (n0 * n0).sum(1) * (n1 * n1).sum(1)
    cos = (n0 * -n1).sum(1) / torch.sqrt(w.clamp_min(1e-16))     # torch 1.13 cosine_similarity, eps = 1e-8
    return (1 - cos).mean()


def face_connectivity(faces: torch.Tensor, n_verts: int) -> torch.Tensor:
    """models/model.py:115-125: for edge id i in range(max_edge_id) with exactly two faces, the (sorted) face pair."""
    _, f2e = edges_of(faces, n_verts)
    f2e_np = f2e.numpy()
    if f2e_np.shape[0] <= 4096:                      # the reference's loop, literally
        out = []
        for i in range(int(f2e_np.max())):
            fs = np.nonzero((f2e_np == i).any(1))[0]
            if len(fs) == 2:
                out.append(fs)
        return torch.from_numpy(np.asarray(out, np.int64).reshape(-1, 2))
    # the same list without the O(edges x faces) scan: (edge, face) incidences grouped by edge id (a face counts once per edge)
    inc = np.unique(np.stack([f2e_np.reshape(-1), np.repeat(np.arange(f2e_np.shape[0]), 3)], 1), axis=0)     # sorted by (edge, face)
    eid, fid = inc[:, 0], inc[:, 1]
    starts = np.flatnonzero(np.r_[True, eid[1:] != eid[:-1]])
    n = np.r_[starts[1:], len(eid)] - starts
    sel = starts[(n == 2) & (eid[starts] < int(f2e_np.max()))]
    return torch.from_numpy(np.stack([fid[sel], fid[sel + 1]], 1).astype(np.int64))


def color_consistency(colors_F3: torch.Tensor, pairs: torch.Tensor) -> torch.Tensor:
    """network_util.py:795-799."""
    return (colors_F3[pairs[:, 0]] - colors_F3[pairs[:, 1]]).abs().mean()
