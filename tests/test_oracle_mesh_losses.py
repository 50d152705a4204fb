"""CPU: oracle/mesh_losses.py against the golden recorded from the reference's own `mesh_laplacian_smoothing`
(utils/network_util.py:669-792) and `mesh_color_consistency` (:795-799); the host branch of train_util against the oracle."""
import os

import numpy as np
import torch

from oracle import mesh_losses as oml


def test_laplacian_oracle_matches_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "mesh_losses.npz"))
    v = torch.from_numpy(g["verts"]).requires_grad_()
    edges, _ = oml.edges_of(torch.from_numpy(g["faces"]), v.shape[0])
    assert np.array_equal(edges.numpy(), g["edges"])
    val = oml.laplacian_smoothing(v, edges)
    val.backward()
    assert abs(float(val.detach()) - float(g["laplacian"])) <= 1e-14
    assert np.abs(v.grad.numpy() - g["laplacian_grad"]).max() <= 1e-15
    # it is the SQUARED norm (network_util.py:789): the un-squared mean is a different number
    L = oml.uniform_laplacian(edges, v.shape[0])
    assert abs(float((L @ v.detach()).norm(dim=1).mean()) - float(g["laplacian"])) > 1e-4


def test_color_consistency_oracle_matches_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "shadow_color.npz"))
    val = oml.color_consistency(torch.from_numpy(g["colors"]), torch.from_numpy(g["pairs"]))
    assert abs(float(val) - float(g["color_consistency"])) <= 1e-7


def test_host_branch_of_train_util_is_the_oracle():
    from gomavatar_amd import synthetic as syn, train_util as tu
    from gomavatar_amd.model import SimpleMesh, mesh_edges, edge_adjacent_face_pairs
    body = syn.icosphere_body(2)
    g = torch.Generator().manual_seed(1)
    v = torch.from_numpy(body["canonical_vertex"]).double() + 0.01 * torch.randn(body["canonical_vertex"].shape, generator=g, dtype=torch.float64)
    faces = torch.from_numpy(body["faces"]).long()
    N = v.shape[0]
    edges, f2e = mesh_edges(faces, N)
    e_o, _ = oml.edges_of(faces, N)
    assert torch.equal(edges, e_o)
    assert torch.equal(edge_adjacent_face_pairs(f2e, skip_last_edge=True), oml.face_connectivity(faces, N))
    pairs_o, _ = oml.edge_face_pairs(faces, N)
    assert torch.equal(edge_adjacent_face_pairs(f2e), torch.sort(pairs_o, 1)[0])
    mesh = SimpleMesh(v, faces, edges)
    assert abs(float(tu.mesh_laplacian_smoothing(mesh)) - float(oml.laplacian_smoothing(v, edges))) <= 1e-14
    assert abs(float(tu.mesh_normal_consistency(mesh)) - float(oml.normal_consistency(v, faces))) <= 1e-14


def test_normal_consistency_flat_and_folded():
    # two triangles sharing an edge: coplanar -> 0; folded by 90 degrees -> 1
    faces = torch.tensor([[0, 1, 2], [1, 0, 3]])
    flat = torch.tensor([[0.0, 0, 0], [1, 0, 0], [0.5, 1, 0], [0.5, -1, 0]], dtype=torch.float64)
    assert abs(float(oml.normal_consistency(flat, faces))) <= 1e-15
    fold = flat.clone(); fold[3] = torch.tensor([0.5, 0.0, -1.0], dtype=torch.float64)
    assert abs(float(oml.normal_consistency(fold, faces)) - 1.0) <= 1e-15
