import numpy as np

from gomavatar_amd import synthetic as syn


def test_smpl_topology_counts():
    b = syn.make_body(0)
    assert b["canonical_vertex"].shape == (6890, 3) and b["faces"].shape == (13776, 3)
    e = np.sort(np.concatenate([b["faces"][:, [0, 1]], b["faces"][:, [1, 2]], b["faces"][:, [2, 0]]]), 1)
    assert len(np.unique(e, axis=0)) == 20664
    # closed, consistently oriented: every directed edge appears once, its reverse once
    d = np.concatenate([b["faces"][:, [0, 1]], b["faces"][:, [1, 2]], b["faces"][:, [2, 0]]])
    assert len(np.unique(d, axis=0)) == len(d)
    w = b["canonical_lbs_weights"]
    assert w.shape == (6890, 24) and np.allclose(w.sum(1), 1, atol=1e-5) and (np.count_nonzero(w, axis=1) <= 4).all()


def test_subdivision_counts_and_child_order():
    b0 = syn.icosphere_body(1)
    v, f = b0["canonical_vertex"].astype(np.float64), b0["faces"]
    v1, f1, _ = syn.subdivide(v, f, None)
    assert len(f1) == 4 * len(f) and len(v1) == len(v) + len(f) * 3 // 2
    # children of face k are rows 4k..4k+3 and tile the parent: same total area
    def area(vv, ff):
        a, b_, c = vv[ff[:, 0]], vv[ff[:, 1]], vv[ff[:, 2]]
        return 0.5 * np.linalg.norm(np.cross(b_ - a, c - a), axis=1)
    np.testing.assert_allclose(area(v1, f1).reshape(-1, 4).sum(1), area(v, f), rtol=1e-9)
    assert (f1[0::4, 0] == f[:, 0]).all() and (f1[1::4, 1] == f[:, 1]).all() and (f1[2::4, 2] == f[:, 2]).all()


def test_frame_schema():
    fr = syn.make_frame(3, 512)
    assert fr["K"].shape == (1, 3, 3) and fr["E"].shape == (1, 4, 4) and fr["cnl_gtfms"].shape == (1, 24, 4, 4)
    assert fr["dst_Rs"].shape == (1, 24, 3, 3) and fr["dst_Ts"].shape == (1, 24, 3) and fr["dst_posevec"].shape == (1, 69)
    assert fr["bgcolor"].shape == (1, 3)
    R = fr["E"][0, :3, :3]
    np.testing.assert_allclose(R @ R.T, np.eye(3), atol=1e-6)
