"""Helpers for the -m gpu tests: run the HIP path through the C ABI and pull
its internal state back for bit-exact comparison with the oracle."""
import numpy as np
import torch

from gomavatar_amd import _lib
from gomavatar_amd import rasterizer as R


def gom_camera(cam: dict) -> _lib.GomCamera:
    return _lib.make_camera(cam["H"], cam["W"], cam["tanfovx"], cam["tanfovy"], np.asarray(cam["viewmatrix"], np.float32).reshape(-1),
                            np.asarray(cam["projmatrix"], np.float32).reshape(-1), np.asarray(cam.get("bg", np.zeros(4)), np.float32).reshape(-1))


def dev(a, dtype=torch.float32):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).cuda()


def hip_forward(cam, means, cov6, colors, op, state=None, requires_grad=False, reuse=False, means2D=False):
    """means2D: also pass a (P, 3) screen-space dummy that requires grad -- the reference's `screenspace_points`, whose .grad is the fifth
    gradient of the boundary (appended to the returned tensor list)."""
    st = state if state is not None else R.RasterState()
    t = [dev(means), dev(cov6), dev(colors), dev(op)]
    if requires_grad:
        for x in t:
            x.requires_grad_()
    m2 = None
    if means2D:
        m2 = torch.zeros((t[0].shape[0], 3), dtype=torch.float32, device="cuda", requires_grad=True)
        t.append(m2)
    out, radii = R.rasterize(t[0], t[1], t[2], t[3], gom_camera(cam), state=st, reuse_binning=reuse, means2D=m2)
    return out, radii, st, t


def export_state(st, P, H, W):
    """All integer/float state of the last forward as numpy arrays."""
    gx, gy = (W + 15) // 16, (H + 15) // 16
    D, overflow = st.poll()
    e = {}
    e["D"], e["overflow"] = D, overflow
    e["depth"] = st.export(_lib.BUF_DEPTH, torch.empty(P, dtype=torch.float32, device="cuda")).cpu().numpy()
    e["xy"] = st.export(_lib.BUF_XY, torch.empty((P, 2), dtype=torch.float32, device="cuda")).cpu().numpy()
    e["conic_opacity"] = st.export(_lib.BUF_CONIC_OPACITY, torch.empty((P, 4), dtype=torch.float32, device="cuda")).cpu().numpy()
    e["tiles_touched"] = st.export(_lib.BUF_TILES_TOUCHED, torch.empty(P, dtype=torch.int32, device="cuda")).cpu().numpy().view(np.uint32)
    e["rect"] = st.export(_lib.BUF_RECT, torch.empty((P, 4), dtype=torch.int16, device="cuda")).cpu().numpy().view(np.uint16).astype(np.int32)
    e["tile_base"] = st.export(_lib.BUF_TILE_BASE, torch.empty(gx * gy + 1, dtype=torch.int32, device="cuda")).cpu().numpy().view(np.uint32)
    n = max(D, 1)
    e["keys"] = st.export(_lib.BUF_KEYS, torch.empty(n, dtype=torch.int64, device="cuda")).cpu().numpy().view(np.uint64)[:D]
    e["point_list"] = st.export(_lib.BUF_POINT_LIST, torch.empty(n, dtype=torch.int32, device="cuda")).cpu().numpy().view(np.uint32)[:D]
    e["final_T"] = st.export(_lib.BUF_FINAL_T, torch.empty((H, W), dtype=torch.float32, device="cuda")).cpu().numpy()
    e["n_contrib"] = st.export(_lib.BUF_N_CONTRIB, torch.empty((H, W), dtype=torch.int32, device="cuda")).cpu().numpy().view(np.uint32)
    return e


def assert_binning_bit_exact(e, f):
    """HIP state `e` vs oracle forward `f`: every integer output identical."""
    assert not e["overflow"]
    assert e["D"] == f["D"]
    np.testing.assert_array_equal(e["tiles_touched"], f["tiles_touched"])
    np.testing.assert_array_equal(e["rect"], f["rect"])
    np.testing.assert_array_equal(e["depth"].view(np.uint32), f["depth"].astype(np.float32).view(np.uint32))
    np.testing.assert_array_equal(e["xy"].view(np.uint32), f["xy"].astype(np.float32).view(np.uint32))
    np.testing.assert_array_equal(e["conic_opacity"].view(np.uint32), f["conic_opacity"].astype(np.float32).view(np.uint32))
    # tile ranges: oracle stores (first, last) per non-empty tile, HIP the exclusive scan
    cnt = (f["ranges"][:, 1] - f["ranges"][:, 0]).astype(np.int64)
    np.testing.assert_array_equal(np.diff(e["tile_base"].astype(np.int64)), cnt)
    nz = cnt > 0
    np.testing.assert_array_equal(e["tile_base"][:-1][nz], f["ranges"][nz, 0])
    # sorted keys: oracle key = tile<<32 | depth ; HIP key = depth<<32 | gaussian inside the tile's range
    np.testing.assert_array_equal(e["point_list"], f["point_list"])
    np.testing.assert_array_equal((e["keys"] >> np.uint64(32)).astype(np.uint32), (f["keys"] & np.uint64(0xffffffff)).astype(np.uint32))
    np.testing.assert_array_equal((e["keys"] & np.uint64(0xffffffff)).astype(np.uint32), f["point_list"])
