"""numpy/ctypes front-end of the C raster oracle (oracle/raster_oracle.c).

TEST INFRASTRUCTURE ONLY -- see the header of raster_oracle.c.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Dict

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS: Dict[str, ctypes.CDLL] = {}


def build(force: bool = False) -> None:
    """Compile the oracle with gcc (idempotent)."""
    out = os.path.join(_HERE, "_build", "liboracle_f32.so")
    src = os.path.join(_HERE, "raster_oracle.c")
    if force or not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])


def _lib(dtype) -> ctypes.CDLL:
    key = "f64" if np.dtype(dtype) == np.float64 else "f32"
    if key not in _LIBS:
        path = os.path.join(_HERE, "_build", f"liboracle_{key}.so")
        if not os.path.exists(path):
            build()
        lib = ctypes.CDLL(path)
        lib.or_scan.restype = ctypes.c_int64
        assert lib.or_real_size() == (8 if key == "f64" else 4)
        _LIBS[key] = lib
    return _LIBS[key]


def set_threads(n: int) -> None:
    for dt in (np.float32, np.float64):
        _lib(dt).or_set_threads(int(n))


def _camera_struct(cam: dict, dtype):
    real = ctypes.c_double if np.dtype(dtype) == np.float64 else ctypes.c_float

    class OrCamera(ctypes.Structure):
        _fields_ = [("H", ctypes.c_int), ("W", ctypes.c_int), ("tanfovx", real), ("tanfovy", real),
                    ("view", real * 16), ("proj", real * 16), ("bg", real * 4)]

    c = OrCamera()
    c.H, c.W = int(cam["H"]), int(cam["W"])
    # tanfov arrives as a python float computed in double (gaussian.py:33-36);
    # the float32 build rounds it once, like the float kernel argument upstream.
    c.tanfovx, c.tanfovy = float(cam["tanfovx"]), float(cam["tanfovy"])
    view = np.asarray(cam["viewmatrix"], dtype=dtype).reshape(16)
    proj = np.asarray(cam["projmatrix"], dtype=dtype).reshape(16)
    bg = np.zeros(4, dtype=dtype)
    b = np.asarray(cam.get("bg", np.zeros(3)), dtype=dtype).reshape(-1)
    bg[: min(4, b.size)] = b[:4]
    for i in range(16):
        c.view[i] = view[i]
        c.proj[i] = proj[i]
    for i in range(4):
        c.bg[i] = bg[i]
    return c


def _p(a: np.ndarray):
    return a.ctypes.data_as(ctypes.c_void_p)


def forward(cam: dict, means3D, cov6, colors, opacity, dtype=np.float32, form: str = "ref", margin: bool = False) -> dict:
    """form: "ref" = the reference's alpha expression, "hip" = the HIP path's formulation of the same alpha (pre-scaled conic, fma chain,
    exp2; raster_oracle.c A.3).  margin=True adds `margin` (H, W): per pixel the smallest relative distance of a branch-deciding quantity to
    its threshold (alpha vs 1/255, T (1 - alpha) vs 1e-4, the exponent's sign): pixels with a margin far above fp32 round-off take the same
    branches in any faithful fp32 implementation; and `roundoff` (H, W): the estimate of what one fp32 rounding of every exponent does to the
    pixel (large only under needle-shaped conics: ill-conditioned in fp32 in any formulation).
    Full forward.  cam: dict(H, W, tanfovx, tanfovy, viewmatrix(4,4) = E^T,
    projmatrix(4,4) = (K_ndc E)^T, bg).  Returns every intermediate:
    depth, radii, xy, conic_opacity, tiles_touched, rect, offsets, D,
    keys(u64), point_list(u32), ranges(tiles,2), color(C,H,W), final_T, n_contrib."""
    lib = _lib(dtype)
    means3D = np.ascontiguousarray(means3D, dtype=dtype)
    cov6 = np.ascontiguousarray(cov6, dtype=dtype)
    colors = np.ascontiguousarray(colors, dtype=dtype)
    opacity = np.ascontiguousarray(opacity, dtype=dtype).reshape(-1)
    P = means3D.shape[0]
    C = colors.shape[1]
    assert 1 <= C <= 4 and cov6.shape == (P, 6) and colors.shape[0] == P and opacity.shape[0] == P
    c = _camera_struct(cam, dtype)
    H, W = c.H, c.W
    gx, gy = (W + 15) // 16, (H + 15) // 16
    depth = np.zeros(P, dtype)
    radii = np.zeros(P, np.int32)
    xy = np.zeros((P, 2), dtype)
    conic_opacity = np.zeros((P, 4), dtype)
    tiles_touched = np.zeros(P, np.uint32)
    rect = np.zeros((P, 4), np.int32)
    lib.or_preprocess(ctypes.byref(c), P, _p(means3D), _p(cov6), _p(opacity), _p(depth), _p(radii), _p(xy),
                      _p(conic_opacity), _p(tiles_touched), _p(rect))
    offsets = np.zeros(P, np.uint32)
    D = int(lib.or_scan(P, _p(tiles_touched), _p(offsets)))
    keys = np.zeros(max(D, 1), np.uint64)
    vals = np.zeros(max(D, 1), np.uint32)
    ranges = np.zeros((gx * gy, 2), np.uint32)
    lib.or_bin(ctypes.byref(c), P, _p(depth), _p(radii), _p(rect), _p(offsets), ctypes.c_int64(D), _p(keys), _p(vals), _p(ranges))
    color = np.zeros((C, H, W), dtype)
    final_T = np.zeros((H, W), dtype)
    n_contrib = np.zeros((H, W), np.uint32)
    assert form in ("ref", "hip")
    mg = np.zeros((H, W), dtype) if margin else None
    ro = np.zeros((H, W), dtype) if margin else None
    lib.or_render_fwd_ex(ctypes.byref(c), C, _p(ranges), _p(vals), _p(xy), _p(conic_opacity), _p(colors), _p(color), _p(final_T), _p(n_contrib),
                         ctypes.c_int(1 if form == "hip" else 0), _p(mg) if margin else None, _p(ro) if margin else None)
    return dict(margin=mg, roundoff=ro, form=form, P=P, C=C, H=H, W=W, D=D, depth=depth, radii=radii, xy=xy, conic_opacity=conic_opacity,
                tiles_touched=tiles_touched, rect=rect, offsets=offsets, keys=keys[:D], point_list=vals[:D],
                ranges=ranges, color=color, final_T=final_T, n_contrib=n_contrib,
                _inputs=(means3D, cov6, colors, opacity), _cam=cam, _dtype=dtype)


def backward(fwd: dict, dL_dcolor) -> dict:
    """Backward of `forward`.  Returns dL_dmeans3D (P,3), dL_dcov6 (P,6),
    dL_dcolors (P,C), dL_dopacity (P,), dL_dmeans2D (P,2), dL_dconic (P,3)."""
    dtype = fwd["_dtype"]
    lib = _lib(dtype)
    means3D, cov6, colors, opacity = fwd["_inputs"]
    c = _camera_struct(fwd["_cam"], dtype)
    P, C = fwd["P"], fwd["C"]
    g = np.ascontiguousarray(dL_dcolor, dtype=dtype)
    assert g.shape == (C, fwd["H"], fwd["W"])
    dcol = np.zeros((P, C), dtype)
    dm2 = np.zeros((P, 2), dtype)
    dcon = np.zeros((P, 3), dtype)
    dop = np.zeros(P, dtype)
    vals = np.ascontiguousarray(fwd["point_list"]) if fwd["D"] > 0 else np.zeros(1, np.uint32)
    lib.or_render_bwd_ex(ctypes.byref(c), C, _p(fwd["ranges"]), _p(vals), _p(fwd["xy"]), _p(fwd["conic_opacity"]), _p(colors),
                         _p(fwd["final_T"]), _p(fwd["n_contrib"]), _p(g), _p(dcol), _p(dm2), _p(dcon), _p(dop),
                         ctypes.c_int(1 if fwd.get("form", "ref") == "hip" else 0))
    dmeans = np.zeros((P, 3), dtype)
    dcov = np.zeros((P, 6), dtype)
    lib.or_preprocess_bwd(ctypes.byref(c), P, _p(means3D), _p(cov6), _p(fwd["radii"]), _p(dcon), _p(dm2), _p(dmeans), _p(dcov))
    return dict(dL_dmeans3D=dmeans, dL_dcov6=dcov, dL_dcolors=dcol, dL_dopacity=dop, dL_dmeans2D=dm2, dL_dconic=dcon)
