"""Builds libgom_hip.so (hand-written HIP for gfx950) in-tree with hipcc.

No torch / cmake involved: the library is a plain C-ABI shared object
(include/gom_hip.h).  hipcc cross-compiles for gfx950 without a GPU.
"""
from __future__ import annotations

import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
_INC = os.path.join(os.path.dirname(_HERE), "include")
LIB_PATH = os.path.join(_HERE, "libgom_hip.so")

# (source, extra flags).  raster_pre.hip carries the bit-exact binning
# contract -> no FMA contraction there.
SOURCES = [
    ("gom_api.hip", []),
    ("raster_pre.hip", ["-ffp-contract=off"]),
    ("raster_render.hip", []),
    ("raster_rank.hip", []),
    ("geom.hip", []),
    ("loss.hip", []),
    ("lpips.hip", []),
    ("metrics.hip", []),
    ("vgg_bf16.hip", []),
    ("lpips_vgg_api.hip", []),
    ("mesh_raster.hip", []),
    ("mesh_losses.hip", []),
    ("posenc.hip", []),
    ("mlp.hip", []),
    ("mlp_mc.hip", []),
    ("frame_parallel.hip", []),
]
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function",
          f"-I{_INC}", f"-I{_CSRC}"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(_CSRC, f) for f in os.listdir(_CSRC)] + [os.path.join(_INC, "gom_hip.h"), os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every translation unit for gfx950 and link libgom_hip.so."""
    if not force and not needs_build():
        return LIB_PATH
    hipcc = _hipcc()
    objdir = os.path.join(_HERE, "_obj")
    os.makedirs(objdir, exist_ok=True)
    procs = []
    objs = []
    for src, extra in SOURCES:
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(obj)
        cmd = [hipcc, *COMMON, *extra, "-c", os.path.join(_CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    failed = []
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed.append(f"--- {src} ---\n{out}")
        elif verbose and out.strip():
            print(out, file=sys.stderr)
    if failed:
        raise RuntimeError("hipcc failed:\n" + "\n".join(failed))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH, *objs]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stdout)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
