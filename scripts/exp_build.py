#!/usr/bin/env python
"""Builds an experiment variant of libgom_hip.so with extra -D flags:
python scripts/exp_build.py NAME -DFOO ...  -> gomavatar_amd/_variants/libgom_hip_NAME.so"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gomavatar_amd import build as B
name, flags = sys.argv[1], sys.argv[2:]
csrc = os.environ.get("GOM_CSRC", B._CSRC)   # (another checkout of csrc/, e.g. `git worktree add /tmp/wt HEAD~1`)
out_dir = os.path.join(B._HERE, "_variants"); os.makedirs(out_dir, exist_ok=True)
objs = []
for src, extra in B.SOURCES:
    obj = os.path.join(out_dir, f"{name}_{src.replace('.hip', '.o')}")
    subprocess.check_call([B._hipcc(), *[c.replace(B._CSRC, csrc) for c in B.COMMON], *extra, *flags, "-c", os.path.join(csrc, src), "-o", obj])
    objs.append(obj)
out = os.path.join(out_dir, f"libgom_hip_{name}.so")
subprocess.check_call([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, *objs])
print(out)
