#!/usr/bin/env python
"""Builds an experiment variant of libgom_hip.so with extra -D flags:
python scripts/exp_build.py NAME -DFOO ...  -> gomavatar_amd/_variants/libgom_hip_NAME.so"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gomavatar_amd import build as B
name, flags = sys.argv[1], sys.argv[2:]
csrc = os.environ.get("GOM_CSRC", B._CSRC)   # (another checkout of csrc/, e.g. `git worktree add /tmp/wt HEAD~1`)
# The development switches (knock-outs GOM_KO_*, workgroup timelines GOM_PHASE_PROF, counters GOM_BLK_STATS) are NOT in the product sources
# (round 6): csrc/lab/dev_switches.patch puts them back into a scratch copy of csrc/ when a flag asks for one.
if any(f.startswith(("-DGOM_KO_", "-DGOM_PHASE_PROF", "-DGOM_BLK_STATS")) for f in flags) and "GOM_CSRC" not in os.environ:
    import shutil, tempfile
    root = tempfile.mkdtemp(prefix="gom_dev_")
    shutil.copytree(os.path.dirname(B._HERE), root, dirs_exist_ok=True, ignore=shutil.ignore_patterns("*.so", "*.o", "_obj", "_variants", "gpurun_out", ".git", "__pycache__", "tests", "profiles"))
    subprocess.check_call(["patch", "-p1", "-s", "-i", os.path.join(B._CSRC, "lab", "dev_switches.patch")], cwd=root)
    csrc = os.path.join(root, "gomavatar_amd", "csrc")
out_dir = os.path.join(B._HERE, "_variants"); os.makedirs(out_dir, exist_ok=True)
objs = []
for src, extra in B.SOURCES:
    obj = os.path.join(out_dir, f"{name}_{src.replace('.hip', '.o')}")
    subprocess.check_call([B._hipcc(), *[c.replace(B._CSRC, csrc) for c in B.COMMON], *extra, *flags, "-c", os.path.join(csrc, src), "-o", obj])
    objs.append(obj)
out = os.path.join(out_dir, f"libgom_hip_{name}.so")
subprocess.check_call([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, *objs])
print(out)
