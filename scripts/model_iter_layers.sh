#!/bin/bash
# On the GPU box: rocprofv3 kernel trace of scripts/model_iter.py -> the LPIPS trunk's launches of one iteration IN ORDER (per layer), averaged over iterations.
# usage: scripts/model_iter_layers.sh TAG [iters] [torch|gom] [bf16x3|bf16]
TAG=${1:-model_iter_layers}; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o it -- python scripts/model_iter.py "$@" > $OUT/run.log 2>&1
tail -1 $OUT/run.log
python - "$OUT" <<'PY'
import csv, glob, sys, collections, os
f = glob.glob(sys.argv[1] + "/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
adam = [i for i, n in enumerate(names) if "k_adam_multi" in n]
seqs = []
for a, b in zip(adam[20:-1], adam[21:]):
    seqs.append(rows[a + 1:b + 1])
n0 = len(seqs[0]); seqs = [s for s in seqs if len(s) == n0]
print("iterations used:", len(seqs), " launches per iteration:", n0)
tot = 0.0
keys = ("conv", "pool", "lpips", "splitk")
for j in range(n0):
    nm = seqs[0][j]["Kernel_Name"]
    d = sum(int(s[j]["End_Timestamp"]) - int(s[j]["Start_Timestamp"]) for s in seqs) / len(seqs) / 1e3
    if os.environ.get("ALL") or any(k in nm for k in keys):
        tot += d
        r = seqs[0][j]
        short = nm.replace("(anonymous namespace)::", "").replace("void ", "")[:64]
        print(f"{j:4d} {short:64s} grid {int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']):6d} x {int(r['Grid_Size_Y']):3d} x {int(r['Grid_Size_Z']):3d}  wg {r['Workgroup_Size_X']:>4s}  {d:8.1f} us")
print("LPIPS launches total: %.1f us" % tot)
PY
rm -rf $OUT/trace
