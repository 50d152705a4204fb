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
print(("ALL launches" if os.environ.get("ALL") else "LPIPS launches") + " total: %.1f us" % tot)
# the summary bench.py's roofline_lpips cites (profiles/<tag>_lpips_launches.json): launches and the convolutions' share of one LPIPS evaluation
import json
lp = [j for j in range(n0) if any(k in seqs[0][j]["Kernel_Name"] for k in keys)]
dur = lambda j: sum(int(s[j]["End_Timestamp"]) - int(s[j]["Start_Timestamp"]) for s in seqs) / len(seqs) / 1e3
conv = [j for j in lp if "k_conv" in seqs[0][j]["Kernel_Name"] or "k_splitk" in seqs[0][j]["Kernel_Name"]]
conv_us = sum(dur(j) for j in conv)
layers = ((3, 64, 1), (64, 64, 1), (64, 128, 2), (128, 128, 2), (128, 256, 4), (256, 256, 4), (256, 256, 4), (256, 512, 8), (512, 512, 8), (512, 512, 8), (512, 512, 16), (512, 512, 16), (512, 512, 16))
flops = 2.0 * sum(ci * co * 9 * (512 // d) ** 2 for ci, co, d in layers) * 3 * 3
json.dump({"what": "LPIPS launches of ONE Model iteration (cfg 2, bf16x3), rocprofv3 kernel trace, average over iterations; conv_only = the 3x3 / 1x1 convolutions + split-K epilogues",
           "iterations_used": len(seqs), "launches_per_evaluation": len(lp), "lpips_kernel_us": round(tot, 1),
           "conv_only": {"launches": len(conv), "us": round(conv_us, 1), "flops_3_walks_3_passes": int(flops), "tflops": round(flops / conv_us / 1e6, 1), "frac_of_2500": round(flops / conv_us / 1e6 / 2500.0, 4)}},
          open(sys.argv[1] + "/lpips_launches.json", "w"), indent=1)
PY
rm -rf $OUT/trace
