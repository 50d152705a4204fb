#!/bin/bash
# usage: pmc_model_iter.sh TAG "COUNTER1 COUNTER2 ..."   -- one rocprofv3 PMC pass (kernel trace only) over 20 cfg-2 Model iterations (scripts/model_iter.py)
TAG=$1; CTRS=$2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
GOM_LPIPS_PREFETCH=0 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $OUT -o pmc -- python scripts/model_iter.py 20 > $OUT/log.txt 2>&1
python - "$OUT" <<'PY'
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
if not f:
    print(open(sys.argv[1] + "/log.txt").read()[-2000:]); sys.exit(0)
tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"]
    if not any(s in k for s in ("conv", "lpips", "splitk")): continue
    k = k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
names = sorted({c for k in tot for c in tot[k]})
print("kernel (per-launch averages)".ljust(44), *[c.rjust(24) for c in names])
for k in sorted(tot):
    print(k[:44].ljust(44), *[("%.4g" % (tot[k][c] / max(1, n[k][c]))).rjust(24) for c in names])
PY
rm -rf $OUT/*/ 2>/dev/null
