#!/bin/bash
# usage: pmc_pass.sh TAG "COUNTER1 COUNTER2 ..." [bench args]   -- one rocprofv3 PMC pass (kernel trace only) over a short bench run
TAG=$1; CTRS=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $OUT -o pmc -- python bench.py --steps 10 --warmup 2 --inflight 1 --no-graph --no-cpu-baseline --no-configs "$@" > $OUT/log.txt 2>&1
python - "$OUT" <<'PY'
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
if not f:
    print(open(sys.argv[1] + "/log.txt").read()[-2000:]); sys.exit(0)
tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"]
    if "anonymous namespace" not in k: continue
    k = k.split("::")[1].split("(")[0]
    tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
names = sorted({c for k in tot for c in tot[k]})
print("kernel".ljust(24), *[c.rjust(22) for c in names])
for k in sorted(tot):
    print(k.ljust(24), *[("%.4g" % (tot[k][c] / max(1, n[k][c]))).rjust(22) for c in names])
PY
