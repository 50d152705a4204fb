#!/bin/bash
# On the GPU box: rocprofv3 kernel trace of scripts/model_iter.py -> per-kernel calls per iteration and average durations.
# usage: scripts/model_iter_prof.sh TAG [iters] [torch|gom] [bf16x3|bf16]
TAG=${1:-model_iter}; shift
IT=${1:-100}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o it -- python scripts/model_iter.py "$@" > $OUT/run.log 2>&1
tail -2 $OUT/run.log
python - "$OUT" "$IT" <<'PY'
import csv, glob, sys, shutil
f = glob.glob(sys.argv[1] + "/trace/**/*kernel_stats.csv", recursive=True)[0]
iters = int(sys.argv[2]) + 10
rows = list(csv.DictReader(open(f)))
shutil.copy(f, sys.argv[1] + "/kernel_stats.csv")
tot = 0.0
for r in rows:
    tot += float(r["TotalDurationNs"])
print("kernel time per iteration: %.3f ms over %d kernel names" % (tot / iters / 1e6, len(rows)))
for r in rows[:70]:
    name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("at::native::", "")
    print(name[:70].ljust(70), ("%.1f" % (int(r["Calls"]) / iters)).rjust(6), ("%.2f" % (float(r["AverageNs"]) / 1e3)).rjust(9), ("%.1f" % (float(r["TotalDurationNs"]) / iters / 1e3)).rjust(9))
PY
rm -rf $OUT/trace
