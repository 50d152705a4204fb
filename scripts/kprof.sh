#!/bin/bash
# On the GPU box: rocprofv3 kernel trace of a short bench run -> per-kernel average durations (top 30 by total time).
# usage: scripts/kprof.sh TAG [bench args]
TAG=${1:-kprof}; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python bench.py --no-modes --no-cpu-baseline --steps 200 --warmup 20 "$@" > $OUT/bench.log 2>&1
grep "^{" $OUT/bench.log | cut -c1-400
python - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/trace/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
import shutil; shutil.copy(f, sys.argv[1] + "/kernel_stats.csv")
for r in rows[:30]:
    name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    print(name[:48].ljust(48), r["Calls"].rjust(6), ("%.2f" % (float(r["AverageNs"]) / 1e3)).rjust(9), r["Percentage"].rjust(7))
PY
rm -rf $OUT/trace
