#!/bin/bash
# On the GPU box: rocprofv3 kernel trace of a short bench run -> the kernels of ONE steady-state step in launch order with their
# start offsets, durations and the idle gap before each (where the step's time goes besides the kernels themselves).
# usage: scripts/timeline.sh TAG [bench args]
TAG=${1:-timeline}; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o bench -- python bench.py --no-modes --no-cpu-baseline --steps 60 --warmup 20 "$@" > $OUT/bench.log 2>&1
grep "^{" $OUT/bench.log | cut -c1-200
python - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:40]
names = [short(r["Kernel_Name"]) for r in rows]
# a step = from one k_fk* launch to the next; take the 50th: inside the timed loop of graph replays (20 warm-up + 60 timed steps; the
# steps after those belong to bench.py's event-bracketed per-kernel pass, launched one by one)
starts = [i for i, n in enumerate(names) if n.startswith("k_fk")]   # k_fk_fwd or k_fk_lbs_fwd: the first kernel of a step
a, b = starts[50], starts[51]
t0 = int(rows[a]["Start_Timestamp"]); prev_end = t0
out = []
for r, n in zip(rows[a:b], names[a:b]):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    out.append(f"{n:40s} start {(s - t0) / 1e3:8.1f}  dur {(e - s) / 1e3:7.1f}  gap {(s - prev_end) / 1e3:6.1f}")
    prev_end = max(prev_end, e)
out.append(f"step span {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us, kernels {b - a}")
open(sys.argv[1] + "/timeline.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
rm -rf $OUT/trace
