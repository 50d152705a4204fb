#!/bin/bash
# On the GPU box: how much of a Model iteration the device is idle.  rocprofv3 kernel trace of scripts/model_iter.py (eager "gom" or "graph") ->
# per iteration (between two optimizer launches): span, union of the kernels' busy intervals, sum of durations, launches, the ten largest gaps.
# usage: scripts/iter_gaps.sh TAG [gom|graph]
TAG=${1:-iter_gaps}; MODE=${2:-gom}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$TAG; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o it -- python scripts/model_iter.py 60 $MODE bf16x3 > $OUT/run.log 2>&1
tail -1 $OUT/run.log
python - "$OUT" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
ad = [i for i, r in enumerate(rows) if "k_adam_multi" in r["Kernel_Name"]]
spans, busy, sums, n = [], [], [], []
gaps = collections.Counter(); gapn = collections.Counter()
for a, b in zip(ad[25:-1], ad[26:]):
    it = rows[a + 1:b + 1]
    s0, e1 = int(rows[a]["End_Timestamp"]), int(it[-1]["End_Timestamp"])
    spans.append((e1 - s0) / 1e3); sums.append(sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in it) / 1e3); n.append(len(it))
    cur_end, u = s0, 0
    prev = rows[a]["Kernel_Name"]
    for r in it:
        st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if st > cur_end:
            g = (st - cur_end) / 1e3
            key = prev.replace("(anonymous namespace)::", "").replace("void ", "")[:36] + " -> " + r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:36]
            gaps[key] += g; gapn[key] += 1
        u += max(0, en - max(st, cur_end))
        if en > cur_end: cur_end = en; prev = r["Kernel_Name"]
    busy.append(u / 1e3)
m = lambda x: sum(x) / len(x)
print(f"iterations {len(spans)}: span {m(spans):.1f} us, device busy (union) {m(busy):.1f} us, idle {m(spans) - m(busy):.1f} us, sum of kernel durations {m(sums):.1f} us, launches {m(n):.0f}")
print("largest idle gaps per iteration (us, between which kernels):")
for k, v in sorted(gaps.items(), key=lambda kv: -kv[1])[:25]:
    print(f"  {v / len(spans):7.2f}  x{gapn[k] / len(spans):4.1f}  {k}")
PY
rm -rf $OUT/trace
