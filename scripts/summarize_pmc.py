#!/usr/bin/env python
"""rocprofv3 counter_collection.csv -> one row per kernel: launches and the per-launch average of every counter.
    python scripts/summarize_pmc.py gpurun_out/pmc_TAG/..._counter_collection.csv out.csv"""
import collections, csv, sys
src, dst = sys.argv[1], sys.argv[2]
tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
for r in csv.DictReader(open(src)):
    tot[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Kernel_Name"]][r["Counter_Name"]] += 1
names = sorted({c for k in tot for c in tot[k]})
with open(dst, "w", newline="") as f:
    w = csv.writer(f); w.writerow(["Kernel_Name", "Launches"] + [c + "_avg_per_launch" for c in names])
    for k in sorted(tot):
        if "at::native" in k or "rocclr" in k or "Cijk" in k:
            continue
        w.writerow([k, max(n[k].values())] + [round(tot[k][c] / max(1, n[k][c]), 3) for c in names])
