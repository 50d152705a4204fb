#!/bin/bash
# Runs on the GPU box (via gpurun): kernel trace + stats of the default bench command, then two separate PMC passes
# (FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950), everything under gpurun_out/prof_$TAG.
TAG=${1:-r04}
BENCH_ARGS=${2:-}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python bench.py $BENCH_ARGS > $OUT/bench_trace.log 2>&1
grep "^{" $OUT/bench_trace.log > $OUT/bench_line.json
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- python bench.py $BENCH_ARGS --steps 20 --warmup 4 --inflight 1 --no-graph --no-cpu-baseline --no-modes --no-configs > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pmc -- python bench.py $BENCH_ARGS --steps 20 --warmup 4 --inflight 1 --no-graph --no-cpu-baseline --no-modes --no-configs > $OUT/pmc_write.log 2>&1
python - "$OUT" <<'PY'
import csv, glob, collections, json, sys
out = sys.argv[1]
def per_kernel(d, counter):
    f = glob.glob(f"{out}/{d}/**/*counter_collection.csv", recursive=True)[0]
    tot = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            tot[r["Kernel_Name"]] += float(r["Counter_Value"]); n[r["Kernel_Name"]] += 1
    return {k: tot[k] / n[k] for k in tot}
fetch, write = per_kernel("pmc_fetch", "FETCH_SIZE"), per_kernel("pmc_write", "WRITE_SIZE")
res = {}
for k in sorted(set(fetch) | set(write)):
    if "anonymous namespace" not in k: continue
    name = k.split("::")[1].split("(")[0].split("<")[0]
    # bench.py's kernel ids: the event pair of "sort" brackets the tile pass, "depth_rank" the bucket scatter + sort, "seg_bwd" whichever backward ran
    name = {"k_tile_rank": "k_sort", "k_seg_bwd_pair": "k_seg_bwd", "k_bucket_scatter": "k_depth_rank", "k_bucket_sort": "k_depth_rank"}.get(name, name)
    # KB per launch; FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950 (MI355X_MICROARCH.md, HBM): doubled.
    # Template instantiations launched once per step each (k_sort<256> + k_sort<1024>) are summed under one name.
    r = res.setdefault(name, {"fetch_kb_raw": 0.0, "write_kb_raw": 0.0, "hbm_bytes": 0.0})
    r["fetch_kb_raw"] += fetch.get(k, 0.0); r["write_kb_raw"] += write.get(k, 0.0)
    r["hbm_bytes"] += (2.0 * fetch.get(k, 0.0) + write.get(k, 0.0)) * 1024.0
line = json.loads(open(f"{out}/bench_line.json").read())
res["batch"] = line["config"]["frames_per_gpu_per_step"]
json.dump(res, open(f"{out}/traffic.json", "w"), indent=1)
for k, v in res.items(): print(k.ljust(20), {a: round(b) for a, b in v.items()} if isinstance(v, dict) else v)
PY
python - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/trace/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:22]:
    print(r["Name"][:60].ljust(60), r["Calls"].rjust(6), ("%.2f" % (float(r["AverageNs"]) / 1e3)).rjust(9), r["Percentage"].rjust(7))
PY
cat $OUT/bench_line.json | cut -c1-600
