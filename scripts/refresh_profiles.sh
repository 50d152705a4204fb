#!/bin/bash
# Regenerates everything under profiles/ for one round tag (run through gpurun; results land in gpurun_out/profiles_$TAG,
# copy them into profiles/ afterwards):   gpurun --timeout 2400 -- 'bash scripts/refresh_profiles.sh r02'
# Runs TWICE the PMC-derived pieces feed bench.py: the bench line of the second pass carries roofline.traffic / roofline.valu read
# from the json files the first pass produced (copy them to profiles/ in between, or simply run this script twice).
TAG=${1:-r06}
cd $GRAFT_REPO_ROOT
# Round 6: the timed step is TWO concurrent 4-frame launch sequences (gom_split_forward_backward).  bench.py's roofline block describes the ONE-sequence
# 8-frame launch (the kernel owns the chip; comparable with rounds 2-5): every pass that feeds it (FETCH / WRITE / SQ counters, the metric-only kernel
# averages) therefore runs `bench.py --split 1`; the default command's own kernel trace -- overlapping 4-frame launches -- is kept beside it
# (<tag>_default_split2_kernel_stats.csv + bench line), and the bench line says which is which (roofline.timed_configuration).
bash scripts/collect_profiles.sh $TAG "--split 1" > gpurun_out/collect_$TAG.log 2>&1            # one 8-frame sequence, one step in flight
bash scripts/collect_profiles.sh ${TAG}_inflight3 "--split 1 --inflight 3" > gpurun_out/collect_${TAG}_inflight3.log 2>&1
( cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/prof_${TAG}_default && rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_default/trace -o bench -- python bench.py --no-modes --no-configs --no-cpu-baseline > gpurun_out/prof_${TAG}_default/bench_trace.log 2>&1; grep "^{" gpurun_out/prof_${TAG}_default/bench_trace.log > gpurun_out/prof_${TAG}_default/bench_line.json )
# the metric workload ALONE (no modes, no other configs): every launch of a kernel in this trace is a launch of the timed loop or of the
# event-bracketed per-kernel pass on it, so the per-kernel averages are the ones roofline.avg_us must agree with
( cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/prof_${TAG}_pure && rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_pure/trace -o bench -- python bench.py --split 1 --no-modes --no-configs --no-cpu-baseline > gpurun_out/prof_${TAG}_pure/bench_trace.log 2>&1; grep "^{" gpurun_out/prof_${TAG}_pure/bench_trace.log > gpurun_out/prof_${TAG}_pure/bench_line.json )
bash scripts/pmc_pass.sh ${TAG}_insts "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" --no-modes --split 1 > gpurun_out/pmc_${TAG}_insts.log 2>&1
bash scripts/pmc_pass.sh ${TAG}_lds "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" --no-modes --split 1 > gpurun_out/pmc_${TAG}_lds.log 2>&1
bash scripts/pmc_pass.sh ${TAG}_busy "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAIT_INST_LDS" --no-modes --split 1 > gpurun_out/pmc_${TAG}_busy.log 2>&1
OUT=gpurun_out/profiles_$TAG; mkdir -p $OUT
cp gpurun_out/prof_$TAG/bench_line.json $OUT/${TAG}_bench_line.json
cp gpurun_out/prof_$TAG/trace/*kernel_stats.csv $OUT/${TAG}_bench_kernel_stats.csv 2>/dev/null || cp $(find gpurun_out/prof_$TAG/trace -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_bench_kernel_stats.csv
cp gpurun_out/prof_$TAG/traffic.json $OUT/${TAG}_traffic.json
cp gpurun_out/prof_${TAG}_inflight3/bench_line.json $OUT/${TAG}_inflight3_bench_line.json
cp gpurun_out/prof_${TAG}_pure/bench_line.json $OUT/${TAG}_metric_only_bench_line.json
cp gpurun_out/prof_${TAG}_default/bench_line.json $OUT/${TAG}_default_split2_bench_line.json
cp $(find gpurun_out/prof_${TAG}_default/trace -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_default_split2_kernel_stats.csv
cp $(find gpurun_out/prof_${TAG}_pure/trace -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_metric_only_kernel_stats.csv
cp $(find gpurun_out/prof_${TAG}_inflight3/trace -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_inflight3_kernel_stats.csv
python scripts/summarize_pmc.py $(find gpurun_out/prof_$TAG/pmc_fetch -name "*counter_collection.csv" | head -1) $OUT/${TAG}_pmc_fetch_per_kernel.csv
python scripts/summarize_pmc.py $(find gpurun_out/prof_$TAG/pmc_write -name "*counter_collection.csv" | head -1) $OUT/${TAG}_pmc_write_per_kernel.csv
python scripts/summarize_pmc.py $(find gpurun_out/pmc_${TAG}_insts -name "*counter_collection.csv" | head -1) $OUT/${TAG}_pmc_sq_insts_per_kernel.csv
python scripts/summarize_pmc.py $(find gpurun_out/pmc_${TAG}_busy -name "*counter_collection.csv" | head -1) $OUT/${TAG}_pmc_sq_busy_per_kernel.csv
python scripts/summarize_pmc.py $(find gpurun_out/pmc_${TAG}_lds -name "*counter_collection.csv" | head -1) $OUT/${TAG}_pmc_sq_lds_per_kernel.csv
# the VALU axis of bench.py's roofline block: per kernel, from the SQ pass (quad-cycle counters summed over the waves of a launch)
python - "$OUT" "$TAG" <<'PY'
import csv, json, sys
out, tag = sys.argv[1], sys.argv[2]
res = {}
for r in csv.DictReader(open(f"{out}/{tag}_pmc_sq_busy_per_kernel.csv")):
    k = r["Kernel_Name"]
    if "anonymous namespace" not in k: continue
    name = k.split("::")[1].split("(")[0].split("<")[0]
    name = {"k_tile_rank": "k_sort", "k_seg_bwd_pair": "k_seg_bwd"}.get(name, name)
    g = lambda c: float(r.get(c + "_avg_per_launch") or 0.0)
    wc = g("SQ_WAVE_CYCLES")
    if not wc: continue
    res[name] = {"valu_insts_per_launch": round(g("SQ_INSTS_VALU")), "wave_quadcycles": round(wc),
                 "wave_parked_frac": round(g("SQ_WAIT_ANY") / wc, 4),          # at s_waitcnt / s_barrier
                 "issue_stall_frac": round(g("SQ_WAIT_INST_ANY") / wc, 4),
                 "inst_active_frac": round(g("SQ_ACTIVE_INST_ANY") / wc, 4),
                 "valu_active_frac_of_wave_time": round(g("SQ_ACTIVE_INST_VALU") / wc, 4),
                 "valu_active_over_busy": round(g("SQ_ACTIVE_INST_VALU") / max(g("SQ_BUSY_CYCLES"), 1.0), 4)}   # SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES as counted (quad-cycles over SE-cycles)
json.dump(res, open(f"{out}/{tag}_valu.json", "w"), indent=1)
print(json.dumps(res.get("k_seg_bwd"), indent=1))
PY
# the rocprofv3 averages of the GRAPH-REPLAYED launches of the metric workload alone, under bench.py's kernel ids (roofline.avg_us_rocprof)
python - "$OUT" "$TAG" <<'PY'
import csv, json, sys
out, tag = sys.argv[1], sys.argv[2]
res = {"batch": json.load(open(f"{out}/{tag}_metric_only_bench_line.json"))["config"]["frames_per_gpu_per_step"],
       "what": "rocprofv3 --kernel-trace --stats of `python bench.py --split 1 --no-modes --no-configs --no-cpu-baseline` (ONE 8-frame launch sequence per step: the configuration of bench.py's roofline block): AverageNs per kernel (graph-replayed launches of the timed loop + the event-bracketed pass)"}
for r in csv.DictReader(open(f"{out}/{tag}_metric_only_kernel_stats.csv")):
    k = r["Name"]
    if "anonymous namespace" not in k: continue
    name = k.split("::")[1].split("(")[0].split("<")[0]
    name = {"k_tile_rank": "k_sort", "k_seg_bwd_pair": "k_seg_bwd"}.get(name, name)
    res.setdefault(name, round(float(r["AverageNs"]) / 1e3, 2))
json.dump(res, open(f"{out}/{tag}_kernel_avg.json", "w"), indent=1)
PY
# the LPIPS launches of one Model iteration in order (both images as one batch at the loss: positions are stable) -> <tag>_lpips_conv.txt + <tag>_lpips_launches.json
GOM_LPIPS_PREFETCH=0 bash scripts/model_iter_layers.sh lpips_layers_$TAG 60 gom bf16x3 > $OUT/${TAG}_lpips_conv.txt 2>&1
cp gpurun_out/lpips_layers_$TAG/lpips_launches.json $OUT/${TAG}_lpips_launches.json
ls -la $OUT; cat $OUT/${TAG}_bench_line.json | cut -c1-600
# gpurun merges at most 64 MiB back: drop the raw traces / counter dumps, the summaries above are what is kept
rm -rf gpurun_out/prof_${TAG}_default/trace gpurun_out/prof_${TAG}_pure/trace gpurun_out/prof_${TAG}/trace gpurun_out/prof_${TAG}/pmc_fetch gpurun_out/prof_${TAG}/pmc_write gpurun_out/prof_${TAG}_inflight3/trace \
       gpurun_out/prof_${TAG}_inflight3/pmc_fetch gpurun_out/prof_${TAG}_inflight3/pmc_write gpurun_out/pmc_${TAG}_insts gpurun_out/pmc_${TAG}_busy gpurun_out/pmc_${TAG}_lds
