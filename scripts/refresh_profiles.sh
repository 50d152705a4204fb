#!/bin/bash
# Regenerates everything under profiles/ for one round tag (run through gpurun; results land in gpurun_out/profiles_$TAG,
# copy them into profiles/ afterwards):   gpurun --timeout 2400 -- 'bash scripts/refresh_profiles.sh r01'
TAG=${1:-r01}
cd $GRAFT_REPO_ROOT
bash scripts/collect_profiles.sh $TAG > gpurun_out/collect_$TAG.log 2>&1
bash scripts/collect_profiles.sh ${TAG}_alone "--inflight 1" > gpurun_out/collect_${TAG}_alone.log 2>&1
bash scripts/pmc_pass.sh ${TAG}_insts "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" > gpurun_out/pmc_${TAG}_insts.log 2>&1
bash scripts/pmc_pass.sh ${TAG}_busy "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" > gpurun_out/pmc_${TAG}_busy.log 2>&1
OUT=gpurun_out/profiles_$TAG; mkdir -p $OUT
cp gpurun_out/prof_$TAG/bench_line.json $OUT/${TAG}_bench_line.json
cp gpurun_out/prof_$TAG/trace/*kernel_stats.csv $OUT/${TAG}_bench_kernel_stats.csv
cp gpurun_out/prof_$TAG/traffic.json $OUT/${TAG}_traffic.json
cp gpurun_out/prof_${TAG}_alone/bench_line.json $OUT/${TAG}_alone_inflight1_bench_line.json
cp gpurun_out/prof_${TAG}_alone/trace/*kernel_stats.csv $OUT/${TAG}_alone_inflight1_kernel_stats.csv
python scripts/summarize_pmc.py gpurun_out/prof_$TAG/pmc_fetch/*counter_collection.csv $OUT/${TAG}_pmc_fetch_per_kernel.csv
python scripts/summarize_pmc.py gpurun_out/prof_$TAG/pmc_write/*counter_collection.csv $OUT/${TAG}_pmc_write_per_kernel.csv
python scripts/summarize_pmc.py gpurun_out/pmc_${TAG}_insts/*counter_collection.csv $OUT/${TAG}_pmc_sq_insts_per_kernel.csv
python scripts/summarize_pmc.py gpurun_out/pmc_${TAG}_busy/*counter_collection.csv $OUT/${TAG}_pmc_sq_busy_per_kernel.csv
ls -la $OUT; cat $OUT/${TAG}_bench_line.json | cut -c1-400
