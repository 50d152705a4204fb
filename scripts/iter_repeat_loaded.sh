#!/bin/bash
# scripts/iter_repeat.py in NP concurrent processes on the one device (each is the others' load): a kernel whose result depends on timing
# (a missing barrier, a cross-stream race) shows up as "NOT bitwise" in some process.  usage: scripts/iter_repeat_loaded.sh [NP] [runs] [img] [subdiv]
NP=${1:-4}; RUNS=${2:-40}; IMG=${3:-256}; SUB=${4:-0}
mkdir -p gpurun_out/repeat
for MC in 1 0; do
  pids=()
  for i in $(seq 1 $NP); do
    GOM_MLP_MATRIX_CORES=$MC python scripts/iter_repeat.py $IMG $SUB $RUNS > gpurun_out/repeat/mc${MC}_p$i.log 2>&1 &
    pids+=($!)
  done
  for p in "${pids[@]}"; do wait $p; done
  echo "== GOM_MLP_MATRIX_CORES=$MC: $NP processes x $RUNS runs"
  grep -c "bitwise$" gpurun_out/repeat/mc${MC}_p*.log
  grep -h "NOT bitwise" gpurun_out/repeat/mc${MC}_p*.log | head -20
  grep -h "Error\|Traceback" gpurun_out/repeat/mc${MC}_p*.log | head -5
done
