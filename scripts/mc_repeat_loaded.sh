#!/bin/bash
# usage: scripts/mc_repeat_loaded.sh [NP] [runs] [img]
NP=${1:-6}; RUNS=${2:-300}; IMG=${3:-256}
mkdir -p gpurun_out/mcrep
for MC in ${MCS:-1 0}; do
  pids=()
  for i in $(seq 1 $NP); do
    python scripts/mc_repeat_loaded.py $IMG $RUNS $MC > gpurun_out/mcrep/mc${MC}_p$i.log 2>&1 &
    pids+=($!)
  done
  for p in "${pids[@]}"; do wait $p; done
  echo "== matrix cores $MC"
  cat gpurun_out/mcrep/mc${MC}_p*.log | cut -c1-500 | head -60
done
