#!/bin/bash
mkdir -p gpurun_out/forensics
for AG in none valu lpips mc; do
  pids=()
  if [ $AG != none ]; then for i in 1 2 3; do python scripts/mc_forensics.py $AG 14 > gpurun_out/forensics/${AG}_aggr$i.log 2>&1 & pids+=($!); done; fi
  python scripts/mc_forensics.py victim 12 > gpurun_out/forensics/${AG}_victim.log 2>&1
  for p in "${pids[@]}"; do wait $p; done
  echo "== aggressors: 3 x $AG"; tail -qn1 gpurun_out/forensics/${AG}_*.log
done
