#!/bin/bash
# A/B of library variants on ONE box: scripts/ab.sh "bench args" NAME1 NAME2 ...  (NAME = "" / product, or a variant of scripts/exp_build.py); 3 alternating rounds.
args="$1"; shift
for r in 1 2 3; do
  for n in "$@"; do
    lib=""; [ "$n" != "product" ] && lib="gomavatar_amd/_variants/libgom_hip_$n.so"
    GOM_HIP_LIB=$lib python bench.py $args --no-modes --no-configs --no-cpu-baseline --steps 300 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); k=j['roofline']['all_kernels_us']
print('%-14s' % '$n', ' '.join('%s %.1f' % (a, b) for a, b in k.items()), '| ms/step %.4f  value %.0f' % (j['ms_per_step'], j['value']))"
  done
done
