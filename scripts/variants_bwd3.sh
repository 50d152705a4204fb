#!/bin/bash
# On the GPU box: scripts/variants.sh in GOM_OPT_BWD_MODE 3 (records), timing only.
cd $GRAFT_REPO_ROOT
export GOM_BENCH_TIMING_ONLY=1
for v in main "$@"; do
  if [ $v = main ]; then unset GOM_HIP_LIB; else export GOM_HIP_LIB=$GRAFT_REPO_ROOT/gomavatar_amd/_variants/libgom_hip_$v.so; fi
  python bench.py --no-modes --no-configs --no-cpu-baseline --steps 150 --warmup 20 --bwd-mode 3 2>&1 | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['all_kernels_us']
print('$v'.ljust(10), 'fps', d['value'], ' '.join(f'{n}={v:.0f}' for n,v in k.items()))"
done
