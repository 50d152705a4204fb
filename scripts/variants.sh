#!/bin/bash
# On the GPU box: the per-kernel HIP-event table of bench.py for the main library and for each experiment variant given.
# usage: scripts/variants.sh name1 name2 ...   (variants built by scripts/exp_build.py; "main" = the in-tree library)
cd $GRAFT_REPO_ROOT
for v in main "$@"; do
  if [ $v = main ]; then unset GOM_HIP_LIB; else export GOM_HIP_LIB=$GRAFT_REPO_ROOT/gomavatar_amd/_variants/libgom_hip_$v.so; fi
  python bench.py --no-modes --no-configs --no-cpu-baseline --steps 150 --warmup 20 2>&1 | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['all_kernels_us']
print('$v'.ljust(10), 'fps', d['value'], ' '.join(f'{n}={v:.0f}' for n,v in k.items()))"
done
