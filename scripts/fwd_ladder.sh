#!/bin/bash
# Round 6: the knock-out ladder of the render FORWARD (k_seg_T, k_seg_fwd), like the backward's (LABBOOK "Round 3, second pass"): development builds
# that leave one part of a kernel out (results invalid, timing valid), each timed by bench.py's per-kernel HIP-event brackets on the one-sequence
# 8-frame launch.  Run on the GPU box from the repository root; writes gpurun_out/fwd_ladder.txt.
set -u
out=gpurun_out/fwd_ladder.txt; : > $out
run() {   # name, library (empty = the product build)
  local lib=""; [ -n "$2" ] && lib="gomavatar_amd/_variants/libgom_hip_$2.so"
  GOM_HIP_LIB=$lib GOM_BENCH_TIMING_ONLY=1 python bench.py --split 1 --no-modes --no-configs --no-cpu-baseline --steps 100 --warmup 10 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); k=j['roofline']['all_kernels_us']
print('%-58s seg_T %6.1f  seg_fwd %6.1f  combine %6.1f  seg_bwd %6.1f   ms/step %.4f' % ('$1', k['seg_T'], k['seg_fwd'], k['combine'], k['seg_bwd'], j['ms_per_step']))" >> $out
}
run "product build" ""
run "k_seg_T: no alpha loop (KO_T=1)" ko_t1
run "k_seg_T: + no entry loads / cull / staging (KO_T=2)" ko_t2
run "k_seg_T: + no stores (KO_T=3: queue, descriptors, barrier)" ko_t3
run "k_seg_fwd: alphas, no serial chain (KO_FWD=3)" ko_f3
run "k_seg_fwd: no compositing loop (KO_FWD=1)" ko_f1
run "k_seg_fwd: every task dead on arrival (KO_FWD=2)" ko_f2
cat $out
