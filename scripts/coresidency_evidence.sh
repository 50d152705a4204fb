#!/bin/bash
# Everything LABBOOK R6.8 quotes, on one box -> gpurun_out/coresidency/
O=${O:-gpurun_out/coresidency}; mkdir -p $O
(cd scripts/ubench && hipcc --offload-arch=gfx950 -O3 -o /tmp/pk pkfma_beside_mfma.hip 2>/dev/null && PK_MORE=1 timeout 300 /tmp/pk) > $O/pkfma_beside_mfma.txt 2>&1
{
  echo "# scripts/mc_forensics.py: victim = the weight-gradient kernels of the shadow MLP (fp32 VALU, csrc/mlp.hip) alone in its process; three aggressor PROCESSES beside it"
  bash scripts/mc_forensics.sh 2>&1 | grep -v amdgpu.ids
  echo "# one process, two streams"
  python scripts/mc_forensics.py inproc_valu 8 2>&1 | grep -v amdgpu.ids
  python scripts/mc_forensics.py inproc_mc 8 2>&1 | grep -v amdgpu.ids
} > $O/wgrad_victim.txt
{
  echo "# scripts/coresidency_product.py: one Model training iteration repeated on the current stream, an aggressor back to back on a side stream of the same process"
  python scripts/coresidency_product.py none 50 2>&1 | tail -1
  python scripts/coresidency_product.py lpips 400 2>&1 | tail -1
  python scripts/coresidency_product.py lpips 200 512 1 2>&1 | tail -1
  python scripts/coresidency_product.py mc 300 2>&1 | cut -c1-400 | tail -4
} > $O/model_iteration.txt
cat $O/*.txt
hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o /tmp/libpkvictim.so scripts/ubench/pkfma_victim.hip 2>/dev/null
python scripts/coresidency_victim.py /tmp/libpkvictim.so 6 2>&1 | grep -v amdgpu.ids > $O/strongest_victim.txt; cat $O/strongest_victim.txt
