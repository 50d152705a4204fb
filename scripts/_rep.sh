cd $GRAFT_REPO_ROOT
for i in $(seq 1 12); do
  python -m pytest tests/test_gpu_model_parallel.py -q -s > gpurun_out/rep3_$i.log 2>&1
  tail -1 gpurun_out/rep3_$i.log >> gpurun_out/rep3.log
done
