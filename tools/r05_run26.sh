#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_driver.py tests/test_gpu_configs.py tests/test_gpu_two_ranks.py -m gpu -q > gpurun_out/r05_suite26.txt 2>&1
grep "passed\|failed\|Error\|error" gpurun_out/r05_suite26.txt | tail -15 > gpurun_out/r05_ab26.txt
for lib in athenapk_amd/libapk_amd_prev.so ""; do
  echo "== lib: ${lib:-default}" >> gpurun_out/r05_ab26.txt
  APK_LIB_PATH=$lib python tools/ot_rate.py 2>&1 | grep blocks >> gpurun_out/r05_ab26.txt
done
cat gpurun_out/r05_ab26.txt
