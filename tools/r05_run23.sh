#!/bin/bash
# Round 5, GPU call 23: boundary-plane fluxes beside the stage kernels: AMR tests, rate with / without, kernel table
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_amr.py tests/test_gpu_configs.py tests/test_amr_ops.py tests/test_gpu_two_ranks.py -m gpu -q -x > gpurun_out/r05_suite23.txt 2>&1
grep "passed\|failed\|error" gpurun_out/r05_suite23.txt | tail -5 > gpurun_out/r05_ab23.txt
for v in 1 "" 1 ""; do
  echo "== APK_AMR_PLANES_INLINE=${v:-unset}" >> gpurun_out/r05_ab23.txt
  if [ -n "$v" ]; then export APK_AMR_PLANES_INLINE=1; else unset APK_AMR_PLANES_INLINE; fi
  python tools/amr_rate.py 2>&1 | grep blocks | head -2 >> gpurun_out/r05_ab23.txt
done
unset APK_AMR_PLANES_INLINE
bash tools/amr_stats.sh 2>&1 | head -20 >> gpurun_out/r05_ab23.txt
cat gpurun_out/r05_ab23.txt
