#!/bin/bash
# Round 5, GPU call 8: refined-mesh kernel table with the predictor's segment length forced to 4 / 6 / 8 / 16 planes
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for k in 0 4 6 8 16; do
  echo "== APK_DC3_KSEG=$k" >> gpurun_out/r05_amr8.txt
  APK_DC3_KSEG=$k bash tools/amr_stats.sh 2>&1 | grep "blocks\|dc3r2\|flux_fix\|kernels, us" >> gpurun_out/r05_amr8.txt
done
cat gpurun_out/r05_amr8.txt
