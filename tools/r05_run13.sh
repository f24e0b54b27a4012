#!/bin/bash
# Round 5, GPU call 13: segment length of the two-row donor-cell predictor on the headline (8 default / 10 / 12 / 16 / 32), and the
# refined-mesh rate with 32-bit index arithmetic in the refinement operators
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export BENCH_ARGS="--no-other-workloads --no-rehearsal --no-cpu-baseline --no-copies-base --sustained 0 --steps 20"
bash tools/r04_ab.sh "kseg12:APK_DC3_KSEG=12" "kseg16:APK_DC3_KSEG=16" "kseg32:APK_DC3_KSEG=32" "kseg6:APK_DC3_KSEG=6" > gpurun_out/r05_ab13.txt 2>&1
for i in 1 2; do python tools/amr_rate.py 2>&1 | grep blocks | head -2 >> gpurun_out/r05_ab13.txt; done
bash tools/amr_stats.sh 2>&1 | grep "blocks\|refine_ops\|flux_fix\|kernels, us" >> gpurun_out/r05_ab13.txt
cat gpurun_out/r05_ab13.txt
