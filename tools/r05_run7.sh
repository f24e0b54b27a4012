#!/bin/bash
# Round 5, GPU call 7: whole GPU suite (merged flux-correction plan, end-of-cycle gather kernel, two-row predictor's segment
# length), then the refined-mesh rate and the headline against the round's starting point on the same box
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 1700 python -m pytest tests -m gpu -q 2>&1 | grep -v "^\.\|^$" | tail -40 ) > gpurun_out/r05_pytest7.txt 2>&1
for lib in athenapk_amd/libapk_amd_base.so "" athenapk_amd/libapk_amd_base.so ""; do
  echo "== lib: ${lib:-default}" >> gpurun_out/r05_ab7.txt
  APK_LIB_PATH=$lib python tools/amr_rate.py 2>&1 | grep blocks | head -2 >> gpurun_out/r05_ab7.txt
done
export BENCH_ARGS="--no-other-workloads --no-rehearsal --no-cpu-baseline --no-copies-base --sustained 0 --steps 20"
bash tools/r04_ab.sh "base:APK_LIB_PATH=athenapk_amd/libapk_amd_base.so" >> gpurun_out/r05_ab7.txt 2>&1
bash tools/amr_stats.sh > gpurun_out/r05_amr_stats7.txt 2>&1
tail -12 gpurun_out/r05_pytest7.txt; cat gpurun_out/r05_ab7.txt; tail -28 gpurun_out/r05_amr_stats7.txt
