#!/bin/bash
# Round 5, GPU call 11: PPM with flat lanes out of the extremum set (variant `flat`) against the default, same box:
# headline (no flat variable), general stage benchmark (two flat variables per direction), Orszag-Tang thin-z (config 3:
# everything flat along x3), the refined blast (config 5: flat ambient medium)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export BENCH_ARGS="--no-other-workloads --no-rehearsal --no-cpu-baseline --no-copies-base --sustained 0 --steps 20"
bash tools/r04_ab.sh "flat:APK_LIB_PATH=athenapk_amd/libapk_amd_flat.so" > gpurun_out/r05_ab11.txt 2>&1
bash tools/r04_ab.sh "flat:APK_LIB_PATH=athenapk_amd/libapk_amd_flat.so" >> gpurun_out/r05_ab11.txt 2>&1
for lib in athenapk_amd/libapk_amd_flat.so "" athenapk_amd/libapk_amd_flat.so ""; do
  echo "== lib: ${lib:-default}" >> gpurun_out/r05_ab11.txt
  APK_LIB_PATH=$lib python tools/amr_rate.py 2>&1 | grep blocks | head -2 >> gpurun_out/r05_ab11.txt
  APK_LIB_PATH=$lib python tools/ot_rate.py 2>&1 | tail -2 >> gpurun_out/r05_ab11.txt
done
cat gpurun_out/r05_ab11.txt
