#!/bin/bash
# Round 5, GPU call 15: HLLD's side operands by masked moves instead of selects (variant `mpick`) against the default, same box
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export BENCH_ARGS="--no-other-workloads --no-rehearsal --no-cpu-baseline --no-copies-base --sustained 0 --steps 20"
bash tools/r04_ab.sh "mpick:APK_LIB_PATH=athenapk_amd/libapk_amd_mpick.so" > gpurun_out/r05_ab15.txt 2>&1
bash tools/r04_ab.sh "mpick:APK_LIB_PATH=athenapk_amd/libapk_amd_mpick.so" >> gpurun_out/r05_ab15.txt 2>&1
export BENCH_ARGS="--no-other-workloads --no-rehearsal --no-cpu-baseline --no-copies-base --sustained 0 --steps 8 --workload mhd_wenoz_hlld_rk3_256"
bash tools/r04_ab.sh "mpick:APK_LIB_PATH=athenapk_amd/libapk_amd_mpick.so" >> gpurun_out/r05_ab15.txt 2>&1
cat gpurun_out/r05_ab15.txt
