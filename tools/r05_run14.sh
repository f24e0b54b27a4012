#!/bin/bash
# Round 5, GPU call 14: the x1 Riemann solve and the x2 reconstruction under the mask of the lanes whose results are used (variant
# `maskdead`) against the default, same box: headline, WENOZ RK3
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export BENCH_ARGS="--no-other-workloads --no-rehearsal --no-cpu-baseline --no-copies-base --sustained 0 --steps 20"
bash tools/r04_ab.sh "maskdead:APK_LIB_PATH=athenapk_amd/libapk_amd_maskdead.so" > gpurun_out/r05_ab14.txt 2>&1
bash tools/r04_ab.sh "maskdead:APK_LIB_PATH=athenapk_amd/libapk_amd_maskdead.so" >> gpurun_out/r05_ab14.txt 2>&1
export BENCH_ARGS="--no-other-workloads --no-rehearsal --no-cpu-baseline --no-copies-base --sustained 0 --steps 8 --workload mhd_wenoz_hlld_rk3_256"
bash tools/r04_ab.sh "maskdead:APK_LIB_PATH=athenapk_amd/libapk_amd_maskdead.so" >> gpurun_out/r05_ab14.txt 2>&1
APK_LIB_PATH=athenapk_amd/libapk_amd_maskdead.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "fused or lean or two_kernel" 2>&1 | tail -3 >> gpurun_out/r05_ab14.txt
cat gpurun_out/r05_ab14.txt
