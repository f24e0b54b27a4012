#!/bin/bash
# Round 5, GPU call 16: masked side pick in the marches only (default) against masked everywhere (variant `mpick`): headline A/B,
# then the parity / driver / edge-case tests
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export BENCH_ARGS="--no-other-workloads --no-rehearsal --no-cpu-baseline --no-copies-base --sustained 0 --steps 20"
bash tools/r04_ab.sh "mpick_all:APK_LIB_PATH=athenapk_amd/libapk_amd_mpick.so" > gpurun_out/r05_ab16.txt 2>&1
bash tools/r04_ab.sh "mpick_all:APK_LIB_PATH=athenapk_amd/libapk_amd_mpick.so" >> gpurun_out/r05_ab16.txt 2>&1
( time timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_driver.py tests/test_edge_cases.py -m gpu -q -x 2>&1 | grep -v "^\.\|^$" | tail -6 ) >> gpurun_out/r05_ab16.txt 2>&1
cat gpurun_out/r05_ab16.txt
