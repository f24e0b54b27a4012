#!/bin/bash
# Round 5, GPU call 1: the whole GPU suite, then same-box A/B of the PPM variants on the bench cycle, then the default bench line.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/r05_pytest1.txt 2>&1
export BENCH_ARGS="--no-other-workloads --no-rehearsal --sustained 0 --steps 20"
bash tools/r04_ab.sh "noflat:APK_LIB_PATH=athenapk_amd/libapk_amd_noflat.so" "pad3:APK_LIB_PATH=athenapk_amd/libapk_amd_pad3.so" \
  "pad5:APK_LIB_PATH=athenapk_amd/libapk_amd_pad5.so" "pad9:APK_LIB_PATH=athenapk_amd/libapk_amd_pad9.so" > gpurun_out/r05_ab1.txt 2>&1
( time python bench.py --steps 20 ) > gpurun_out/r05_bench1.json 2> gpurun_out/r05_bench1.err
tail -3 gpurun_out/r05_pytest1.txt; cat gpurun_out/r05_ab1.txt; tail -4 gpurun_out/r05_bench1.err
