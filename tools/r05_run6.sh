#!/bin/bash
# Round 5, GPU call 6: the tests repaired after call 5, then same-box A/B of where the finishing march requests u1 / d3
# (APK_M12F_LOADS_MODE 1 / 2 against the default) on the headline, the WENOZ RK3 cycle and the general stage
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_driver.py tests/test_gpu_two_ranks.py -m gpu -q --tb=short \
  -k "takes_its_input_from_the_conserved_state or density_floor_that_fires or (sharing_one_gpu and sod_outflow)" 2>&1 | tail -30 > gpurun_out/r05_pytest6.txt
export BENCH_ARGS="--no-other-workloads --no-rehearsal --no-cpu-baseline --no-copies-base --sustained 0 --steps 20"
bash tools/r04_ab.sh "loads1:APK_LIB_PATH=athenapk_amd/libapk_amd_loads1.so" "loads2:APK_LIB_PATH=athenapk_amd/libapk_amd_loads2.so" "base:APK_LIB_PATH=athenapk_amd/libapk_amd_base.so" > gpurun_out/r05_ab6.txt 2>&1
export BENCH_ARGS="--no-other-workloads --no-rehearsal --no-cpu-baseline --no-copies-base --sustained 0 --steps 8 --workload mhd_wenoz_hlld_rk3_256"
bash tools/r04_ab.sh "loads1:APK_LIB_PATH=athenapk_amd/libapk_amd_loads1.so" "loads2:APK_LIB_PATH=athenapk_amd/libapk_amd_loads2.so" >> gpurun_out/r05_ab6.txt 2>&1
tail -8 gpurun_out/r05_pytest6.txt; cat gpurun_out/r05_ab6.txt
