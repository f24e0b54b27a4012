#!/bin/bash
# Round 5, GPU call 5: tracebacks of the tests that failed at the start of the session, the whole GPU suite with the row-addressed
# marches, then same-box A/B: round-5 start (base) / x1 stencil by DPP (x1dpp) / default, on the headline, WENOZ RK3 and hydro cycles
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_driver.py -m gpu -q -x --tb=short \
  -k "(takes_its_input_from_the_conserved_state and dt_only and rough and strict and ppm and filled and input_u1) or (density_floor_that_fires and rk3_ppm and fma)" 2>&1 | tail -60 > gpurun_out/r05_pytest5a.txt
timeout 600 python -m pytest tests/test_gpu_driver.py -m gpu -q --tb=short -k "density_floor_that_fires" 2>&1 | tail -40 > gpurun_out/r05_pytest5b.txt
( time timeout 1700 python -m pytest tests -m gpu -q 2>&1 | grep -v "^\.\|^$" | tail -40 ) > gpurun_out/r05_pytest5.txt 2>&1
export BENCH_ARGS="--no-other-workloads --no-rehearsal --no-cpu-baseline --no-copies-base --sustained 0 --steps 20"
bash tools/r04_ab.sh "base:APK_LIB_PATH=athenapk_amd/libapk_amd_base.so" "x1dpp:APK_LIB_PATH=athenapk_amd/libapk_amd_x1dpp.so" > gpurun_out/r05_ab5.txt 2>&1
export BENCH_ARGS="--no-other-workloads --no-rehearsal --no-cpu-baseline --no-copies-base --sustained 0 --steps 8 --workload mhd_wenoz_hlld_rk3_256"
bash tools/r04_ab.sh "base:APK_LIB_PATH=athenapk_amd/libapk_amd_base.so" >> gpurun_out/r05_ab5.txt 2>&1
export BENCH_ARGS="--no-other-workloads --no-rehearsal --no-cpu-baseline --no-copies-base --sustained 0 --steps 20 --workload hydro_plm_hllc_rk2_256"
bash tools/r04_ab.sh "base:APK_LIB_PATH=athenapk_amd/libapk_amd_base.so" >> gpurun_out/r05_ab5.txt 2>&1
for lib in athenapk_amd/libapk_amd_base.so ""; do echo "== stage_time lib: ${lib:-default}" >> gpurun_out/r05_ab5.txt; APK_LIB_PATH=$lib python tools/stage_time.py --gam0 0.5 --fill 2 --dt --reps 6 2>&1 | tail -3 >> gpurun_out/r05_ab5.txt; done
cat gpurun_out/r05_pytest5a.txt | tail -30; tail -15 gpurun_out/r05_pytest5.txt; cat gpurun_out/r05_ab5.txt
