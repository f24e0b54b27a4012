#!/bin/bash
# Round 5, GPU call 2: whole GPU suite (all failures), A/B of the flat-stencil test (headline, cfg5 mesh, cfg3 thin-z), single-march hydro stage A/B
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 1700 python -m pytest tests -m gpu -q 2>&1 | tail -25 ) > gpurun_out/r05_pytest2.txt 2>&1
export BENCH_ARGS="--no-other-workloads --no-rehearsal --sustained 0 --steps 20"
bash tools/r04_ab.sh "noflat:APK_LIB_PATH=athenapk_amd/libapk_amd_noflat.so" > gpurun_out/r05_ab2.txt 2>&1
export BENCH_ARGS="--no-other-workloads --no-rehearsal --sustained 0 --steps 20 --workload hydro_plm_hllc_rk2_256"
bash tools/r04_ab.sh "two_kernel:APK_S3=0" "s3_kseg8:APK_S3_KSEG=8" "s3_kseg32:APK_S3_KSEG=32" >> gpurun_out/r05_ab2.txt 2>&1
for lib in "" athenapk_amd/libapk_amd_noflat.so; do
  echo "== lib: ${lib:-default}" >> gpurun_out/r05_ab2.txt
  APK_LIB_PATH=$lib python tools/amr_rate.py 2>&1 | head -2 >> gpurun_out/r05_ab2.txt
  APK_LIB_PATH=$lib python tools/ot_rate.py 2>&1 | tail -2 >> gpurun_out/r05_ab2.txt
done
( time python bench.py --steps 20 ) > gpurun_out/r05_bench2.json 2> gpurun_out/r05_bench2.err
tail -12 gpurun_out/r05_pytest2.txt; cat gpurun_out/r05_ab2.txt; tail -4 gpurun_out/r05_bench2.err
