#!/bin/bash
# Round 5, GPU call 9: PPM limiter branches without the wave's invalid edge lanes -- parity / driver / AMR tests, then
# same-box A/B against the commit before (prev) and the round's start (base): headline, Orszag-Tang thin-z, refined mesh;
# SQ_INSTS_VALU of the stage kernels
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 1700 python -m pytest tests/test_gpu_parity.py tests/test_gpu_driver.py tests/test_edge_cases.py tests/test_gpu_configs.py tests/test_gpu_amr.py -m gpu -q -x 2>&1 | grep -v "^\.\|^$" | tail -12 ) > gpurun_out/r05_pytest9.txt 2>&1
export BENCH_ARGS="--no-other-workloads --no-rehearsal --no-cpu-baseline --no-copies-base --sustained 0 --steps 20"
bash tools/r04_ab.sh "prev:APK_LIB_PATH=athenapk_amd/libapk_amd_prev.so" "base:APK_LIB_PATH=athenapk_amd/libapk_amd_base.so" > gpurun_out/r05_ab9.txt 2>&1
for lib in athenapk_amd/libapk_amd_prev.so "" athenapk_amd/libapk_amd_prev.so ""; do
  echo "== lib: ${lib:-default}" >> gpurun_out/r05_ab9.txt
  APK_LIB_PATH=$lib python tools/amr_rate.py 2>&1 | grep blocks | head -2 >> gpurun_out/r05_ab9.txt
  APK_LIB_PATH=$lib python tools/ot_rate.py 2>&1 | tail -2 >> gpurun_out/r05_ab9.txt
done
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for lib in athenapk_amd/libapk_amd_prev.so ""; do
  rm -rf $R/gpurun_out/sq9; 
  ( cd $R && APK_LIB_PATH=$lib timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/sq9 -o s -- python bench.py $BENCH_ARGS --steps 4 --warmup 1 --regions 1 > /dev/null 2>&1 )
  echo "== SQ counters, lib: ${lib:-default}" >> $R/gpurun_out/r05_ab9.txt
  ( cd $R && python tools/pmc_csv_summary.py gpurun_out/sq9/s_counter_collection.csv fused 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
for k,v in d.items():
    print('%-50s calls %3d avg_us %8.1f VALU %.4e lanes/inst %.1f'%(k[:50], v['calls'], v['avg_us'], v['SQ_INSTS_VALU'], v['SQ_THREAD_CYCLES_VALU']/v['SQ_INSTS_VALU']))
" ) >> $R/gpurun_out/r05_ab9.txt 2>&1
done
cd $R; tail -5 gpurun_out/r05_pytest9.txt; cat gpurun_out/r05_ab9.txt
