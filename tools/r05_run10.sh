#!/bin/bash
# Round 5, GPU call 10: lane validity as the threshold of PPM's extremum tests (no extra branch, no scalar registers) against
# the commit before (prev), same box: parity tests, headline A/B, SQ counters
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_driver.py tests/test_edge_cases.py -m gpu -q -x 2>&1 | grep -v "^\.\|^$" | tail -8 ) > gpurun_out/r05_pytest10.txt 2>&1
export BENCH_ARGS="--no-other-workloads --no-rehearsal --no-cpu-baseline --no-copies-base --sustained 0 --steps 20"
bash tools/r04_ab.sh "prev:APK_LIB_PATH=athenapk_amd/libapk_amd_prev.so" > gpurun_out/r05_ab10.txt 2>&1
bash tools/r04_ab.sh "prev:APK_LIB_PATH=athenapk_amd/libapk_amd_prev.so" >> gpurun_out/r05_ab10.txt 2>&1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for lib in athenapk_amd/libapk_amd_prev.so ""; do
  rm -rf $R/gpurun_out/sq10
  ( cd $R && APK_LIB_PATH=$lib timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $R/gpurun_out/sq10 -o s -- python bench.py $BENCH_ARGS --steps 4 --warmup 1 --regions 1 > /dev/null 2>&1 )
  echo "== SQ counters, lib: ${lib:-default}" >> $R/gpurun_out/r05_ab10.txt
  ( cd $R && python tools/pmc_csv_summary.py gpurun_out/sq10/s_counter_collection.csv fused_m12f fused_march 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
for k,v in d.items():
    print('%-48s us %7.1f VALU %.4e lanes %.1f SALU %.3e BR %.3e act/inst %.3f waitinst %.3f'%(k[:48], v['avg_us'], v['SQ_INSTS_VALU'], v['SQ_THREAD_CYCLES_VALU']/v['SQ_INSTS_VALU'], v.get('SQ_INSTS_SALU',0), v.get('SQ_INSTS_BRANCH',0), v['SQ_ACTIVE_INST_VALU']/v['SQ_INSTS_VALU'], v.get('SQ_WAIT_INST_ANY',0)/v['SQ_WAVE_CYCLES']))
" ) >> $R/gpurun_out/r05_ab10.txt 2>&1
done
cd $R; tail -4 gpurun_out/r05_pytest10.txt; cat gpurun_out/r05_ab10.txt
