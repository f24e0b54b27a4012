#!/bin/bash
# Round 5, GPU call 17: PPM's limiter deferred to one masked pass per direction in the finishing march, parking by masked moves
# (variant `defer`) against the default, same box: parity tests on the variant, headline A/B, SQ counters
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
APK_LIB_PATH=athenapk_amd/libapk_amd_defer.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_edge_cases.py -m gpu -q -x -k "fma or product or not strict" 2>&1 | tail -3 > gpurun_out/r05_ab17.txt
export BENCH_ARGS="--no-other-workloads --no-rehearsal --no-cpu-baseline --no-copies-base --sustained 0 --steps 20"
bash tools/r04_ab.sh "defer:APK_LIB_PATH=athenapk_amd/libapk_amd_defer.so" >> gpurun_out/r05_ab17.txt 2>&1
bash tools/r04_ab.sh "defer:APK_LIB_PATH=athenapk_amd/libapk_amd_defer.so" >> gpurun_out/r05_ab17.txt 2>&1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for lib in athenapk_amd/libapk_amd_defer.so ""; do
  rm -rf $R/gpurun_out/sq17
  ( cd $R && APK_LIB_PATH=$lib timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $R/gpurun_out/sq17 -o s -- python bench.py $BENCH_ARGS --steps 4 --warmup 1 --regions 1 > /dev/null 2>&1 )
  echo "== SQ counters, lib: ${lib:-default}" >> $R/gpurun_out/r05_ab17.txt
  ( cd $R && python tools/pmc_csv_summary.py gpurun_out/sq17/s_counter_collection.csv fused_m12f fused_march 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
for k,v in d.items():
    print('%-48s us %7.1f VALU %.4e lanes %.1f SALU %.3e BR %.3e'%(k[:48], v['avg_us'], v['SQ_INSTS_VALU'], v['SQ_THREAD_CYCLES_VALU']/v['SQ_INSTS_VALU'], v.get('SQ_INSTS_SALU',0), v.get('SQ_INSTS_BRANCH',0)))
" ) >> $R/gpurun_out/r05_ab17.txt 2>&1
done
rm -rf $R/gpurun_out/sq17
cd $R; cat gpurun_out/r05_ab17.txt
