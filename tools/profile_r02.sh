#!/bin/bash
# round-2 profile collection (run on the GPU box from the repo root): kernel statistics, HBM traffic
# counters (separate passes, --kernel-trace only), SQ counters, effective clock; the summaries go
# under gpurun_out/r02/ and are copied into profiles/ by hand afterwards.
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02
mkdir -p $O
cd $R
python bench.py > $O/bench_plain.json 2> $O/bench_plain.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2> /dev/null
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o s -- python bench.py --no-cpu-baseline --steps 4 --warmup 1 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o s -- python bench.py --no-cpu-baseline --steps 4 --warmup 1 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/sq -o s -- python bench.py --no-cpu-baseline --steps 4 --warmup 1 > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/clk -o s -- python bench.py --no-cpu-baseline --steps 4 --warmup 1 > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/clk_ubench -o s -- tools/ubench/ubench_fp64 500 > $O/ubench_under_rocprof.jsonl 2>&1
tools/ubench/ubench_fp64 > $O/ubench_fp64.jsonl 2>&1
find $O -name "*.csv" | head -40
ls -la $O
