#!/bin/bash
# round-4 profile collection (run on the GPU box from the repo root): kernel statistics, HBM traffic counters
# (separate passes, --kernel-trace only), SQ counters of the headline and the WENOZ RK3 workloads, instruction-mix
# and lane-activity counters of the north-star stage, the refined-mesh kernel statistics, dependent-chain issue rates.
# Summaries go under gpurun_out/r04/; tools/collect_profiles_r04.py turns them into profiles/r04_*.
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04
mkdir -p $O
cd $R
B="python bench.py --no-cpu-baseline --no-copies-base --no-rehearsal --no-other-workloads"
python bench.py > $O/bench_plain.json 2> $O/bench_plain.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- $B > $O/bench_under_rocprof.json 2> /dev/null
pmc() { tag=$1; shift; n=$1; shift; timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$tag -o s -- $CMD > $O/$tag.log 2>&1; echo "$tag rc=$?" >> $O/passes.txt; }
CMD="$B --steps 4 --warmup 1"
pmc fetch 0 FETCH_SIZE
pmc write 0 WRITE_SIZE
pmc sq 0 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
pmc clk 0 GRBM_GUI_ACTIVE
CMD="$B --steps 3 --warmup 1 --workload mhd_wenoz_hlld_rk3_256"
pmc sq_wenoz 0 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
CMD="python tools/stage_time.py --gam0 0.5 --fill 2 --dt --reps 6"
pmc mix 0 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT64 SQ_THREAD_CYCLES_VALU
pmc other 0 SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_FLAT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU
python bench.py --no-cpu-baseline --no-rehearsal --workload mhd_wenoz_hlld_rk3_256 > $O/bench_wenoz.json 2> /dev/null
python bench.py --no-cpu-baseline --no-rehearsal --workload hydro_plm_hllc_rk2_256 > $O/bench_hydro.json 2> /dev/null
python bench.py --no-cpu-baseline --no-copies-base --no-rehearsal --no-other-workloads --amr-extra > $O/bench_amr_extra.json 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/amr_stats -o s -- python tools/amr_prof.py > $O/amr_prof.txt 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/turb_stats -o s -- python tools/turb_prof.py > $O/turb_prof.txt 2> /dev/null
tools/ubench/ubench_march_traffic > $O/ubench_march_traffic.jsonl 2>&1
# sustained rate: 500 cycles of the headline workload
python bench.py --no-cpu-baseline --no-copies-base --no-rehearsal --no-other-workloads --steps 500 --warmup 5 > $O/bench_sustained_500.json 2> /dev/null
find $O -name "*.csv" | head -40
cat $O/passes.txt
