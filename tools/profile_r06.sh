#!/bin/bash
# round-6 profile collection (run on the GPU box from the repo root): kernel statistics, HBM traffic counters
# (separate passes, --kernel-trace only), SQ counters of the headline, the WENOZ RK3 and the hydro PLM+HLLC workloads,
# instruction-mix and lane-activity counters of the north-star stage, the refined-mesh and forced-turbulence kernel
# statistics.  Summaries go under gpurun_out/r06/; tools/collect_profiles_r06.py turns them into profiles/r06_*.
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06
mkdir -p $O
cd $R
B="python bench.py --no-cpu-baseline --no-copies-base --no-rehearsal --no-other-workloads --sustained 0"
python bench.py --steps 20 > $O/bench_plain.json 2> $O/bench_plain.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- $B --steps 20 > $O/bench_under_rocprof.json 2> /dev/null
pmc() { tag=$1; shift; timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$tag -o s -- $CMD > $O/$tag.log 2>&1; echo "$tag rc=$?" >> $O/passes.txt; }
CMD="$B --steps 4 --warmup 1 --regions 1"
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
pmc sq SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
pmc clk GRBM_GUI_ACTIVE
CMD="$B --steps 3 --warmup 1 --regions 1 --workload mhd_wenoz_hlld_rk3_256"
pmc sq_wenoz SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
CMD="$B --steps 4 --warmup 1 --regions 1 --workload hydro_plm_hllc_rk2_256"
pmc fetch_hydro FETCH_SIZE
pmc write_hydro WRITE_SIZE
pmc sq_hydro SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
pmc mix_hydro SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT64 SQ_THREAD_CYCLES_VALU
CMD="python tools/stage_time.py --gam0 0.5 --fill 2 --dt --reps 6"
pmc mix SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT64 SQ_THREAD_CYCLES_VALU
pmc other SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_FLAT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU
python bench.py --no-cpu-baseline --no-rehearsal --sustained 0 --steps 10 --workload mhd_wenoz_hlld_rk3_256 > $O/bench_wenoz.json 2> /dev/null
python bench.py --no-cpu-baseline --no-rehearsal --sustained 0 --steps 40 --workload hydro_plm_hllc_rk2_256 > $O/bench_hydro.json 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/hydro_stats -o s -- $B --steps 20 --workload hydro_plm_hllc_rk2_256 > $O/bench_hydro_under_rocprof.json 2> /dev/null
python bench.py --no-cpu-baseline --no-copies-base --no-rehearsal --no-other-workloads --sustained 0 --amr-extra > $O/bench_amr_extra.json 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/amr_stats -o s -- python tools/amr_prof.py > $O/amr_prof.txt 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/turb_stats -o s -- python tools/turb_prof.py > $O/turb_prof.txt 2> /dev/null
# the one-GPU rehearsal of an 8-GPU rank (x1 strips in the buffers / packed): kernel statistics
REPS=1 CYCLES=20 rocprofv3 --kernel-trace --stats --output-format csv -d $O/reh_stats -o s -- python tools/reh_run.py > $O/reh_prof.txt 2> /dev/null
APK_X1_DIRECT=0 REPS=1 CYCLES=20 rocprofv3 --kernel-trace --stats --output-format csv -d $O/reh_packed_stats -o s -- python tools/reh_run.py > $O/reh_packed_prof.txt 2> /dev/null
# the scheme's issue floor (csrc/bench_floor.hip): its instruction mix
CMD="python tools/floor_run.py"
pmc mix_floor SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT64 SQ_THREAD_CYCLES_VALU
python tools/floor_run.py > $O/floor.txt 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ot_stats -o s -- python tools/wl_rate.py orszag_tang_512x512x4_vl2 > $O/ot_prof.txt 2> /dev/null
find $O -name "*.csv" | head -60
cat $O/passes.txt
