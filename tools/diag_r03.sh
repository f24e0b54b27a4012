#!/bin/bash
# round-3 stall diagnosis of the PPM+HLLD general stage (run on the GPU box from the repo root):
# SQ counter passes on tools/stage_time.py, then (last, bounded) a PC-sampling attempt.
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/diag
mkdir -p $O
cd $R
CMD="python tools/stage_time.py --gam0 0.5 --fill 2 --dt --reps 6"
$CMD > $O/plain.txt 2>&1
rocprofv3 -L > $O/counters.txt 2>&1
pass() { tag=$1; shift; timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$tag -o s -- $CMD > $O/$tag.log 2>&1; echo "$tag rc=$?" >> $O/passes.txt; }
pass p1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
pass p2 SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_FLAT SQ_INSTS_VALU_TRANS_F64
pass p3 SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_IFETCH
pass p4 SQ_WAVE_CYCLES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_IFETCH_LEVEL SQ_INSTS_VMEM_RD
pass p5 SQ_WAVE_CYCLES SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_MISSES SQ_INSTS_VMEM_WR SQ_WAVE_DEP_WAIT
pass p6 SQ_WAVE_CYCLES SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_THREAD_CYCLES_VALU
for p in p1 p2 p3 p4 p5 p6; do f=$O/$p/s_counter_collection.csv; [ -f $f ] && python tools/pmc_csv_summary.py $f fused > $O/$p.json; done
# PC sampling (beta): bounded, last
timeout 180 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method stochastic --pc-sampling-unit cycles --pc-sampling-interval 65536 --kernel-trace --output-format csv -d $O/pcs -o s -- $CMD > $O/pcs.log 2>&1
echo "pcs stochastic rc=$?" >> $O/passes.txt
if [ ! -s $O/pcs/s_pc_sampling_stochastic.csv ]; then
timeout 180 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method host_trap --pc-sampling-unit time --pc-sampling-interval 10 --kernel-trace --output-format csv -d $O/pch -o s -- $CMD > $O/pch.log 2>&1
echo "pcs host_trap rc=$?" >> $O/passes.txt
fi
ls -la $O $O/pcs $O/pch 2>/dev/null | head -60
du -sh $O
cat $O/passes.txt
