#!/bin/bash
# Round 5, GPU call 12: segment length of the single-march hydro stage (16 default / 32 / 64 / 128 planes), same box; what the
# remaining blits of a refined-mesh cycle are (kernel trace around them)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export BENCH_ARGS="--no-other-workloads --no-rehearsal --no-cpu-baseline --no-copies-base --sustained 0 --steps 20 --workload hydro_plm_hllc_rk2_256"
bash tools/r04_ab.sh "kseg32:APK_S3_KSEG=32" "kseg64:APK_S3_KSEG=64" "kseg128:APK_S3_KSEG=128" "kseg8:APK_S3_KSEG=8" > gpurun_out/r05_ab12.txt 2>&1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/amr_trace; mkdir -p $R/gpurun_out/amr_trace
( cd $R && rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/gpurun_out/amr_trace -o t -- python tools/amr_prof.py > /dev/null 2>&1 )
cd $R
python - <<'P' >> gpurun_out/r05_ab12.txt 2>&1
import csv, glob
k = sorted(csv.DictReader(open(glob.glob('gpurun_out/amr_trace/*kernel_trace.csv')[0])), key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'][:60] for r in k]
# one cycle in the middle: from a dc3r2 launch to the next
idx = [i for i, n in enumerate(names) if 'fused_dc3r2' in n]
a, b = idx[30], idx[31]
t0 = int(k[a]['Start_Timestamp'])
for r in k[a:b]:
    print("%8.1f %7.1f %s" % ((int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r['Kernel_Name'][:90]))
print("cycle us:", (int(k[b]['Start_Timestamp']) - t0) / 1e3)
P
rm -rf gpurun_out/amr_trace
cat gpurun_out/r05_ab12.txt
