#!/bin/bash
# Development aid (run on the GPU box from the repo root): rocprofv3 kernel statistics of tools/amr_prof.py (the refined-mesh
# MHD blast of BASELINE config 5's shape, 60 cycles) as microseconds per cycle.  Extra arguments go to amr_prof.py.
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/amr_stats
rm -rf $O; mkdir -p $O
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o s -- python tools/amr_prof.py "$@" > $O/amr_prof.txt 2> /dev/null
cat $O/amr_prof.txt
python - $(find $O -name "*kernel_stats.csv" | head -1) <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:24]:
    print("%-64s calls %5s  us/cycle %7.1f  avg us %7.1f  %5.1f%%" % (r['Name'][:64], r['Calls'], float(r['TotalDurationNs']) / 63e3, float(r['AverageNs']) / 1e3, 100 * float(r['TotalDurationNs']) / tot))
print("kernels, us per cycle: %.1f" % (tot / 63e3))
P
