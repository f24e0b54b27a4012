#!/bin/bash
# Development aid (GPU box, repo root): rocprofv3 kernel statistics of tools/reh_run.py (environment passed through), us per cycle.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${TAG:-prof}
O=$R/gpurun_out/reh_$TAG
rm -rf $O; mkdir -p $O
cd $R
REPS=1 CYCLES=20 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o s -- python tools/reh_run.py > $O/prof.txt 2> /dev/null
grep ms/cycle $O/prof.txt
python - $(find $O -name "*kernel_stats.csv" | head -1) <<'Q'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:12]:
    print("%-72s calls %5s  us/cycle %7.1f  avg us %7.1f" % (r['Name'][:72], r['Calls'], float(r['TotalDurationNs']) / 23e3, float(r['AverageNs']) / 1e3))
print("kernels, us per cycle: %.1f" % (tot / 23e3))
Q
