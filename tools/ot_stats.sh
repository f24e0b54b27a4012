#!/bin/bash
# Development aid (GPU box, repo root): rocprofv3 kernel statistics of the Orszag-Tang 512 x 512 x 4 cycle (BASELINE config 3's
# shape in 128 x 128 x 4 meshblocks, VL2 PPM+HLLD), microseconds per cycle.
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/ot_stats
rm -rf $O; mkdir -p $O
cd $R
cat > /tmp/ot_prof.py <<'P'
import sys, time, torch
sys.path.insert(0, ".")
from athenapk_amd import decks, driver
import os
ov = ([] if os.environ.get("OT_2D") else ["parthenon/mesh/nx3=4", "parthenon/meshblock/nx3=4"]) + ["parthenon/meshblock/nx1=128", "parthenon/meshblock/nx2=128", "hydro/first_order_flux_correct=false"]
s = driver.Simulation(decks.load("orszag_tang"), ov + sys.argv[1:]).initialize()
for _ in range(3): s.step()
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(60): s.step()
torch.cuda.synchronize(); dt = time.perf_counter() - t
print("ms/cycle %.3f  cell-updates/s %.3e" % (dt / 60 * 1e3, s.refresh_info().zones_total * 60 / dt), flush=True)
P
python /tmp/ot_prof.py "$@"
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o s -- python /tmp/ot_prof.py "$@" > $O/ot_prof.txt 2> /dev/null
cat $O/ot_prof.txt
python - $(find $O -name "*kernel_stats.csv" | head -1) <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:24]:
    print("%-64s calls %5s  us/cycle %7.1f  avg us %7.1f  %5.1f%%" % (r['Name'][:64], r['Calls'], float(r['TotalDurationNs']) / 63e3, float(r['AverageNs']) / 1e3, 100 * float(r['TotalDurationNs']) / tot))
print("kernels, us per cycle: %.1f" % (tot / 63e3))
P
