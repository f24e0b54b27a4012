#!/bin/bash
# Development aid (GPU box, repo root): rocprofv3 kernel statistics of the one-GPU rehearsal of an 8-GPU rank (bench.py
# rehearsal_8gpu_rank: the headline workload with the brick's outer faces treated as remote), microseconds per cycle.
# OVERLAP=0: exchanges synchronous.
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/rehearsal_stats
rm -rf $O; mkdir -p $O
cd $R
cat > /tmp/reh_prof.py <<'P'
import os, sys, time, torch
sys.path.insert(0, ".")
from athenapk_amd import decks, driver
import bench
deck, fluid, integrator, recon, riemann, brick, mb, desc = bench.WORKLOADS["mhd_ppm_hlld_vl2_256"] if "mhd_ppm_hlld_vl2_256" in bench.WORKLOADS else list(bench.WORKLOADS.values())[0]
ov = ["parthenon/mesh/nx%d=%d" % (d + 1, brick) for d in range(3)] + ["parthenon/meshblock/nx%d=%d" % (d + 1, mb) for d in range(3)]
ov += ["parthenon/time/integrator=%s" % integrator, "hydro/reconstruction=%s" % recon, "hydro/riemann=%s" % riemann]
if os.environ.get("REHEARSE", "1") == "1": ov += ["apk_amd/rehearse_remote_faces=true"]
s = driver.Simulation(decks.load(deck), ov, strict=False)
s.set_overlap(os.environ.get("OVERLAP", "1") == "1")
s.initialize()
for _ in range(3): s.step()
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(20): s.step()
torch.cuda.synchronize(); dt = time.perf_counter() - t
print("ms/cycle %.3f" % (dt / 20 * 1e3), flush=True)
P
python /tmp/reh_prof.py
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o s -- python /tmp/reh_prof.py > $O/prof.txt 2> /dev/null
cat $O/prof.txt
python - $(find $O -name "*kernel_stats.csv" | head -1) <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:24]:
    print("%-64s calls %5s  us/cycle %7.1f  avg us %7.1f  %5.1f%%" % (r['Name'][:64], r['Calls'], float(r['TotalDurationNs']) / 23e3, float(r['AverageNs']) / 1e3, 100 * float(r['TotalDurationNs']) / tot))
print("kernels, us per cycle: %.1f" % (tot / 23e3))
P
