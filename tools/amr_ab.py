"""Development aid: the refined MHD blast of BASELINE config 5's shape, 40 cycles after 3, repeated; prints the median
rate of the repeats (one process = one setting of the environment switches)."""
import sys, time, statistics, torch
sys.path.insert(0, ".")
from athenapk_amd import decks, driver
ov = ["parthenon/mesh/nx1=64", "parthenon/mesh/nx2=64", "parthenon/mesh/nx3=64", "parthenon/meshblock/nx1=16", "parthenon/meshblock/nx2=16",
      "parthenon/meshblock/nx3=16", "parthenon/mesh/numlevel=4", "parthenon/time/tlim=0.02", "hydro/fluid=glmmhd", "hydro/riemann=hlld",
      "hydro/reconstruction=ppm", "parthenon/mesh/nghost=4", "problem/blast/pressure_ambient=1.0", "problem/blast/pressure_ratio=100"]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
rates = []
for _ in range(reps):
    s = driver.Simulation(decks.load("blast_3d_amr"), ov).initialize()
    for _ in range(3):
        s.step()
    torch.cuda.synchronize()
    z0 = s.amr_stats()[3]
    t = time.perf_counter()
    for _ in range(40):
        s.step()
    torch.cuda.synchronize()
    rates.append((s.amr_stats()[3] - z0) / (time.perf_counter() - t))
    del s
print("median %.4e  min %.4e  max %.4e  (%d x 40 cycles)" % (statistics.median(rates), min(rates), max(rates), reps), flush=True)
