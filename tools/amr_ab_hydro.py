"""Development aid: the refined hydro blast of BASELINE config 5 as decked (PLM+HLLE VL2, 16^3 blocks, 4 levels), 40 cycles after 3,
repeated; prints the median rate of the repeats (one process = one setting of the environment switches)."""
import sys, time, statistics, torch
sys.path.insert(0, ".")
from athenapk_amd import decks, driver
mb = int(sys.argv[2]) if len(sys.argv) > 2 else 16  # (meshblock size; the root mesh is four blocks wide)
ov = ["parthenon/mesh/nx%d=%d" % (d, 4 * mb) for d in (1, 2, 3)] + ["parthenon/meshblock/nx%d=%d" % (d, mb) for d in (1, 2, 3)] + [
      "parthenon/mesh/numlevel=%d" % (4 if mb <= 16 else 3), "parthenon/time/tlim=0.02"]
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
    nb = s.refresh_info().nblocks_total
    del s
print("median %.4e  min %.4e  max %.4e  (%d x 40 cycles, %d blocks)" % (statistics.median(rates), min(rates), max(rates), reps, nb), flush=True)
