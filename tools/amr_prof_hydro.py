"""Development aid: the adaptive hydro PLM+HLLE VL2 blast of BASELINE config 5 as decked (near-vacuum ambient), 60 cycles (for rocprofv3)."""
import sys, time, torch
sys.path.insert(0, ".")
from athenapk_amd import decks, driver
ov = ["parthenon/mesh/nx%d=64" % d for d in (1, 2, 3)] + ["parthenon/meshblock/nx%d=16" % d for d in (1, 2, 3)] + [
    "parthenon/mesh/numlevel=4", "parthenon/time/tlim=1.0"] + sys.argv[1:]
s = driver.Simulation(decks.load("blast_3d_amr"), ov).initialize()
for _ in range(3): s.step()
torch.cuda.synchronize(); z0 = s.amr_stats()[3]; t = time.perf_counter(); n = 0
while n < 60: s.step(); n += 1
torch.cuda.synchronize(); dt = time.perf_counter() - t
i = s.refresh_info()
print("blocks", i.nblocks_total, "zone-cycles/s %.3e" % ((s.amr_stats()[3] - z0) / dt), "ms/cycle %.3f" % (dt / n * 1e3), flush=True)
