"""Development aid: a larger adaptive mesh (root 128^3 in 16^3 meshblocks, 4 levels, a few thousand blocks)."""
import sys, time, torch
sys.path.insert(0, ".")
from athenapk_amd import decks, driver
ov = ["parthenon/mesh/nx%d=128" % d for d in (1, 2, 3)] + ["parthenon/meshblock/nx%d=16" % d for d in (1, 2, 3)] + [
    "parthenon/mesh/numlevel=4", "problem/blast/radius_outer=0.25", "problem/blast/radius_inner=0.2", "problem/blast/pressure_ambient=0.1",
    "problem/blast/pressure_ratio=100"]
s = driver.Simulation(decks.load("blast_3d_amr"), ov).initialize()
i = s.refresh_info()
print("blocks", i.nblocks_total, "cells %.2e" % i.zones_total, "levels", 1 + max(s.block_level(lb) for lb in range(i.nblocks_total)), flush=True)
for _ in range(3):
    s.step()
torch.cuda.synchronize(); z0 = s.amr_stats()[3]; t = time.perf_counter()
for _ in range(30):
    s.step()
torch.cuda.synchronize(); dt = time.perf_counter() - t
i = s.refresh_info()
print("after 33 cycles: blocks", i.nblocks_total, "zone-cycles/s %.3e" % ((s.amr_stats()[3] - z0) / dt), "ms/cycle %.2f" % (dt / 30 * 1e3), "refined/merged", s.amr_stats()[:2], flush=True)
h = s.history()
print("mass %.15f" % h[0])
