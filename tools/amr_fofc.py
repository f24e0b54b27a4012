"""Development aid: the adaptive blast as decked (near-vacuum ambient medium, pressure ratio 1.6e8) with the
high-order GLM-MHD scheme and first_order_flux_correct on."""
import sys, time, torch
sys.path.insert(0, ".")
from athenapk_amd import decks, driver
ov = ["parthenon/mesh/nx%d=64" % d for d in (1, 2, 3)] + ["parthenon/meshblock/nx%d=16" % d for d in (1, 2, 3)] + ["parthenon/mesh/numlevel=4",
      "hydro/fluid=glmmhd", "hydro/riemann=hlld", "hydro/reconstruction=ppm", "parthenon/mesh/nghost=4", "hydro/first_order_flux_correct=true"]
s = driver.Simulation(decks.load("blast_3d_amr"), ov).initialize()
n = 0
t0 = time.perf_counter()
try:
    while n < 300 and s.time < 0.1:
        s.step(); n += 1
    torch.cuda.synchronize()
    i = s.refresh_info()
    print("ok cycles", n, "t=%.3e" % s.time, "blocks", i.nblocks_total, "fofc cells", s.fofc_count, "zone-cycles/s %.3e" % (s.amr_stats()[3] / (time.perf_counter() - t0)), "mass %.15f" % s.history()[0])
except Exception as e:
    print("FAILED at cycle", n, str(e)[:80])
