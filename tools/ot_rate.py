"""Development aid: cycle rate of the Orszag-Tang problem (BASELINE config 3), 2-D 512^2 and thin-z 512 x 512 x 4."""
import sys, time, torch
sys.path.insert(0, ".")
from athenapk_amd import decks, driver
for name, ov in (("2-D 512^2", []), ("512x512x4", ["parthenon/mesh/nx3=4", "parthenon/meshblock/nx3=4"]),
                 ("512x512x4 in 128x128x4", ["parthenon/mesh/nx3=4", "parthenon/meshblock/nx3=4", "parthenon/meshblock/nx1=128", "parthenon/meshblock/nx2=128"])):
    s = driver.Simulation(decks.load("orszag_tang"), ov + ["hydro/first_order_flux_correct=false"] + sys.argv[1:]).initialize()
    for _ in range(3): s.step()
    torch.cuda.synchronize(); t = time.perf_counter(); n = 0
    while n < 40: s.step(); n += 1
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    i = s.refresh_info()
    print("%-24s blocks %3d  ms/cycle %.3f  cell-updates/s %.3e" % (name, i.nblocks_total, dt / n * 1e3, i.zones_total * n / dt), flush=True)
