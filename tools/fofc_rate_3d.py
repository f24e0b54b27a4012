"""Development aid: 3-D MHD PPM+HLLD VL2 256^3 with first_order_flux_correct on: optimistic fused vs flux-array."""
import sys, time, torch
sys.path.insert(0, ".")
from athenapk_amd import decks, driver
ov = ["parthenon/mesh/nx1=256", "parthenon/mesh/nx2=256", "parthenon/mesh/nx3=256", "parthenon/meshblock/nx1=128",
      "parthenon/meshblock/nx2=128", "parthenon/meshblock/nx3=128"]
for label, extra, fused in (("fofc on, optimistic fused", ["hydro/first_order_flux_correct=true"], True),
                            ("fofc on, flux-array path", ["hydro/first_order_flux_correct=true"], False),
                            ("fofc off, fused", ["hydro/first_order_flux_correct=false"], True)):
    s = driver.Simulation(decks.load("synthetic_mhd"), ov + extra)
    s.set_fused(fused)
    s.initialize()
    for _ in range(3):
        s.step()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10):
        s.step()
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    i = s.refresh_info()
    print(label, "zone-cycles/s %.3e" % (i.zones_total * 10 / dt), "fallbacks", s.fofc_fallback_stages, flush=True)
    s.close()
