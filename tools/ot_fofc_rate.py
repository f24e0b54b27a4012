"""Development aid: Orszag-Tang 512^2 (BASELINE config 3 as decked: first_order_flux_correct on) cycle rate."""
import sys, time, torch
sys.path.insert(0, ".")
from athenapk_amd import decks, driver
for label, extra, fused in (("fofc on, optimistic fused", ["hydro/first_order_flux_correct=true"], True),
                            ("fofc on, flux-array path", ["hydro/first_order_flux_correct=true"], False),
                            ("fofc off, fused", ["hydro/first_order_flux_correct=false"], True)):
    s = driver.Simulation(decks.load("orszag_tang"), extra)
    s.set_fused(fused)
    s.initialize()
    for _ in range(5):
        s.step()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(100):
        s.step()
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    i = s.refresh_info()
    print(label, "zone-cycles/s %.3e" % (i.zones_total * 100 / dt), "fallbacks", s.fofc_fallback_stages, flush=True)
