import sys
sys.path.insert(0, ".")
from athenapk_amd import decks, driver
for cfl in ("0.4", "0.399", "0.401", "0.395", "0.39"):
    for strict in (False, True):
        for fused in (True, False):
            s = driver.Simulation(decks.load("orszag_tang"), ["parthenon/time/cfl=" + cfl], strict=strict)
            s.set_fused(fused)
            s.initialize()
            n = 0
            try:
                while s.time < s.tlim and n < 6000:
                    s.step(); n += 1
                print("cfl", cfl, "strict", strict, "fused", fused, "ok", n, "fofc", s.fofc_count, "fallbacks", s.fofc_fallback_stages, flush=True)
            except Exception as e:
                print("cfl", cfl, "strict", strict, "fused", fused, "FAILED at", n + 1, "t %.4f" % s.time, "fofc", s.fofc_count, "fallbacks", s.fofc_fallback_stages, flush=True)
            s.close()
