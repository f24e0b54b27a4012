import sys
sys.path.insert(0, ".")
import numpy as np
from athenapk_amd import decks, driver
for strict in (False, True):
    for fused in (True, False):
        s = driver.Simulation(decks.load("orszag_tang"), [], strict=strict)
        s.set_fused(fused)
        s.initialize()
        n = 0
        try:
            while s.time < s.tlim and n < 5000:
                s.step(); n += 1
            print("strict", strict, "fused", fused, "ok cycles", n, "t", s.time, "fofc", s.fofc_count, "fallbacks", s.fofc_fallback_stages, "mass %.15f" % s.history()[0], flush=True)
        except Exception as e:
            print("strict", strict, "fused", fused, "FAILED at cycle", n, "t", s.time, "fofc", s.fofc_count, "fallbacks", s.fofc_fallback_stages, str(e)[:60], flush=True)
        s.close()
