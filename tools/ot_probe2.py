import sys
sys.path.insert(0, ".")
import numpy as np
from athenapk_amd import decks, driver
s = driver.Simulation(decks.load("orszag_tang"), [], strict=False)
s.initialize()
n = 0; last = (0, 0)
try:
    while s.time < s.tlim and n < 5000:
        s.step(); n += 1
        cur = (s.fofc_count, s.fofc_fallback_stages)
        if cur != last:
            print("cycle", n, "t %.6f" % s.time, "fofc", cur[0], "fallbacks", cur[1], flush=True); last = cur
    print("ok", n)
except Exception as e:
    print("FAILED at cycle", n + 1, "t", s.time, "fofc", s.fofc_count, "fallbacks", s.fofc_fallback_stages, str(e)[:50], flush=True)
    w = s.gather("prim"); u = s.gather("cons")
    print("min prim p", w[4].min(), "at", np.unravel_index(np.argmin(w[4]), w[4].shape), "min rho", w[0].min())
    ep = u[4] - 0.5 * (u[1] ** 2 + u[2] ** 2 + u[3] ** 2) / u[0] - 0.5 * (u[5] ** 2 + u[6] ** 2 + u[7] ** 2)
    print("min cons-derived p*", ep.min(), "at", np.unravel_index(np.argmin(ep), ep.shape))
