import sys, time, torch
sys.path.insert(0, ".")
from athenapk_amd import decks, driver
base = ["parthenon/mesh/nx%d=64" % d for d in (1, 2, 3)] + ["parthenon/meshblock/nx%d=16" % d for d in (1, 2, 3)] + ["parthenon/mesh/numlevel=4",
      "hydro/fluid=glmmhd", "hydro/riemann=hlld", "hydro/reconstruction=ppm", "parthenon/mesh/nghost=4", "parthenon/time/tlim=1.0"]
for fofc in ("true", "false"):
    s = driver.Simulation(decks.load("blast_3d_amr"), base + ["hydro/first_order_flux_correct=" + fofc]).initialize()
    n = 0
    try:
        while n < 400:
            s.step(); n += 1
            if n % 50 == 0:
                print(fofc, "cycle", n, "t=%.3e" % s.time, "blocks", s.refresh_info().nblocks_total, "fofc cells", s.fofc_count, "fallback stages", s.fofc_fallback_stages, flush=True)
    except Exception as e:
        print(fofc, "FAILED at cycle", n, str(e)[:100])
    s.close()
