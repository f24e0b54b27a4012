"""Development aid: does the order / instance of simulations in ONE process change the rate?  N = 1 and rehearsal alternating."""
import os, sys, time, torch
sys.path.insert(0, ".")
from athenapk_amd import decks, driver
import bench
deck, fluid, integrator, recon, riemann, brick, mb, desc = bench.WORKLOADS["mhd_ppm_hlld_vl2_256"]
ov = ["parthenon/mesh/nx%d=%d" % (d + 1, brick) for d in range(3)] + ["parthenon/meshblock/nx%d=%d" % (d + 1, mb) for d in range(3)]
ov += ["parthenon/time/integrator=%s" % integrator, "hydro/reconstruction=%s" % recon, "hydro/riemann=%s" % riemann]
def run(tag, extra, keep=None):
    s = driver.Simulation(decks.load(deck), ov + extra, strict=False).initialize()
    for _ in range(3): s.step()
    cyc, reg, med = bench.timed_regions(s.step, torch.cuda.synchronize, probe_cycles=3)
    print("%-12s ms/cycle %s" % (tag, " ".join("%.3f" % (r / cyc * 1e3) for r in reg)), flush=True)
    if keep is None: s.close()
    return s
R = ["apk_amd/rehearse_remote_faces=true"]
for rep in range(3):
    run("n1", [])
    run("rehearsal", R)
a = run("n1 kept", [], keep=True)
run("n1 second", [])
run("rehearsal", R)
a.close()
