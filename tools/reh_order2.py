import os, sys, time, torch
sys.path.insert(0, ".")
from athenapk_amd import decks, driver
import bench
deck, fluid, integrator, recon, riemann, brick, mb, desc = bench.WORKLOADS["mhd_ppm_hlld_vl2_256"]
ov = ["parthenon/mesh/nx%d=%d" % (d + 1, brick) for d in range(3)] + ["parthenon/meshblock/nx%d=%d" % (d + 1, mb) for d in range(3)]
ov += ["parthenon/time/integrator=%s" % integrator, "hydro/reconstruction=%s" % recon, "hydro/riemann=%s" % riemann]
def run(tag, extra, n=0):
    s = driver.Simulation(decks.load(deck), ov + extra, strict=False).initialize()
    for _ in range(3): s.step()
    cyc, reg, med = bench.timed_regions(s.step, torch.cuda.synchronize, probe_cycles=3)
    print("%-12s ms/cycle %s  mem alloc %.2f GB reserved %.2f GB" % (tag, " ".join("%.3f" % (r / cyc * 1e3) for r in reg), torch.cuda.memory_allocated() / 1e9, torch.cuda.memory_reserved() / 1e9), flush=True)
    for _ in range(n): s.step()
    torch.cuda.synchronize()
    if n:
        cyc, reg, med = bench.timed_regions(s.step, torch.cuda.synchronize, probe_cycles=3)
        print("%-12s after %d more cycles: ms/cycle %s" % (tag, n, " ".join("%.3f" % (r / cyc * 1e3) for r in reg)), flush=True)
    s.close()
run("n1", [], 500)
print(bench.general_stage_bench("ppm", "hlld")["ms_per_stage"], "general stage; reserved %.2f GB" % (torch.cuda.memory_reserved() / 1e9), flush=True)
run("n1 after", [])
run("rehearsal", ["apk_amd/rehearse_remote_faces=true"])
torch.cuda.empty_cache()
run("n1 emptied", [])
run("rehearsal", ["apk_amd/rehearse_remote_faces=true"])
