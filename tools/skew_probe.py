"""Development aid (GPU box): the headline brick at N = 1 with the fields' base addresses skewed against each other
(APK_ALLOC_SKEW, driver.py) -- do the relative offsets of the arrays a kernel streams together matter?"""
import os, sys, time, torch
sys.path.insert(0, ".")
from athenapk_amd import decks, driver
import bench
wl = os.environ.get("WORKLOAD", "mhd_ppm_hlld_vl2_256")
deck, fluid, integrator, recon, riemann, brick, mb, desc = bench.WORKLOADS[wl]
ov = ["parthenon/mesh/nx%d=%d" % (d + 1, brick) for d in range(3)] + ["parthenon/meshblock/nx%d=%d" % (d + 1, mb) for d in range(3)]
ov += ["parthenon/time/integrator=%s" % integrator, "hydro/reconstruction=%s" % recon, "hydro/riemann=%s" % riemann]
def run(tag, skew):
    os.environ["APK_ALLOC_SKEW"] = skew
    s = driver.Simulation(decks.load(deck), ov, strict=False).initialize()
    for _ in range(3): s.step()
    cyc, reg, med = bench.timed_regions(s.step, torch.cuda.synchronize, probe_cycles=3)
    s.kernel_timing(True); s.read_kernel_timing()
    for _ in range(6): s.step()
    torch.cuda.synchronize()
    t = s.read_kernel_timing()
    per = {k: (v[0] / v[1] if v[1] else 0.0) for k, v in t.items()}
    print("%-34s ms/cycle %s | dc %.3f x3 %.3f m12f %.3f" % (tag, " ".join("%.3f" % (r / cyc * 1e3) for r in reg), per["fused_dc_x1"], per["fused_x3"], per["fused_x1"]), flush=True)
    s.close()
K = 1024
def pat(step): return "cons:%d,prim:%d,u1:%d,prim2:%d" % (0, step, 2 * step, 3 * step)
for rep in range(2):
    run("none (padded allocs)", "cons:0")
    for name, step in (("256 B", 256), ("4 KB", 4 * K), ("68 KB", 68 * K), ("1 MB + 4 KB", K * K + 4 * K), ("2 MB", 2 * K * K), ("6 MB", 6 * K * K)):
        run("steps of " + name, pat(step))
