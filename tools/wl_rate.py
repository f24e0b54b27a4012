"""Development aid (GPU box, repo root): one bench workload, three regions of >= 50 ms, REPS times; per-kernel HIP-event averages.
   python tools/wl_rate.py hydro_plm_hllc_rk2_256 [override ...]"""
import os, sys, torch
sys.path.insert(0, ".")
from athenapk_amd import decks, driver
import bench
wl = sys.argv[1]
if wl in bench.EXTRA_WORKLOADS:
    deck, fluid, integrator, recon, riemann, mesh, mbs, desc, extra = bench.EXTRA_WORKLOADS[wl]
else:
    deck, fluid, integrator, recon, riemann, brick, mb, desc = bench.WORKLOADS[wl]
    mesh, mbs, extra = (brick,) * 3, (mb,) * 3, []
ov = ["parthenon/mesh/nx%d=%d" % (d + 1, mesh[d]) for d in range(3)] + ["parthenon/meshblock/nx%d=%d" % (d + 1, mbs[d]) for d in range(3)]
ov += ["parthenon/time/integrator=%s" % integrator, "hydro/reconstruction=%s" % recon, "hydro/riemann=%s" % riemann] + extra + sys.argv[2:]
for rep in range(int(os.environ.get("REPS", "2"))):
    s = driver.Simulation(decks.load(deck), ov, strict=False).initialize()
    for _ in range(3): s.step()
    cyc, reg, med = bench.timed_regions(s.step, torch.cuda.synchronize, probe_cycles=3)
    s.kernel_timing(True); s.read_kernel_timing()
    for _ in range(8): s.step()
    torch.cuda.synchronize()
    t = s.read_kernel_timing()
    per = {k: (v[0] / v[1], v[1] / 8.0) for k, v in t.items() if v[1]}
    print("%s %s: ms/cycle %s = %.4g cell-updates/s | %s" % (wl, " ".join(sys.argv[2:]), " ".join("%.3f" % (r / cyc * 1e3) for r in reg), s.info.zones_total * cyc / reg[med],
                                                          " ".join("%s %.3fx%.1f" % (k, v[0], v[1]) for k, v in sorted(per.items()) if v[0] * v[1] > 0.01)), flush=True)
    s.close()
