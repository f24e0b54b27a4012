"""Development aid (GPU box, repo root): workloads with natural rows and with rows at a line-aligned pitch
(apk_amd/row_pitch = aligned: 144 instead of 134 doubles for 128-cell blocks, interior cells on 128-byte boundaries),
alternating in one process; ms per cycle (three regions) and the stage kernels' HIP-event averages."""
import os, sys, time, torch
sys.path.insert(0, ".")
from athenapk_amd import decks, driver
import bench
def run(wl, pitch, extra=()):
    deck, fluid, integrator, recon, riemann, brick, mb, desc = bench.WORKLOADS[wl]
    ov = ["parthenon/mesh/nx%d=%d" % (d + 1, brick) for d in range(3)] + ["parthenon/meshblock/nx%d=%d" % (d + 1, mb) for d in range(3)]
    ov += ["parthenon/time/integrator=%s" % integrator, "hydro/reconstruction=%s" % recon, "hydro/riemann=%s" % riemann,
           "apk_amd/row_pitch=%s" % pitch] + list(extra)
    s = driver.Simulation(decks.load(deck), ov, strict=False).initialize()
    for _ in range(3): s.step()
    cyc, reg, med = bench.timed_regions(s.step, torch.cuda.synchronize, probe_cycles=3)
    s.kernel_timing(True); s.read_kernel_timing()
    for _ in range(6): s.step()
    torch.cuda.synchronize()
    t = s.read_kernel_timing()
    per = {k: (v[0] / v[1] if v[1] else 0.0) for k, v in t.items() if v[1]}
    print("%-24s %-8s %s ms/cycle %s | %s" % (wl, pitch, "rehearsal" if extra else "", " ".join("%.3f" % (r / cyc * 1e3) for r in reg),
                                           " ".join("%s %.3f" % (k, v) for k, v in sorted(per.items()) if v > 0.02)), flush=True)
    s.close()
R = ["apk_amd/rehearse_remote_faces=true"]
for rep in range(2):
    for wl in ("mhd_ppm_hlld_vl2_256", "hydro_plm_hllc_rk2_256", "mhd_wenoz_hlld_rk3_256"):
        for pitch in ("natural", "aligned"):
            run(wl, pitch)
    for pitch in ("natural", "aligned"):
        run("mhd_ppm_hlld_vl2_256", pitch, R)
