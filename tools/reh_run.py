"""Development aid (GPU box, repo root): the headline brick for a few cycles -- plain (REHEARSE=0) or as the one-GPU
rehearsal of an 8-GPU rank -- printing ms per cycle; REPS timed regions of CYCLES cycles.  OVERLAP=0: exchanges
synchronous; APK_X1_DIRECT=0 / other library switches through the environment."""
import os, sys, time, torch
sys.path.insert(0, ".")
from athenapk_amd import decks, driver
import bench
wl = os.environ.get("WORKLOAD", "mhd_ppm_hlld_vl2_256")
deck, fluid, integrator, recon, riemann, brick, mb, desc = bench.WORKLOADS[wl]
ov = ["parthenon/mesh/nx%d=%d" % (d + 1, brick) for d in range(3)] + ["parthenon/meshblock/nx%d=%d" % (d + 1, mb) for d in range(3)]
ov += ["parthenon/time/integrator=%s" % integrator, "hydro/reconstruction=%s" % recon, "hydro/riemann=%s" % riemann]
if os.environ.get("REHEARSE", "1") == "1": ov += ["apk_amd/rehearse_remote_faces=true"]
s = driver.Simulation(decks.load(deck), ov, strict=False)
s.set_overlap(os.environ.get("OVERLAP", "1") == "1")
s.initialize()
for _ in range(3): s.step()
out = []
cycles = int(os.environ.get("CYCLES", "20"))
for rep in range(int(os.environ.get("REPS", "3"))):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(cycles): s.step()
    torch.cuda.synchronize(); out.append((time.perf_counter() - t) / cycles * 1e3)
print("%s rehearse=%s overlap=%s x1=%s: ms/cycle %s  x1_direct_exchanges %d thin %d" % (
    wl, os.environ.get("REHEARSE", "1"), os.environ.get("OVERLAP", "1"), os.environ.get("APK_X1_DIRECT", "1"),
    " ".join("%.3f" % v for v in out), s.x1_direct_exchanges(), s.thin_exchanges()), flush=True)
