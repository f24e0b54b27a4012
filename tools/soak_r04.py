import sys, time, numpy as np, torch
sys.path.insert(0, ".")
from athenapk_amd import decks, driver
def run(deck, ov, n, label):
    s = driver.Simulation(decks.load(deck), ov, strict=False).initialize()
    t = time.perf_counter()
    for _ in range(n): s.step()
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    u = s.gather(); h = s.history()
    print(label, "cycles", n, "ms/cycle %.3f" % (dt / n * 1e3), "finite", bool(np.isfinite(u).all()), "rho min %.4g" % u[0].min(), "mass %.15g" % h[0], flush=True)
    s.close()
b = lambda n, m: ["parthenon/mesh/nx%d=%d" % (d, n) for d in (1, 2, 3)] + ["parthenon/meshblock/nx%d=%d" % (d, m) for d in (1, 2, 3)]
run("sod", b(256, 128) + ["parthenon/time/integrator=rk2"], 400, "hydro rk2 plm hllc sod 256^3")
run("synthetic_mhd", b(256, 128) + ["parthenon/time/integrator=rk3", "hydro/reconstruction=wenoz"], 150, "mhd rk3 wenoz 256^3")
run("synthetic_mhd", b(256, 128) + ["apk_amd/rehearse_remote_faces=true"], 300, "mhd vl2 ppm rehearsed remote faces 256^3")
run("synthetic_mhd", b(256, 128) + ["parthenon/time/integrator=rk3", "hydro/reconstruction=ppm", "apk_amd/rehearse_remote_faces=true"], 100, "mhd rk3 ppm rehearsed remote faces 256^3")
