"""Development aid: longer runs of the paths of round 6 in the product build -- finite states and conserved mass after
hundreds of cycles (x1 strips in the exchange buffers on the one-GPU rehearsal of an 8-GPU rank, VL2 and RK3; the prim-free
cycle of refined meshes with regridding, MHD and hydro; forced turbulence on the RK3 rehearsal)."""
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
from athenapk_amd import decks, driver


def run(deck, ov, n, label, refined=False):
    s = driver.Simulation(decks.load(deck), ov, strict=False).initialize()
    m0 = s.history()[0]
    t = time.perf_counter()
    for _ in range(n):
        s.step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    h = s.history()
    if refined:
        nb = s.refresh_info().nblocks_total
        fin = all(bool(np.isfinite(s.read_block(lb)).all()) for lb in range(0, nb, 5))
        extra = "blocks %d passes skipped %d" % (nb, s.amr_c2p_passes_skipped())
    else:
        fin = bool(np.isfinite(s.gather()).all())
        extra = "x1 direct %d" % s.x1_direct_exchanges()
    print(label, "cycles", n, "ms/cycle %.3f" % (dt / n * 1e3), "finite", fin, "mass drift %.2e" % (abs(h[0] - m0) / abs(m0)), extra, flush=True)
    assert fin
    s.close()


b = lambda n, m: ["parthenon/mesh/nx%d=%d" % (d, n) for d in (1, 2, 3)] + ["parthenon/meshblock/nx%d=%d" % (d, m) for d in (1, 2, 3)]
run("synthetic_mhd", b(256, 128) + ["apk_amd/rehearse_remote_faces=true"], 400, "mhd vl2 ppm, rehearsed remote faces 256^3")
run("synthetic_mhd", b(256, 128) + ["parthenon/time/integrator=rk3", "hydro/reconstruction=wenoz", "apk_amd/rehearse_remote_faces=true"], 120,
    "mhd rk3 wenoz, rehearsed remote faces 256^3")
amr = ["parthenon/mesh/nx%d=64" % d for d in (1, 2, 3)] + ["parthenon/meshblock/nx%d=16" % d for d in (1, 2, 3)] + ["parthenon/mesh/numlevel=4", "parthenon/time/tlim=10.0"]
run("blast_3d_amr", amr + ["hydro/fluid=glmmhd", "hydro/riemann=hlld", "hydro/reconstruction=ppm", "parthenon/mesh/nghost=4",
                           "problem/blast/pressure_ambient=1.0", "problem/blast/pressure_ratio=100"], 1500, "refined mhd blast (config 5's mesh)", refined=True)
run("blast_3d_amr", amr, 1500, "refined hydro blast as decked", refined=True)

# forced turbulence (config 4's scheme) and Orszag-Tang with first-order flux correction (config 3's deck): mass to round-off
b3 = lambda n, m: ["parthenon/mesh/nx%d=%d" % (d, n) for d in (1, 2, 3)] + ["parthenon/meshblock/nx%d=%d" % (d, m) for d in (1, 2, 3)]
run("turbulence", b3(128, 64) + ["parthenon/time/tlim=100.0", "parthenon/mesh/nghost=3", "hydro/reconstruction=wenoz", "hydro/riemann=hlld", "parthenon/time/integrator=rk3"], 300,
    "forced turbulence mhd rk3 wenoz 128^3")
run("orszag_tang", ["parthenon/mesh/nx1=512", "parthenon/mesh/nx2=512", "parthenon/mesh/nx3=4", "parthenon/meshblock/nx1=128", "parthenon/meshblock/nx2=128",
                    "parthenon/meshblock/nx3=4", "parthenon/time/tlim=100.0", "hydro/first_order_flux_correct=true"], 4000,
    "orszag-tang 512 x 512 x 4 vl2 with first-order flux correction (past t = 0.94)")

# a uniform-mesh blast (flat ambient state: ties in PPM's extremum tests everywhere) in blocks that meet through the face
# table: every face between two blocks is solved once by each of them -- mass to round-off
for integ, recon in (("vl2", "ppm"), ("rk3", "wenoz"), ("rk2", "ppm")):
    run("blast", b3(128, 64) + ["parthenon/mesh/nghost=3", "hydro/fluid=glmmhd", "hydro/riemann=hlld", "hydro/reconstruction=%s" % recon,
                                "parthenon/time/integrator=%s" % integ, "parthenon/time/tlim=10.0", "problem/blast/pressure_ambient=1.0",
                                "problem/blast/pressure_ratio=100"] + ["parthenon/mesh/ix%d_bc=periodic" % d for d in (1, 2, 3)] +
        ["parthenon/mesh/ox%d_bc=periodic" % d for d in (1, 2, 3)], 400, "uniform mhd blast 128^3 in 64^3 blocks, %s %s" % (integ, recon))
run("blast", b3(128, 64) + ["parthenon/time/integrator=rk2", "parthenon/time/tlim=10.0"] + ["parthenon/mesh/ix%d_bc=periodic" % d for d in (1, 2, 3)] +
    ["parthenon/mesh/ox%d_bc=periodic" % d for d in (1, 2, 3)], 400, "uniform hydro blast 128^3 in 64^3 blocks as decked, rk2")
# ... and the same with its outer faces exchanged like those between the bricks of an 8-GPU run (split stages: windows and
# slabs solve the faces between them twice, in different launches)
run("blast", b3(128, 64) + ["parthenon/mesh/nghost=3", "hydro/fluid=glmmhd", "hydro/riemann=hlld", "hydro/reconstruction=ppm",
                            "parthenon/time/integrator=vl2", "parthenon/time/tlim=10.0", "problem/blast/pressure_ambient=1.0",
                            "problem/blast/pressure_ratio=100", "apk_amd/rehearse_remote_faces=true"] + ["parthenon/mesh/ix%d_bc=periodic" % d for d in (1, 2, 3)] +
    ["parthenon/mesh/ox%d_bc=periodic" % d for d in (1, 2, 3)], 400, "uniform mhd blast 128^3 in 64^3 blocks, vl2 ppm, rehearsed remote faces")
