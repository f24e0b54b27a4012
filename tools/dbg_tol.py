import os, sys, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from oracle import oracle
from athenapk_amd import decks, driver
G = 1.666666666666667
s = driver.Simulation(decks.load("linear_wave3d"), [], strict=False).initialize()
o = oracle.Sim(fluid="euler", recon="plm", riemann="hlle", integrator="rk2", nx=(64, 32, 32), ng=2, xmax=(3.0, 1.5, 1.5), cfl=0.3, gamma=G, nthreads=os.cpu_count())
o.pgen("linear_wave", wave_flag=0, amp=1e-6)
s.run(); o.run(o.period)
rms, l1, _ = s.linear_wave_errors(); rms_o, l1_o, _ = o.linear_wave_errors()
print("lw rms", rms, rms_o, abs(rms - rms_o), abs(rms - rms_o) / rms_o, "l1 max abs", np.abs(l1 - l1_o).max(), "state", np.abs(s.gather("cons") - o.gather_cons()).max())
ov = ["parthenon/mesh/nx1=64", "parthenon/mesh/nx2=64", "parthenon/meshblock/nx1=64", "parthenon/meshblock/nx2=64", "parthenon/time/tlim=0.1"]
s = driver.Simulation(decks.load("orszag_tang"), ov, strict=False).initialize()
o = oracle.Sim(fluid="glmmhd", recon="ppm", riemann="hlld", integrator="vl2", nx=(64, 64, 1), ng=3, xmin=(-0.5, -0.5, -0.5), xmax=(0.5, 0.5, 0.5), cfl=0.4, gamma=G).pgen("orszag_tang")
n = s.run(); o.run(0.1)
u, uo = s.gather(), o.gather_cons()
print("OT cycles", n, "max rel", np.max(np.abs(u - uo)) / np.max(np.abs(uo)))
