"""Development aid: cycle rate of the adaptive blast problem (BASELINE config 5 shape) on one GPU."""
import sys, time, torch
sys.path.insert(0, ".")
from athenapk_amd import decks, driver
ov=["parthenon/mesh/nx1=64","parthenon/mesh/nx2=64","parthenon/mesh/nx3=64","parthenon/meshblock/nx1=16","parthenon/meshblock/nx2=16","parthenon/meshblock/nx3=16","parthenon/mesh/numlevel=4","parthenon/time/tlim=0.02"]
for extra in ([], ["hydro/fluid=glmmhd","hydro/riemann=hlld","hydro/reconstruction=ppm","parthenon/mesh/nghost=4", "problem/blast/pressure_ambient=1.0", "problem/blast/pressure_ratio=100"], ["parthenon/meshblock/nx1=8","parthenon/meshblock/nx2=8","parthenon/meshblock/nx3=8"]):
    s=driver.Simulation(decks.load("blast_3d_amr"), ov+extra).initialize()
    for _ in range(3): s.step()
    torch.cuda.synchronize(); z0=s.amr_stats()[3]; t=time.perf_counter(); n=0
    while n<40: s.step(); n+=1
    torch.cuda.synchronize(); dt=time.perf_counter()-t
    i=s.refresh_info(); print(extra[:1], "blocks",i.nblocks_total,"zone-cycles/s %.3e"%((s.amr_stats()[3]-z0)/dt), "ms/cycle %.2f"%(dt/n*1e3), flush=True)
