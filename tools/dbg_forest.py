import sys, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import helpers as H
from oracle import oracle
from athenapk_amd import decks, driver
from test_amr_mesh import SMR3, _bc, _forest_oracle
fluid, riemann, recon, ng, integrator = sys.argv[1:6]; ng = int(ng)
ov = SMR3 + _bc("periodic") + ["hydro/fluid=%s" % fluid, "hydro/riemann=%s" % riemann, "hydro/reconstruction=%s" % recon,
   "parthenon/mesh/nghost=%d" % ng, "parthenon/time/integrator=%s" % integrator,
   "problem/blast/radius_outer=0.2", "problem/blast/pressure_ratio=100", "problem/blast/x3_0=0.1",
   "problem/blast/x1_0=0.013", "problem/blast/x2_0=-0.021", "problem/blast/radius_inner=0.1", "problem/blast/pressure_ambient=1.0"]
s = driver.Simulation(decks.load("blast"), ov, strict=True); s.set_fused(False); s.initialize()
nb = s.info.nblocks_total
fo = _forest_oracle(s, oracle, fluid, recon, riemann, integrator)
fo.initialize([s.read_block(lb) for lb in range(nb)])
print("dt0", fo.dt, s.dt)
for lb in range(nb):
    a, b = s.read_block(lb), fo.cons[lb]
    if not np.array_equal(a, b): print("init cons differs block", lb, fo.leaves[lb], np.abs(a-b).max())
    a, b = s.read_block(lb, "prim"), fo.prim[lb]
    if not np.array_equal(a, b, equal_nan=True): print("init prim differs block", lb, fo.leaves[lb], np.nanmax(np.abs(a-b)))
s.step(); fo.step()
print("dt1", fo.dt, s.dt, "c_h", fo.c_h, s.c_h)
nbad = 0
for lb in range(nb):
    a, b = s.read_block(lb), fo.cons[lb]
    if not np.array_equal(a, b):
        d = np.abs(a - b)
        idx = np.argwhere(d > 0)
        inter = d[:, ng:-ng, ng:-ng, ng:-ng]
        nbad += 1
        if nbad < 12: print("block", lb, fo.leaves[lb], "max", d.max(), "interior max", inter.max(), "ncells", len(idx), "first", idx[0], "last", idx[-1])
print("blocks differing", nbad, "of", nb)
