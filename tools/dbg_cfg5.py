"""Debug aid: BASELINE config 5 as decked (MHD PPM+HLLD on the adaptive blast); first cycle at which a field component
appears (B must stay exactly zero), per library / switch.   python tools/dbg_cfg5.py [ncycles]"""
import os
import sys

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from athenapk_amd import decks, driver  # noqa: E402

CFG5 = ["parthenon/mesh/nx%d=64" % d for d in (1, 2, 3)] + ["parthenon/meshblock/nx%d=16" % d for d in (1, 2, 3)] + [
    "parthenon/mesh/numlevel=4", "hydro/fluid=glmmhd", "hydro/riemann=hlld", "hydro/reconstruction=ppm",
    "parthenon/mesh/nghost=4", "parthenon/time/tlim=1.0"]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
s = driver.Simulation(decks.load("blast_3d_amr"), CFG5, strict=False).initialize()
first = None
from amr_emulator import placement


def symmetric():
    locs = {(p[0], tuple(p[1])) for p in placement(s)}
    for lev, loc in list(locs):
        n1 = 4 * 2 ** lev
        if (lev, (n1 - 1 - loc[0], loc[1], loc[2])) not in locs:
            return False
        if (lev, (loc[1], loc[0], loc[2])) not in locs or (lev, (loc[0], loc[2], loc[1])) not in locs:
            return False
    return True


if os.environ.get("DBG_SYMMETRY"):
    broke = []
    for c in range(n):
        s.step()
        if not symmetric():
            broke.append(c)
    print("lib=%s NO_LEAN=%s cycles after which the forest is not octant-symmetric: %r (nblocks %d at the end)" % (
        os.environ.get("APK_LIB_PATH", "default"), os.environ.get("APK_NO_LEAN"), broke, s.refresh_info().nblocks_total))
    sys.exit(0)
for c in range(n):
    s.step()
    i = s.refresh_info()
    g = i.ng
    mx = 0.0
    where = None
    for lb in range(i.nblocks_local):
        u = s.read_block(lb, "cons")[:, g:-g, g:-g, g:-g]
        m = np.abs(u[5:9]).max()
        if not np.isfinite(m) or m > mx:
            mx = m
            where = (lb, np.unravel_index(np.argmax(np.abs(u[5:9])), u[5:9].shape))
    if mx != 0.0 and first is None:
        first = c
        print("cycle %d: max |B, psi| = %r at %r, nblocks %d" % (c, mx, where, i.nblocks_total))
        break
print("lib=%s NO_LEAN=%s first nonzero cycle: %r" % (os.environ.get("APK_LIB_PATH", "default"), os.environ.get("APK_NO_LEAN"), first))
