"""Development aid: what the exchanges of the stage loop move on the refined MHD blast of BASELINE config 5's shape -- copy
regions by (source, destination) kind and the restriction / prolongation operators, in cells."""
import sys, collections
sys.path.insert(0, ".")
from athenapk_amd import decks, driver
ov = ["parthenon/mesh/nx%d=64" % d for d in (1, 2, 3)] + ["parthenon/meshblock/nx%d=16" % d for d in (1, 2, 3)] + [
    "parthenon/mesh/numlevel=4", "hydro/fluid=glmmhd", "hydro/riemann=hlld", "hydro/reconstruction=ppm", "parthenon/mesh/nghost=4",
    "problem/blast/pressure_ambient=1.0", "problem/blast/pressure_ratio=100"]
s = driver.Simulation(decks.load("blast_3d_amr"), ov).initialize()
for _ in range(3):
    s.step()
n = s.refresh_info().nblocks_total
print("blocks", n, "levels", dict(collections.Counter(s.block_level(lb) for lb in range(n))))
KN = {0: "block", 1: "send", 2: "recv", 3: "coarse"}
for ph in ("my_fill_faces", "my_fill_direct", "my_fill_shell"):
    cnt, cells = collections.Counter(), collections.Counter()
    for r in s.regions(ph):
        key = (KN.get(r.src_kind, r.src_kind), KN.get(r.dst_kind, r.dst_kind))
        cnt[key] += 1
        cells[key] += r.ext[0] * r.ext[1] * r.ext[2]
    print(ph, {k: (cnt[k], cells[k]) for k in cnt}, "total cells", sum(cells.values()))
for w in ("my_restrict_own", "my_prolongate_faces", "my_prolongate_shell", "my_prolongate"):
    ops = s.amr_ops(w)
    cells = sum((o.hi[0] - o.lo[0] + 1) * (o.hi[1] - o.lo[1] + 1) * (o.hi[2] - o.lo[2] + 1) for o in ops)
    print(w, len(ops), "ops", cells, "cells")
