import sys
sys.path.insert(0, ".")
import numpy as np
from athenapk_amd import decks, driver
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ncyc = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ov = ["parthenon/mesh/nx%d=%d" % (d, n) for d in (1, 2, 3)] + ["parthenon/meshblock/nx%d=%d" % (d, n // 2) for d in (1, 2, 3)]
out, meta = [], []
for strict in (False, True):
    s = driver.Simulation(decks.load("synthetic_mhd"), ov, strict=strict).initialize()
    m = [("init", repr(s.dt), repr(s.c_h))]
    for _ in range(ncyc):
        s.step()
        m.append((repr(s.time), repr(s.dt), repr(s.c_h)))
    out.append(s.gather())
    meta.append(m)
    s.close()
for a, b in zip(*meta):
    print("product", a, "| strict", b)
d = np.abs(out[0] - out[1])
v = int(np.argmax(d.reshape(9, -1).max(1)))
idx = np.unravel_index(np.argmax(d[v]), d[v].shape)
print("max diff %.3e in var %d at (k,j,i)=%s; count of cells with diff > 1e-13: %d of %d" % (d.max(), v, idx, int((d[v] > 1e-13).sum()), d[v].size))
big = np.argwhere(d[v] > 0.5 * d[v].max())
print("cells within 2x of the max:", big[:10].tolist(), "..." if len(big) > 10 else "")
