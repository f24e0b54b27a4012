"""max |product - parity| after a few cycles of the synthetic GLM-MHD benchmark state (A/B aid)"""
import sys
sys.path.insert(0, ".")
import numpy as np
from athenapk_amd import decks, driver
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ncyc = int(sys.argv[2]) if len(sys.argv) > 2 else 2
ov = ["parthenon/mesh/nx%d=%d" % (d, n) for d in (1, 2, 3)] + ["parthenon/meshblock/nx%d=%d" % (d, n // 2) for d in (1, 2, 3)]
out = []
for strict in (False, True):
    s = driver.Simulation(decks.load("synthetic_mhd"), ov, strict=strict).initialize()
    for _ in range(ncyc):
        s.step()
    out.append(s.gather())
    s.close()
d = np.abs(out[0] - out[1])
print("n=%d cycles=%d max abs diff %.3e (scale %.3f) per variable:" % (n, ncyc, d.max(), np.abs(out[1]).max()), ["%.1e" % d[v].max() for v in range(9)])
