"""Development aid: how far the product build (FMA contraction, reciprocal-based divides / roots) is from
the bit-parity build on the driver level."""
import sys
import numpy as np
sys.path.insert(0, ".")
from athenapk_amd import decks, driver

def run(deck, ov, strict, ncyc=None):
    s = driver.Simulation(decks.load(deck), ov, strict=strict).initialize()
    if ncyc is None:
        s.run()
    else:
        for _ in range(ncyc):
            s.step()
    return s

ot = ["parthenon/mesh/nx1=64", "parthenon/mesh/nx2=64", "parthenon/meshblock/nx1=32", "parthenon/meshblock/nx2=32"]
a, b = run("orszag_tang", ot, False, 5).gather("cons"), run("orszag_tang", ot, True, 5).gather("cons")
print("Orszag-Tang 64^2 PPM+HLLD VL2, 5 cycles: max |fast - strict| / max|strict| per variable",
      ["%.1e" % (np.abs(a[n] - b[n]).max() / (np.abs(b[n]).max() + 1e-300)) for n in range(9)])
a, b = run("orszag_tang", ot, False, 200).gather("cons"), run("orszag_tang", ot, True, 200).gather("cons")
print("  200 cycles:", ["%.1e" % (np.abs(a[n] - b[n]).max() / (np.abs(b[n]).max() + 1e-300)) for n in range(9)])
lw = ["parthenon/meshblock/nx1=64", "parthenon/meshblock/nx2=32", "parthenon/meshblock/nx3=32", "parthenon/time/integrator=rk2"]
ea, eb = run("linear_wave3d", lw, False).linear_wave_errors(), run("linear_wave3d", lw, True).linear_wave_errors()
print("linear wave 64x32x32 one period: RMS L1 error fast %.16e strict %.16e diff %.1e" % (ea[0], eb[0], abs(ea[0] - eb[0])))
print("  per-variable L1 diff", ["%.1e" % abs(x - y) for x, y in zip(ea[1], eb[1])])
