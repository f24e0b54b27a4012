"""Runs the oracle on the configuration of the reference's turbulence regression test
(tst/regression/test_suites/turbulence/turbulence.py:44-52: 64^3 GLM-MHD PLM+HLLE VL2 driven to
t = 5, final sonic Mach in (0.45, 0.50) and Alfvenic Mach in (12.8, 13.6)) and records the result
in turbulence_pin.json.  ~5 CPU-minutes on 8 cores.

  python tests/golden/make_turbulence_pin.py
"""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402


def deck_modes(path):
    kv = {}
    block = None
    for line in open(path):
        line = line.split("#")[0].strip()
        if line.startswith("<"):
            block = line.strip("<>")
        elif "=" in line and block == "modes":
            k, v = [x.strip() for x in line.split("=")]
            kv[k] = int(v)
    n = len(kv) // 3
    return np.array([[kv["k_%d_%d" % (m + 1, d)] for m in range(n)] for d in range(3)], dtype=np.float64)


def main():
    k_vec = deck_modes(os.path.join(ROOT, "inputs", "turbulence.in"))
    t0 = time.time()
    s = O.Sim(fluid="glmmhd", recon="plm", riemann="hlle", integrator="vl2", nx=(64, 64, 64), mb=(32, 32, 32),
              ng=2, cfl=0.3, gamma=1.0001, nthreads=os.cpu_count())
    s.pgen("turbulence", k_vec=k_vec)
    n = s.run(5.0)
    ms, ma, pb = s.turb_history()
    out = {"cycles": n, "Ms": ms, "Ma": ma, "plasma_beta": pb, "reference_window_Ms": [0.45, 0.50],
           "reference_window_Ma": [12.8, 13.6], "history": list(s.history()), "wall_s": round(time.time() - t0, 1)}
    print(out)
    with open(os.path.join(HERE, "turbulence_pin.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
