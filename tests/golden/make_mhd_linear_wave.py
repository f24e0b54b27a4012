"""Freezes the oracle's result of the MHD linear-wave problem (src/pgen/linear_wave_mhd.cpp:177-363) for five wave
families at 32 x 16 x 16 in two meshblocks, PPM + HLLD + Dedner, RK3, one wave period -- the configuration of
tests/test_gpu_driver.py::test_mhd_linear_wave_matches_oracle -- in mhd_linear_wave.json: cycle count, final time step,
RMS-L1 error, the L1 and maximum error norms of d, M1..3, E, B1..3 (doubles as hex strings: exact), and a checksum of
the final conserved state.  The GPU suite checks both HIP builds against these numbers without the live oracle
(strict: bit for bit; product: L1 norms within north_star's 1e-12).

  python tests/golden/make_mhd_linear_wave.py          # ~1 min
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

CASES = [("fast", 0, 0.0), ("alfven", 1, 0.0), ("slow", 2, 0.0), ("entropy", 3, 1.0), ("fast_plus", 6, 0.0)]
GAMMA_DECK = 1.666666666666667


def main():
    out = {"configuration": "inputs/linear_wave_mhd3d.in at 32 x 16 x 16 in 16^3 meshblocks, glmmhd ppm hlld rk3, cfl 0.3, amp 1e-6, one period",
           "cases": {}}
    for name, flag, vflow in CASES:
        o = O.Sim(fluid="glmmhd", recon="ppm", riemann="hlld", integrator="rk3", nx=(32, 16, 16), mb=(16, 16, 16), ng=3,
                  xmax=(3.0, 1.5, 1.5), cfl=0.3, gamma=GAMMA_DECK, nthreads=os.cpu_count())
        o.pgen("linear_wave_mhd", wave_flag=flag, amp=1e-6, vflow=vflow)
        n = o.run(o.period)
        rms, l1, mx = o.linear_wave_errors()
        u = np.ascontiguousarray(o.gather_cons())
        out["cases"][name] = dict(wave_flag=flag, vflow=vflow, cycles=int(n), dt=float(o.dt).hex(), rms_l1=float(rms).hex(),
                                  l1=[float(x).hex() for x in l1], max=[float(x).hex() for x in mx],
                                  rms_l1_printed="%e" % rms, cons_sha256=hashlib.sha256(u.tobytes()).hexdigest())
        print(name, n, "%e" % rms, flush=True)
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "mhd_linear_wave.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
