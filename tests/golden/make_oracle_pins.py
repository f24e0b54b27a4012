"""Runs the oracle on the two configurations whose results the reference's own regression
suite pins and records the outcome in oracle_pins.json.

  python tests/golden/make_oracle_pins.py            # ~15 s + ~8 min on 8 cores

The hydro bound 1.547584e-08 is the reference's own result printed with %e
(linear_wave.cpp prints rms_err with "%e"), so agreement to 7 significant digits pins the
whole pipeline: reconstruction, HLLE, flux divergence, VL2 coefficients, dt control, cell
centres and the error norm.
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

CASES = [
    dict(name="hydro_vl2_plm_hlle_128x64x64", fluid="euler", recon="plm", riemann="hlle",
         integrator="vl2", nx=(128, 64, 64), mb=(32, 32, 32), ng=2, bound=1.547584e-08),
    dict(name="glmmhd_rk3_wenoz_hlle_256x128x128", fluid="glmmhd", recon="wenoz", riemann="hlle",
         integrator="rk3", nx=(256, 128, 128), mb=(64, 64, 64), ng=3, bound=6.14e-12),
]


def main():
    out = {}
    for c in CASES:
        t0 = time.time()
        s = O.Sim(fluid=c["fluid"], recon=c["recon"], riemann=c["riemann"], integrator=c["integrator"],
                  nx=c["nx"], mb=c["mb"], ng=c["ng"], xmax=(3.0, 1.5, 1.5), cfl=0.3, nthreads=os.cpu_count())
        s.pgen("linear_wave", wave_flag=0, amp=1e-6)
        n = s.run(1.0 * s.period)
        rms, l1, mx = s.linear_wave_errors()
        out[c["name"]] = dict(cycles=n, rms_l1=rms, l1=list(l1), reference_bound=c["bound"],
                              printed_like_reference="%e" % rms, wall_s=round(time.time() - t0, 1))
        print(c["name"], out[c["name"]], flush=True)
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_pins.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
