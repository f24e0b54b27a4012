"""Generates tests/golden/block_fixture.npz: seeded inputs and the oracle's outputs for one
GLM-MHD PPM+HLLD block and one hydro PLM+HLLC block (flux arrays, one full RK stage with Dedner
source, ConsToPrim, hyperbolic dt).  The fixture freezes today's oracle so that later edits to
either the oracle or the HIP kernels are caught bit for bit.

  python tests/golden/make_block_fixture.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
import helpers as H  # noqa: E402
from oracle import oracle as O  # noqa: E402

GAMMA, C_H, DX = 5.0 / 3.0, 1.9, (0.1, 0.07, 0.13)
CASES = {"mhd": ("glmmhd", "ppm", "hlld", (10, 5, 4), 3, 1), "hydro": ("euler", "plm", "hllc", (12, 6, 5), 2, 0)}


def main():
    out = {}
    for tag, (fluid, recon, riemann, nx, ng, ded) in CASES.items():
        prim = H.random_prim(fluid, nx, ng, seed=2024, kind="shock", nblocks=1)
        g = H.geom(fluid, nx, ng, 0, DX)
        cons = H.prim_to_cons(fluid, prim, GAMMA)
        fl = H.orc_fluxes(fluid, recon, riemann, g, prim, GAMMA, C_H)
        stage = H.orc_stage(fluid, recon, riemann, g, cons, cons, prim, GAMMA, C_H, 0.25, 0.75, 0.004, dedner=ded,
                            alpha=0.1, mindx=0.07)
        c2, p2, bad = H.orc_c2p(fluid, g, stage, O.make_eos(GAMMA))
        assert bad == 0
        out[tag + "_prim"] = prim
        out[tag + "_cons"] = cons
        for d in range(3):
            out[tag + "_flux%d" % (d + 1)] = fl[d]
        out[tag + "_stage"] = stage
        out[tag + "_prim_after"] = p2
        out[tag + "_min_dt"] = np.array([H.orc_min_dt(fluid, g, p2, GAMMA)])
    np.savez_compressed(os.path.join(HERE, "block_fixture.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
