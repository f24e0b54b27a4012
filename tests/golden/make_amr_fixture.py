"""Generates tests/golden/amr_fixture.npz: a small statically refined periodic mesh (one octant of a 2 x 2 x 2 root
grid refined: 15 leaves, coarse-fine faces in every direction, across the periodic seam too), a state on its
leaves and the state after three cycles of the refined-mesh stage loop as restated by tests/amr_oracle.py (multilevel
ghost exchange, coarse-fine flux correction before the flux divergence, per-level cell widths, time step and c_h) --
GLM-MHD PPM + HLLD + Dedner VL2 (nghost 4) and hydro PLM + HLLC RK3 (nghost 2).

Stored as DATA: the forest (level and position of every leaf, in the driver's Z order), the interiors of the final
conserved states, the time steps taken (the initial state is a closed-form function of the cell centres, initial_state()
below, which the tests evaluate again).  tests/test_amr_mesh.py checks that the oracle keeps reproducing it (CPU),
tests/test_gpu_amr.py that the device does (flux-array task order, parity build: bit for bit) -- a reference the HIP
path has to meet that does not move when the live oracle or the driver's plans do.

    python tests/golden/make_amr_fixture.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import oracle as O  # noqa: E402
from amr_oracle import RefinedMeshOracle  # noqa: E402
from amr_emulator import placement  # noqa: E402

MESH = ["parthenon/mesh/refinement=static", "parthenon/mesh/nx1=16", "parthenon/mesh/nx2=16", "parthenon/mesh/nx3=16",
        "parthenon/meshblock/nx1=8", "parthenon/meshblock/nx2=8", "parthenon/meshblock/nx3=8",
        "parthenon/static_refinement0/x1min=0.05", "parthenon/static_refinement0/x1max=0.2",
        "parthenon/static_refinement0/x2min=0.05", "parthenon/static_refinement0/x2max=0.2",
        "parthenon/static_refinement0/x3min=0.05", "parthenon/static_refinement0/x3max=0.2",
        "parthenon/static_refinement0/level=1"] + \
       ["parthenon/mesh/%sx%d_bc=periodic" % (io, d) for d in (1, 2, 3) for io in "io"]
CASES = {"mhd_ppm_hlld_vl2": ("glmmhd", "ppm", "hlld", "vl2", 4), "hydro_plm_hllc_rk3": ("euler", "plm", "hllc", "rk3", 2)}
NCYCLES, GAMMA, CFL = 3, 5.0 / 3.0, 0.3


def overrides(case):
    fluid, recon, riemann, integ, ng = CASES[case]
    return MESH + ["hydro/fluid=%s" % fluid, "hydro/reconstruction=%s" % recon, "hydro/riemann=%s" % riemann,
                   "parthenon/time/integrator=%s" % integ, "parthenon/mesh/nghost=%d" % ng, "hydro/gamma=%r" % GAMMA,
                   "parthenon/time/cfl=%r" % CFL]


NRB = (2, 2, 2)


def initial_state(fluid, leaves, mb, ng):
    """positive state per leaf: a closed-form function of the cell centres (coarse and fine data are consistent samples
    of one field), with a short-wavelength term so that the limiters have something to do"""
    nv = 9 if fluid == "glmmhd" else 5
    out = []
    for level, lx in leaves:
        n = [mb[d] + 2 * ng for d in range(3)]
        dx = [1.0 / (NRB[d] * mb[d] * 2 ** level) for d in range(3)]
        ax = [(-0.5 + (lx[d] * mb[d] + np.arange(n[d]) - ng + 0.5) * dx[d]) for d in range(3)]
        z, y, x = np.meshgrid(ax[2], ax[1], ax[0], indexing="ij")
        w = np.zeros((nv,) + x.shape)
        w[0] = 1.0 + 0.3 * np.sin(2 * np.pi * x) * np.cos(2 * np.pi * y) + 0.1 * np.sin(4 * np.pi * z)
        w[1] = 0.4 * np.cos(2 * np.pi * y) + 0.1 * np.sin(2 * np.pi * z)
        w[2] = -0.3 * np.sin(2 * np.pi * x)
        w[3] = 0.2 * np.sin(2 * np.pi * (x + y))
        w[4] = 1.0 + 0.2 * np.cos(2 * np.pi * (x - z))
        if fluid == "glmmhd":
            w[5] = 0.5 * np.sin(2 * np.pi * y)
            w[6] = 0.4 * np.sin(2 * np.pi * z) + 0.2
            w[7] = -0.3 * np.cos(2 * np.pi * x)
            w[8] = 0.05 * np.sin(2 * np.pi * (x + z))
        w[0] *= 1.0 + 0.02 * np.sin(14 * np.pi * x) * np.sin(10 * np.pi * y + 0.3) * np.cos(12 * np.pi * z)
        w[4] *= 1.0 + 0.02 * np.cos(10 * np.pi * x + 0.7) * np.sin(14 * np.pi * (y - z))
        u = np.zeros_like(w)
        u[0] = w[0]
        u[1:4] = w[0] * w[1:4]
        u[4] = w[4] / (GAMMA - 1.0) + 0.5 * w[0] * (w[1] ** 2 + w[2] ** 2 + w[3] ** 2)
        if fluid == "glmmhd":
            u[5:8] = w[5:8]
            u[4] += 0.5 * (w[5] ** 2 + w[6] ** 2 + w[7] ** 2)
            u[8] = w[8]
        out.append(u)
    return out


def forest():
    from athenapk_amd import decks, driver
    v = driver.HostPlan(decks.load("blast"), overrides("hydro_plm_hllc_rk3"))
    return [(p[0], tuple(p[1])) for p in placement(v)]


def run(case, leaves):
    fluid, recon, riemann, integ, ng = CASES[case]
    fo = RefinedMeshOracle(O, fluid, recon, riemann, integ, NRB, (8, 8, 8), ng, (-0.5, -0.5, -0.5), (0.5, 0.5, 0.5), leaves,
                           GAMMA, CFL)
    u0 = initial_state(fluid, leaves, (8, 8, 8), ng)
    fo.initialize(u0)
    dts = []
    for _ in range(NCYCLES):
        dts.append(fo.dt)
        fo.step()
    sl = (slice(None),) + (slice(ng, -ng),) * 3
    return np.stack([u[sl] for u in fo.cons]), np.array(dts), fo.time


def main():
    leaves = forest()
    out = {"levels": np.array([l for l, _ in leaves]), "lx": np.array([lx for _, lx in leaves])}
    for case in CASES:
        u1, dts, t = run(case, leaves)
        out[case + "_final"], out[case + "_dt"], out[case + "_time"] = u1, dts, np.array(t)
    assert sorted(set(out["levels"])) == [0, 1]
    np.savez_compressed(os.path.join(HERE, "amr_fixture.npz"), **out)
    print("leaves", len(leaves), {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
