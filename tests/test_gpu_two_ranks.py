"""The N>1 driver path on a real GPU: two (or four) processes share cuda:0, each owns its Morton
share of the meshblocks, halo messages go rank to rank (gloo here, staged through the host --
RCCL needs one GPU per rank and the test box has one), dt / c_h / history / turbulence sums are
all-reduced.  Results must equal the oracle's single-process run: bit for bit for the pure
hydro path in the strict build, to round-off where global sums are involved."""
import os
import sys

import numpy as np
import pytest

from _spawn import spawn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


CASES = {
    # deck, overrides, oracle kwargs, pgen, pgen kwargs, cycles
    "mhd_ppm_hlld_vl2": ("synthetic_mhd",
                         ["parthenon/mesh/nx1=32", "parthenon/mesh/nx2=32", "parthenon/mesh/nx3=32",
                          "parthenon/meshblock/nx1=16", "parthenon/meshblock/nx2=16", "parthenon/meshblock/nx3=16"],
                         dict(fluid="glmmhd", recon="ppm", riemann="hlld", integrator="vl2", nx=(32, 32, 32),
                              mb=(16, 16, 16), ng=3, cfl=0.3, gamma=1.666666666666667), "synthetic", {}, 4),
    # (rows of 32 cells: hydro PLM's single march, fused3_kernel.hpp; of 16: the two-kernel form)
    "sod_outflow": ("sod",
                    ["parthenon/mesh/nx1=128", "parthenon/mesh/nx2=8", "parthenon/mesh/nx3=8",
                     "parthenon/meshblock/nx1=32", "parthenon/meshblock/nx2=8", "parthenon/meshblock/nx3=8"],
                    dict(fluid="euler", recon="plm", riemann="hllc", integrator="rk2", nx=(128, 8, 8), mb=(32, 8, 8),
                         ng=2, bc=("outflow", "periodic", "periodic"), xmin=(0.0, -0.5, -0.5), xmax=(1.0, 0.5, 0.5),
                         gamma=1.4, cfl=0.3), "sod", {}, 6),
    "sod_outflow_narrow_blocks": ("sod",
                    ["parthenon/mesh/nx1=64", "parthenon/mesh/nx2=8", "parthenon/mesh/nx3=8",
                     "parthenon/meshblock/nx1=16", "parthenon/meshblock/nx2=8", "parthenon/meshblock/nx3=8"],
                    dict(fluid="euler", recon="plm", riemann="hllc", integrator="rk2", nx=(64, 8, 8), mb=(16, 8, 8),
                         ng=2, bc=("outflow", "periodic", "periodic"), xmin=(0.0, -0.5, -0.5), xmax=(1.0, 0.5, 0.5),
                         gamma=1.4, cfl=0.3), "sod", {}, 6),
    "mhd_wenoz_hlld_rk3": ("synthetic_mhd",
                           ["parthenon/mesh/nx1=32", "parthenon/mesh/nx2=32", "parthenon/mesh/nx3=16",
                            "parthenon/meshblock/nx1=16", "parthenon/meshblock/nx2=16", "parthenon/meshblock/nx3=8",
                            "parthenon/time/integrator=rk3", "hydro/reconstruction=wenoz"],
                           dict(fluid="glmmhd", recon="wenoz", riemann="hlld", integrator="rk3", nx=(32, 32, 16),
                                mb=(16, 16, 8), ng=3, cfl=0.3, gamma=1.666666666666667), "synthetic", {}, 3),
    # passive scalars ride the fused stages (mass-flux workspace + scalar kernels), also when split
    "mhd_scalars_vl2": ("synthetic_mhd",
                        ["parthenon/mesh/nx1=32", "parthenon/mesh/nx2=32", "parthenon/mesh/nx3=32",
                         "parthenon/meshblock/nx1=16", "parthenon/meshblock/nx2=16", "parthenon/meshblock/nx3=16",
                         "hydro/nscalars=2"],
                        dict(fluid="glmmhd", recon="ppm", riemann="hlld", integrator="vl2", nx=(32, 32, 32),
                             mb=(16, 16, 16), ng=3, cfl=0.3, gamma=1.666666666666667, nscalars=2), "synthetic", {}, 3),
    # 2-D: the donor-cell predictor has no single-kernel form there, so only the exchange before the
    # high-order stage is overlapped
    "ot_2d": ("orszag_tang",
              ["parthenon/mesh/nx1=64", "parthenon/mesh/nx2=64", "parthenon/meshblock/nx1=32", "parthenon/meshblock/nx2=32",
               "hydro/first_order_flux_correct=false"],
              dict(fluid="glmmhd", recon="ppm", riemann="hlld", integrator="vl2", nx=(64, 64, 1), mb=(32, 32, 1), ng=3,
                   xmin=(-0.5, -0.5, -0.5), xmax=(0.5, 0.5, 0.5), cfl=0.4, gamma=1.666666666666667), "orszag_tang", {}, 6),
    # first-order flux correction: flux-array path, synchronous exchanges
    "ot_2d_fofc": ("orszag_tang",
                   ["parthenon/mesh/nx1=64", "parthenon/mesh/nx2=64", "parthenon/meshblock/nx1=32",
                    "parthenon/meshblock/nx2=32", "hydro/first_order_flux_correct=true"],
                   dict(fluid="glmmhd", recon="ppm", riemann="hlld", integrator="vl2", nx=(64, 64, 1), mb=(32, 32, 1),
                        ng=3, xmin=(-0.5, -0.5, -0.5), xmax=(0.5, 0.5, 0.5), cfl=0.4, gamma=1.666666666666667,
                        fofc=True), "orszag_tang", {}, 6),
    # reflecting walls on every side: all faces of the blocks are "late"
    "lw_implode_2d": ("lw_implode",
                      ["parthenon/mesh/nx1=64", "parthenon/mesh/nx2=64", "parthenon/meshblock/nx1=32",
                       "parthenon/meshblock/nx2=32"],
                      dict(fluid="euler", recon="plm", riemann="hllc", integrator="vl2", nx=(64, 64, 1), mb=(32, 32, 1),
                           ng=3, bc=("reflecting", "reflecting", "periodic"), xmin=(0.0, 0.0, -0.5),
                           xmax=(0.3, 0.3, 0.5), cfl=0.4, gamma=1.4), "lw_implode", {}, 40),
}
# meshblocks wide enough (nx1 >= 32) for the two-kernel stage: the split runs the x3 sweep on plane
# windows while the halo messages fly, then the finishing x1 + x2 march (apk_stage_split_axis == 3)
CASES["mhd_ppm_two_kernel"] = ("synthetic_mhd",
                               ["parthenon/mesh/nx1=64", "parthenon/mesh/nx2=32", "parthenon/mesh/nx3=32",
                                "parthenon/meshblock/nx1=32", "parthenon/meshblock/nx2=16", "parthenon/meshblock/nx3=16"],
                               dict(fluid="glmmhd", recon="ppm", riemann="hlld", integrator="vl2", nx=(64, 32, 32),
                                    mb=(32, 16, 16), ng=3, cfl=0.3, gamma=1.666666666666667), "synthetic", {}, 3)
CASES["mhd_wenoz_rk3_two_kernel"] = ("synthetic_mhd",
                                     ["parthenon/mesh/nx1=64", "parthenon/mesh/nx2=32", "parthenon/mesh/nx3=32",
                                      "parthenon/meshblock/nx1=32", "parthenon/meshblock/nx2=16", "parthenon/meshblock/nx3=16",
                                      "parthenon/time/integrator=rk3", "hydro/reconstruction=wenoz"],
                                     dict(fluid="glmmhd", recon="wenoz", riemann="hlld", integrator="rk3", nx=(64, 32, 32),
                                          mb=(32, 16, 16), ng=3, cfl=0.3, gamma=1.666666666666667), "synthetic", {}, 2)
# ... the same with line-aligned rows (apk_amd/row_pitch = aligned: a pitch of 48 doubles for these 38-cell rows, 13 doubles in
# front of every block): pack / unpack plans, x1 strips in the buffers and the accessors on the padded layout
CASES["mhd_ppm_two_kernel_padded_rows"] = (CASES["mhd_ppm_two_kernel"][0], CASES["mhd_ppm_two_kernel"][1] + ["apk_amd/row_pitch=aligned"]) + CASES["mhd_ppm_two_kernel"][2:]
# the layout of the 8-GPU benchmark in small: 64 meshblocks, a 2x2x2 brick of them per rank, 7 peers
CASES["mhd_8_ranks"] = ("synthetic_mhd",
                        ["parthenon/mesh/nx1=64", "parthenon/mesh/nx2=64", "parthenon/mesh/nx3=64",
                         "parthenon/meshblock/nx1=16", "parthenon/meshblock/nx2=16", "parthenon/meshblock/nx3=16"],
                        dict(fluid="glmmhd", recon="ppm", riemann="hlld", integrator="vl2", nx=(64, 64, 64),
                             mb=(16, 16, 16), ng=3, cfl=0.3, gamma=1.666666666666667), "synthetic", {}, 3)
# overlapped exchanges after ncyc cycles with overlap on (default: every exchange but the initial one)
EXPECT_OVERLAPPED = {"ot_2d": lambda nst, ncyc: ncyc, "ot_2d_fofc": lambda nst, ncyc: 0,
                     "lw_implode_2d": lambda nst, ncyc: ncyc}
# 3-D VL2 (round 4): the exchange in flight at the start of a cycle is completed before the donor-cell predictor, which
# then runs whole (two rows per lane, one launch) instead of on seven windows; the one before the corrector is overlapped
for _case in ("mhd_ppm_hlld_vl2", "mhd_scalars_vl2", "mhd_ppm_two_kernel", "mhd_ppm_two_kernel_padded_rows", "mhd_8_ranks"):
    EXPECT_OVERLAPPED[_case] = lambda nst, ncyc: ncyc


# hydro PLM in a prim-free RK cycle runs every stage as ONE march (fused3_kernel.hpp); such stages are left whole -- the
# split into plane windows would put the two-kernel form back (can_overlap_next, host/sim.cpp) -- so nothing is overlapped
EXPECT_OVERLAPPED["sod_outflow"] = lambda nst, ncyc: 0


# one-layer exchanges (apk_sim_set_thin_exchange): the exchange at the end of every cycle of VL2 on a periodic 3-D mesh
# without passive scalars; the blocks compared below include their ghost zones, which the accessor completes first
EXPECT_THIN = {"mhd_ppm_hlld_vl2", "mhd_ppm_two_kernel", "mhd_ppm_two_kernel_padded_rows", "mhd_8_ranks"}
# ... whose x1 strips bypass the pack / unpack kernels in both exchanges of a cycle (apk_sim_set_x1_direct) where the
# meshblocks are wide enough for the two-kernel stage: real messages between ranks, laid out by the pack plans of the peer
EXPECT_X1_DIRECT = {"mhd_ppm_two_kernel", "mhd_ppm_two_kernel_padded_rows", "mhd_8_ranks"}
# (the RK integrators: every exchange but the one after the very first stage, which reads stored primitives)
EXPECT_X1_DIRECT_RK = {"mhd_wenoz_rk3_two_kernel": 3}


def _worker(rank, world, port, case, outdir, overlap=True):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from athenapk_amd import decks, driver

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        deck, ov, _, _, _, ncyc = CASES[case]
        s = driver.Simulation(decks.load(deck), ov, rank=rank, nranks=world, strict=True)
        s.set_overlap(overlap)
        s.initialize()
        for _ in range(ncyc):
            s.step()
        thin = s.thin_exchanges()
        blocks = {s.block_gid(lb)[0]: s.read_block(lb, "cons") for lb in range(s.info.nblocks_local)}
        np.savez(os.path.join(outdir, "rank%d.npz" % rank), time=s.time, dt=s.dt, hist=s.history(),
                 overlapped=s.overlapped_exchanges, thin=thin, x1_direct=s.x1_direct_exchanges(),
                 **{"b%d" % g: a for g, a in blocks.items()})
        s.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("world,overlap", [(2, True), (4, True), (2, False), (8, True)])
@pytest.mark.parametrize("case", sorted(CASES))
def test_ranks_sharing_one_gpu_match_oracle(oracle, tmp_path, case, world, overlap):
    """overlap=True: between the stages of a cycle the halo messages stay in flight while the next
    stage's x1 sweep runs on the cells that do not need them (apk_stage_args.phase); the result
    must not change by a bit."""
    import torch.multiprocessing as mp
    if (world == 8) != (case == "mhd_8_ranks"):
        pytest.skip("the 8-rank layout has its own case")
    deck, ov, okw, pgen, pkw, ncyc = CASES[case]
    nstages = {"rk1": 1, "rk2": 2, "vl2": 2, "rk3": 3}[okw["integrator"]]
    o = oracle.Sim(nthreads=os.cpu_count(), **okw)
    o.pgen(pgen, **pkw)
    for _ in range(ncyc):
        o.step()
    spawn(_worker, lambda port: (world, port, case, str(tmp_path), overlap), world)
    seen = set()
    for r in range(world):
        z = np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))
        assert z["time"] == o.time and z["dt"] == o.dt
        # every exchange but the one of the initialisation is overlapped with the stage that follows
        # it (the x1 sweep of a high-order stage / the single-kernel donor-cell stage of VL2),
        # across cycle boundaries as well
        want_ov = EXPECT_OVERLAPPED.get(case, lambda nst, n: nst * n - 1)(nstages, ncyc)
        assert int(z["overlapped"]) == (want_ov if overlap else 0)
        assert int(z["thin"]) == (ncyc if case in EXPECT_THIN else 0)
        if case in EXPECT_X1_DIRECT:
            assert int(z["x1_direct"]) == 2 * ncyc
        elif case in EXPECT_X1_DIRECT_RK:
            assert int(z["x1_direct"]) == EXPECT_X1_DIRECT_RK[case] * ncyc - 1
        else:
            assert int(z["x1_direct"]) in (0, nstages * ncyc, nstages * ncyc - 1)
        np.testing.assert_allclose(z["hist"], o.history(), rtol=1e-13, atol=1e-15)
        for key in z.files:
            if key.startswith("b"):
                gid = int(key[1:])
                seen.add(gid)
                assert np.array_equal(z[key], o.cons(gid)), "rank %d block %d" % (r, gid)
    assert seen == set(range(o.nblocks))


def _turb_worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from athenapk_amd import decks, driver
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        ov = ["parthenon/mesh/nx1=32", "parthenon/mesh/nx2=32", "parthenon/mesh/nx3=32", "parthenon/meshblock/nx1=16",
              "parthenon/meshblock/nx2=16", "parthenon/meshblock/nx3=16"]
        s = driver.Simulation(decks.load("turbulence"), ov, rank=rank, nranks=world, strict=True)
        s.initialize()
        for _ in range(8):
            s.step()
        blocks = {s.block_gid(lb)[0]: s.read_block(lb, "cons") for lb in range(s.info.nblocks_local)}
        np.savez(os.path.join(outdir, "rank%d.npz" % rank), time=s.time, turb=s.turbulence_history(), hist=s.history(),
                 var_hat=s.fmft_var_hat(), overlapped=s.overlapped_exchanges, **{"b%d" % g: a for g, a in blocks.items()})
        s.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_turbulence_driver_on_two_ranks(oracle, tmp_path):
    """The forcing's global sums (mean momentum, RMS normalisation) and the Ms / Ma history are
    all-reduced over ranks; every rank evolves the same spectral state from the same host RNG."""
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_gpu_driver import _turb_k_vec
    o = oracle.Sim(fluid="glmmhd", recon="plm", riemann="hlle", integrator="vl2", nx=(32, 32, 32), mb=(16, 16, 16), ng=2,
                   cfl=0.3, gamma=1.0001)
    o.pgen("turbulence", k_vec=_turb_k_vec())
    for _ in range(8):
        o.step()
    spawn(_turb_worker, lambda port: (2, port, str(tmp_path)), 2)
    for r in range(2):
        z = np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))
        assert np.array_equal(z["var_hat"], o.var_hat())
        assert abs(z["time"] - o.time) <= 1e-13 * o.time
        np.testing.assert_allclose(z["turb"], o.turb_history(), rtol=1e-10)
        np.testing.assert_allclose(z["hist"], o.history(), rtol=1e-11, atol=1e-14)
        # the exchange before each corrector stage: 8 (round 4: the one in flight at the start of a cycle is completed
        # before the donor-cell predictor, which runs whole instead of on windows)
        assert int(z["overlapped"]) == 8
        for key in z.files:
            if key.startswith("b"):
                np.testing.assert_allclose(z[key], o.cons(int(key[1:])), rtol=1e-11, atol=1e-13)
