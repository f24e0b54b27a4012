"""BASELINE.json configs 3, 4 and 5 AS SPECIFIED on the HIP path (round-1 verdict, "What's missing" 1-2):

  cfg3  Orszag-Tang 512^2 as a thin-z 3-D mesh (512 x 512 x 4: the 3-D kernels incl. the x3 sweep),
        MHD PPM + HLLD + Dedner, VL2
  cfg4  driven turbulence, MHD WENOZ + HLLD, RK3, forcing ON (inputs/turbulence.in with the scheme
        overrides; wenoz => nghost = 3, src/hydro/hydro.cpp:334-336; driver src/pgen/turbulence.cpp:373-482)
        -- against the oracle at 32^3 on 1 and 2 ranks, by properties at the full 512^3
  cfg5  blast_3d_amr: 4 levels, 64^3 root in 16^3 meshblocks, GLM-MHD PPM + HLLD, the deck's own
        near-vacuum ambient medium and pressure ratio 1.6e8 (inputs/blast_3d_amr.in:12-56, IC
        src/pgen/blast.cpp:151-199).  The deck sets no first_order_flux_correct and the run needs none
        (400 cycles below); a second test switches the option on and must reproduce the run without it
        while no cell fails FirstOrderFluxCorrect's test
"""
import os
import sys

import numpy as np
import pytest

from _spawn import spawn

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GAMMA_DECK = 1.666666666666667


def _sim(deck, overrides, strict=True, **kw):
    from athenapk_amd import decks, driver
    return driver.Simulation(decks.load(deck), overrides, strict=strict, **kw)


# ---- config 3 as "thin-z 3-D" ----------------------------------------------------------------------------
OT3D = ["parthenon/mesh/nx3=4", "parthenon/meshblock/nx3=4", "hydro/first_order_flux_correct=false"]


@pytest.mark.parametrize("mb", [256, 128])
def test_config3_thin_z_3d_equals_the_2d_run_plane_by_plane(mb):
    """512 x 512 x 4 runs the 3-D kernels (single-kernel donor-cell predictor, x1 sweep + x2 march +
    finishing x3 march).  Every plane is the 2-D problem and the x3 flux difference of identical
    planes is exactly zero, so in the parity build each plane must equal the 2-D run bit for bit
    (x3 is 128 x wider than x1/x2 cells: it never limits dt and never sets min dx)."""
    ov2 = ["parthenon/meshblock/nx1=%d" % mb, "parthenon/meshblock/nx2=%d" % mb, "hydro/first_order_flux_correct=false"]
    a = _sim("orszag_tang", ov2).initialize()
    b = _sim("orszag_tang", ov2 + OT3D).initialize()
    assert a.info.zones_total == 512 * 512 and b.info.zones_total == 512 * 512 * 4
    assert b.info.nblocks_local == (512 // mb) ** 2
    ncyc = 12
    for _ in range(ncyc):
        a.step()
        b.step()
    assert a.time == b.time and a.dt == b.dt
    ua, ub = a.gather(), b.gather()
    assert ua.shape[1] == 1 and ub.shape[1] == 4
    for k in range(4):
        assert np.array_equal(ub[:, k], ua[:, 0]), "plane %d" % k
    assert np.all(ub[3] == 0.0) and np.all(ub[7] == 0.0)          # no m3, no B3 ever appears


def test_config3_thin_z_3d_product_build_conserves_and_stays_symmetric():
    s = _sim("orszag_tang", OT3D, strict=False).initialize()
    h0 = s.history()
    s.run(nlim=20)
    h1 = s.history()
    assert h1[0] == pytest.approx(h0[0], rel=1e-13) and h1[5] == pytest.approx(h0[5], rel=1e-12)
    # (FMA build: a*f - a*f contracts to fma(a, f, -(a*f)), the rounding error of the product, not 0)
    assert abs(h1[1]) < 1e-12 and abs(h1[2]) < 1e-12 and abs(h1[3]) < 1e-15
    u = s.gather()
    assert np.max(np.abs(u[:, 1:] - u[:, :1])) <= 1e-13       # planes stay alike (FMA build: round-off)
    rot = u[:, 0, ::-1, ::-1]
    assert np.max(np.abs(u[0, 0] - rot[0])) < 1e-11 and np.max(np.abs(u[1, 0] + rot[1])) < 1e-11


# ---- config 4: forced turbulence with the config's scheme ---------------------------------------------------
CFG4_SCHEME = ["parthenon/time/integrator=rk3", "hydro/reconstruction=wenoz", "hydro/riemann=hlld", "parthenon/mesh/nghost=3"]
CFG4_SMALL = ["parthenon/mesh/nx1=32", "parthenon/mesh/nx2=32", "parthenon/mesh/nx3=32", "parthenon/meshblock/nx1=16",
              "parthenon/meshblock/nx2=16", "parthenon/meshblock/nx3=16"] + CFG4_SCHEME
CFG4_CYCLES = 12


def _cfg4_oracle(oracle):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_gpu_driver import _turb_k_vec
    o = oracle.Sim(fluid="glmmhd", recon="wenoz", riemann="hlld", integrator="rk3", nx=(32, 32, 32), mb=(16, 16, 16),
                   ng=3, cfl=0.3, gamma=1.0001, nthreads=os.cpu_count())
    o.pgen("turbulence", k_vec=_turb_k_vec())
    return o


@pytest.mark.parametrize("strict", [True, False], ids=["strict", "fma"])
def test_config4_forced_turbulence_wenoz_hlld_rk3_matches_oracle(oracle, strict):
    """32^3 in 8 meshblocks of 16^3, 12 driven cycles.  The two global sums of Perturb are reduced in a
    different order than the oracle's, so the fields (and through dt the Ornstein-Uhlenbeck blend of the
    spectral state, same host RNG draws) agree to round-off, not bit for bit, in either build."""
    s = _sim("turbulence", CFG4_SMALL, strict=strict).initialize()
    o = _cfg4_oracle(oracle)
    assert s.info.ng == 3 and s.info.nblocks_local == 8
    np.testing.assert_allclose(s.gather("cons"), o.gather_cons(), rtol=1e-14, atol=0)
    for _ in range(CFG4_CYCLES):
        s.step()
        o.step()
    # (round-4 advisor: no stage of this cycle stores primitives and the kick after the last one estimates the time step
    # without storing them either -- apk_turb_apply_dt, once per cycle -- so they are stale until an accessor asks)
    assert s.prim_is_stale and s.turb_dt_kicks() == CFG4_CYCLES
    np.testing.assert_allclose(s.fmft_var_hat(), o.var_hat(), rtol=1e-12, atol=1e-14)
    assert abs(s.time - o.time) <= 1e-13 * o.time
    np.testing.assert_allclose(s.gather("cons"), o.gather_cons(), rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(s.turbulence_history(), o.turb_history(), rtol=1e-10)
    np.testing.assert_allclose(s.history(), o.history(), rtol=1e-11, atol=1e-14)
    # the forcing did something: the box was at rest
    assert s.turbulence_history()[0] > 1e-3


def _cfg4_worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from athenapk_amd import decks, driver
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        s = driver.Simulation(decks.load("turbulence"), CFG4_SMALL, rank=rank, nranks=world, strict=True)
        s.initialize()
        for _ in range(CFG4_CYCLES):
            s.step()
        blocks = {s.block_gid(lb)[0]: s.read_block(lb, "cons") for lb in range(s.info.nblocks_local)}
        np.savez(os.path.join(outdir, "rank%d.npz" % rank), time=s.time, turb=s.turbulence_history(), hist=s.history(),
                 var_hat=s.fmft_var_hat(), **{"b%d" % g: a for g, a in blocks.items()})
        s.close()
    finally:
        dist.destroy_process_group()


def test_config4_forced_turbulence_on_two_ranks_matches_oracle(oracle, tmp_path):
    import torch.multiprocessing as mp
    o = _cfg4_oracle(oracle)
    for _ in range(CFG4_CYCLES):
        o.step()
    spawn(_cfg4_worker, lambda port: (2, port, str(tmp_path)), 2)
    seen = set()
    for r in range(2):
        z = np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))
        np.testing.assert_allclose(z["var_hat"], o.var_hat(), rtol=1e-12, atol=1e-14)
        assert abs(z["time"] - o.time) <= 1e-13 * o.time
        np.testing.assert_allclose(z["turb"], o.turb_history(), rtol=1e-10)
        np.testing.assert_allclose(z["hist"], o.history(), rtol=1e-11, atol=1e-14)
        for key in z.files:
            if key.startswith("b"):
                seen.add(int(key[1:]))
                np.testing.assert_allclose(z[key], o.cons(int(key[1:])), rtol=1e-11, atol=1e-13)
    assert seen == set(range(8))


def test_config4_full_size_512_cubed_forced_turbulence_properties():
    """The whole 512^3 mesh of config 4 (64 meshblocks of 128^3, WENOZ + HLLD RK3, forcing on) on one
    GPU for 2 cycles.  Size-independent properties: zone count, mass conserved to round-off, the
    forcing removes its own mean momentum (total momentum stays at round-off of the rms momentum),
    kinetic energy grows from rest, the state stays finite."""
    ov = ["parthenon/mesh/nx1=512", "parthenon/mesh/nx2=512", "parthenon/mesh/nx3=512", "parthenon/meshblock/nx1=128",
          "parthenon/meshblock/nx2=128", "parthenon/meshblock/nx3=128"] + CFG4_SCHEME
    s = _sim("turbulence", ov, strict=False).initialize()
    assert s.info.zones_total == 512 ** 3 and s.info.nblocks_local == 64 and s.info.ng == 3
    h0 = s.history()
    assert h0[4] == 0.0                                     # at rest
    for _ in range(2):
        s.step()
    h1 = s.history()
    assert np.all(np.isfinite(h1))
    assert abs(h1[0] - h0[0]) <= 1e-13 * h0[0]             # mass
    ms, ma, pb = s.turbulence_history()
    assert ms > 0.0 and np.isfinite(ma) and np.isfinite(pb)
    assert h1[4] > 0.0                                      # kinetic energy from the forcing
    vrms = np.sqrt(2.0 * h1[4] / h1[0])
    assert np.abs(h1[1:4]).max() <= 1e-10 * vrms           # Perturb subtracts the mean momentum (turbulence.cpp:395-430)
    assert s.ncycle == 2


# ---- config 5: the adaptive blast as specified ---------------------------------------------------------------
CFG5 = ["parthenon/mesh/nx%d=64" % d for d in (1, 2, 3)] + ["parthenon/meshblock/nx%d=16" % d for d in (1, 2, 3)] + [
    "parthenon/mesh/numlevel=4", "hydro/fluid=glmmhd", "hydro/riemann=hlld", "hydro/reconstruction=ppm",
    "parthenon/mesh/nghost=4"]


def _forest_is_octant_symmetric(s, root_blocks):
    from amr_emulator import placement
    locs = {(p[0], tuple(p[1])) for p in placement(s)}
    for lev, loc in list(locs):
        n1 = root_blocks * 2 ** lev
        if (lev, (n1 - 1 - loc[0], loc[1], loc[2])) not in locs:
            return False
        if (lev, (loc[1], loc[0], loc[2])) not in locs or (lev, (loc[0], loc[2], loc[1])) not in locs:
            return False
    return True


def _forest_octant_symmetric_within(s, root_blocks, cycles=2):
    """The blast is symmetric under the reflections and axis permutations of its octants, but floating-point
    arithmetic is not exactly so: the flux differences are summed in the order x1, x2, x3 ((d1 + d2) + d3 is not
    (d1 + d3) + d2 in the last bit), and in the product build a*b - c*d is fma(a, b, -(c*d)), whose mirror image rounds
    differently.  A refinement criterion that sits at its threshold can therefore flag a block one cycle before its
    mirror image: the forest is symmetric except for single cycles in which one of the two has not followed yet
    (tools/dbg_cfg5.py lists them: 3 of 420 cycles on config 5, as in round 3).  So: symmetric now, or after at most
    `cycles` more cycles."""
    for extra in range(cycles + 1):
        if _forest_is_octant_symmetric(s, root_blocks):
            return True
        if extra < cycles:
            s.step()
    return False


def test_config5_adaptive_mhd_blast_as_decked():
    """4 levels over a 64^3 root in 16^3 meshblocks, GLM-MHD PPM + HLLD (nghost = 4 on refined meshes),
    the deck's own ambient pressure 1e-3 and pressure ratio 1.6e8, 400 cycles (the mesh regrids as the
    shock leaves the initial patch).  No oracle exists for the refined-mesh stage loop (Parthenon is
    un-vendored), so this checks what must hold for ANY correct implementation: finest level around
    the hot sphere, no negative state (the driver raises on the device flags), finite positive fields,
    mass and energy conserved to round-off across all coarse-fine faces and regrids, B stays exactly
    zero, octant symmetry of the forest."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_gpu_amr import _totals
    from amr_emulator import placement
    ov = CFG5 + ["parthenon/time/tlim=1.0"]
    s = _sim("blast_3d_amr", ov, strict=False).initialize()
    i = s.refresh_info()
    pl = placement(s)
    assert max(p[0] for p in pl) == 3 and i.ng == 4 and tuple(i.mb) == (16, 16, 16)
    nb0 = i.nblocks_total
    assert nb0 > 64
    assert _forest_is_octant_symmetric(s, 4)
    t0 = _totals(s)
    ncyc = 400
    for _ in range(ncyc):
        s.step()                                            # raises on a negative density / pressure flag
    i = s.refresh_info()
    t1 = _totals(s)
    assert np.all(np.isfinite(t1))
    refined, merged, maxlev, zc = s.amr_stats()
    assert maxlev == 3 and zc >= ncyc * 64 * 16 ** 3 and refined > 0
    assert i.nblocks_total > nb0                            # the refined patch has grown with the shock
    assert abs(t1[0] - t0[0]) < 1e-12 * t0[0] and abs(t1[4] - t0[4]) < 1e-12 * t0[4]
    assert np.abs(t1[5:8]).max() == 0.0                     # no field, and PPM + HLLD keeps B = 0 exactly
    assert _forest_octant_symmetric_within(s, 4)
    i = s.refresh_info()
    g = i.ng
    for lb in range(i.nblocks_local):
        w = s.read_block(lb, "prim")[:, g:-g, g:-g, g:-g]
        assert np.all(np.isfinite(w)) and w[0].min() > 0.0 and w[4].min() > 0.0


def _forest_is_reflection_symmetric(s, root_blocks):
    from amr_emulator import placement
    locs = {(p[0], tuple(p[1])) for p in placement(s)}
    for lev, loc in locs:
        n1 = root_blocks * 2 ** lev
        for d in range(3):
            m = list(loc)
            m[d] = n1 - 1 - m[d]
            if (lev, tuple(m)) not in locs:
                return False
    return True


def test_config5_parity_build_keeps_the_forest_mirror_symmetric_every_cycle():
    """The tolerance of _forest_octant_symmetric_within is for what floating point does NOT keep: axis permutations
    (the flux differences are summed x1, x2, x3) and, in the product build, mirror images of contracted a*b - c*d.  The
    parity build contracts nothing and the reference's groupings are mirror symmetric (the "KGF" parentheses of PPM,
    ppm_simple.hpp:53-63), so there the forest must equal its three mirror images after EVERY cycle -- no slack."""
    ov = CFG5 + ["parthenon/time/tlim=1.0"]
    s = _sim("blast_3d_amr", ov, strict=True).initialize()
    assert _forest_is_reflection_symmetric(s, 4)
    for _ in range(40):
        s.step()
        assert _forest_is_reflection_symmetric(s, 4)
    assert s.amr_stats()[0] > 0


def test_config5_with_first_order_flux_correct_enabled():
    """hydro/first_order_flux_correct = true on the same deck: every stage with gam0 = 0 runs as the
    optimistic fused stage and is tested like FirstOrderFluxCorrect's trial update before the
    coarse-fine correction.  While no cell fails the test the run must equal the run without the
    option bit for bit (parity build); the number of corrected cells is reported either way."""
    ov = CFG5 + ["parthenon/time/tlim=1.0"]
    a = _sim("blast_3d_amr", ov + ["hydro/first_order_flux_correct=true"], strict=True).initialize()
    b = _sim("blast_3d_amr", ov + ["hydro/first_order_flux_correct=false"], strict=True).initialize()
    for _ in range(60):
        a.step()
        b.step()
    assert a.time == b.time
    assert a.refresh_info().nblocks_total == b.refresh_info().nblocks_total
    if a.fofc_count == 0 and a.fofc_fallback_stages == 0:
        for lb in range(a.info.nblocks_local):
            assert np.array_equal(a.read_block(lb, "cons"), b.read_block(lb, "cons"))


def test_config5_mesh_without_flux_correction_conserves_to_round_off():
    """the same 4-level 64^3 / 16^3 MHD PPM + HLLD mesh and scheme on a blast the scheme keeps positive
    by itself (ambient pressure 1, ratio 100, a uniform oblique field): no first-order correction, so
    mass, momentum-free energy and the magnetic flux are conserved to round-off across all
    coarse-fine faces while the mesh regrids"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_gpu_amr import _totals
    ov = [o for o in CFG5 if "first_order" not in o] + ["problem/blast/pressure_ambient=1.0",
                                                         "problem/blast/pressure_ratio=100.0", "parthenon/time/tlim=1.0"]
    s = _sim("blast_3d_amr", ov, strict=False).initialize()
    t0 = _totals(s)
    nb0 = s.refresh_info().nblocks_total
    for _ in range(40):
        s.step()
    t1 = _totals(s)
    refined, merged, maxlev, zc = s.amr_stats()
    assert maxlev == 3 and refined > 0
    assert abs(t1[0] - t0[0]) < 1e-12 * t0[0] and abs(t1[4] - t0[4]) < 1e-12 * t0[4]
    assert _forest_octant_symmetric_within(s, 4)
    assert s.refresh_info().nblocks_total != nb0 or refined > 0


# ---- round-1 advisor findings: regression tests -----------------------------------------------------------
def test_fofc_with_pressure_floor_tests_the_unfloored_trial_update(oracle):
    """FirstOrderFluxCorrect tests the trial update BEFORE any floor acts (hydro.cpp:1283-1306; floors
    belong to the ConsToPrim after the stage).  With hydro/pfloor set the optimistic fused stage must
    therefore not floor its result before the test: the default (fused) driver, the flux-array driver
    and the oracle agree bit for bit, and corrections do happen."""
    pf = 1e-6
    ov = ["parthenon/mesh/nx1=32", "parthenon/mesh/nx2=32", "parthenon/mesh/nx3=32", "parthenon/meshblock/nx1=16",
          "parthenon/meshblock/nx2=16", "parthenon/meshblock/nx3=16", "hydro/first_order_flux_correct=true",
          "problem/blast/radius_outer=0.1", "problem/blast/radius_inner=0.0", "problem/blast/pressure_ratio=1e12",
          "problem/blast/density_ratio=1.0", "hydro/fluid=glmmhd", "hydro/reconstruction=ppm", "hydro/riemann=hlld",
          "parthenon/time/integrator=vl2", "parthenon/mesh/nghost=3", "parthenon/time/cfl=0.45", "hydro/pfloor=%r" % pf]
    a = _sim("blast", ov, strict=True).initialize()
    b = _sim("blast", ov, strict=True)
    b.set_fused(False)
    b.initialize()
    o = oracle.Sim(fluid="glmmhd", recon="ppm", riemann="hlld", integrator="vl2", nx=(32, 32, 32), mb=(16, 16, 16), ng=3,
                   xmin=(-0.5, -0.5, -0.5), xmax=(0.5, 0.5, 0.5), cfl=0.45, fofc=True, nthreads=os.cpu_count(),
                   eos=oracle.make_eos(GAMMA_DECK, pfloor=pf))
    o.pgen("blast", radius_outer=0.1, radius_inner=0.0, pressure_ambient=0.001, pressure_ratio=1e12, density_ratio=1.0)
    for _ in range(25):
        a.step()
        b.step()
        o.step()
    assert a.fofc_count == o.fofc_count == b.fofc_count and a.fofc_count > 0
    assert a.time == o.time and a.dt == o.dt
    ua = a.gather("cons")
    assert np.array_equal(ua, b.gather("cons"))
    assert np.array_equal(ua, o.gather_cons())


def test_reflecting_walls_are_refused_for_mhd():
    """src/bvals/boundary_conditions_apk.hpp:47-50"""
    from athenapk_amd import lib as L
    with pytest.raises(L.ApkError, match="Reflecting boundary conditions for MHD"):
        _sim("orszag_tang", ["parthenon/mesh/ix1_bc=reflecting", "parthenon/mesh/ox1_bc=reflecting"])
