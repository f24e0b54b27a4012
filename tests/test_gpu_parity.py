"""GPU parity tests: every call goes through the C-ABI of libapk_amd[_strict].so and is
compared with the CPU oracle on the same seeded inputs.

  strict build (-ffp-contract=off): BIT-EXACT agreement with the oracle is required.
  default build (FMA contraction):  |diff| <= 1e-12 * scale  (north_star tolerance).
"""
import numpy as np
import pytest

import helpers as H
from helpers import FUSABLE, NGHOST, NHYDRO, REGISTRY

pytestmark = pytest.mark.gpu

GAMMA = 5.0 / 3.0
C_H = 1.9
FAST_TOL = 1e-12


def _ctx(request, strict):
    return request.getfixturevalue("gpu_ctx_strict" if strict else "gpu_ctx_fast")


def _cmp(got, want, strict, what):
    if strict:
        # NaNs (e.g. sqrt of a negative reconstructed pressure on rough data) must appear in the
        # same places on both sides; everything else must agree bit for bit
        if not np.array_equal(got, want, equal_nan=True):
            bad = np.argwhere((got != want) & ~(np.isnan(got) & np.isnan(want)))
            raise AssertionError("%s: %d entries differ bitwise, first at %s: %r vs %r" % (
                what, len(bad), bad[0], got[tuple(bad[0])], want[tuple(bad[0])]))
    else:
        scale = np.max(np.abs(want)) + 1e-300
        err = np.max(np.abs(got - want)) / scale
        assert err <= FAST_TOL, "%s: max scaled error %.3e" % (what, err)


def _case(fluid, recon, nx, kind="smooth", nscalars=0, nblocks=2, seed=11, dx=(0.1, 0.07, 0.13)):
    ng = NGHOST[recon] if recon != "dc" else 2
    prim = H.random_prim(fluid, nx, ng, nscalars=nscalars, seed=seed, kind=kind, nblocks=nblocks)
    g = H.geom(fluid, nx, ng, nscalars, dx)
    return ng, prim, g


# ---- Hydro::CalculateFluxes: every entry of the registry, 3-D ------------------------------------
@pytest.mark.parametrize("strict", [True, False], ids=["strict", "fma"])
@pytest.mark.parametrize("fluid,recon,riemann", REGISTRY)
def test_calculate_fluxes_registry_3d(request, fluid, recon, riemann, strict):
    from athenapk_amd import hydro
    ctx = _ctx(request, strict)
    nx = (16, 8, 8)
    ng, prim, g = _case(fluid, recon, nx, kind="smooth")
    md = hydro.MeshData(ctx, nx, ng, NHYDRO[fluid], dx=tuple(g.dx), nblocks=prim.shape[0], prim=prim)
    hydro.CalculateFluxes(md, fluid, recon, riemann, hydro.L.make_eos(GAMMA), C_H)
    want = H.orc_fluxes(fluid, recon, riemann, g, prim, GAMMA, C_H)
    for d in range(3):
        _cmp(md.flux_host(d), want[d], strict, "flux%d %s/%s/%s" % (d + 1, fluid, recon, riemann))


@pytest.mark.parametrize("nx", [(16, 8, 8), (8, 8, 8), (40, 12, 1), (72, 6, 4)], ids=["16x8x8", "8x8x8", "2d", "wide"])
@pytest.mark.parametrize("fluid,recon,riemann", [("euler", "plm", "hlle"), ("glmmhd", "ppm", "hlld")])
def test_calculate_fluxes_tight_equals_the_reference_extents_on_interior_faces(request, fluid, recon, riemann, nx):
    """apk_calculate_fluxes_tight: same values on every face of an interior cell, nothing written
    beyond CalculateFluxesTight's loop limits; narrow blocks (flattened lanes) and wide ones"""
    from athenapk_amd import hydro
    ctx = _ctx(request, True)
    ng, prim, g = _case(fluid, recon, nx, kind="rough", seed=5)
    eos = hydro.L.make_eos(GAMMA)
    full = hydro.MeshData(ctx, nx, ng, NHYDRO[fluid], dx=tuple(g.dx), nblocks=prim.shape[0], prim=prim)
    tight = hydro.MeshData(ctx, nx, ng, NHYDRO[fluid], dx=tuple(g.dx), nblocks=prim.shape[0], prim=prim)
    hydro.CalculateFluxes(full, fluid, recon, riemann, eos, C_H)
    hydro.CalculateFluxes(tight, fluid, recon, riemann, eos, C_H, tight=True)
    ndim = 3 if nx[2] > 1 else (2 if nx[1] > 1 else 1)
    act = [True, nx[1] > 1, nx[2] > 1]
    for d in range(ndim):
        a, b = full.flux_host(d), tight.flux_host(d)
        sl = [slice(None), slice(None)]
        for q in (2, 1, 0):          # array axes k, j, i
            if not act[q]:
                sl.append(slice(None))
            else:
                sl.append(slice(ng, ng + nx[q] + (1 if q == d else 0)))
        sl = tuple(sl)
        assert np.array_equal(a[sl], b[sl], equal_nan=True)   # (rough data: NaNs in the same places)
        # (CalculateFluxesTight is one loop nest over [s, e + 1] in every active direction)
        wide = tuple(sl[:2]) + tuple(slice(None) if not act[q] else slice(ng, ng + nx[q] + 1) for q in (2, 1, 0))
        mask = np.ones(b.shape, bool)
        mask[wide] = False
        assert np.all(b[mask] == 0.0) and np.array_equal(a[wide], b[wide], equal_nan=True)


@pytest.mark.parametrize("kind", ["rough", "shock"])
@pytest.mark.parametrize("fluid,recon,riemann", [("euler", "ppm", "hllc"), ("euler", "wenoz", "hlle"),
                                                 ("glmmhd", "ppm", "hlld"), ("glmmhd", "limo3", "hlld"),
                                                 ("glmmhd", "plm", "hlle")])
def test_calculate_fluxes_discontinuous_data_bitexact(request, fluid, recon, riemann, kind):
    """Limiter / solver branches (PPM extremum logic, HLLD degenerate states, supersonic
    upwinding) on rough and shocked data."""
    from athenapk_amd import hydro
    ctx = _ctx(request, True)
    nx = (24, 6, 5)
    ng, prim, g = _case(fluid, recon, nx, kind=kind, seed=23)
    md = hydro.MeshData(ctx, nx, ng, NHYDRO[fluid], dx=tuple(g.dx), nblocks=prim.shape[0], prim=prim)
    hydro.CalculateFluxes(md, fluid, recon, riemann, hydro.L.make_eos(GAMMA), C_H)
    want = H.orc_fluxes(fluid, recon, riemann, g, prim, GAMMA, C_H)
    for d in range(3):
        _cmp(md.flux_host(d), want[d], True, "flux%d" % (d + 1))


@pytest.mark.parametrize("nx", [(20, 12, 1), (33, 1, 1)], ids=["2d", "1d"])
@pytest.mark.parametrize("fluid,recon,riemann", [("euler", "plm", "hlle"), ("glmmhd", "ppm", "hlld"),
                                                 ("glmmhd", "dc", "llf")])
def test_calculate_fluxes_collapsed_dimensions(request, nx, fluid, recon, riemann):
    from athenapk_amd import hydro
    ctx = _ctx(request, True)
    ng, prim, g = _case(fluid, recon, nx, kind="rough", seed=5)
    md = hydro.MeshData(ctx, nx, ng, NHYDRO[fluid], dx=tuple(g.dx), nblocks=prim.shape[0], prim=prim)
    hydro.CalculateFluxes(md, fluid, recon, riemann, hydro.L.make_eos(GAMMA), C_H)
    want = H.orc_fluxes(fluid, recon, riemann, g, prim, GAMMA, C_H)
    ndim = 2 if nx[1] > 1 else 1
    for d in range(ndim):
        _cmp(md.flux_host(d), want[d], True, "flux%d" % (d + 1))


@pytest.mark.parametrize("fluid,recon,riemann", [("euler", "plm", "hllc"), ("glmmhd", "wenoz", "hlld"),
                                                 ("euler", "dc", "llf")])
def test_passive_scalar_fluxes(request, fluid, recon, riemann):
    """hydro.cpp:1088-1097: F_n = F_rho * (F_rho >= 0 ? wl_n : wr_n)."""
    from athenapk_amd import hydro
    ctx = _ctx(request, True)
    nx = (12, 6, 4)
    ng, prim, g = _case(fluid, recon, nx, kind="smooth", nscalars=3, seed=9)
    md = hydro.MeshData(ctx, nx, ng, NHYDRO[fluid], nscalars=3, dx=tuple(g.dx), nblocks=prim.shape[0], prim=prim)
    hydro.CalculateFluxes(md, fluid, recon, riemann, hydro.L.make_eos(GAMMA), C_H)
    want = H.orc_fluxes(fluid, recon, riemann, g, prim, GAMMA, C_H)
    for d in range(3):
        _cmp(md.flux_host(d), want[d], True, "flux%d" % (d + 1))


# ---- UpdateWithFluxDivergence / DednerSource / ConsToPrim / dt / history ------------------------------
@pytest.mark.parametrize("strict", [True, False], ids=["strict", "fma"])
@pytest.mark.parametrize("nx", [(16, 8, 8), (16, 10, 1), (24, 1, 1)], ids=["3d", "2d", "1d"])
def test_update_with_flux_divergence(request, strict, nx):
    from athenapk_amd import hydro
    ctx = _ctx(request, strict)
    fluid, ng = "glmmhd", 3
    g = H.geom(fluid, nx, ng, 1, (0.1, 0.07, 0.13))
    rng = np.random.default_rng(3)
    shp = (2,) + H.block_shape(nx, ng, 10)
    u0, u1 = rng.uniform(-1, 1, shp), rng.uniform(-1, 1, shp)
    fl = [rng.uniform(-1, 1, shp) for _ in range(3)]
    m0 = hydro.MeshData(ctx, nx, ng, 9, nscalars=1, dx=tuple(g.dx), nblocks=2, cons=u0)
    for d in range(m0.ndim):
        m0.flux[d].copy_(hydro.torch.from_numpy(fl[d]))
    m1 = hydro.MeshData(ctx, nx, ng, 9, nscalars=1, dx=tuple(g.dx), nblocks=2, cons=u1, with_flux=False)
    hydro.UpdateWithFluxDivergence(m0, m1, 0.25, 0.75, 0.0123)
    want = H.orc_update(g, u0, u1, fl, 0.25, 0.75, 0.0123)
    _cmp(m0.cons_host(), want, strict, "update")


@pytest.mark.parametrize("extended", [False, True], ids=["plain", "extended"])
@pytest.mark.parametrize("nx", [(16, 8, 8), (16, 10, 1)], ids=["3d", "2d"])
def test_dedner_source(request, extended, nx):
    from athenapk_amd import hydro
    ctx = _ctx(request, True)
    ng = 2
    g = H.geom("glmmhd", nx, ng, 0, (0.1, 0.07, 0.13))
    prim = H.random_prim("glmmhd", nx, ng, seed=4, kind="rough", nblocks=2)
    cons = H.prim_to_cons("glmmhd", prim, GAMMA)
    md = hydro.MeshData(ctx, nx, ng, 9, dx=tuple(g.dx), nblocks=2, cons=cons, prim=prim, with_flux=False)
    hydro.DednerSource(md, extended, 0.1, C_H, 0.07, 0.011)
    want = H.orc_dedner(g, cons, prim, extended, 0.1, C_H, 0.07, 0.011)
    _cmp(md.cons_host(), want, True, "dedner")


@pytest.mark.parametrize("fluid", ["euler", "glmmhd"])
@pytest.mark.parametrize("floors", ["off", "on"])
def test_cons_to_prim(request, oracle, fluid, floors):
    from athenapk_amd import hydro
    ctx = _ctx(request, True)
    nx, ng = (12, 6, 5), 2
    g = H.geom(fluid, nx, ng, 2)
    w = H.random_prim(fluid, nx, ng, nscalars=2, seed=8, kind="rough", nblocks=2)
    u = H.prim_to_cons(fluid, w, 1.4)
    kw = {}
    if floors == "on":
        u[0, 4, 2, 3, 4] = 1e-7      # negative pressure -> pressure floor rewrites E
        u[1, 0, 1, 1, 1] = 1e-9      # tiny density -> density floor
        u[0, 1, 3, 2, 5] = 50.0      # huge momentum -> velocity ceiling
        kw = dict(pfloor=1e-3, dfloor=1e-2, vceil=10.0)
    md = hydro.MeshData(ctx, nx, ng, NHYDRO[fluid], nscalars=2, nblocks=2, cons=u, with_flux=False)
    ctx.poll_flags()
    hydro.ConservedToPrimitive(md, fluid, hydro.L.make_eos(1.4, **kw))
    u_want, w_want, bad = H.orc_c2p(fluid, g, u, oracle.make_eos(1.4, **kw))
    assert bad == 0
    _cmp(md.prim_host(), w_want, True, "prim")
    _cmp(md.cons_host(), u_want, True, "cons after floors")
    assert ctx.poll_flags() == 0


@pytest.mark.parametrize("fluid", ["euler", "glmmhd"])
def test_copy_plans_with_and_without_cons_to_prim(request, oracle, fluid):
    """apk_copy_plan_run / _run_c2p / _run_c2p_prim_only on an x1 ghost strip (3 cells of every row: the worst box for
    the memory system), an x3 slab filled from a compact message buffer and a reflecting strip with the normal momentum
    flipped: the copies bit for bit, the primitives the oracle's ConservedToPrimitive of the copied values, and with
    prim_only nothing written at the destination itself."""
    import torch
    from athenapk_amd import hydro
    ctx = _ctx(request, True)
    nx, ng, nv = (12, 6, 5), 3, NHYDRO[fluid]
    g = H.geom(fluid, nx, ng, 0)
    w = H.random_prim(fluid, nx, ng, nscalars=0, seed=21, kind="rough", nblocks=2)
    u = H.prim_to_cons(fluid, w, 1.4)
    ni, nj, nk = nx[0] + 2 * ng, nx[1] + 2 * ng, nx[2] + 2 * ng
    sj, sk, sn = ni, ni * nj, ni * nj * nk
    per = nv * sn
    msg_ext = (nx[0], nx[1], ng)
    msg_cells = msg_ext[0] * msg_ext[1] * msg_ext[2]
    msg = H.prim_to_cons(fluid, H.random_prim(fluid, (msg_ext[0], msg_ext[1], msg_ext[2]), 0, nscalars=0, seed=22, kind="rough", nblocks=1), 1.4)
    msg = np.ascontiguousarray(msg.reshape(nv, msg_ext[2], msg_ext[1], msg_ext[0]))
    eos = hydro.L.make_eos(1.4)

    def run(mode):
        cons = torch.from_numpy(u.copy()).cuda()
        prim = torch.full_like(cons, -777.0)
        buf = torch.from_numpy(msg.copy()).cuda()
        base = cons.data_ptr()
        blk = lambda b, i, j, k: base + 8 * (b * per + k * sk + j * sj + i)
        big = (1, sj, sk, sn)
        regs = [
            # block 0's upper x1 ghost strip <- block 1's first interior columns
            dict(src=blk(1, ng, ng, ng), dst=blk(0, ng + nx[0], ng, ng), ext=(ng, nx[1], nx[2]), nvar=nv, src_stride=big, dst_stride=big),
            # block 1's lower x3 ghost slab <- a compact message
            dict(src=buf.data_ptr(), dst=blk(1, ng, ng, 0), ext=msg_ext, nvar=nv,
                 src_stride=(1, msg_ext[0], msg_ext[0] * msg_ext[1], msg_cells), dst_stride=big),
            # block 1's lower x2 ghost strip: reflecting (mirror of the first interior rows, normal momentum flipped)
            dict(src=blk(1, ng, 2 * ng - 1, ng), dst=blk(1, ng, 0, ng), ext=(nx[0], ng, nx[2]), nvar=nv,
                 src_stride=(1, -sj, sk, sn), dst_stride=big, flip_var=2),
        ]
        plan = hydro.CopyPlan(ctx, regs)
        ctx.poll_flags()
        if mode == "copy":
            plan.run()
        else:
            plan.run_c2p(fluid, eos, (prim.data_ptr() - cons.data_ptr()) // 8, prim_only=(mode == "prim_only"))
        torch.cuda.synchronize()
        assert ctx.poll_flags() == 0
        return cons.cpu().numpy(), prim.cpu().numpy()

    want = u.copy()
    want[0, :, ng:ng + nx[2], ng:ng + nx[1], ng + nx[0]:] = u[1, :, ng:ng + nx[2], ng:ng + nx[1], ng:2 * ng]
    want[1, :, 0:ng, ng:ng + nx[1], ng:ng + nx[0]] = msg
    want[1, :, ng:ng + nx[2], 0:ng, ng:ng + nx[0]] = u[1, :, ng:ng + nx[2], 2 * ng - 1:ng - 1:-1, ng:ng + nx[0]]
    want[1, 2, ng:ng + nx[2], 0:ng, ng:ng + nx[0]] *= -1.0
    touched = want != u
    _, w_want, bad = H.orc_c2p(fluid, g, want, oracle.make_eos(1.4))
    assert bad == 0 and touched.any()
    cell_touched = touched.any(axis=1, keepdims=True) & np.ones_like(touched)
    got_c, got_p = run("copy")
    assert np.array_equal(got_c, want) and np.all(got_p == -777.0)
    got_c, got_p = run("c2p")
    assert np.array_equal(got_c, want)
    assert np.array_equal(got_p[cell_touched], w_want[cell_touched]) and np.all(got_p[~cell_touched] == -777.0)
    got_c, got_p = run("prim_only")
    assert np.array_equal(got_c, u)  # (nothing stored at the destination itself)
    assert np.array_equal(got_p[cell_touched], w_want[cell_touched]) and np.all(got_p[~cell_touched] == -777.0)


@pytest.mark.parametrize("nx", [(12, 6, 5), (16, 8, 1)], ids=["3d", "2d"])
def test_cons_to_prim_faces_converts_everything_but_edges_and_corners(request, nx):
    """apk_cons_to_prim_faces: the full ConsToPrim's bits on every cell with at most one ghost coordinate, nothing
    written (and no flag raised by the garbage there) behind edges and corners"""
    from athenapk_amd import hydro
    ctx = _ctx(request, True)
    ng = 2
    w = H.random_prim("glmmhd", nx, ng, seed=12, kind="rough", nblocks=2)
    u = H.prim_to_cons("glmmhd", w, 1.4)
    eos = hydro.L.make_eos(1.4)
    full = hydro.MeshData(ctx, nx, ng, 9, nblocks=2, cons=u, with_flux=False)
    hydro.ConservedToPrimitive(full, "glmmhd", eos)
    act = [True, nx[1] > 1, nx[2] > 1]
    K, J, I = np.meshgrid(*[np.arange(n + 2 * ng if a else 1) for n, a in zip(nx[::-1], act[::-1])], indexing="ij")
    nghost = sum(((c < ng) | (c >= ng + n)).astype(int) if a else 0 for c, n, a in ((I, nx[0], True), (J, nx[1], act[1]), (K, nx[2], act[2])))
    corner = np.broadcast_to(nghost > 1, u.shape[2:])
    u2 = u.copy()
    u2[:, 0][:, corner] = -1.0                       # negative densities behind edges and corners
    md = hydro.MeshData(ctx, nx, ng, 9, nblocks=2, cons=u2, prim=np.full_like(u, -3.0), with_flux=False)
    ctx.poll_flags()
    hydro.ConservedToPrimitiveFaces(md, "glmmhd", eos)
    got, want = md.prim_host(), full.prim_host()
    assert np.array_equal(got[:, :, ~corner], want[:, :, ~corner]) and np.all(got[:, :, corner] == -3.0)
    assert ctx.poll_flags() == 0 and corner.any()
    if nx[2] == 1:
        return
    # apk_cons_to_prim_faces_skip: nor behind the faces a face table marks as read from the neighbour's interior
    import torch
    tab = np.array([[1, -1, -1, 0, -1, -1], [-1, 0, 1, -1, -1, 1]], dtype=np.int32)
    behind = np.zeros((2,) + u.shape[2:], dtype=bool)
    zones = ((I < ng), (I >= ng + nx[0]), (J < ng), (J >= ng + nx[1]), (K < ng), (K >= ng + nx[2]))
    for b in range(2):
        for f in range(6):
            if tab[b, f] >= 0:
                behind[b] |= np.broadcast_to(zones[f] & (nghost == 1), u.shape[2:])
    u3 = u2.copy()
    for b in range(2):
        u3[b, 0][behind[b]] = -1.0                   # garbage in the zones nobody fills
    md = hydro.MeshData(ctx, nx, ng, 9, nblocks=2, cons=u3, prim=np.full_like(u, -3.0), with_flux=False)
    hydro.ConservedToPrimitiveFaces(md, "glmmhd", eos, face_neighbor=torch.from_numpy(tab).cuda())
    got = md.prim_host()
    for b in range(2):
        skip = corner | behind[b]
        assert np.array_equal(got[b][:, ~skip], want[b][:, ~skip]) and np.all(got[b][:, skip] == -3.0) and behind[b].any()
    assert ctx.poll_flags() == 0


@pytest.mark.parametrize("strict", [True, False], ids=["strict", "fma"])
def test_cons_to_prim_latches_negative_state_flags(request, strict):
    """adiabatic_hydro.hpp:77-79,111-113: PARTHENON_REQUIRE -> latched device flag.  A NaN state
    must be flagged too, also in the default build (compiled with -fno-honor-nans)."""
    from athenapk_amd import hydro
    ctx = _ctx(request, strict)
    nx, ng = (8, 4, 4), 2
    w = H.random_prim("euler", nx, ng, seed=8, kind="rough")
    u = H.prim_to_cons("euler", w, 1.4)
    u[0, 0, 2, 2, 2] = -1.0
    md = hydro.MeshData(ctx, nx, ng, 5, cons=u, with_flux=False)
    ctx.poll_flags()
    hydro.ConservedToPrimitive(md, "euler", hydro.L.make_eos(1.4))
    assert ctx.poll_flags() & hydro.L.FLAG_NEG_DENSITY
    u[0, 0, 2, 2, 2] = 1.0
    u[0, 4, 2, 2, 2] = -5.0
    md = hydro.MeshData(ctx, nx, ng, 5, cons=u, with_flux=False)
    hydro.ConservedToPrimitive(md, "euler", hydro.L.make_eos(1.4))
    assert ctx.poll_flags() == hydro.L.FLAG_NEG_PRESSURE
    u[0, 4, 2, 2, 2] = 5.0
    u[0, 0, 3, 2, 1] = np.nan
    md = hydro.MeshData(ctx, nx, ng, 5, cons=u, with_flux=False)
    hydro.ConservedToPrimitive(md, "euler", hydro.L.make_eos(1.4))
    assert ctx.poll_flags() & hydro.L.FLAG_NEG_DENSITY
    u[0, 0, 3, 2, 1] = 1.0
    u[0, 4, 3, 2, 1] = np.nan
    md = hydro.MeshData(ctx, nx, ng, 5, cons=u, with_flux=False)
    hydro.ConservedToPrimitive(md, "euler", hydro.L.make_eos(1.4))
    assert ctx.poll_flags() & hydro.L.FLAG_NEG_PRESSURE


@pytest.mark.parametrize("fluid", ["euler", "glmmhd"])
@pytest.mark.parametrize("nx", [(16, 8, 8), (16, 10, 1), (40, 1, 1)], ids=["3d", "2d", "1d"])
def test_estimate_timestep(request, fluid, nx):
    from athenapk_amd import hydro
    ctx = _ctx(request, True)
    ng = 2
    g = H.geom(fluid, nx, ng, 0, (0.1, 0.07, 0.13))
    prim = H.random_prim(fluid, nx, ng, seed=6, kind="rough", nblocks=3)
    md = hydro.MeshData(ctx, nx, ng, NHYDRO[fluid], dx=tuple(g.dx), nblocks=3, prim=prim, with_flux=False)
    dt = hydro.EstimateTimestep(md, fluid, hydro.L.make_eos(GAMMA), 0.3)
    assert dt == 0.3 * H.orc_min_dt(fluid, g, prim, GAMMA)


@pytest.mark.parametrize("strict", [True, False], ids=["strict", "fma"])
@pytest.mark.parametrize("fluid", ["euler", "glmmhd"])
@pytest.mark.parametrize("nx", [(16, 8, 8), (64, 6, 5), (16, 10, 1), (40, 1, 1)], ids=["3d", "3d_wide", "2d", "1d"])
def test_cons_to_prim_with_time_step_estimate(request, fluid, nx, strict):
    """apk_cons_to_prim_dt = apk_cons_to_prim + apk_estimate_timestep in one pass: the same primitives in every cell
    (ghost zones included), and the minimum over the INTERIOR cells only -- the ghost zones hold faster states here --
    equal to the oracle's estimate on the oracle's primitives (bit for bit in the parity build)."""
    from athenapk_amd import hydro
    ctx = _ctx(request, strict)
    ng = 2
    g = H.geom(fluid, nx, ng, 0, (0.1, 0.07, 0.13))
    w = H.random_prim(fluid, nx, ng, seed=16, kind="rough", nblocks=3)
    act = [True, nx[1] > 1, nx[2] > 1]
    inner = tuple(slice(ng, -ng) if a else slice(None) for a in act[::-1])
    fast = w.copy()
    fast[:, 1:4] *= 50.0                               # ghost cells that would win the minimum if they were read
    fast[(slice(None), slice(None)) + inner] = w[(slice(None), slice(None)) + inner]
    u = H.prim_to_cons(fluid, fast, GAMMA)
    eos = hydro.L.make_eos(GAMMA)
    ref = hydro.MeshData(ctx, nx, ng, NHYDRO[fluid], dx=tuple(g.dx), nblocks=3, cons=u, with_flux=False)
    hydro.ConservedToPrimitive(ref, fluid, eos)
    md = hydro.MeshData(ctx, nx, ng, NHYDRO[fluid], dx=tuple(g.dx), nblocks=3, cons=u, with_flux=False)
    dt = hydro.ConservedToPrimitiveDt(md, fluid, eos, 0.3)
    assert np.array_equal(md.prim_host(), ref.prim_host())
    want = hydro.EstimateTimestep(ref, fluid, eos, 0.3)
    if strict:
        _, w_orc, bad = H.orc_c2p(fluid, g, u, H.O.make_eos(GAMMA))
        assert bad == 0 and np.array_equal(w_orc, ref.prim_host())
        assert dt == want == 0.3 * H.orc_min_dt(fluid, g, w_orc, GAMMA)
    else:
        assert abs(dt - want) <= 4e-16 * want
    # ghost_depth = 1: the same estimate; primitives only in the cells at most one layer outside the interior
    K, J, I = np.meshgrid(*[np.arange(n + 2 * ng if a else 1) for n, a in zip(nx[::-1], act[::-1])], indexing="ij")
    deep = np.zeros(u.shape[2:], dtype=bool)
    for c, n, a in ((I, nx[0], True), (J, nx[1], act[1]), (K, nx[2], act[2])):
        if a:
            deep |= (c < ng - 1) | (c > ng + n)
    md = hydro.MeshData(ctx, nx, ng, NHYDRO[fluid], dx=tuple(g.dx), nblocks=3, cons=u, prim=np.full_like(u, -3.0), with_flux=False)
    assert hydro.ConservedToPrimitiveDt(md, fluid, eos, 0.3, ghost_depth=1) == dt
    got = md.prim_host()
    assert np.array_equal(got[:, :, ~deep], ref.prim_host()[:, :, ~deep]) and np.all(got[:, :, deep] == -3.0) and deep.any()
    # apk_cons_to_prim_dt_skip: nor the ghost cells straight behind the faces a face table joins to another block
    import torch
    tab = np.array([[1, -1, 2, 0, -1, -1], [-1, 0, -1, -1, 1, 2], [2, 2, -1, 1, 0, -1]], dtype=np.int32)
    nghost = ((I < ng) | (I >= ng + nx[0])).astype(int) + (act[1] & ((J < ng) | (J >= ng + nx[1]))) + (act[2] & ((K < ng) | (K >= ng + nx[2])))
    zones = ((I < ng), (I >= ng + nx[0]), act[1] & (J < ng), act[1] & (J >= ng + nx[1]), act[2] & (K < ng), act[2] & (K >= ng + nx[2]))
    for depth in (1, -1):
        md = hydro.MeshData(ctx, nx, ng, NHYDRO[fluid], dx=tuple(g.dx), nblocks=3, cons=u, prim=np.full_like(u, -3.0), with_flux=False)
        assert hydro.ConservedToPrimitiveDt(md, fluid, eos, 0.3, ghost_depth=depth, face_neighbor=torch.from_numpy(tab).cuda()) == dt
        got = md.prim_host()
        for b in range(3):
            left = np.broadcast_to(deep if depth == 1 else np.zeros_like(deep), u.shape[2:]).copy()
            for f in range(6):
                if tab[b, f] >= 0:
                    left |= np.broadcast_to(zones[f] & (nghost == 1), u.shape[2:])
            assert np.array_equal(got[b][:, ~left], ref.prim_host()[b][:, ~left]) and np.all(got[b][:, left] == -3.0) and left.any()
    # apk_cons_to_prim_dt_select: the same estimate, and of the primitives only those named -- in the cells of the box
    for depth, mask in ((1, 1 << 4), (0, 0), (1, 0b00110), (1, (1 << NHYDRO[fluid]) - 1)):
        md = hydro.MeshData(ctx, nx, ng, NHYDRO[fluid], dx=tuple(g.dx), nblocks=3, cons=u, prim=np.full_like(u, -3.0), with_flux=False)
        assert hydro.ConservedToPrimitiveDt(md, fluid, eos, 0.3, ghost_depth=depth, store_vars=mask) == dt
        got = md.prim_host()
        box = ~deep if depth == 1 else np.broadcast_to(((I >= ng) & (I < ng + nx[0])) & (~np.bool_(act[1]) | ((J >= ng) & (J < ng + nx[1])))
                                                       & (~np.bool_(act[2]) | ((K >= ng) & (K < ng + nx[2]))), deep.shape)
        for n in range(NHYDRO[fluid]):
            if (mask >> n) & 1:
                assert np.array_equal(got[:, n][:, box], ref.prim_host()[:, n][:, box]) and np.all(got[:, n][:, ~box] == -3.0)
            else:
                assert np.all(got[:, n] == -3.0)
    with pytest.raises(hydro.L.ApkError):  # (a floor writes conserved values back: the full pass only)
        hydro.ConservedToPrimitiveDt(md, fluid, hydro.L.make_eos(GAMMA, dfloor=1e-9), 0.3, ghost_depth=1, store_vars=1 << 4)
    # apk_cons_to_prim_faces_dt: what apk_cons_to_prim_faces converts, and the same estimate
    a = hydro.MeshData(ctx, nx, ng, NHYDRO[fluid], dx=tuple(g.dx), nblocks=3, cons=u, prim=np.full_like(u, -3.0), with_flux=False)
    b = hydro.MeshData(ctx, nx, ng, NHYDRO[fluid], dx=tuple(g.dx), nblocks=3, cons=u, prim=np.full_like(u, -3.0), with_flux=False)
    hydro.ConservedToPrimitiveFaces(a, fluid, eos)
    assert hydro.ConservedToPrimitiveFacesDt(b, fluid, eos, 0.3) == dt
    assert np.array_equal(a.prim_host(), b.prim_host())


@pytest.mark.parametrize("fluid", ["euler", "glmmhd"])
def test_history(request, fluid):
    from athenapk_amd import hydro
    ctx = _ctx(request, True)
    nx, ng = (16, 8, 8), 2
    g = H.geom(fluid, nx, ng, 0, (0.1, 0.07, 0.13))
    prim = H.random_prim(fluid, nx, ng, seed=12, kind="smooth", nblocks=2)
    cons = H.prim_to_cons(fluid, prim, GAMMA)
    md = hydro.MeshData(ctx, nx, ng, NHYDRO[fluid], dx=tuple(g.dx), nblocks=2, cons=cons, with_flux=False)
    got = hydro.HydroHst(md, fluid)
    want = H.orc_history(fluid, g, cons)
    # a sum: order of additions differs (as it does between Kokkos back ends)
    np.testing.assert_allclose(got, want, rtol=1e-13, atol=1e-15)


# ---- FirstOrderFluxCorrect ------------------------------------------------------------------------
@pytest.mark.parametrize("fluid", ["euler", "glmmhd"])
def test_first_order_flux_correct(request, fluid):
    from athenapk_amd import hydro
    ctx = _ctx(request, True)
    nx, ng = (16, 8, 6), 2
    g = H.geom(fluid, nx, ng, 0, (0.1, 0.1, 0.1))
    prim = H.random_prim(fluid, nx, ng, seed=31, kind="shock", nblocks=2)
    prim[:, 4] *= 1e-3  # very cold: a large explicit step drives the trial pressure negative
    cons = H.prim_to_cons(fluid, prim, GAMMA)
    fl = H.orc_fluxes(fluid, "plm", "hlle", g, prim, GAMMA, C_H)
    beta_dt = 0.3 if fluid == "euler" else 0.06  # large enough to drive trial states negative
    want_fl, want_n = H.orc_fofc(fluid, g, cons, prim, cons, fl, GAMMA, C_H, 0.0, 1.0, beta_dt)
    assert want_n > 0, "test data must trigger corrections"
    m0 = hydro.MeshData(ctx, nx, ng, NHYDRO[fluid], dx=tuple(g.dx), nblocks=2, cons=cons, prim=prim)
    for d in range(3):
        m0.flux[d].copy_(hydro.torch.from_numpy(fl[d]))
    m1 = hydro.MeshData(ctx, nx, ng, NHYDRO[fluid], dx=tuple(g.dx), nblocks=2, cons=cons, with_flux=False)
    n = hydro.FirstOrderFluxCorrect(m0, m1, fluid, hydro.L.make_eos(GAMMA), C_H, 0.0, 1.0, beta_dt)
    assert n == want_n
    for d in range(3):
        _cmp(m0.flux_host(d), want_fl[d], True, "corrected flux%d" % (d + 1))


# ---- fused stage = CalculateFluxes + UpdateWithFluxDivergence + DednerSource -------------------------
@pytest.mark.parametrize("strict", [True, False], ids=["strict", "fma"])
@pytest.mark.parametrize("fluid,recon,riemann", FUSABLE)
def test_fused_stage_registry_3d(request, fluid, recon, riemann, strict):
    from athenapk_amd import hydro
    ctx = _ctx(request, strict)
    nx = (70, 9, 7)  # > 64 wide: exercises the x1 wave-overlap scheme and idle lanes
    ng, prim, g = _case(fluid, recon, nx, kind="smooth", seed=17)
    cons = H.prim_to_cons(fluid, prim, GAMMA)
    rng = np.random.default_rng(2)
    u1 = cons * (1.0 + 1e-3 * rng.standard_normal(cons.shape))
    ded = 1 if fluid == "glmmhd" else 0
    m0 = hydro.MeshData(ctx, nx, ng, NHYDRO[fluid], dx=tuple(g.dx), nblocks=prim.shape[0], cons=cons, prim=prim,
                        with_flux=False)
    m1 = hydro.MeshData(ctx, nx, ng, NHYDRO[fluid], dx=tuple(g.dx), nblocks=prim.shape[0], cons=u1, with_flux=False)
    hydro.StageFused(m0, m1, fluid, recon, riemann, hydro.L.make_eos(GAMMA), C_H, 0.25, 0.75, 0.004, dedner=ded,
                     glmmhd_alpha=0.1, mindx=0.07)
    want = H.orc_stage(fluid, recon, riemann, g, cons, u1, prim, GAMMA, C_H, 0.25, 0.75, 0.004, dedner=ded,
                       alpha=0.1, mindx=0.07)
    _cmp(m0.cons_host(), want, strict, "fused stage %s/%s/%s" % (fluid, recon, riemann))


@pytest.mark.parametrize("nx", [(130, 12, 1), (200, 1, 1)], ids=["2d", "1d"])
@pytest.mark.parametrize("dedner", [1, 2], ids=["plain", "extended"])
def test_fused_stage_collapsed_dimensions_and_dedner(request, nx, dedner):
    from athenapk_amd import hydro
    ctx = _ctx(request, True)
    fluid, recon, riemann = "glmmhd", "ppm", "hlld"
    ng, prim, g = _case(fluid, recon, nx, kind="rough", seed=19)
    cons = H.prim_to_cons(fluid, prim, GAMMA)
    m0 = hydro.MeshData(ctx, nx, ng, 9, dx=tuple(g.dx), nblocks=prim.shape[0], cons=cons, prim=prim, with_flux=False)
    m1 = hydro.MeshData(ctx, nx, ng, 9, dx=tuple(g.dx), nblocks=prim.shape[0], cons=cons, with_flux=False)
    hydro.StageFused(m0, m1, fluid, recon, riemann, hydro.L.make_eos(GAMMA), C_H, 0.0, 1.0, 0.002, dedner=dedner,
                     glmmhd_alpha=0.1, mindx=0.07)
    want = H.orc_stage(fluid, recon, riemann, g, cons, cons, prim, GAMMA, C_H, 0.0, 1.0, 0.002, dedner=dedner,
                       alpha=0.1, mindx=0.07)
    _cmp(m0.cons_host(), want, True, "fused stage")


def test_fused_stage_3d_extended_dedner_and_first_stage_gam0_zero(request):
    from athenapk_amd import hydro
    ctx = _ctx(request, True)
    fluid, recon, riemann = "glmmhd", "wenoz", "hlld"
    nx = (64, 8, 8)
    ng, prim, g = _case(fluid, recon, nx, kind="shock", seed=29)
    cons = H.prim_to_cons(fluid, prim, GAMMA)
    m0 = hydro.MeshData(ctx, nx, ng, 9, dx=tuple(g.dx), nblocks=prim.shape[0], cons=cons, prim=prim, with_flux=False)
    m1 = hydro.MeshData(ctx, nx, ng, 9, dx=tuple(g.dx), nblocks=prim.shape[0], cons=cons, with_flux=False)
    hydro.StageFused(m0, m1, fluid, recon, riemann, hydro.L.make_eos(GAMMA), C_H, 0.0, 1.0, 0.003, dedner=2,
                     glmmhd_alpha=0.1, mindx=0.07)
    want = H.orc_stage(fluid, recon, riemann, g, cons, cons, prim, GAMMA, C_H, 0.0, 1.0, 0.003, dedner=2,
                       alpha=0.1, mindx=0.07)
    _cmp(m0.cons_host(), want, True, "fused stage")


# ---- error conventions of the boundary ----------------------------------------------------------------
def test_boundary_error_codes(request):
    from athenapk_amd import hydro
    from athenapk_amd import lib as L
    ctx = _ctx(request, True)
    nx = (8, 4, 4)
    prim = H.random_prim("euler", nx, 2, seed=1)
    md = hydro.MeshData(ctx, nx, 2, 5, prim=prim)
    with pytest.raises(L.ApkError) as e:  # ppm needs 3 ghost zones (hydro.cpp:444-447)
        hydro.CalculateFluxes(md, "euler", "ppm", "hlle", L.make_eos(1.4))
    assert e.value.code == L.APK_ERR_NGHOST
    with pytest.raises(L.ApkError) as e:  # hlld is not a hydro solver (registry hydro.cpp:386-416)
        hydro.CalculateFluxes(md, "euler", "plm", "hlld", L.make_eos(1.4))
    assert e.value.code == L.APK_ERR_UNSUPPORTED
    with pytest.raises(L.ApkError) as e:  # llf only with dc (hydro.cpp:347-349)
        hydro.CalculateFluxes(md, "euler", "plm", "llf", L.make_eos(1.4))
    assert e.value.code == L.APK_ERR_UNSUPPORTED
    with pytest.raises(L.ApkError) as e:  # fluid does not match the pack
        hydro.CalculateFluxes(md, "glmmhd", "plm", "hlle", L.make_eos(1.4))
    assert e.value.code == L.APK_ERR_INVALID


# ---- fused stage with FillDerived + dt estimate folded into the finishing sweep -----------------------
@pytest.mark.parametrize("strict", [True, False], ids=["strict", "fma"])
@pytest.mark.parametrize("fill", [0, 2], ids=["nofill", "outofplace"])
@pytest.mark.parametrize("fluid,recon,riemann,nx", [("glmmhd", "ppm", "hlld", (16, 16, 16)), ("glmmhd", "ppm", "hlld", (40, 9, 16)),
                                                    ("euler", "plm", "hllc", (16, 16, 16)), ("glmmhd", "dc", "hlld", (16, 16, 16)),
                                                    ("glmmhd", "wenoz", "hlle", (32, 8, 16))])
def test_fused_stage_with_more_ghost_layers_than_the_stencil(request, oracle, fluid, recon, riemann, nx, fill, strict):
    """Refined meshes run PPM / WENO-Z with nghost = 4 (Parthenon wants it even) and donor-cell / PLM stages on the same
    blocks: the stage kernels -- the two-kernel form on these shapes -- must not depend on the ghost zone being exactly
    as deep as the stencil.  Against the oracle's tasks on the same blocks."""
    from athenapk_amd import hydro
    ctx = _ctx(request, strict)
    ng = 4
    prim = H.random_prim(fluid, nx, ng, seed=29, kind="smooth", nblocks=3)
    g = H.geom(fluid, nx, ng, 0, (0.1, 0.07, 0.13))
    cons = H.prim_to_cons(fluid, prim, GAMMA)
    rng = np.random.default_rng(6)
    u1 = cons * (1.0 + 1e-3 * rng.standard_normal(cons.shape))
    ded = 1 if fluid == "glmmhd" else 0
    m0 = hydro.MeshData(ctx, nx, ng, NHYDRO[fluid], dx=tuple(g.dx), nblocks=3, cons=cons, prim=prim, with_flux=False)
    m1 = hydro.MeshData(ctx, nx, ng, NHYDRO[fluid], dx=tuple(g.dx), nblocks=3, cons=u1, with_flux=False)
    hydro.StageFused(m0, m1, fluid, recon, riemann, hydro.L.make_eos(GAMMA), C_H, 0.25, 0.75, 0.004, dedner=ded,
                     glmmhd_alpha=0.1, mindx=0.07, fill_derived=fill, estimate_dt=bool(fill))
    want = H.orc_stage(fluid, recon, riemann, g, cons, u1, prim, GAMMA, C_H, 0.25, 0.75, 0.004, dedner=ded, alpha=0.1, mindx=0.07)
    _cmp(H.interior(m0.cons_host(), nx, ng), H.interior(want, nx, ng), strict, "cons")
    if fill:
        want_cons, want_prim, bad = H.orc_c2p(fluid, g, want, oracle.make_eos(GAMMA))
        assert bad == 0
        _cmp(H.interior(m1.prim_host(), nx, ng), H.interior(want_prim, nx, ng), strict, "prim (out of place)")
        dt, want_dt = hydro.StageDt(ctx, 0.3), 0.3 * H.orc_min_dt(fluid, g, want_prim, GAMMA)
        assert dt == want_dt if strict else dt == pytest.approx(want_dt, rel=1e-12)


@pytest.mark.parametrize("strict", [True, False], ids=["strict", "fma"])
@pytest.mark.parametrize("fluid,recon,riemann,nx", [("glmmhd", "ppm", "hlld", (70, 9, 7)),
                                                    ("euler", "plm", "hllc", (66, 10, 6)),
                                                    ("glmmhd", "dc", "hlld", (64, 8, 8)),
                                                    ("glmmhd", "ppm", "hlld", (130, 12, 1))])
def test_fused_stage_fill_derived_and_dt(request, oracle, fluid, recon, riemann, nx, strict):
    """finishing sweep = update + Dedner + ConsToPrim (in place) + hyperbolic dt, vs the oracle's
    separate tasks (hydro_driver.cpp:534-577,605-613)."""
    from athenapk_amd import hydro
    ctx = _ctx(request, strict)
    ng, prim, g = _case(fluid, recon, nx, kind="smooth", seed=41)
    cons = H.prim_to_cons(fluid, prim, GAMMA)
    ded = 1 if fluid == "glmmhd" else 0
    eos_kw = dict(pfloor=1e-6, dfloor=1e-6)
    m0 = hydro.MeshData(ctx, nx, ng, NHYDRO[fluid], dx=tuple(g.dx), nblocks=prim.shape[0], cons=cons, prim=prim,
                        with_flux=False)
    m1 = hydro.MeshData(ctx, nx, ng, NHYDRO[fluid], dx=tuple(g.dx), nblocks=prim.shape[0], cons=cons, with_flux=False)
    ctx.poll_flags()
    hydro.StageFused(m0, m1, fluid, recon, riemann, hydro.L.make_eos(GAMMA, **eos_kw), C_H, 0.0, 1.0, 0.004,
                     dedner=ded, glmmhd_alpha=0.1, mindx=0.07, fill_derived=True, estimate_dt=True)
    dt = hydro.StageDt(ctx, 0.3)
    want_cons = H.orc_stage(fluid, recon, riemann, g, cons, cons, prim, GAMMA, C_H, 0.0, 1.0, 0.004, dedner=ded,
                            alpha=0.1, mindx=0.07)
    want_cons, want_prim, bad = H.orc_c2p(fluid, g, want_cons, oracle.make_eos(GAMMA, **eos_kw))
    assert bad == 0 and ctx.poll_flags() == 0
    want_dt = 0.3 * H.orc_min_dt(fluid, g, want_prim, GAMMA)
    _cmp(H.interior(m0.cons_host(), nx, ng), H.interior(want_cons, nx, ng), strict, "cons")
    _cmp(H.interior(m0.prim_host(), nx, ng), H.interior(want_prim, nx, ng), strict, "prim (interior, in place)")
    if strict:
        assert dt == want_dt
    else:
        assert dt == pytest.approx(want_dt, rel=1e-12)
    # ghost zones of prim are untouched by the stage and converted by the companion call
    ghosts_before = m0.prim_host().copy()
    hydro.ConservedToPrimitiveGhosts(m0, fluid, hydro.L.make_eos(GAMMA, **eos_kw))
    full_cons, full_prim, _ = H.orc_c2p(fluid, g, m0.cons_host(), oracle.make_eos(GAMMA, **eos_kw))
    _cmp(m0.prim_host(), full_prim, strict, "prim after ghost conversion")
    assert np.array_equal(H.interior(ghosts_before, nx, ng), H.interior(m0.prim_host(), nx, ng))


@pytest.mark.gpu
@pytest.mark.parametrize("strict", [True, False], ids=["strict", "fma"])
@pytest.mark.parametrize("fluid,recon,riemann,nx,gam0", [("glmmhd", "dc", "hlld", (64, 8, 34), 0.0),
                                                         ("glmmhd", "dc", "hlle", (70, 9, 7), 0.5),
                                                         ("euler", "dc", "hllc", (66, 10, 6), 0.0),
                                                         ("euler", "dc", "hlle", (64, 8, 8), 0.0),
                                                         ("glmmhd", "ppm", "hlld", (70, 9, 7), 0.0),
                                                         ("glmmhd", "plm", "hlld", (66, 9, 8), -2.0),
                                                         ("euler", "plm", "hllc", (64, 34, 36), 0.5),
                                                         ("euler", "plm", "hllc", (66, 10, 1), 0.5)])
def test_fused_stage_fill_derived_out_of_place(request, oracle, fluid, recon, riemann, nx, gam0, strict):
    """fill_derived = 2: the new primitives land in u1's prim arrays, u0.prim stays as it was.
    For 3-D donor cell this is the single-march kernel (fused_dc3_kernel), incl. its k segments
    (nx3 = 34 -> 4 x 8 + 2 planes); the (64, 34, 36) case runs the x2 and the finishing x3 march in
    two segments each (small packs are cut so that the launch fills the machine)."""
    from athenapk_amd import hydro
    ctx = _ctx(request, strict)
    ng, prim, g = _case(fluid, recon, nx, kind="smooth", seed=43)
    cons = H.prim_to_cons(fluid, prim, GAMMA)
    ded = 1 if fluid == "glmmhd" else 0
    if gam0 < 0.0:   # marker: extended Dedner source (reads neighbouring primitives), gam0 = 0
        ded, gam0 = 2, 0.0
    u1c = cons * 1.01 if gam0 != 0.0 else cons
    eos_kw = dict(pfloor=1e-6, dfloor=1e-6)
    sentinel = np.full_like(prim, -7.0)
    m0 = hydro.MeshData(ctx, nx, ng, NHYDRO[fluid], dx=tuple(g.dx), nblocks=prim.shape[0], cons=cons, prim=prim,
                        with_flux=False)
    m1 = hydro.MeshData(ctx, nx, ng, NHYDRO[fluid], dx=tuple(g.dx), nblocks=prim.shape[0], cons=u1c, prim=sentinel,
                        with_flux=False)
    ctx.poll_flags()
    hydro.StageFused(m0, m1, fluid, recon, riemann, hydro.L.make_eos(GAMMA, **eos_kw), C_H, gam0, 1.0 - gam0, 0.004,
                     dedner=ded, glmmhd_alpha=0.1, mindx=0.07, fill_derived=2, estimate_dt=True)
    dt = hydro.StageDt(ctx, 0.3)
    want_cons = H.orc_stage(fluid, recon, riemann, g, cons, u1c, prim, GAMMA, C_H, gam0, 1.0 - gam0, 0.004,
                            dedner=ded, alpha=0.1, mindx=0.07)
    want_cons, want_prim, bad = H.orc_c2p(fluid, g, want_cons, oracle.make_eos(GAMMA, **eos_kw))
    assert bad == 0 and ctx.poll_flags() == 0
    _cmp(H.interior(m0.cons_host(), nx, ng), H.interior(want_cons, nx, ng), strict, "cons")
    _cmp(H.interior(m1.prim_host(), nx, ng), H.interior(want_prim, nx, ng), strict, "u1.prim (interior)")
    assert np.array_equal(m0.prim_host(), prim), "u0.prim must not be touched"
    got1 = m1.prim_host()
    mask = np.ones(got1.shape, dtype=bool)
    H.interior(mask, nx, ng)[...] = False
    assert np.all(got1[mask] == -7.0), "ghost zones of u1.prim must not be touched"
    want_dt = 0.3 * H.orc_min_dt(fluid, g, want_prim, GAMMA)
    if strict:
        assert dt == want_dt
    else:
        assert dt == pytest.approx(want_dt, rel=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("strict", [True, False], ids=["strict", "fma"])
@pytest.mark.parametrize("cons_store", [0, 1, 2], ids=["all_cells", "shell", "none"])
@pytest.mark.parametrize("fluid,recon,riemann,nx,gam0", [("glmmhd", "dc", "hlld", (64, 8, 34), 0.0),    # two rows per lane
                                                         ("glmmhd", "dc", "hlld", (40, 9, 10), 0.0),    # odd nx2: one row per lane
                                                         ("euler", "dc", "hllc", (66, 12, 9), 0.0),
                                                         ("glmmhd", "dc", "hlle", (34, 10, 8), 0.5),
                                                         ("glmmhd", "ppm", "hlld", (70, 9, 7), 0.25),    # lean finishing march
                                                         ("glmmhd", "wenoz", "hlld", (64, 8, 8), 0.0)])
def test_lean_stage_forms_and_conserved_store_modes(request, oracle, fluid, recon, riemann, nx, gam0, cons_store, strict):
    """The stage kernels' LEAN forms (default equation of state: no floors, no ceilings; plain Dedner source; nothing
    optional asked for) -- the finishing x1 + x2 march, the donor-cell march with one and with two rows per lane --
    against the oracle, and apk_stage_args.cons_store: 1 leaves the updated conserved state in the nghost-deep shell of
    every block only, 2 nowhere (the primitives are the stage's product); the cells not stored keep what they held."""
    from athenapk_amd import hydro
    ctx = _ctx(request, strict)
    ng, prim, g = _case(fluid, recon, nx, kind="smooth", seed=143, nblocks=2)
    cons = H.prim_to_cons(fluid, prim, GAMMA)
    ded = 1 if fluid == "glmmhd" else 0
    u1c = cons * 1.01 if gam0 != 0.0 else cons
    sentinel = np.full_like(prim, -7.0)
    m0 = hydro.MeshData(ctx, nx, ng, NHYDRO[fluid], dx=tuple(g.dx), nblocks=2, cons=cons, prim=prim, with_flux=False)
    m1 = hydro.MeshData(ctx, nx, ng, NHYDRO[fluid], dx=tuple(g.dx), nblocks=2, cons=u1c, prim=sentinel, with_flux=False)
    ctx.poll_flags()
    hydro.StageFused(m0, m1, fluid, recon, riemann, hydro.L.make_eos(GAMMA), C_H, gam0, 1.0 - gam0, 0.004,
                     dedner=ded, glmmhd_alpha=0.1, mindx=0.07, fill_derived=2, estimate_dt=True, cons_store=cons_store)
    dt = hydro.StageDt(ctx, 0.3)
    want_cons = H.orc_stage(fluid, recon, riemann, g, cons, u1c, prim, GAMMA, C_H, gam0, 1.0 - gam0, 0.004,
                            dedner=ded, alpha=0.1, mindx=0.07)
    want_cons, want_prim, bad = H.orc_c2p(fluid, g, want_cons, oracle.make_eos(GAMMA))
    assert bad == 0 and ctx.poll_flags() == 0
    _cmp(H.interior(m1.prim_host(), nx, ng), H.interior(want_prim, nx, ng), strict, "u1.prim (interior)")
    assert np.array_equal(m0.prim_host(), prim), "u0.prim must not be touched"
    want_dt = 0.3 * H.orc_min_dt(fluid, g, want_prim, GAMMA)
    assert dt == want_dt if strict else dt == pytest.approx(want_dt, rel=1e-12)
    got = H.interior(m0.cons_host(), nx, ng)
    full = H.interior(want_cons, nx, ng)
    old = H.interior(cons, nx, ng)
    honoured = recon == "dc" and nx[2] > 1          # the single-march donor-cell stage in its lean form
    if cons_store == 0 or not honoured:
        _cmp(got, full, strict, "cons")
    else:
        shell = np.ones(got.shape, dtype=bool)
        if cons_store == 1:
            shell[..., ng:-ng, ng:-ng, ng:-ng] = False
        else:
            shell[...] = False
        if shell.any():
            _cmp(got[shell], full[shell], strict, "cons (shell)")
        assert np.array_equal(got[~shell], old[~shell]), "cells outside the shell must keep their old contents"
        assert (~shell).any() or cons_store == 1


@pytest.mark.gpu
@pytest.mark.parametrize("strict", [True, False], ids=["strict", "fma"])
@pytest.mark.parametrize("fluid,riemann,nx", [("glmmhd", "hlld", (64, 8, 34)), ("glmmhd", "hlld", (40, 9, 10)),
                                              ("euler", "hllc", (66, 12, 9)), ("glmmhd", "hlle", (34, 10, 8))])
def test_donor_cell_stage_takes_its_input_from_the_conserved_state(request, oracle, fluid, riemann, nx, strict):
    """apk_stage_args.prim_from_cons: u0.prim holds garbage, u1.cons the state; the lean single-march donor-cell stage
    (one and two rows per lane) derives the primitives itself and must give what the stage gives when handed them."""
    from athenapk_amd import hydro
    ctx = _ctx(request, strict)
    ng, prim, g = _case(fluid, "dc", nx, kind="smooth", seed=151, nblocks=2)
    cons = H.prim_to_cons(fluid, prim, GAMMA)
    # (the primitives the finishing sweep of the previous stage would have stored: ConsToPrim of the conserved state)
    _, prim_of_cons, bad = H.orc_c2p(fluid, g, cons.copy(), oracle.make_eos(GAMMA))
    assert bad == 0
    ded = 1 if fluid == "glmmhd" else 0
    garbage = np.full_like(prim, np.nan)
    m0 = hydro.MeshData(ctx, nx, ng, NHYDRO[fluid], dx=tuple(g.dx), nblocks=2, cons=cons, prim=garbage, with_flux=False)
    m1 = hydro.MeshData(ctx, nx, ng, NHYDRO[fluid], dx=tuple(g.dx), nblocks=2, cons=cons, prim=np.full_like(prim, -7.0), with_flux=False)
    ctx.poll_flags()
    hydro.StageFused(m0, m1, fluid, "dc", riemann, hydro.L.make_eos(GAMMA), C_H, 0.0, 1.0, 0.004, dedner=ded, glmmhd_alpha=0.1,
                     mindx=0.07, fill_derived=2, prim_from_cons=True)
    want_cons = H.orc_stage(fluid, "dc", riemann, g, cons, cons, prim_of_cons, GAMMA, C_H, 0.0, 1.0, 0.004, dedner=ded, alpha=0.1, mindx=0.07)
    want_cons, want_prim, bad = H.orc_c2p(fluid, g, want_cons, oracle.make_eos(GAMMA))
    assert bad == 0 and ctx.poll_flags() == 0
    _cmp(H.interior(m0.cons_host(), nx, ng), H.interior(want_cons, nx, ng), strict, "cons")
    _cmp(H.interior(m1.prim_host(), nx, ng), H.interior(want_prim, nx, ng), strict, "u1.prim")
    # a non-lean stage refuses
    with pytest.raises(Exception):
        hydro.StageFused(m0, m1, fluid, "dc", riemann, hydro.L.make_eos(GAMMA, pfloor=1e-6), C_H, 0.0, 1.0, 0.004, dedner=ded,
                         glmmhd_alpha=0.1, mindx=0.07, fill_derived=2, prim_from_cons=True)


@pytest.mark.gpu
@pytest.mark.parametrize("strict", [True, False], ids=["strict", "fma"])
@pytest.mark.parametrize("recon,gam0", [("ppm", 0.0), ("wenoz", 0.25)])
def test_time_step_estimate_without_stored_primitives(request, oracle, recon, gam0, strict):
    """fill_derived = 3: the finishing march of the lean two-kernel stage computes the primitives of the cells it updates
    for the time-step estimate only -- same estimate, same conserved state as fill_derived = 2, and neither prim array
    is written."""
    from athenapk_amd import hydro
    ctx = _ctx(request, strict)
    fluid, riemann, nx = "glmmhd", "hlld", (70, 9, 7)
    ng, prim, g = _case(fluid, recon, nx, kind="smooth", seed=157, nblocks=2)
    cons = H.prim_to_cons(fluid, prim, GAMMA)
    u1c = cons * 1.01 if gam0 != 0.0 else cons
    out = {}
    for fd in (2, 3):
        m0 = hydro.MeshData(ctx, nx, ng, 9, dx=tuple(g.dx), nblocks=2, cons=cons, prim=prim, with_flux=False)
        m1 = hydro.MeshData(ctx, nx, ng, 9, dx=tuple(g.dx), nblocks=2, cons=u1c, prim=np.full_like(prim, -7.0), with_flux=False)
        hydro.StageFused(m0, m1, fluid, recon, riemann, hydro.L.make_eos(GAMMA), C_H, gam0, 1.0 - gam0, 0.004, dedner=1,
                         glmmhd_alpha=0.1, mindx=0.07, fill_derived=fd, estimate_dt=True)
        out[fd] = (hydro.StageDt(ctx, 0.3), m0.cons_host(), m0.prim_host(), m1.prim_host())
    assert out[3][0] == out[2][0] and np.array_equal(out[3][1], out[2][1])
    assert np.array_equal(out[3][2], prim) and np.all(out[3][3] == -7.0), "fill_derived = 3 must not write primitives"
    assert not np.all(out[2][3] == -7.0)
    with pytest.raises(Exception):   # nothing to do without the estimate
        hydro.StageFused(m0, m1, fluid, recon, riemann, hydro.L.make_eos(GAMMA), C_H, gam0, 1.0 - gam0, 0.004, dedner=1,
                         glmmhd_alpha=0.1, mindx=0.07, fill_derived=3, estimate_dt=False)


@pytest.mark.gpu
@pytest.mark.parametrize("strict", [True, False], ids=["strict", "fma"])
@pytest.mark.parametrize("fluid,recon,riemann,nx", [("glmmhd", "ppm", "hlld", (70, 9, 7)),
                                                    ("glmmhd", "wenoz", "hlld", (64, 8, 8)),
                                                    ("euler", "plm", "hllc", (66, 10, 1))])
def test_fused_stage_split_around_exchange(request, oracle, fluid, recon, riemann, nx, strict):
    """phase 1 (x1 sweep on column windows: everything but the cells next to a 'remote' face,
    then the thin slabs) + phase 2 (remaining sweeps) == the unsplit stage, bit for bit."""
    import torch
    from athenapk_amd import hydro
    ctx = _ctx(request, strict)
    ng, prim, g = _case(fluid, recon, nx, kind="smooth", seed=47, nblocks=3)
    cons = H.prim_to_cons(fluid, prim, GAMMA)
    ded = 1 if fluid == "glmmhd" else 0
    eos = hydro.L.make_eos(GAMMA, pfloor=1e-6, dfloor=1e-6)
    kw = dict(dedner=ded, glmmhd_alpha=0.1, mindx=0.07, fill_derived=True, estimate_dt=True)

    def packs():
        a = hydro.MeshData(ctx, nx, ng, NHYDRO[fluid], dx=tuple(g.dx), nblocks=3, cons=cons, prim=prim, with_flux=False)
        b = hydro.MeshData(ctx, nx, ng, NHYDRO[fluid], dx=tuple(g.dx), nblocks=3, cons=cons * 1.01, with_flux=False)
        return a, b
    r0, r1 = packs()
    hydro.StageFused(r0, r1, fluid, recon, riemann, eos, C_H, 0.5, 0.5, 0.004, **kw)
    dt_ref = hydro.StageDt(ctx, 0.3)
    m0, m1 = packs()
    is_, ie, ni, W = ng, ng + nx[0] - 1, nx[0] + 2 * ng, ng
    jk = [ng, ng + nx[1] - 1, ng if nx[2] > 1 else 0, ng + nx[2] - 1 if nx[2] > 1 else 0]  # full j, k ranges
    # block 0: data missing on the low x1 side, block 1: on the high side, block 2: on both
    missing = [(1, 0), (0, 1), (1, 1)]
    main = [[0, ni, is_ + W * lo, ie - W * hi] + jk for lo, hi in missing]
    slab_lo = [([is_ - 2, W + 3, is_, is_ + W - 1] if lo else [0, 0, 0, -1]) + jk for lo, hi in missing]
    slab_hi = [([ie - W - 1, W + 3, ie - W + 1, ie] if hi else [0, 0, 0, -1]) + jk for lo, hi in missing]
    for win in (main, slab_lo, slab_hi):
        t = torch.tensor(win, dtype=torch.int32, device="cuda")
        hydro.StageFused(m0, m1, fluid, recon, riemann, eos, C_H, 0.5, 0.5, 0.004, phase=1, window=t, **kw)
    hydro.StageFused(m0, m1, fluid, recon, riemann, eos, C_H, 0.5, 0.5, 0.004, phase=2, **kw)
    dt = hydro.StageDt(ctx, 0.3)
    if strict:
        assert np.array_equal(m0.cons_host(), r0.cons_host()) and np.array_equal(m0.prim_host(), r0.prim_host())
        assert dt == dt_ref
    else:  # the window kernels are the same code: identical in the FMA build as well
        assert np.array_equal(m0.cons_host(), r0.cons_host())
    want = H.orc_stage(fluid, recon, riemann, g, cons, cons * 1.01, prim, GAMMA, C_H, 0.5, 0.5, 0.004, dedner=ded,
                       alpha=0.1, mindx=0.07)
    _cmp(H.interior(m0.cons_host(), nx, ng), H.interior(want, nx, ng), strict, "cons")


@pytest.mark.gpu
@pytest.mark.parametrize("strict", [True, False], ids=["strict", "fma"])
@pytest.mark.parametrize("fluid,recon,riemann,nx,gam0", [("glmmhd", "ppm", "hlld", (70, 9, 14), 0.0),
                                                        ("glmmhd", "wenoz", "hlld", (64, 8, 13), 0.5),
                                                        ("euler", "plm", "hllc", (40, 10, 9), 0.5)])
def test_two_kernel_stage_split_on_plane_windows(request, oracle, fluid, recon, riemann, nx, gam0, strict):
    """The two-kernel 3-D stage (x3 sweep writing its flux difference, then ONE march doing x1 + x2 and
    finishing; apk_stage_split_axis == 3): unsplit == oracle, and phase 1 on plane windows (all planes
    farther than nghost from a 'remote' x3 face, then the slabs) + phase 2 == unsplit, bit for bit."""
    import ctypes as C
    import torch
    from athenapk_amd import hydro
    ctx = _ctx(request, strict)
    ng, prim, g = _case(fluid, recon, nx, kind="smooth", seed=59, nblocks=3)
    cons = H.prim_to_cons(fluid, prim, GAMMA)
    ded = 1 if fluid == "glmmhd" else 0
    eos = hydro.L.make_eos(GAMMA)
    gam1 = 1.0 - gam0 if gam0 else 1.0
    kw = dict(dedner=ded, glmmhd_alpha=0.1, mindx=0.07, fill_derived=2, estimate_dt=True)
    sentinel = np.full_like(prim, -7.0)

    def packs():
        a = hydro.MeshData(ctx, nx, ng, NHYDRO[fluid], dx=tuple(g.dx), nblocks=3, cons=cons, prim=prim, with_flux=False)
        b = hydro.MeshData(ctx, nx, ng, NHYDRO[fluid], dx=tuple(g.dx), nblocks=3, cons=cons * 1.01, prim=sentinel, with_flux=False)
        return a, b
    r0, r1 = packs()
    cfg = hydro._cfg(fluid, recon, riemann)
    assert ctx.lib.apk_stage_split_axis(r0.h, C.byref(cfg), 2) == 3      # this IS the two-kernel form
    hydro.StageFused(r0, r1, fluid, recon, riemann, eos, C_H, gam0, gam1, 0.004, **kw)
    dt_ref = hydro.StageDt(ctx, 0.3)
    want = H.orc_stage(fluid, recon, riemann, g, cons, cons * 1.01, prim, GAMMA, C_H, gam0, gam1, 0.004, dedner=ded,
                       alpha=0.1, mindx=0.07)
    _cmp(H.interior(r0.cons_host(), nx, ng), H.interior(want, nx, ng), strict, "cons")
    assert np.all(H.interior(r1.prim_host(), nx, ng) != -7.0)            # FillDerived went out of place, everywhere
    m0, m1 = packs()
    is_, ie, ni, W = ng, ng + nx[0] - 1, nx[0] + 2 * ng, ng
    js, je, ks, ke = ng, ng + nx[1] - 1, ng, ng + nx[2] - 1
    missing = [(1, 0), (0, 1), (1, 1)]   # block 0: data missing below, block 1: above, block 2: both
    main = [[0, ni, is_, ie, js, je, ks + W * lo, ke - W * hi] for lo, hi in missing]
    slab_lo = [[0, ni if lo else 0, is_, ie, js, je, ks, ks + W - 1] for lo, hi in missing]
    slab_hi = [[0, ni if hi else 0, is_, ie, js, je, ke - W + 1, ke] for lo, hi in missing]
    for win in (main, slab_lo, slab_hi):
        t = torch.tensor(win, dtype=torch.int32, device="cuda")
        hydro.StageFused(m0, m1, fluid, recon, riemann, eos, C_H, gam0, gam1, 0.004, phase=1, window=t, **kw)
    hydro.StageFused(m0, m1, fluid, recon, riemann, eos, C_H, gam0, gam1, 0.004, phase=2, **kw)
    assert np.array_equal(m0.cons_host(), r0.cons_host())
    assert np.array_equal(H.interior(m1.prim_host(), nx, ng), H.interior(r1.prim_host(), nx, ng))
    assert hydro.StageDt(ctx, 0.3) == dt_ref


@pytest.mark.gpu
@pytest.mark.parametrize("strict", [True, False], ids=["strict", "fma"])
@pytest.mark.parametrize("fluid,riemann,nx", [("glmmhd", "hlld", (64, 10, 34)), ("euler", "hllc", (66, 9, 7))])
def test_donor_cell_stage_split_around_exchange(request, oracle, fluid, riemann, nx, strict):
    """The single-kernel 3-D donor-cell stage on index windows: everything but the one-cell layers
    next to 'late' faces first, then the six slabs (disjoint, covering every cell once)."""
    import torch
    from athenapk_amd import hydro
    ctx = _ctx(request, strict)
    ng, prim, g = _case(fluid, "dc", nx, kind="smooth", seed=53, nblocks=3)
    cons = H.prim_to_cons(fluid, prim, GAMMA)
    ded = 1 if fluid == "glmmhd" else 0
    eos = hydro.L.make_eos(GAMMA, pfloor=1e-6, dfloor=1e-6)
    kw = dict(dedner=ded, glmmhd_alpha=0.1, mindx=0.07, fill_derived=2)
    sentinel = np.full_like(prim, -7.0)

    def packs():
        a = hydro.MeshData(ctx, nx, ng, NHYDRO[fluid], dx=tuple(g.dx), nblocks=3, cons=cons, prim=prim, with_flux=False)
        b = hydro.MeshData(ctx, nx, ng, NHYDRO[fluid], dx=tuple(g.dx), nblocks=3, cons=cons, prim=sentinel, with_flux=False)
        return a, b
    r0, r1 = packs()
    hydro.StageFused(r0, r1, fluid, "dc", riemann, eos, C_H, 0.0, 1.0, 0.004, **kw)
    m0, m1 = packs()
    s_ = [ng, ng, ng]
    e_ = [ng + nx[0] - 1, ng + nx[1] - 1, ng + nx[2] - 1]
    ni = nx[0] + 2 * ng
    # late faces per block: (x lo, x hi, y lo, y hi, z lo, z hi)
    late = [(1, 0, 0, 1, 0, 0), (0, 1, 1, 0, 1, 1), (1, 1, 1, 1, 1, 1)]

    def rng(d, L):  # interior range of direction d minus the late layers
        return [s_[d] + L[2 * d], e_[d] - L[2 * d + 1]]
    tables = [[[0, ni] + rng(0, L) + rng(1, L) + rng(2, L) for L in late]]
    empty = [0, 0, 0, -1, 0, -1, 0, -1]
    full_i = lambda: [0, ni, s_[0], e_[0]]  # noqa: E731
    for side in (0, 1):  # z slabs: one plane, all rows and columns
        kk = s_[2] if side == 0 else e_[2]
        tables.append([full_i() + [s_[1], e_[1], kk, kk] if L[4 + side] else empty for L in late])
    for side in (0, 1):  # y slabs: one row, the planes not covered by the z slabs
        jj = s_[1] if side == 0 else e_[1]
        tables.append([full_i() + [jj, jj] + rng(2, L) if L[2 + side] else empty for L in late])
    for side in (0, 1):  # x slabs: one column (+ a halo lane each side)
        ii = s_[0] if side == 0 else e_[0]
        tables.append([[ii - 1, 3, ii, ii] + rng(1, L) + rng(2, L) if L[side] else empty for L in late])
    for win in tables:
        t = torch.tensor(win, dtype=torch.int32, device="cuda")
        hydro.StageFused(m0, m1, fluid, "dc", riemann, eos, C_H, 0.0, 1.0, 0.004, phase=1, window=t, **kw)
    hydro.StageFused(m0, m1, fluid, "dc", riemann, eos, C_H, 0.0, 1.0, 0.004, phase=2, **kw)   # no-op
    # (the whole stage runs two rows per lane, the windows one: the same operations in the parity build; the product
    # build contracts each kernel's expressions its own way -- last-bit differences)
    if strict:
        assert np.array_equal(m0.cons_host(), r0.cons_host()) and np.array_equal(m1.prim_host(), r1.prim_host())
    else:
        np.testing.assert_allclose(m0.cons_host(), r0.cons_host(), rtol=1e-12, atol=1e-14)
        np.testing.assert_allclose(m1.prim_host(), r1.prim_host(), rtol=1e-12, atol=1e-14)
    want = H.orc_stage(fluid, "dc", riemann, g, cons, cons, prim, GAMMA, C_H, 0.0, 1.0, 0.004, dedner=ded, alpha=0.1,
                       mindx=0.07)
    _cmp(H.interior(m0.cons_host(), nx, ng), H.interior(want, nx, ng), strict, "cons")


def test_fused_fill_derived_rejected_where_unsafe(request):
    from athenapk_amd import hydro
    from athenapk_amd import lib as L
    ctx = _ctx(request, True)
    nx = (64, 1, 1)
    ng, prim, g = _case("glmmhd", "ppm", nx, seed=3)
    cons = H.prim_to_cons("glmmhd", prim, GAMMA)
    m0 = hydro.MeshData(ctx, nx, ng, 9, nblocks=prim.shape[0], cons=cons, prim=prim, with_flux=False)
    with pytest.raises(L.ApkError) as e:  # 1-D: the finishing sweep is the x1 sweep, lanes share prim
        hydro.StageFused(m0, m0, "glmmhd", "ppm", "hlld", L.make_eos(GAMMA), C_H, 0.0, 1.0, 1e-3, dedner=1,
                         mindx=0.1, fill_derived=True)
    assert e.value.code == L.APK_ERR_UNSUPPORTED
    with pytest.raises(L.ApkError) as e:  # out of place needs a second prim array
        hydro.StageFused(m0, m0, "glmmhd", "ppm", "hlld", L.make_eos(GAMMA), C_H, 0.0, 1.0, 1e-3, dedner=1,
                         mindx=0.1, fill_derived=2)
    assert e.value.code == L.APK_ERR_INVALID


# ---- few-modes turbulence driver kernels ------------------------------------------------------------
def _turb_case(oracle, nblocks_axis=2, n=16, ng=2, nmodes=8):
    """an n^3 box split in nblocks_axis^3 blocks; returns geometry, spectral state, phases per block"""
    rng = np.random.default_rng(5)
    kv = np.array([[1, 0, 0, 1, 2, 0, -1, 2], [0, 1, 0, 1, -1, 2, 1, 2], [0, 0, 1, -1, 0, 1, 2, 2]], dtype=np.float64)
    if nmodes != 8:
        kv = np.concatenate([kv, rng.integers(-3, 4, size=(3, nmodes - 8)).astype(np.float64)], axis=1)
        kv[:, np.all(kv == 0, axis=0)] = 1.0
    f = oracle.Fmft(oracle.load(), kv, k_peak=2.0, sol_weight=0.7, t_corr=0.5, rseed=7)
    f.evolve(0.1)
    f.evolve(0.1)
    mb = n // nblocks_axis
    g = H.geom("glmmhd", (mb, mb, mb), ng, 0, (1.0 / n,) * 3)
    phases = []
    for bk in range(nblocks_axis):
        for bj in range(nblocks_axis):
            for bi in range(nblocks_axis):
                phases.append((f.phases(0, mb, bi * mb, n), f.phases(1, mb, bj * mb, n), f.phases(2, mb, bk * mb, n)))
    return f, g, mb, phases


@pytest.mark.gpu
@pytest.mark.parametrize("strict", [True, False], ids=["strict", "fma"])
@pytest.mark.parametrize("case", [(2, 16, 8), (1, 136, 23)], ids=["8cubed_blocks_8_modes", "136cubed_block_23_modes"])
def test_fmft_inverse(request, oracle, strict, case):
    """few_modes_ft.cpp:330-347.  The product build's kernel keeps a cell column's phases in registers, ten modes at a
    time, two columns per lane: 8-cell rows (one lane in eight at work), a row of 136 cells (two passes, the second one
    with its upper columns off the row) and mode counts that are not a multiple of ten (zero-padded coefficients)."""
    from athenapk_amd import hydro
    ctx = _ctx(request, strict)
    f, g, mb, phases = _turb_case(oracle, nblocks_axis=case[0], n=case[1], nmodes=case[2])
    nb = len(phases)
    md = hydro.MeshData(ctx, (mb, mb, mb), 2, 9, dx=tuple(g.dx), nblocks=nb, with_flux=False, row_pitch="natural")  # (the acceleration field's layout)
    drv = hydro.FewModesFT(md, [[p.transpose(2, 1, 0) for p in blk] for blk in phases])
    drv.Inverse(f.var_hat())
    got = drv.acc_host()
    want = np.stack([f.inverse(g, *phases[b]) for b in range(nb)])
    _cmp(got, want, strict, "acc")


@pytest.mark.gpu
@pytest.mark.parametrize("strict", [True, False], ids=["strict", "fma"])
@pytest.mark.parametrize("floors", [False, True], ids=["nofloor", "floors"])
def test_turbulence_kick_with_fill_derived_and_dt(request, oracle, strict, floors):
    """apk_turb_apply_fill = apk_turb_apply, then ConsToPrim of the interior, then the dt estimate: the same
    bits in cons, acc, the interior primitives (ghost zones untouched) and the reduced time step."""
    from athenapk_amd import hydro
    ctx = _ctx(request, strict)
    f, g, mb, phases = _turb_case(oracle)
    nb = len(phases)
    nx = (mb, mb, mb)
    prim = H.random_prim("glmmhd", nx, 2, seed=23, kind="smooth", nblocks=nb)
    cons = H.prim_to_cons("glmmhd", prim, GAMMA)
    eos = hydro.L.make_eos(GAMMA, pfloor=0.9, dfloor=0.95) if floors else hydro.L.make_eos(GAMMA)

    def run(fill):
        md = hydro.MeshData(ctx, nx, 2, 9, dx=tuple(g.dx), nblocks=nb, cons=cons, prim=np.full_like(prim, -3.0), with_flux=False, row_pitch="natural")
        drv = hydro.FewModesFT(md, [[p.transpose(2, 1, 0) for p in blk] for blk in phases])
        drv.Inverse(f.var_hat())
        if fill:
            drv.Perturb(0.01, 0.5, 1.0, fill=("glmmhd", eos, True))
            dt = hydro.StageDt(ctx, 0.3)
        else:
            drv.Perturb(0.01, 0.5, 1.0)
            hydro.ConservedToPrimitive(md, "glmmhd", eos)
            dt = hydro.EstimateTimestep(md, "glmmhd", eos, 0.3)
        ctx.poll_flags()
        return md.cons_host(), md.prim_host(), drv.acc_host(), dt
    a, b = run(True), run(False)
    # apk_turb_apply_dt: the same kick and estimate, primitives left alone
    md = hydro.MeshData(ctx, nx, 2, 9, dx=tuple(g.dx), nblocks=nb, cons=cons, prim=np.full_like(prim, -3.0), with_flux=False, row_pitch="natural")
    drv = hydro.FewModesFT(md, [[p.transpose(2, 1, 0) for p in blk] for blk in phases])
    drv.Inverse(f.var_hat())
    drv.Perturb(0.01, 0.5, 1.0, fill=("glmmhd", eos, True, False))
    dt_only = hydro.StageDt(ctx, 0.3)
    ctx.poll_flags()
    assert np.array_equal(md.cons_host(), a[0]) and np.array_equal(drv.acc_host(), a[2]) and dt_only == a[3]
    assert np.all(md.prim_host() == -3.0)
    assert np.array_equal(H.interior(a[0], nx, 2), H.interior(b[0], nx, 2)) and np.array_equal(a[2], b[2])
    # (product build: ConsToPrim inlined behind the kick contracts into FMAs differently from the separate kernel)
    _cmp(H.interior(a[1], nx, 2), H.interior(b[1], nx, 2), strict, "prim")
    assert a[3] == b[3] if strict else abs(a[3] - b[3]) <= 1e-12 * b[3]
    ghost = np.ones(a[1].shape, bool)
    ghost[..., 2:-2, 2:-2, 2:-2] = False
    assert np.all(a[1][ghost] == -3.0) and np.array_equal(a[0][ghost], cons[ghost])   # ghost zones untouched
    if floors:
        assert not np.array_equal(H.interior(a[0], nx, 2)[:, 0], H.interior(run_unfloored(ctx, hydro, f, g, phases, nx, cons, prim), nx, 2)[:, 0])


def run_unfloored(ctx, hydro, f, g, phases, nx, cons, prim):
    md = hydro.MeshData(ctx, nx, 2, 9, dx=tuple(g.dx), nblocks=len(phases), cons=cons, prim=prim, with_flux=False, row_pitch="natural")
    drv = hydro.FewModesFT(md, [[p.transpose(2, 1, 0) for p in blk] for blk in phases])
    drv.Inverse(f.var_hat())
    drv.Perturb(0.01, 0.5, 1.0)
    return md.cons_host()


@pytest.mark.gpu
@pytest.mark.parametrize("strict", [True, False], ids=["strict", "fma"])
def test_turbulence_perturb_and_history(request, oracle, strict):
    from athenapk_amd import hydro
    ctx = _ctx(request, strict)
    f, g, mb, phases = _turb_case(oracle)
    nb = len(phases)
    nx = (mb, mb, mb)
    prim = H.random_prim("glmmhd", nx, 2, seed=21, kind="smooth", nblocks=nb)
    cons = H.prim_to_cons("glmmhd", prim, GAMMA)
    md = hydro.MeshData(ctx, nx, 2, 9, dx=tuple(g.dx), nblocks=nb, cons=cons, prim=prim, with_flux=False, row_pitch="natural")
    drv = hydro.FewModesFT(md, [[p.transpose(2, 1, 0) for p in blk] for blk in phases])
    drv.Inverse(f.var_hat())
    acc0 = drv.acc_host()
    drv.Perturb(0.01, 0.5, 1.0)
    want_u, want_a = H.orc_turb_perturb(g, cons, acc0, 0.01, 0.5, 1.0)
    # the two global sums are accumulated in a different order than the oracle's serial loop
    np.testing.assert_allclose(drv.acc_host(), want_a, rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(md.cons_host(), want_u, rtol=1e-12, atol=1e-14)
    # ghost zones untouched
    assert np.array_equal(md.cons_host()[:, :, 0], cons[:, :, 0])
    # RMS of the normalised field is accel_rms, mass-weighted mean is zero
    a = drv.acc_host()[:, :, 2:-2, 2:-2, 2:-2]
    assert abs(np.sqrt((a ** 2).sum(axis=1).mean()) - 0.5) < 1e-12
    for fluid in ("glmmhd", "euler"):
        got = hydro.TurbulenceHst(md, fluid, GAMMA)
        want = H.orc_turb_history(fluid, g, prim, GAMMA)
        np.testing.assert_allclose(got, want, rtol=1e-13, atol=1e-15)


# ---- passive scalars through the fused stage ---------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("strict", [True, False], ids=["strict", "fma"])
@pytest.mark.parametrize("fill", [0, 1, 2], ids=["nofill", "inplace", "outofplace"])
@pytest.mark.parametrize("fluid,recon,riemann,nx", [("glmmhd", "ppm", "hlld", (70, 9, 7)),
                                                    ("euler", "plm", "hllc", (66, 10, 1)),
                                                    ("glmmhd", "dc", "hlld", (64, 8, 18)),
                                                    ("euler", "wenoz", "hlle", (40, 1, 1)),
                                                    ("euler", "limo3", "hllc", (16, 16, 16))])
def test_fused_stage_with_passive_scalars(request, oracle, fluid, recon, riemann, nx, fill, strict):
    """nscalars = 2 through apk_stage_fused: the sweeps leave the mass fluxes behind and one light
    kernel upwinds the reconstructed concentrations (hydro.cpp:1088-1097) and updates the scalar
    densities; equal to CalculateFluxes + UpdateWithFluxDivergence + DednerSource (+ ConsToPrim)."""
    from athenapk_amd import hydro, lib as L
    ctx = _ctx(request, strict)
    ndim = sum(1 for n in nx if n > 1)
    if fill and ndim == 1:
        pytest.skip("FillDerived is not fused in 1-D")
    ng, prim, g = _case(fluid, recon, nx, kind="smooth", nscalars=2, seed=61)
    cons = H.prim_to_cons(fluid, prim, GAMMA)
    u1c = cons * 1.02
    ded = 1 if fluid == "glmmhd" else 0
    eos_kw = dict(pfloor=1e-6, dfloor=1e-6)
    nh = NHYDRO[fluid]
    m0 = hydro.MeshData(ctx, nx, ng, nh, nscalars=2, dx=tuple(g.dx), nblocks=prim.shape[0], cons=cons, prim=prim,
                        with_flux=False)
    m1 = hydro.MeshData(ctx, nx, ng, nh, nscalars=2, dx=tuple(g.dx), nblocks=prim.shape[0], cons=u1c,
                        prim=np.full_like(prim, -7.0), with_flux=False)
    hydro.StageFused(m0, m1, fluid, recon, riemann, L.make_eos(GAMMA, **eos_kw), C_H, 0.5, 0.5, 0.004, dedner=ded,
                     glmmhd_alpha=0.1, mindx=0.07, fill_derived=fill)
    want = H.orc_stage(fluid, recon, riemann, g, cons, u1c, prim, GAMMA, C_H, 0.5, 0.5, 0.004, dedner=ded, alpha=0.1,
                       mindx=0.07)
    _cmp(H.interior(m0.cons_host(), nx, ng), H.interior(want, nx, ng), strict, "cons incl. scalar densities")
    if fill:
        _, want_prim, _ = H.orc_c2p(fluid, g, want, oracle.make_eos(GAMMA, **eos_kw))
        got = (m1 if fill == 2 else m0).prim_host()
        _cmp(H.interior(got, nx, ng), H.interior(want_prim, nx, ng), strict, "prim incl. concentrations")


# ---- direct neighbour addressing (apk_stage_args.face_neighbor) ------------------------------------------------
def _fill_faces_from_neighbors(a, table, nx, ng):
    """host reference of the table's meaning: the ghost zone behind face f of block b (interior extent in the other
    two directions) := the adjacent interior layers of block table[b][f], shifted by one block length"""
    out = a.copy()
    I = [slice(ng, ng + nx[0]), slice(ng, ng + nx[1]), slice(ng, ng + nx[2])]   # interior ranges along x1, x2, x3
    for b, row in enumerate(table):
        for f, nb in enumerate(row):
            if nb < 0:
                continue
            d, hi = f // 2, f % 2
            dst, src = list(I), list(I)
            dst[d] = slice(ng + nx[d], ng + nx[d] + ng) if hi else slice(0, ng)
            src[d] = slice(ng, 2 * ng) if hi else slice(nx[d], nx[d] + ng)
            out[b][:, dst[2], dst[1], dst[0]] = a[nb][:, src[2], src[1], src[0]]
    return out


def _poison_faces(a, table, nx, ng):
    out = a.copy()
    I = [slice(ng, ng + nx[0]), slice(ng, ng + nx[1]), slice(ng, ng + nx[2])]
    for b, row in enumerate(table):
        for f, nb in enumerate(row):
            if nb < 0:
                continue
            d, hi = f // 2, f % 2
            dst = list(I)
            dst[d] = slice(ng + nx[d], ng + nx[d] + ng) if hi else slice(0, ng)
            out[b][:, dst[2], dst[1], dst[0]] = np.nan
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("strict", [True, False], ids=["strict", "fma"])
@pytest.mark.parametrize("fluid,recon,riemann,gam0", [("glmmhd", "dc", "hlld", 0.0), ("glmmhd", "ppm", "hlld", 0.0),
                                                      ("glmmhd", "ppm", "hlld", 0.5), ("glmmhd", "wenoz", "hlld", 0.25),
                                                      ("euler", "plm", "hllc", 0.0), ("euler", "dc", "hlle", 0.0),
                                                      ("glmmhd", "plm", "hlle", 0.5)])
@pytest.mark.parametrize("nx", [(36, 9, 10), (64, 8, 8)], ids=["36x9x10", "64x8x8"])
def test_direct_neighbor_addressing_equals_filled_ghost_zones(request, fluid, recon, riemann, gam0, nx, strict):
    """apk_stage_args.face_neighbor: the stage that reads its neighbours' interiors through the table, with the
    ghost zones behind those faces poisoned, equals the stage on ghost zones filled from the same neighbours -- bit
    for bit, in the updated state, the out-of-place primitives and the time step.  The table mixes other blocks, a
    block that is its own periodic neighbour and faces without an entry (their ghost zones are read as before)."""
    import ctypes as C
    import torch
    from athenapk_amd import hydro
    ctx = _ctx(request, strict)
    ng, prim, g = _case(fluid, recon, nx, kind="smooth", seed=67, nblocks=3)
    table = [[1, 1, 2, -1, 0, -1],
             [0, 2, -1, 2, 1, 1],
             [-1, 0, 1, 0, -1, 2]]
    prim_ref = _fill_faces_from_neighbors(prim, table, nx, ng)
    prim_dir = _poison_faces(prim_ref, table, nx, ng)
    cons = H.prim_to_cons(fluid, prim_ref, GAMMA)
    ded = 1 if fluid == "glmmhd" else 0
    eos = hydro.L.make_eos(GAMMA)
    gam1 = 1.0 - gam0 if gam0 else 1.0
    last = recon != "dc"
    kw = dict(dedner=ded, glmmhd_alpha=0.1, mindx=0.07, fill_derived=2, estimate_dt=last)
    cfg = hydro._cfg(fluid, recon, riemann)

    def run(w, tab):
        a = hydro.MeshData(ctx, nx, ng, NHYDRO[fluid], dx=tuple(g.dx), nblocks=3, cons=cons, prim=w, with_flux=False)
        b = hydro.MeshData(ctx, nx, ng, NHYDRO[fluid], dx=tuple(g.dx), nblocks=3, cons=cons * 1.01, prim=np.full_like(w, -7.0),
                           with_flux=False)
        if recon != "dc":
            assert ctx.lib.apk_stage_split_axis(a.h, C.byref(cfg), 2) == 3
        hydro.StageFused(a, b, fluid, recon, riemann, eos, C_H, gam0, gam1, 0.004, face_neighbor=tab, **kw)
        dt = hydro.StageDt(ctx, 0.3) if last else 0.0
        return H.interior(a.cons_host(), nx, ng), H.interior(b.prim_host(), nx, ng), dt
    want = run(prim_ref, None)
    got = run(prim_dir, torch.tensor(table, dtype=torch.int32, device="cuda"))
    assert np.all(np.isfinite(got[0])) and np.all(np.isfinite(got[1]))
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]) and got[2] == want[2]
    # and the unsplit stage on filled ghost zones is the oracle's
    ref = H.orc_stage(fluid, recon, riemann, g, cons, cons * 1.01, prim_ref, GAMMA, C_H, gam0, gam1, 0.004, dedner=ded,
                      alpha=0.1, mindx=0.07)
    _cmp(want[0], H.interior(ref, nx, ng), strict, "cons")


@pytest.mark.gpu
@pytest.mark.parametrize("strict", [True, False], ids=["strict", "fma"])
@pytest.mark.parametrize("fluid,recon,riemann,from_cons", [("glmmhd", "ppm", "hlld", False), ("glmmhd", "dc", "hlld", True),
                                                           ("glmmhd", "dc", "hlld", False), ("euler", "plm", "hllc", False),
                                                           ("euler", "dc", "hlle", True), ("glmmhd", "wenoz", "hlld", False),
                                                           ("glmmhd", "ppm", "hlld", True), ("glmmhd", "wenoz", "hlld", True)])
def test_x1_strips_in_exchange_buffers_equal_filled_ghost_zones(request, fluid, recon, riemann, from_cons, strict):
    """apk_stage_args.x1_halo: a stage whose x1 ghost columns live in receive-buffer segments ([nvar][nx3][nx2][depth];
    the ghost zones behind those faces poisoned) equals the stage on filled ghost zones bit for bit -- updated state,
    out-of-place primitives, time step -- and the send segments hold the x1 boundary columns of what it stored: the
    conserved state one layer deep (the corrector's place in a VL2 cycle) or the new primitives nghost deep (the
    predictor's).  A face without a segment is read from the block as before.  from_cons: the stage derives its input from
    u1's conserved state -- the segments then hold conserved values (the donor-cell predictor of a prim-free VL2 cycle, the
    finishing marches of the RK integrators)."""
    import torch
    from athenapk_amd import hydro
    ctx = _ctx(request, strict)
    nx = (36, 8, 10)
    ng, prim, g = _case(fluid, recon, nx, kind="smooth", seed=71, nblocks=2)
    table = [[1, 1, -1, -1, -1, -1], [0, -1, -1, -1, -1, -1]]  # x1 faces only; block 1's upper face keeps its ghost zone
    prim_ref = _fill_faces_from_neighbors(prim, table, nx, ng)
    cons_ref = H.prim_to_cons(fluid, prim_ref, GAMMA)
    ded = 1 if fluid == "glmmhd" else 0
    eos = hydro.L.make_eos(GAMMA)
    dc = recon == "dc"
    nv = NHYDRO[fluid]
    rdepth, sdepth, sfield = (1, ng, 1) if dc else (ng, 1, 0)
    kw = dict(dedner=ded, glmmhd_alpha=0.1, mindx=0.07, fill_derived=2, estimate_dt=not dc, prim_from_cons=from_cons)
    assert hydro.StageFollowsX1Halo(hydro.MeshData(ctx, nx, ng, nv, dx=tuple(g.dx), nblocks=2, cons=cons_ref, prim=prim_ref, with_flux=False),
                                    fluid, recon, riemann, eos, 2, ded, prim_from_cons=int(from_cons))
    src = cons_ref if from_cons else prim_ref  # the array the stage takes its input from
    ks, js = slice(ng, ng + nx[2]), slice(ng, ng + nx[1])

    def ghost_cols(b, side, depth):  # the segment of block b's lower / upper x1 ghost strip: the `depth` columns next to the face
        i0 = ng + nx[0] if side else ng - depth
        return np.ascontiguousarray(src[b][:, ks, js, i0:i0 + depth])

    def run(halo):
        w, u = prim_ref, cons_ref
        if halo:
            w, u = _poison_faces(prim_ref, table, nx, ng), (_poison_faces(cons_ref, table, nx, ng) if from_cons else cons_ref)
        a = hydro.MeshData(ctx, nx, ng, nv, dx=tuple(g.dx), nblocks=2, cons=cons_ref * 0.99, prim=w, with_flux=False)
        b = hydro.MeshData(ctx, nx, ng, nv, dx=tuple(g.dx), nblocks=2, cons=u, prim=np.full_like(w, -7.0), with_flux=False)
        xh, send = None, None
        if halo:
            recv = [tuple(torch.tensor(ghost_cols(blk, side, rdepth), device="cuda") if table[blk][side] >= 0 else None for side in range(2))
                    for blk in range(2)]
            send = [tuple(torch.full((nv, nx[2], nx[1], sdepth), -3.0, dtype=torch.float64, device="cuda") if (blk, side) != (1, 0) else None
                          for side in range(2)) for blk in range(2)]
            xh = dict(recv=recv, send=send, recv_depth=rdepth, send_depth=sdepth, send_field=sfield)
        hydro.StageFused(a, b, fluid, recon, riemann, eos, C_H, 0.0, 1.0, 0.004, x1_halo=xh, **kw)
        dt = hydro.StageDt(ctx, 0.3) if not dc else 0.0
        return a.cons_host(), b.prim_host(), dt, send
    want = run(False)
    got = run(True)
    I = lambda x: H.interior(x, nx, ng)
    assert np.all(np.isfinite(I(got[0]))) and np.all(np.isfinite(I(got[1])))
    if strict:
        assert np.array_equal(I(got[0]), I(want[0])) and np.array_equal(I(got[1]), I(want[1])) and got[2] == want[2]
    else:  # (the product build contracts each kernel form's expressions its own way: last-bit differences)
        np.testing.assert_allclose(I(got[0]), I(want[0]), rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(I(got[1]), I(want[1]), rtol=1e-12, atol=1e-13)
        assert abs(got[2] - want[2]) <= 1e-12 * abs(want[2])
    stored = got[1] if sfield else got[0]
    for blk in range(2):
        for side in range(2):
            seg = got[3][blk][side]
            if seg is None:
                continue
            i0 = ng + nx[0] - sdepth if side else ng
            assert np.array_equal(seg.cpu().numpy(), stored[blk][:, ks, js, i0:i0 + sdepth]), (blk, side)
    # (and what it is compared with is the oracle's stage -- on the primitives the oracle derives from the conserved
    # state where the stage derived its own)
    w_in = H.orc_c2p(fluid, g, cons_ref, H.O.make_eos(GAMMA))[1] if from_cons else prim_ref
    ref = H.orc_stage(fluid, recon, riemann, g, cons_ref * 0.99, cons_ref, w_in, GAMMA, C_H, 0.0, 1.0, 0.004, dedner=ded, alpha=0.1, mindx=0.07)
    _cmp(I(want[0]), I(ref), strict, "cons")


@pytest.mark.gpu
@pytest.mark.parametrize("strict", [True, False], ids=["strict", "fma"])
@pytest.mark.parametrize("pitch", ["aligned", 57], ids=["line_aligned", "odd_pitch"])
@pytest.mark.parametrize("fluid,recon,riemann", [("glmmhd", "ppm", "hlld"), ("glmmhd", "dc", "hlld"), ("euler", "plm", "hllc")])
def test_packs_with_explicit_strides_match_the_oracle(request, fluid, recon, riemann, pitch, strict):
    """apk_pack_desc.stride: `pack(b)(v,k,j,i)` is stride-agnostic in the reference (hydro.cpp:1041-1073).  Rows at a
    pitch of their own -- the line-aligned layout of the standalone driver (pitch a multiple of 16 doubles, the first
    interior cell on a 128-byte boundary) and an odd one -- with NaNs in the padding: the flux-array tasks and the fused
    stage (FillDerived out of place + dt) give the oracle's values, bit for bit in the parity build; the turbulence
    driver, whose acceleration field has a layout of its own, refuses such packs."""
    from athenapk_amd import hydro
    ctx = _ctx(request, strict)
    nx = (36, 8, 10)
    ng, prim, g = _case(fluid, recon, nx, kind="smooth", seed=23, nblocks=2)
    cons = H.prim_to_cons(fluid, prim, GAMMA)
    nv, ded, eos = NHYDRO[fluid], (1 if fluid == "glmmhd" else 0), hydro.L.make_eos(GAMMA)
    a = hydro.MeshData(ctx, nx, ng, nv, dx=tuple(g.dx), nblocks=2, cons=cons, prim=prim, row_pitch=pitch)
    assert a.pitch >= nx[0] + 2 * ng and (pitch != "aligned" or (a.pitch % 16 == 0 and (a.cons.data_ptr() // 8 + ng) % 16 == 0))
    hydro.CalculateFluxes(a, fluid, recon, riemann, eos, C_H)
    want = H.orc_fluxes(fluid, recon, riemann, g, prim, GAMMA, C_H)
    for d in range(3):
        _cmp(a.flux_host(d), want[d], strict, "flux%d" % (d + 1))
    b = hydro.MeshData(ctx, nx, ng, nv, dx=tuple(g.dx), nblocks=2, cons=cons * 1.01, prim=np.full_like(prim, -7.0), with_flux=False, row_pitch=pitch)
    hydro.StageFused(a, b, fluid, recon, riemann, eos, C_H, 0.5, 0.5, 0.004, dedner=ded, glmmhd_alpha=0.1, mindx=0.07, fill_derived=2,
                     estimate_dt=recon != "dc")
    ref = H.orc_stage(fluid, recon, riemann, g, cons, cons * 1.01, prim, GAMMA, C_H, 0.5, 0.5, 0.004, dedner=ded, alpha=0.1, mindx=0.07)
    _cmp(H.interior(a.cons_host(), nx, ng), H.interior(ref, nx, ng), strict, "cons")
    _, w_ref, _ = H.orc_c2p(fluid, g, ref, H.O.make_eos(GAMMA))
    _cmp(H.interior(b.prim_host(), nx, ng), H.interior(w_ref, nx, ng), strict, "prim")
    with pytest.raises(hydro.L.ApkError):  # (u0 and u1 must share their strides)
        hydro.StageFused(a, hydro.MeshData(ctx, nx, ng, nv, dx=tuple(g.dx), nblocks=2, cons=cons, with_flux=False, row_pitch="natural"), fluid, recon,
                         riemann, eos, C_H, 0.5, 0.5, 0.004, dedner=ded, glmmhd_alpha=0.1, mindx=0.07)


@pytest.mark.gpu
def test_stage_forms_that_do_not_follow_the_x1_table_say_so(request):
    """apk_stage_x1_halo / APK_ERR_UNSUPPORTED: the one-row donor-cell march (odd nx2), blocks narrower than two strips,
    the flux-array solvers and stages with passive scalars do not follow apk_stage_args.x1_halo -- the query says so and
    the stage refuses instead of reading stale ghost columns."""
    import torch
    from athenapk_amd import hydro
    ctx = _ctx(request, True)
    eos = hydro.L.make_eos(GAMMA)
    for fluid, recon, riemann, nx, ok in [("glmmhd", "dc", "hlld", (36, 9, 10), False), ("glmmhd", "dc", "hlld", (36, 8, 10), True),
                                          ("glmmhd", "ppm", "hlld", (36, 8, 10), True), ("glmmhd", "ppm", "hlld", (12, 8, 10), False),
                                          ("euler", "plm", "llf", (36, 8, 10), False)]:
        ng, prim, g = _case(fluid, recon, nx, kind="smooth", seed=5, nblocks=1)
        cons = H.prim_to_cons(fluid, prim, GAMMA)
        nv = NHYDRO[fluid]
        a = hydro.MeshData(ctx, nx, ng, nv, dx=tuple(g.dx), nblocks=1, cons=cons, prim=prim, with_flux=False)
        b = hydro.MeshData(ctx, nx, ng, nv, dx=tuple(g.dx), nblocks=1, cons=cons, prim=prim.copy(), with_flux=False)
        ded = 1 if fluid == "glmmhd" else 0
        assert hydro.StageFollowsX1Halo(a, fluid, recon, riemann, eos, 2, ded) == ok, (fluid, recon, riemann, nx)
        if not ok and riemann != "llf":
            seg = torch.zeros((nv, nx[2], nx[1], 1), dtype=torch.float64, device="cuda")
            with pytest.raises(hydro.L.ApkError):
                hydro.StageFused(a, b, fluid, recon, riemann, eos, C_H, 0.0, 1.0, 0.004, dedner=ded, glmmhd_alpha=0.1, mindx=0.07,
                                 fill_derived=2, x1_halo=dict(recv=[(seg, None)], send=[(None, None)], recv_depth=1, send_depth=0))


@pytest.mark.gpu
@pytest.mark.parametrize("strict", [True, False], ids=["strict", "fma"])
@pytest.mark.parametrize("kind", ["smooth", "rough"])
@pytest.mark.parametrize("faces", ["filled_ghosts", "face_table"])
@pytest.mark.parametrize("mode", ["input_u1", "input_u1_dt_only", "input_u0_third_buffer", "input_u0_third_buffer_dt_only"])
@pytest.mark.parametrize("fluid,recon,riemann", [("glmmhd", "ppm", "hlld"), ("glmmhd", "wenoz", "hlld"), ("euler", "plm", "hllc")])
def test_two_kernel_stage_takes_its_input_from_the_conserved_state(request, oracle, fluid, recon, riemann, mode, faces, kind, strict):
    """apk_stage_fused in the forms a prim-free RK cycle is made of (round-4 review, item 6), each against
    orc_stage + orc_c2p through the C-ABI:
      prim_from_cons = 1   the input state is u1.cons (gam0 = 0); u0.prim holds NaN, u0.cons something unrelated;
      prim_from_cons = 2   the input state is u0.cons itself (gam0 != 0): the result goes to a third buffer
                           (cons_out_delta) and u0.cons stays as it was, bit for bit;
      fill_derived   = 3   ... and the primitives of the updated cells feed the time-step estimate only: no prim array is
                           written.
    With filled ghost zones, and through a face table with the ghost zones behind its faces poisoned."""
    import ctypes as C
    import torch
    from athenapk_amd import hydro
    if kind == "rough" and not strict:
        pytest.skip("uniform random states reconstruct negative pressures here and there: NaNs, which only the parity build "
                    "is asked to reproduce place for place (as in the registry tests on rough data)")
    ctx = _ctx(request, strict)
    # (hydro PLM takes the single-march form of these stages -- fused3_kernel.hpp -- which wants an even number of x2 rows)
    nx = (36, 10, 9) if fluid == "euler" else (36, 9, 10)
    ng, prim, g = _case(fluid, recon, nx, kind=kind, seed=211, nblocks=3)
    table = [[1, 1, 2, -1, 0, -1],
             [0, 2, -1, 2, 1, 1],
             [-1, 0, 1, 0, -1, 2]]
    use_table = faces == "face_table"
    if use_table:
        prim = _fill_faces_from_neighbors(prim, table, nx, ng)
    state = H.prim_to_cons(fluid, prim, GAMMA)                      # the stage's input, as a conserved state
    _, prim_of_state, bad = H.orc_c2p(fluid, g, state.copy(), oracle.make_eos(GAMMA))
    assert bad == 0
    state_dev = _poison_faces(state, table, nx, ng) if use_table else state
    tab = torch.tensor(table, dtype=torch.int32, device="cuda") if use_table else None
    own_input = mode.startswith("input_u0")
    dt_only = mode.endswith("dt_only")
    gam0 = 0.25 if own_input else 0.0
    bdt = 0.004 if kind == "smooth" else 0.0004       # (uniform random states: a small step keeps the update admissible)
    nh, ded = NHYDRO[fluid], (1 if fluid == "glmmhd" else 0)
    nan = np.full_like(prim, np.nan)
    if own_input:
        u0c, u1c = state_dev, state * 1.01
    else:
        u0c, u1c = np.full_like(state, 123.0), state_dev
    m0 = hydro.MeshData(ctx, nx, ng, nh, dx=tuple(g.dx), nblocks=3, cons=u0c, prim=nan, with_flux=False)
    m1 = hydro.MeshData(ctx, nx, ng, nh, dx=tuple(g.dx), nblocks=3, cons=u1c, prim=np.full_like(prim, -7.0), with_flux=False)
    m2 = hydro.MeshData(ctx, nx, ng, nh, dx=tuple(g.dx), nblocks=3, cons=np.full_like(state, -7.0), with_flux=False) if own_input else None
    cfg = hydro._cfg(fluid, recon, riemann)
    assert ctx.lib.apk_stage_split_axis(m0.h, C.byref(cfg), 0) == 3
    ctx.poll_flags()
    eos = hydro.L.make_eos(GAMMA)
    kw = dict(dedner=ded, glmmhd_alpha=0.1, mindx=0.07, face_neighbor=tab)
    slots = {name: q for q, name in enumerate(hydro.L.TIMING_SLOTS)}

    def launches(name):
        ms, cnt = C.c_double(0.0), C.c_longlong(0)
        assert ctx.lib.apk_kernel_timing_read(ctx.h, slots[name], C.byref(ms), C.byref(cnt)) == 0
        return cnt.value
    ctx.lib.apk_kernel_timing_enable(ctx.h, 1)
    launches("fused_x1"), launches("fused_x3")                        # (read = reset)
    hydro.StageFused(m0, m1, fluid, recon, riemann, eos, C_H, gam0, 1.0 - gam0, bdt, fill_derived=3 if dt_only else 0,
                     estimate_dt=dt_only, prim_from_cons=2 if own_input else 1, cons_out=m2, **kw)
    dt = hydro.StageDt(ctx, 0.3) if dt_only else None
    n1, n3 = launches("fused_x1"), launches("fused_x3")
    ctx.lib.apk_kernel_timing_enable(ctx.h, 0)
    # the form that ran: one march for hydro PLM, the x3 sweep + the finishing march for the others
    assert (n1, n3) == ((1, 0) if (fluid, recon) == ("euler", "plm") else (1, 1))
    rough = kind == "rough"                                            # (NaN cells raise the flags, rightly)
    assert rough or ctx.poll_flags() == 0
    # (gam0 = 0 in the u1-input forms: the oracle's u0 only lends its ghost zones to the ConsToPrim of the whole block below)
    want = H.orc_stage(fluid, recon, riemann, g, state, state * 1.01 if own_input else state,
                       prim_of_state, GAMMA, C_H, gam0, 1.0 - gam0, bdt, dedner=ded, alpha=0.1, mindx=0.07)
    if dt_only:
        # FillDerived acts on the conserved state it converts (adiabatic_hydro.hpp:81: the density "floor" of -1 replaces
        # a NaN density, `(u_d > floor) ? u_d : floor`): the stored state is the oracle's AFTER its ConsToPrim
        want = H.orc_c2p(fluid, g, want, oracle.make_eos(GAMMA))[0]
    got = (m2 if own_input else m0).cons_host()
    _cmp(H.interior(got, nx, ng), H.interior(want, nx, ng), strict, "updated conserved state")
    if own_input:
        assert np.array_equal(m0.cons_host(), u0c, equal_nan=True), "u0.cons is the stage's input: it must stay as it was"
        ghosts = np.ones(got.shape, dtype=bool)
        H.interior(ghosts, nx, ng)[...] = False
        assert np.all(got[ghosts] == -7.0), "interior cells only"
    ctx.poll_flags()
    assert np.all(np.isnan(m0.prim_host())) and np.all(m1.prim_host() == -7.0), "no primitives are stored in these forms"
    if dt_only and not rough:
        _, want_prim, bad = H.orc_c2p(fluid, g, want.copy(), oracle.make_eos(GAMMA))
        want_dt = 0.3 * H.orc_min_dt(fluid, g, want_prim, GAMMA)
        assert bad == 0 and (dt == want_dt if strict else dt == pytest.approx(want_dt, rel=1e-12))
    # refusals: an equation of state with limits the lean forms do not compile; the own-input form without a third buffer
    with pytest.raises(Exception):
        hydro.StageFused(m0, m1, fluid, recon, riemann, hydro.L.make_eos(GAMMA, pfloor=1e-6), C_H, gam0, 1.0 - gam0, bdt,
                         prim_from_cons=2 if own_input else 1, cons_out=m2, **kw)
    if own_input:
        with pytest.raises(Exception):
            hydro.StageFused(m0, m1, fluid, recon, riemann, eos, C_H, gam0, 1.0 - gam0, bdt, prim_from_cons=2, **kw)


@pytest.mark.gpu
def test_direct_neighbor_addressing_rejects_stage_forms_that_read_ghost_zones(request):
    import torch
    from athenapk_amd import hydro
    ctx = _ctx(request, True)
    nx = (8, 8, 8)                                   # narrower than the two-kernel stage wants: three sweeps
    ng, prim, g = _case("glmmhd", "ppm", nx, nblocks=1)
    cons = H.prim_to_cons("glmmhd", prim, GAMMA)
    a = hydro.MeshData(ctx, nx, ng, 9, dx=tuple(g.dx), nblocks=1, cons=cons, prim=prim, with_flux=False)
    b = hydro.MeshData(ctx, nx, ng, 9, dx=tuple(g.dx), nblocks=1, cons=cons, prim=prim, with_flux=False)
    tab = torch.zeros((1, 6), dtype=torch.int32, device="cuda")
    with pytest.raises(Exception):
        hydro.StageFused(a, b, "glmmhd", "ppm", "hlld", hydro.L.make_eos(GAMMA), C_H, 0.0, 1.0, 0.004, dedner=1,
                         fill_derived=2, face_neighbor=tab)
    with pytest.raises(Exception):                   # the extended Dedner source reads neighbouring primitives
        nx2 = (64, 8, 8)
        ng2, p2, g2 = _case("glmmhd", "ppm", nx2, nblocks=1)
        c2 = H.prim_to_cons("glmmhd", p2, GAMMA)
        a2 = hydro.MeshData(ctx, nx2, ng2, 9, dx=tuple(g2.dx), nblocks=1, cons=c2, prim=p2, with_flux=False)
        b2 = hydro.MeshData(ctx, nx2, ng2, 9, dx=tuple(g2.dx), nblocks=1, cons=c2, prim=p2, with_flux=False)
        hydro.StageFused(a2, b2, "glmmhd", "ppm", "hlld", hydro.L.make_eos(GAMMA), C_H, 0.0, 1.0, 0.004, dedner=2,
                         fill_derived=2, face_neighbor=tab)


@pytest.mark.parametrize("strict", [True, False], ids=["strict", "fma"])
@pytest.mark.parametrize("d", [1, 2, 3])
def test_hip_hlld_resolves_isolated_contact_and_rotational_discontinuities(request, strict, d):
    """The HIP HLLD kernels against the SOLVER'S DEFINING PROPERTY instead of against the oracle (whose PPM / HLLD the
    reference pins with no numbers): an isolated contact and an isolated Alfven (rotational) discontinuity are resolved
    exactly -- the face flux is the physical flux of the upwind state (Miyoshi & Kusano 2005, sections 5.2 - 5.3) --
    in every sweep direction, through the flux-array kernel (donor-cell states = the cell values)."""
    from athenapk_amd import hydro
    from test_oracle_cpu import _mhd_state, _phys_flux
    ctx = _ctx(request, strict)
    gamma, c_h, ng = 5.0 / 3.0, 2.0, 2
    dens, p, bn = 1.44, 0.7, 0.9
    ca = abs(bn) / np.sqrt(dens)
    cases = []
    # contact moving right / left: density jump only
    for u in (0.37, -0.29):
        v, b = (u, 0.21, -0.13), (0.8, 0.6, -0.4)
        cases.append((_mhd_state(1.3, v, p, b), _mhd_state(0.4, v, p, b), u > 0.0))
    # rotational discontinuity of the (vn + ca) family drifting right / left
    for drift in (0.2, -0.2):
        vn = drift - ca
        btl = np.array([0.5, 0.3])
        ang = 1.1
        btr = np.array([np.cos(ang) * btl[0] - np.sin(ang) * btl[1], np.sin(ang) * btl[0] + np.cos(ang) * btl[1]])
        vtl = np.array([0.1, -0.2])
        vtr = vtl - np.sign(bn) * (btr - btl) / np.sqrt(dens)
        cases.append((_mhd_state(dens, (vn, vtl[0], vtl[1]), p, (bn, btl[0], btl[1])),
                      _mhd_state(dens, (vn, vtr[0], vtr[1]), p, (bn, btr[0], btr[1])), drift > 0.0))

    def rot(w):     # the x1-normal state turned so that direction d is the normal: (n, t1, t2) -> components (d, d+1, d+2)
        r = w.copy()
        for base in (1, 5):
            for c in range(3):
                r[base + (d - 1 + c) % 3] = w[base + c]
        return r

    n = len(cases)
    nx = [2, 2, 2]
    nx[d - 1] = 2 * n
    N = [m + 2 * ng for m in nx]
    line = np.zeros((9, N[d - 1]))
    for c, (wl, wr, _) in enumerate(cases):
        line[:, ng + 2 * c] = rot(wl)
        line[:, ng + 2 * c + 1] = rot(wr)
    line[:, :ng] = line[:, ng:ng + 1]
    line[:, -ng:] = line[:, -ng - 1:-ng]
    shape = [1, 1, 1]
    shape[3 - d] = N[d - 1]
    w = np.zeros((1, 9, N[2], N[1], N[0]))
    w[0] = line.reshape((9,) + tuple(shape))
    md = hydro.MeshData(ctx, tuple(nx), ng, 9, dx=(0.1, 0.1, 0.1), prim=w)
    hydro.CalculateFluxes(md, "glmmhd", "dc", "hlld", hydro.L.make_eos(gamma), c_h)
    f = md.flux_host(d - 1)[0]
    ivx = d
    for c, (wl, wr, upwind_left) in enumerate(cases):
        at = [ng, ng, ng]
        at[d - 1] = ng + 2 * c + 1
        got = f[:, at[2], at[1], at[0]]
        exact = _phys_flux("glmmhd", rot(wl if upwind_left else wr), gamma, ivx, c_h)
        np.testing.assert_allclose(got, exact, rtol=1e-13, atol=3e-14, err_msg="case %d direction %d" % (c, d))
