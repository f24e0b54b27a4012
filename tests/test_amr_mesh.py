"""Mesh refinement, host logic (no GPU): the block forest built from the deck, the 2:1 balance,
and the index-box plans of the multilevel ghost exchange executed on the host with the oracle's
operators (tests/amr_emulator.py)."""
import os

import numpy as np
import pytest

from amr_emulator import Emulator, exchange_on_ranks, flux_correction_on_ranks, placement

SMR3 = ["parthenon/mesh/refinement=static", "parthenon/mesh/nx1=32", "parthenon/mesh/nx2=32", "parthenon/mesh/nx3=32",
        "parthenon/meshblock/nx1=8", "parthenon/meshblock/nx2=8", "parthenon/meshblock/nx3=8",
        "parthenon/static_refinement0/x1min=-0.05", "parthenon/static_refinement0/x1max=0.05",
        "parthenon/static_refinement0/x2min=-0.05", "parthenon/static_refinement0/x2max=0.05",
        "parthenon/static_refinement0/x3min=0.05", "parthenon/static_refinement0/x3max=0.2",
        "parthenon/static_refinement0/level=2"]
SMR2 = ["parthenon/mesh/refinement=static", "parthenon/mesh/nx1=32", "parthenon/mesh/nx2=24", "parthenon/mesh/nx3=1",
        "parthenon/meshblock/nx1=8", "parthenon/meshblock/nx2=8", "parthenon/meshblock/nx3=1",
        "parthenon/static_refinement0/x1min=-0.45", "parthenon/static_refinement0/x1max=-0.3",
        "parthenon/static_refinement0/x2min=0.3", "parthenon/static_refinement0/x2max=0.45",
        "parthenon/static_refinement0/level=3"]


SMR3_NG4 = [o for o in SMR3] + ["parthenon/mesh/nghost=4", "hydro/reconstruction=ppm"]
SMR1 = ["parthenon/mesh/refinement=static", "parthenon/mesh/nx1=64", "parthenon/mesh/nx2=1", "parthenon/mesh/nx3=1",
        "parthenon/meshblock/nx1=8", "parthenon/meshblock/nx2=1", "parthenon/meshblock/nx3=1",
        "parthenon/static_refinement0/x1min=0.1", "parthenon/static_refinement0/x1max=0.2",
        "parthenon/static_refinement0/level=3"]


def _bc(kind):
    return ["parthenon/mesh/%sx%d_bc=%s" % (io, d, kind) for d in (1, 2, 3) for io in "io"]


def _view(overrides, rank=0, nranks=1):
    from athenapk_amd import decks, driver
    return driver.HostPlan(decks.load("blast"), overrides, rank=rank, nranks=nranks)


def _cell_centres(view, lb, pl):
    i = view.info
    lev, loc, x0, dx = pl[lb]
    act = [True, i.mb[1] > 1, i.mb[2] > 1]
    ax = []
    for d in range(3):
        n = i.mb[d] + 2 * i.ng if act[d] else 1
        g = i.ng if act[d] else 0
        ax.append(x0[d] + (np.arange(n) - g + 0.5) * dx[d])
    return np.meshgrid(ax[2], ax[1], ax[0], indexing="ij")   # z, y, x arrays of shape (nk, nj, ni)


@pytest.mark.parametrize("ov,nblocks,levels", [(SMR3, None, 2), (SMR2, None, 3)])
def test_static_refinement_builds_a_balanced_forest(ov, nblocks, levels):
    v = _view(ov)
    pl = placement(v)
    i = v.info
    assert max(p[0] for p in pl) == levels
    # the leaves tile the domain exactly once
    vol = sum(np.prod([i.mb[d] * p[3][d] for d in range(i.ndim)]) for p in pl)
    dom = np.prod([i.xmax[d] - i.xmin[d] for d in range(i.ndim)])
    assert abs(vol - dom) < 1e-12
    assert i.zones_total == i.nblocks_total * i.mb[0] * i.mb[1] * i.mb[2]
    # 2:1 balance across faces, edges and corners: neighbouring leaves differ by at most one level
    boxes = []
    for lev, loc, x0, dx in pl:
        boxes.append((lev, np.array(x0[:i.ndim]), np.array([x0[d] + i.mb[d] * dx[d] for d in range(i.ndim)])))
    for a, (la, lo_a, hi_a) in enumerate(boxes):
        for lb, lo_b, hi_b in boxes[a + 1:]:
            touch = np.all(lo_a <= hi_b + 1e-12) and np.all(lo_b <= hi_a + 1e-12)
            if touch:
                assert abs(la - lb) <= 1
    # Z-order: a block's children are contiguous
    levs = [p[0] for p in pl]
    assert levs.count(levels) % (2 ** i.ndim) == 0
    # the region asked for is covered at the finest level
    key = "parthenon/static_refinement0/"
    reg = {o.split("=")[0][len(key):]: float(o.split("=")[1]) for o in ov if o.startswith(key) and "level" not in o}
    mid = [0.5 * (reg["x%dmin" % (d + 1)] + reg["x%dmax" % (d + 1)]) for d in range(i.ndim)]
    inside = [p for p, (lev, lo, hi) in zip(pl, boxes) if np.all(lo <= mid) and np.all(np.array(mid) < hi)]
    assert len(inside) == 1 and inside[0][0] == levels


@pytest.mark.parametrize("ov", [SMR3, SMR2, SMR3_NG4, SMR1], ids=["3d", "2d", "3d_ng4", "1d"])
@pytest.mark.parametrize("bc", ["outflow", "reflecting", "periodic"])
def test_multilevel_exchange_is_exact_for_linear_data(oracle, ov, bc):
    """cell averages of a linear function are its values at the cell centres on every level, the
    restriction of those is exact, and so is the minmod prolongation: after the exchange every ghost
    cell inside the domain must hold the function's value at its own centre"""
    v = _view(ov + _bc(bc))
    pl = placement(v)
    em = Emulator(v, oracle)
    i = v.info
    coef = np.array([[1.0, 0.3, -0.2, 0.5], [2.0, -1.1, 0.7, 0.25], [-0.5, 0.05, 0.9, -0.6], [0.1, 1.0, 1.0, 1.0],
                     [3.0, -0.4, 0.2, 0.8]])
    ng = i.ng
    sl = (slice(None), slice(ng, -ng) if i.mb[2] > 1 else slice(None), slice(ng, -ng) if i.mb[1] > 1 else slice(None),
          slice(ng, -ng))
    exact = []
    for lb in range(em.nb):
        z, y, x = _cell_centres(v, lb, pl)
        f = np.stack([c[0] + c[1] * x + c[2] * y + c[3] * z for c in coef])
        exact.append(f)
        em.cons[lb][:] = np.nan
        em.cons[lb][sl] = f[sl]
    em.exchange()
    nchecked = 0
    for lb in range(em.nb):
        z, y, x = _cell_centres(v, lb, pl)
        # (not next to a physical boundary: there the limited slope of the coarse cell sees the
        # boundary condition's ghost value -- or, across a periodic boundary, the jump of the
        # linear function -- and the prolongation is first order)
        inside = np.ones(x.shape, bool)
        for d, c in enumerate((x, y, z)):
            if d < i.ndim:
                inside &= (c > i.xmin[d] + 2 * pl[lb][3][d]) & (c < i.xmax[d] - 2 * pl[lb][3][d])
        got = em.cons[lb]
        assert not np.isnan(got).any(), "block %d has unfilled ghost cells" % lb
        err = np.abs(got - exact[lb])[:, inside]
        assert err.max() < 5e-14, "block %d (level %d): %.3e" % (lb, pl[lb][0], err.max())
        nchecked += inside.sum()
    assert nchecked > em.nb * i.mb[0] * i.mb[1] * i.mb[2]


def test_multilevel_exchange_outflow_and_reflecting_ghosts(oracle):
    """outside the domain: outflow copies the last interior cell, reflecting mirrors with the normal
    momentum flipped -- also on blocks whose ghost zones were prolongated"""
    for bc in ("outflow", "reflecting"):
        v = _view(SMR2 + _bc(bc))
        pl = placement(v)
        em = Emulator(v, oracle)
        i = v.info
        ng = i.ng
        rng = np.random.default_rng(5)
        for lb in range(em.nb):
            em.cons[lb][:] = np.nan
            em.cons[lb][:, :, ng:-ng, ng:-ng] = rng.uniform(1.0, 2.0, (em.nvar, 1, i.mb[1], i.mb[0]))
        em.exchange()
        for lb in range(em.nb):
            lev, loc, x0, dx = pl[lb]
            u = em.cons[lb]
            assert not np.isnan(u).any()
            if loc[0] == 0:                      # inner x1 boundary
                for g in range(ng):
                    src = u[:, :, :, ng] if bc == "outflow" else u[:, :, :, 2 * ng - 1 - g]
                    want = src.copy()
                    if bc == "reflecting":
                        want[1] = -want[1]
                    assert np.array_equal(u[:, :, :, g], want)
            if loc[1] == (i.nx[1] // i.mb[1]) * 2 ** lev - 1:   # outer x2 boundary
                e = ng + i.mb[1] - 1
                for g in range(ng):
                    src = u[:, :, e, :] if bc == "outflow" else u[:, :, e - g, :]
                    want = src.copy()
                    if bc == "reflecting":
                        want[2] = -want[2]
                    assert np.array_equal(u[:, :, e + 1 + g, :], want)


def test_flux_correction_plan_matches_fine_fluxes(oracle):
    """after the correction the coarse block's face flux equals the area average of the fine
    fluxes on every coarse-fine face, and is untouched elsewhere"""
    v = _view(SMR3)
    pl = placement(v)
    em = Emulator(v, oracle)
    i = v.info
    rng = np.random.default_rng(11)
    for d in range(3):
        for lb in range(em.nb):
            em.flux[d][lb][:] = rng.standard_normal(em.shape)
    before = [[f.copy() for f in em.flux[d]] for d in range(3)]
    em.flux_correction()
    ng, mb = i.ng, i.mb[0]
    changed = 0
    for d in range(3):
        regs = v.regions("amr_flux%d" % (d + 1))
        assert len(regs) > 0
        touched = [np.zeros(em.shape, bool) for _ in range(em.nb)]
        for reg in regs:
            cb, fb = reg.dst_block, reg.src_block
            assert pl[fb][0] == pl[cb][0] + 1
            # destination plane in (k, j, i)
            off = reg.dst_off
            k0, rem = divmod(off, em.shape[2] * em.shape[3])
            j0, i0 = divmod(rem, em.shape[3])
            ext = list(reg.ext)
            sel = (slice(None), slice(k0, k0 + ext[2]), slice(j0, j0 + ext[1]), slice(i0, i0 + ext[0]))
            touched[cb][sel] = True
            # the fine block's face: its low face if it sits on my high side
            lo_side = (i0 if d == 0 else (j0 if d == 1 else k0)) == ng   # my low face -> child's high face
            fidx = ng + mb if lo_side else ng
            ff = before[d][fb]
            fsel = [slice(None), slice(ng, ng + mb), slice(ng, ng + mb), slice(ng, ng + mb)]
            fsel[3 - d] = fidx
            plane = ff[tuple(fsel)]          # (nvar, a, b) over the two transverse directions (slower, faster)
            avg = 0.25 * (plane[:, 0::2, 0::2] + plane[:, 1::2, 0::2] + plane[:, 0::2, 1::2] + plane[:, 1::2, 1::2])
            got = np.squeeze(em.flux[d][cb][sel], axis=3 - d)
            assert np.allclose(got, avg, rtol=1e-14, atol=1e-15)
            changed += 1
        for lb in range(em.nb):
            assert np.array_equal(em.flux[d][lb][~touched[lb]], before[d][lb][~touched[lb]])
    assert changed > 0


@pytest.mark.parametrize("nranks", [2, 3, 7])
@pytest.mark.parametrize("ov,bc", [(SMR3, "periodic"), (SMR2, "outflow"), (SMR3_NG4, "reflecting")], ids=["3d", "2d", "3d_ng4"])
def test_distributed_plans_reproduce_the_one_rank_exchange(oracle, ov, bc, nranks):
    """the forest's blocks in Z-order ranges over nranks ranks, every rank executing its share of the
    plans and one halo / one flux-correction message per peer: same ghost zones and corrected fluxes as
    on one rank, bit for bit; equal shares; symmetric message sizes"""
    one = _view(ov + _bc(bc))
    ref = Emulator(one, oracle)
    rng = np.random.default_rng(17)
    for lb in range(ref.nb):
        ref.cons[lb][:] = rng.uniform(0.5, 2.0, ref.shape)
        for d in range(3):
            ref.flux[d][lb][:] = rng.standard_normal(ref.shape)
    where = {(one.block_level(lb), one.block_gid(lb)[1]): lb for lb in range(ref.nb)}
    views = [_view(ov + _bc(bc), rank=r, nranks=nranks) for r in range(nranks)]
    ems = [Emulator(v, oracle) for v in views]
    counts = [e.nb for e in ems]
    assert sum(counts) == ref.nb and max(counts) - min(counts) <= 1
    first = 0
    for v, e in zip(views, ems):
        for lb in range(e.nb):
            assert v.block_gid(lb)[0] == first + lb          # contiguous Z-order ranges
            g = where[(v.block_level(lb), v.block_gid(lb)[1])]
            assert g == first + lb
            e.cons[lb][:] = ref.cons[g]
            for d in range(3):
                e.flux[d][lb][:] = ref.flux[d][g]
        first += e.nb
    ref.exchange()
    ref.flux_correction()
    exchange_on_ranks(ems)
    flux_correction_on_ranks(ems)
    first = 0
    for e in ems:
        for lb in range(e.nb):
            assert np.array_equal(e.cons[lb], ref.cons[first + lb])
            for d in range(ref.info.ndim):
                assert np.array_equal(e.flux[d][lb], ref.flux[d][first + lb])
        first += e.nb
    # the halo message carries strips of every kind, the flux message only planes: much smaller
    halo = sum(sc for v in views for _, sc, _ in v.messages("halo"))
    flux = sum(sc for v in views for _, sc, _ in v.messages("flux"))
    assert halo > 0 and 0 < flux < halo


SMR2_NG4 = [o for o in SMR2] + ["parthenon/mesh/nghost=4", "hydro/reconstruction=ppm"]


@pytest.mark.parametrize("nranks", [1, 3])
@pytest.mark.parametrize("bc", ["periodic", "reflecting"])
@pytest.mark.parametrize("ov", [SMR3_NG4, SMR2_NG4], ids=["3d", "2d"])
def test_stage_loop_exchanges_fill_what_they_promise(oracle, ov, bc, nranks):
    """The three cheaper exchanges of the refined-mesh stage loop against the complete one on the same random state
    (four ghost layers, three / four levels, blocks over 1 and 3 ranks).  Faces only: every ghost cell straight behind a
    face as in the complete exchange, nothing else written.  Direct: the same, minus exactly the zones behind faces shared
    with a same-rank block of the same level.  Shell: every ghost cell at most two layers outside the interior --
    edges and corners included -- as in the complete exchange, nothing deeper written (on the periodic mesh; physical
    boundaries copy whole transverse extents)."""
    from amr_emulator import stage_loop_exchange_on_ranks
    views = [_view(ov + _bc(bc), rank=r, nranks=nranks) for r in range(nranks)]
    info = views[0].refresh_info()
    ng, mb = info.ng, info.mb
    act = [True, mb[1] > 1, mb[2] > 1]

    def fresh():
        ems = [Emulator(v, oracle) for v in views]
        r2 = np.random.default_rng(23)
        for e in ems:
            for lb in range(e.nb):
                e.cons[lb][:] = r2.uniform(0.5, 2.0, e.shape)
        return ems

    ref = fresh()
    exchange_on_ranks(ref)
    start = fresh()
    K, J, I = np.meshgrid(*[np.arange(mb[d] + 2 * ng if act[d] else 1) for d in (2, 1, 0)], indexing="ij")
    depth = [np.maximum(ng - c, 0) + np.maximum(c - (ng + n - 1), 0) if a else 0 * c
             for c, n, a in ((I, mb[0], act[0]), (J, mb[1], act[1]), (K, mb[2], act[2]))]
    nghost = sum((d > 0).astype(int) for d in depth)
    zone = [depth[0] * (I < ng) > 0, depth[0] * (I >= ng) > 0, depth[1] * (J < ng) > 0, depth[1] * (J >= ng) > 0,
            depth[2] * (K < ng) > 0, depth[2] * (K >= ng) > 0]
    # the same-level same-rank neighbours, from the forest
    where = [{(v.block_level(lb), v.block_gid(lb)[1]): lb for lb in range(e.nb)} for v, e in zip(views, start)]
    nroot = [info.nx[d] // mb[d] for d in range(3)]
    for mode in ("faces", "direct", "shell"):
        ems = fresh()
        stage_loop_exchange_on_ranks(ems, mode)
        skipped = 0
        for r, (v, e) in enumerate(zip(views, ems)):
            for lb in range(e.nb):
                if mode == "shell":
                    filled = (nghost > 0) & (np.maximum(np.maximum(depth[0], depth[1]), depth[2]) <= 2)
                else:
                    filled = nghost == 1
                    if mode == "direct":
                        lev, loc = v.block_level(lb), v.block_gid(lb)[1]
                        for f in range(6):
                            if not act[f // 2]:
                                continue
                            nloc = list(loc)
                            nloc[f // 2] += 1 if f % 2 else -1
                            n = nroot[f // 2] << lev
                            if bc == "periodic":
                                nloc[f // 2] %= n
                            if (lev, tuple(nloc)) in where[r]:
                                filled = filled & ~zone[f]
                                skipped += 1
                got, want, was = e.cons[lb], ref[r].cons[lb], start[r].cons[lb]
                assert np.array_equal(got[:, filled], want[:, filled]), (mode, r, lb)
                untouched = (nghost > 0) & ~filled
                if bc == "periodic":  # (physical boundaries copy whole transverse extents, stale rows included)
                    assert np.array_equal(got[:, untouched], was[:, untouched]), (mode, r, lb)
                assert np.array_equal(got[:, nghost == 0], was[:, nghost == 0])
        assert mode != "direct" or skipped > 0
    if nranks > 1:  # the shallow exchange sends less
        full = sum(sc for v in views for _, sc, _ in v.messages("halo"))
        faces = sum(sc for v in views for _, sc, _ in v.messages("halo_faces"))
        shell = sum(sc for v in views for _, sc, _ in v.messages("halo_shell"))
        assert 0 < faces < full and 0 < shell < full


def _check_forest(v):
    """tiling, 2:1 balance across faces / edges / corners (periodic wrap included)"""
    pl = placement(v)
    i = v.info
    nd = i.ndim
    vol = sum(np.prod([i.mb[d] * p[3][d] for d in range(nd)]) for p in pl)
    assert abs(vol - np.prod([i.xmax[d] - i.xmin[d] for d in range(nd)])) < 1e-12
    leaves = {(p[0],) + tuple(p[1][:nd]) for p in pl}
    assert len(leaves) == len(pl)
    nroot = [i.nx[d] // i.mb[d] for d in range(nd)]
    finest = max(p[0] for p in pl)
    # occupancy map at the finest level -> level of the covering leaf
    shape = [nroot[d] << finest for d in range(nd)]
    lev = -np.ones(shape, int)
    for p in pl:
        w = 1 << (finest - p[0])
        sl = tuple(slice(p[1][d] * w, (p[1][d] + 1) * w) for d in range(nd))
        assert np.all(lev[sl] == -1)
        lev[sl] = p[0]
    assert np.all(lev >= 0)
    for shift in np.ndindex(*([3] * nd)):
        off = [s - 1 for s in shift]
        nb = np.roll(lev, off, axis=tuple(range(nd)))
        assert np.abs(nb - lev).max() <= 1
    return pl


@pytest.mark.parametrize("dims", [2, 3])
def test_random_regridding_keeps_the_forest_balanced(oracle, dims):
    """random refine / derefine requests for 12 rounds (derefine_count = 2): after every round the
    leaves tile the domain, touching leaves differ by at most one level (periodic wrap included), blocks
    only merge after two consecutive requests of all siblings, and the multilevel exchange built for
    the new forest is still exact for linear data"""
    ov = ["parthenon/mesh/refinement=adaptive", "parthenon/mesh/numlevel=4", "parthenon/mesh/derefine_count=2",
          "parthenon/mesh/nx1=32", "parthenon/mesh/nx2=32", "parthenon/meshblock/nx1=8", "parthenon/meshblock/nx2=8"]
    ov += ["parthenon/mesh/nx3=16", "parthenon/meshblock/nx3=8"] if dims == 3 else ["parthenon/mesh/nx3=1", "parthenon/meshblock/nx3=1"]
    v = _view(ov + _bc("periodic"))        # (the balance check wraps around: neighbours across the boundary count)
    rng = np.random.default_rng(100 + dims)
    sizes = []
    for rnd in range(12):
        n = v.refresh_info().nblocks_total
        p_ref = 0.15 if rnd < 5 else 0.03
        tags = rng.choice([1, 0, -1], size=n, p=[p_ref, 0.35, 0.65 - p_ref])
        before = {(p[0],) + tuple(p[1]) for p in placement(v)}
        v.apply_tags(tags)
        pl = _check_forest(v)
        after = {(p[0],) + tuple(p[1]) for p in pl}
        if rnd == 0:
            assert not any(k[0] < min(b[0] for b in before) for k in after)    # nothing merges in the first round
        sizes.append(len(pl))
    assert max(sizes) > sizes[0] and max(p[0] for p in pl) >= 1 and min(sizes[6:]) < max(sizes)
    # the plans of the final forest: linear data exact
    pl = placement(v)
    em = Emulator(v, oracle)
    i = v.info
    ng = i.ng
    sl = (slice(None), slice(ng, -ng) if i.mb[2] > 1 else slice(None), slice(ng, -ng), slice(ng, -ng))
    for lb in range(em.nb):
        z, y, x = _cell_centres(v, lb, pl)
        f = np.stack([1.0 + 0.3 * x - 0.2 * y + 0.5 * z * (dims == 3)] * em.nvar)
        em.cons[lb][:] = np.nan
        em.cons[lb][sl] = f[sl]
    em.exchange()
    for lb in range(em.nb):
        z, y, x = _cell_centres(v, lb, pl)
        f = 1.0 + 0.3 * x - 0.2 * y + 0.5 * z * (dims == 3)
        inside = np.ones(x.shape, bool)
        for d, c in enumerate((x, y, z)):
            if d < i.ndim:
                inside &= (c > i.xmin[d] + 2 * pl[lb][3][d]) & (c < i.xmax[d] - 2 * pl[lb][3][d])
        assert not np.isnan(em.cons[lb]).any()
        assert np.abs(em.cons[lb][0] - f)[inside].max() < 5e-14


# ---- the plans against a restatement that knows only the forest (tests/amr_oracle.py) ------------------------------
def _forest_oracle(view, oracle, fluid="euler", recon="plm", riemann="hlle", integrator="vl2"):
    from amr_oracle import RefinedMeshOracle
    i = view.refresh_info()
    pl = placement(view)
    nrb = [i.nx[d] // i.mb[d] for d in range(3)]
    return RefinedMeshOracle(oracle, fluid, recon, riemann, integrator, nrb, tuple(i.mb), i.ng, tuple(i.xmin), tuple(i.xmax),
                             [(p[0], p[1]) for p in pl], i.gamma, i.cfl)


@pytest.mark.parametrize("ov", [SMR3, SMR3_NG4], ids=["3d", "3d_ng4"])
def test_plans_reproduce_the_forest_oracle(oracle, ov):
    """The driver's index-box plans (executed by the emulator) and the restatement that classifies every ghost region
    by looking up who covers the neighbouring slot fill every ghost cell, every coarse-buffer cell the prolongation
    reads and every coarse face flux with the same bits."""
    v = _view(ov + _bc("periodic"))
    em = Emulator(v, oracle)
    fo = _forest_oracle(v, oracle)
    assert fo.cng == em.cng and fo.shape == em.shape and max(l for l, _ in fo.leaves) == 2
    rng = np.random.default_rng(5)
    for lb in range(em.nb):
        u = rng.uniform(0.5, 2.0, em.shape)
        em.cons[lb][:] = u
        fo.cons[lb][:] = u
    em.exchange()
    fo.exchange()
    for lb in range(em.nb):
        assert np.array_equal(em.cons[lb], fo.cons[lb]), "block %d (level %d)" % (lb, fo.leaves[lb][0])
    flux = [[rng.standard_normal(em.shape) for _ in range(em.nb)] for _ in range(3)]
    for d in range(3):
        for lb in range(em.nb):
            em.flux[d][lb][:] = flux[d][lb]
    em.flux_correction()
    fo.flux_correction(flux)
    nfix = 0
    for d in range(3):
        for lb in range(em.nb):
            assert np.array_equal(em.flux[d][lb], flux[d][lb]), "flux%d of block %d" % (d + 1, lb)
    assert any(fo.classify(l, tuple(lx[q] + (1 if q == 0 else 0) for q in range(3)))[0] == "finer" for l, lx in fo.leaves)


@pytest.mark.parametrize("fluid,recon,riemann,integ,ng", [("euler", "plm", "hlle", "vl2", 2), ("glmmhd", "ppm", "hlld", "vl2", 4),
                                                          ("glmmhd", "wenoz", "hlle", "rk3", 4)])
def test_forest_oracle_on_a_one_level_forest_is_the_uniform_mini_driver(oracle, fluid, recon, riemann, integ, ng):
    """pins the time loop of tests/amr_oracle.py: on a forest of root blocks only it must reproduce the oracle's
    uniform-mesh mini-driver (oracle/sim.c) bit for bit -- state, time step and c_h"""
    from amr_oracle import RefinedMeshOracle
    box = dict(xmin=(-0.5, -0.5, -0.5), xmax=(0.5, 0.5, 0.5))
    o = oracle.Sim(fluid=fluid, recon=recon, riemann=riemann, integrator=integ, nx=(16, 16, 16), mb=(8, 8, 8), ng=ng, cfl=0.3,
                   gamma=5.0 / 3.0, **box)
    o.pgen("blast", radius_outer=0.2, radius_inner=0.1, pressure_ratio=100.0, x1_0=0.013, x2_0=-0.021, x3_0=0.1)
    nb = o.nblocks
    leaves = [(0, (b % 2, (b // 2) % 2, b // 4)) for b in range(nb)]
    fo = RefinedMeshOracle(oracle, fluid, recon, riemann, integ, (2, 2, 2), (8, 8, 8), ng, box["xmin"], box["xmax"], leaves,
                           5.0 / 3.0, 0.3)
    fo.initialize([np.array(o.cons(b)) for b in range(nb)])
    for _ in range(3):
        assert fo.dt == o.dt
        o.step()
        fo.step()
        assert fo.c_h == o.c_h
        for b in range(nb):
            assert np.array_equal(fo.cons[b], o.cons(b)) and np.array_equal(fo.prim[b], o.prim(b), equal_nan=True)


# ---- frozen refined-mesh fixture (tests/golden/amr_fixture.npz, made by tests/golden/make_amr_fixture.py) -----------
def _amr_fixture():
    import importlib.util
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("make_amr_fixture", os.path.join(here, "golden", "make_amr_fixture.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod, np.load(os.path.join(here, "golden", "amr_fixture.npz"))


@pytest.mark.parametrize("case", ["mhd_ppm_hlld_vl2", "hydro_plm_hllc_rk3"])
def test_forest_oracle_reproduces_the_frozen_refined_mesh_fixture(oracle, case):
    """the driver's tree still builds the stored forest, and the live refined-mesh oracle still produces the stored
    state and time steps from the closed-form initial state -- bit for bit"""
    mod, gold = _amr_fixture()
    leaves = mod.forest()
    assert [l for l, _ in leaves] == list(gold["levels"]) and [list(lx) for _, lx in leaves] == gold["lx"].tolist()
    final, dts, t = mod.run(case, leaves)
    assert np.array_equal(final, gold[case + "_final"]) and np.array_equal(dts, gold[case + "_dt"]) and t == float(gold[case + "_time"])
