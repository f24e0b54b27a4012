"""The crafted special cases of SURVEY 8(c)(2) (tests/golden/edge_cases.npz + .json, made by
tests/golden/make_edge_cases.py): identical states, Bx = 0, B = 0, supersonic either way, every HLLD
region and degenerate branch, HLLC's clipped contact pressure, PPM's extremum / round-off /
overshoot limiters, every floor and ceiling of ConsToPrim -- each with the list of reference
branches it takes (oracle branch tracing).

  CPU : the oracle reproduces every stored output and branch mask, and the cases reach every branch
  GPU : both HIP builds against the STORED outputs (not the live oracle), through the real kernels:
        Riemann cases as neighbouring cells of donor-cell blocks (flux-array kernels and the fused
        donor-cell stage), PPM stencils as the pencils of a block (flux arrays + one general stage
        of the two-kernel path), ConsToPrim cases as cells of a block."""
import json
import os

import numpy as np
import pytest

import helpers as H

GDIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GDIR, "edge_cases.npz"))


@pytest.fixture(scope="module")
def meta():
    with open(os.path.join(GDIR, "edge_cases.json")) as f:
        return json.load(f)


REQUIRED = ["hlld_fl", "hlld_fr", "hlld_lstar", "hlld_ldstar", "hlld_rdstar", "hlld_rstar", "hlld_deg_l", "hlld_deg_r",
            "hlld_deg_dst", "hllc_am_pos", "hllc_cp_clip", "hllc_ql", "hllc_qr", "ppm_lim_m", "ppm_lim_p", "ppm_extremum",
            "ppm_roundoff", "ppm_ratio_big", "ppm_over_m", "ppm_over_p", "c2p_dfloor", "c2p_vceil", "c2p_pfloor",
            "c2p_efloor", "c2p_eceil"]


# ---- CPU: the oracle against the fixture ---------------------------------------------------------------------
def test_cases_reach_every_special_branch(meta):
    hits = meta["branch_hit_counts"]
    assert all(hits.get(b, 0) >= 1 for b in REQUIRED), {b: hits.get(b, 0) for b in REQUIRED}
    # the block made of the crafted stencils meets the PPM branches in every sweep direction
    bc = meta["block"]["branch_hit_counts"]
    for d in (1, 2, 3):
        for b in ("ppm_lim_m", "ppm_lim_p", "ppm_extremum", "ppm_ratio_big", "ppm_over_m", "ppm_over_p"):
            assert bc["dir%d_%s" % (d, b)] > 0


def test_oracle_reproduces_riemann_cases_and_branches(oracle, gold, meta):
    for n, c in enumerate(meta["riemann"]):
        for ivx in (1, 2, 3):
            f, m = oracle.riemann_many_traced(c["fluid"], c["riemann"], ivx, gold["riemann_%02d_wl_dir%d" % (n, ivx)],
                                              gold["riemann_%02d_wr_dir%d" % (n, ivx)], meta["gamma"], meta["c_h"])
            assert np.array_equal(f[0], gold["riemann_%02d_flux_dir%d" % (n, ivx)], equal_nan=True), c["label"]
            assert int(m[0]) == c["mask"] and oracle.trace_names(int(m[0])) == c["branches"], c["label"]


def test_oracle_reproduces_ppm_stencils_and_branches(oracle, gold, meta):
    ql, qr, m = oracle.recon_many_traced("ppm", gold["ppm_q"])
    assert np.array_equal(ql, gold["ppm_ql"]) and np.array_equal(qr, gold["ppm_qr"])
    assert [int(x) for x in m] == [c["mask"] for c in meta["ppm"]]
    for rec in ("plm", "wenoz", "weno3", "limo3"):
        a, b = oracle.recon_many(rec, gold["ppm_q"], dx=0.1, n=0)
        assert np.array_equal(a, gold["%s_ql" % rec], equal_nan=True) and np.array_equal(b, gold["%s_qr" % rec], equal_nan=True)


def test_oracle_reproduces_cons_to_prim_cases_and_branches(oracle, gold, meta):
    for n, c in enumerate(meta["c2p"]):
        u2, w, st, m = oracle.c2p_many_traced("glmmhd", oracle.make_eos(meta["gamma"], **c["eos"]), [gold["c2p_%02d_u" % n]])
        assert np.array_equal(u2[0], gold["c2p_%02d_u_after" % n]) and np.array_equal(w[0], gold["c2p_%02d_w" % n]), c["label"]
        assert int(m[0]) == c["mask"] and int(st[0]) == c["status"], c["label"]


def test_oracle_reproduces_the_crafted_block(oracle, gold, meta):
    b = meta["block"]
    g = H.geom("glmmhd", tuple(b["nx"]), b["ng"], 0, tuple(b["dx"]))
    w = gold["block_prim"]
    fl = H.orc_fluxes("glmmhd", "ppm", "hlld", g, w, meta["gamma"], meta["c_h"])
    for d in range(3):
        assert np.array_equal(fl[d], gold["block_flux%d" % (d + 1)], equal_nan=True)
    cons = H.prim_to_cons("glmmhd", w, meta["gamma"])
    st = H.orc_stage("glmmhd", "ppm", "hlld", g, cons, cons * 1.01, w, meta["gamma"], meta["c_h"], b["gam0"], b["gam1"],
                     b["beta_dt"], dedner=1, alpha=b["alpha"], mindx=b["mindx"])
    assert np.array_equal(st, gold["block_stage"])


# ---- GPU: both builds against the stored outputs -----------------------------------------------------------------
def _close(got, want, strict, what):
    if strict:
        assert np.array_equal(got, want, equal_nan=True), "%s: max abs diff %.3e" % (what, np.nanmax(np.abs(got - want)))
    else:
        scale = np.nanmax(np.abs(want)) + 1e-300
        assert np.nanmax(np.abs(got - want)) <= 1e-12 * scale, "%s: %.3e of scale %.3e" % (what, np.nanmax(np.abs(got - want)), scale)


def _riemann_block(cases, d, nv, gold, ng):
    """donor-cell block with the cases laid along direction d: interior cells 2c, 2c+1 hold (wl, wr) of
    case c; two cells wide in the other directions; ghost zones repeat the nearest interior cell"""
    n = len(cases)
    nx = [2, 2, 2]
    nx[d - 1] = 2 * n
    N = [m + 2 * ng for m in nx]
    w = np.zeros((1, nv, N[2], N[1], N[0]))
    line = np.zeros((nv, N[d - 1]))
    for c, idx in enumerate(cases):
        line[:, ng + 2 * c] = gold["riemann_%02d_wl_dir%d" % (idx, d)]
        line[:, ng + 2 * c + 1] = gold["riemann_%02d_wr_dir%d" % (idx, d)]
    line[:, :ng] = line[:, ng:ng + 1]
    line[:, -ng:] = line[:, -ng - 1:-ng]
    shape = [1, 1, 1]
    shape[3 - d] = N[d - 1]
    w[0] = line.reshape((nv,) + tuple(shape))
    return tuple(nx), w


@pytest.mark.gpu
@pytest.mark.parametrize("strict", [True, False], ids=["strict", "fma"])
@pytest.mark.parametrize("fluid,riemann", [("glmmhd", "hlld"), ("glmmhd", "hlle"), ("glmmhd", "llf"), ("euler", "hllc"),
                                           ("euler", "hlle"), ("euler", "llf")])
def test_hip_riemann_solvers_on_crafted_states(request, gold, meta, fluid, riemann, strict):
    from athenapk_amd import hydro
    ctx = request.getfixturevalue("gpu_ctx_strict" if strict else "gpu_ctx_fast")
    cases = [n for n, c in enumerate(meta["riemann"]) if c["fluid"] == fluid and c["riemann"] == riemann]
    assert cases
    nv, ng = H.NHYDRO[fluid], 2
    eos = hydro.L.make_eos(meta["gamma"])
    for d in (1, 2, 3):
        nx, w = _riemann_block(cases, d, nv, gold, ng)
        md = hydro.MeshData(ctx, nx, ng, nv, dx=(0.1, 0.1, 0.1), prim=w)
        hydro.CalculateFluxes(md, fluid, "dc", riemann, eos, meta["c_h"], tight=(riemann == "llf"))
        f = md.flux_host(d - 1)[0]
        for c, idx in enumerate(cases):
            at = [ng, ng, ng]
            at[d - 1] = ng + 2 * c + 1                      # lower d-face of the cell holding wr
            got = f[:, at[2], at[1], at[0]]
            _close(got, gold["riemann_%02d_flux_dir%d" % (idx, d)], strict, "%s dir %d" % (meta["riemann"][idx]["label"], d))


@pytest.mark.gpu
@pytest.mark.parametrize("strict", [True, False], ids=["strict", "fma"])
def test_hip_kernels_on_the_block_of_crafted_stencils(request, gold, meta, strict):
    """flux-array kernels and one general stage (gam0 = gam1 = 1/2, Dedner) of the two-kernel fused
    path (nx1 = 32) on the block whose pencils are the crafted PPM stencils"""
    import ctypes as C
    from athenapk_amd import hydro
    ctx = request.getfixturevalue("gpu_ctx_strict" if strict else "gpu_ctx_fast")
    b = meta["block"]
    nx, ng, dx = tuple(b["nx"]), b["ng"], tuple(b["dx"])
    w = gold["block_prim"]
    eos = hydro.L.make_eos(meta["gamma"])
    md = hydro.MeshData(ctx, nx, ng, 9, dx=dx, prim=w)
    hydro.CalculateFluxes(md, "glmmhd", "ppm", "hlld", eos, meta["c_h"])
    for d in range(3):
        _close(md.flux_host(d), gold["block_flux%d" % (d + 1)], strict, "flux %d" % (d + 1))
    cons = H.prim_to_cons("glmmhd", w, meta["gamma"])
    m0 = hydro.MeshData(ctx, nx, ng, 9, dx=dx, cons=cons, prim=w, with_flux=False)
    m1 = hydro.MeshData(ctx, nx, ng, 9, dx=dx, cons=cons * 1.01, with_flux=False)
    cfg = hydro._cfg("glmmhd", "ppm", "hlld")
    assert ctx.lib.apk_stage_split_axis(m0.h, C.byref(cfg), 0) == 3          # the two-kernel form
    hydro.StageFused(m0, m1, "glmmhd", "ppm", "hlld", eos, meta["c_h"], b["gam0"], b["gam1"], b["beta_dt"], dedner=1,
                     glmmhd_alpha=b["alpha"], mindx=b["mindx"])
    _close(H.interior(m0.cons_host(), nx, ng), H.interior(gold["block_stage"], nx, ng), strict, "stage")


@pytest.mark.gpu
@pytest.mark.parametrize("strict", [True, False], ids=["strict", "fma"])
def test_hip_cons_to_prim_floors_and_ceilings(request, gold, meta, strict):
    from athenapk_amd import hydro
    ctx = request.getfixturevalue("gpu_ctx_strict" if strict else "gpu_ctx_fast")
    for n, c in enumerate(meta["c2p"]):
        nx, ng = (4, 1, 1), 2
        u = np.zeros((1, 9, 1, 1, nx[0] + 2 * ng))
        u[0, :, 0, 0, :] = gold["c2p_%02d_u" % n][:, None]
        md = hydro.MeshData(ctx, nx, ng, 9, cons=u, prim=np.zeros_like(u), with_flux=False)
        hydro.ConservedToPrimitive(md, "glmmhd", hydro.L.make_eos(meta["gamma"], **c["eos"]))
        _close(md.prim_host()[0, :, 0, 0, ng], gold["c2p_%02d_w" % n], strict, c["label"] + " (prim)")
        _close(md.cons_host()[0, :, 0, 0, ng], gold["c2p_%02d_u_after" % n], strict, c["label"] + " (cons)")
        if c["status"] == 0:
            assert ctx.poll_flags() == 0
        else:
            ctx.poll_flags()
