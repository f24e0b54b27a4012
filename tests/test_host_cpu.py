"""CPU tests of the host side (no GPU, no compute calls): the C-ABI library loads and exports
every symbol the headers declare, deck parsing mirrors Hydro::Initialize's option handling,
and the Morton partition + ghost-exchange plans are correct (executed here with numpy and
compared with the oracle's whole-mesh ghost fill)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(apk_[a-z0-9_]+)\s*\(", text)))


@pytest.mark.parametrize("strict", [False, True], ids=["fma", "strict"])
def test_library_loads_and_exports_every_declared_symbol(strict):
    from athenapk_amd import lib as L
    lib = L.load(strict)
    declared = _declared_functions("apk_amd.h") + _declared_functions("apk_host.h")
    assert len(declared) > 40
    for name in declared:
        assert hasattr(lib, name), "%s declared in include/ but not exported" % name
    assert set(declared) == set(L.SYMBOLS)  # the ctypes table binds exactly the declared API
    assert lib.apk_version() == 1
    assert lib.apk_fp_strict() == int(strict)


def test_no_device_fails_loudly_without_fallback():
    """On a box without a gfx950 GPU the context cannot be created: the product has no CPU path."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from athenapk_amd import lib as L
    lib = L.load()
    h = C.c_void_p()
    assert lib.apk_create(C.byref(h)) == L.APK_ERR_NO_DEVICE
    assert not h.value


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "athenapk_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".hpp", ".h")) or f == "Makefile":
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.replace("the oracle", "").replace("CPU oracle", ""), (
                    "%s mentions the oracle" % os.path.join(dirpath, f))


# ---- deck handling (Hydro::Initialize) -------------------------------------------------------------------
def _plan(deck, overrides=(), rank=0, nranks=1):
    from athenapk_amd import decks, driver
    return driver.HostPlan(decks.load(deck), list(overrides), rank=rank, nranks=nranks)


def test_deck_defaults_and_overrides():
    from athenapk_amd import lib as L
    p = _plan("linear_wave3d")
    i = p.info
    assert (i.fluid, i.recon, i.riemann, i.integrator) == (L.FLUID["euler"], L.RECON["plm"], L.RIEMANN["hlle"],
                                                           L.INTEGRATOR["rk2"])
    assert list(i.nx) == [64, 32, 32] and i.ng == 2 and i.nhydro == 5 and i.cfl == 0.3
    assert i.nblocks_total == 1 and i.ndim == 3
    # "test = true" reinterprets tlim as wave periods (linear_wave.cpp:169-175): lambda = 1, a = 1
    assert p.tlim == pytest.approx(1.0, rel=1e-14)
    p = _plan("linear_wave3d", ["hydro/fluid=glmmhd", "hydro/reconstruction=wenoz", "parthenon/mesh/nghost=3",
                                "parthenon/time/integrator=rk3", "hydro/riemann=hlld"])
    assert p.info.nhydro == 9 and p.info.recon == L.RECON["wenoz"] and p.info.riemann == L.RIEMANN["hlld"]
    assert p.info.glmmhd_alpha == 0.1 and p.info.dedner_extended == 0  # defaults hydro.cpp:285-295


@pytest.mark.parametrize("overrides,msg", [
    (["hydro/reconstruction=ppm"], "Need more ghost zones"),                 # hydro.cpp:444-447
    (["hydro/reconstruction=foo"], "Unknown reconstruction"),                # hydro.cpp:338
    (["hydro/riemann=roe"], "Unknown riemann"),                              # hydro.cpp:366
    (["hydro/riemann=llf"], "LLF Riemann solver only implemented with DC"),  # hydro.cpp:347-349
    (["hydro/fluid=mhd"], "Unknown fluid"),                                  # hydro.cpp:299
    (["hydro/riemann=hlld"], "no flux function"),                            # registry hydro.cpp:386-420
    (["parthenon/meshblock/nx1=48"], "multiple of the meshblock"),
    (["parthenon/mesh/refinement=octree"], "none, static or adaptive"),
    (["parthenon/mesh/refinement=adaptive", "parthenon/mesh/numlevel=2", "parthenon/mesh/nghost=3"], "even number of ghost"),
    (["job/problem_id=cluster"], "unknown job/problem_id"),
])
def test_deck_errors_are_reported_not_fatal(overrides, msg):
    from athenapk_amd import lib as L
    with pytest.raises(L.ApkError) as e:
        _plan("linear_wave3d", overrides)
    assert e.value.code == L.APK_ERR_INVALID and msg in str(e.value)


def test_rehearsal_of_an_eight_rank_brick_has_that_ranks_messages():
    """apk_amd/rehearse_remote_faces on ONE rank (bench.py's one-GPU rehearsal of the 2 x 2 x 2 run): the faces of the
    brick count as faces to other ranks, so the plan has the 7 peers -- 3 faces, 3 edges, 1 corner -- with exactly
    the message sizes rank 0 of the real 8-rank run has towards ranks 1 .. 7, send = receive."""
    mb = ["parthenon/meshblock/nx%d=16" % d for d in (1, 2, 3)]
    real = _plan("synthetic_mhd", ["parthenon/mesh/nx%d=64" % d for d in (1, 2, 3)] + mb, rank=0, nranks=8)
    one = _plan("synthetic_mhd", ["parthenon/mesh/nx%d=32" % d for d in (1, 2, 3)] + mb + ["apk_amd/rehearse_remote_faces=true"])
    plain = _plan("synthetic_mhd", ["parthenon/mesh/nx%d=32" % d for d in (1, 2, 3)] + mb)
    assert plain.peers() == [] and one.info.nblocks_local == real.info.nblocks_local == 8
    want = sorted((s, r) for _, s, r in real.peers())
    got = sorted((s, r) for _, s, r in one.peers())
    assert len(got) == 7 and got == want and all(s == r for s, r in got)
    # pseudo ranks are outside the real rank range and distinct
    assert sorted(q[0] for q in one.peers()) == list(range(1, 8))
    with pytest.raises(Exception):
        _plan("synthetic_mhd", ["parthenon/mesh/nx%d=64" % d for d in (1, 2, 3)] + mb + ["apk_amd/rehearse_remote_faces=true"], rank=0, nranks=8)


def test_plans_without_x1_faces_keep_the_message_layout():
    """PH_PACK_NOX1 .. PH_UNPACK_THIN_NOX1 (apk_sim_set_x1_direct: the x1 strips are stored into / read from the buffers by
    the stage kernels): the same regions at the same places in the same messages, minus exactly the x1 FACES -- one strip
    of nghost (or one) columns over the whole interior extent in x2 and x3 per block and remote x1 side."""
    mb = ["parthenon/meshblock/nx%d=16" % d for d in (1, 2, 3)]
    p = _plan("synthetic_mhd", ["parthenon/mesh/nx%d=64" % d for d in (1, 2, 3)] + mb, rank=3, nranks=8)
    ng = p.info.ng
    key = lambda r: (r.src_kind, r.src_block, r.dst_kind, r.dst_block, r.src_off, r.dst_off, tuple(r.ext), r.nvar, tuple(r.src_stride), tuple(r.dst_stride))
    for full, cut, depth in (("pack", "pack_nox1", ng), ("unpack", "unpack_nox1", ng), ("pack_thin", "pack_thin_nox1", 1),
                             ("unpack_thin", "unpack_thin_nox1", 1)):
        a, b = [key(r) for r in p.regions(full)], [key(r) for r in p.regions(cut)]
        gone = [r for r in a if r not in set(b)]
        assert set(b) <= set(a) and len(gone) == len(a) - len(b)
        # every block of the 2 x 2 x 2 brick has ONE x1 face on another rank
        assert len(gone) == 8 and all(r[6] == (depth, 16, 16) for r in gone)
        assert sorted({r[1] if "unpack" not in full else r[3] for r in gone}) == list(range(8))
        assert not any(r[6] == (depth, 16, 16) for r in b)


def test_morton_partition_gives_bricks():
    """4x4x4 meshblocks over 8 ranks: each rank owns a 2x2x2 brick."""
    ov = ["parthenon/mesh/nx1=64", "parthenon/mesh/nx2=64", "parthenon/mesh/nx3=64",
          "parthenon/meshblock/nx1=16", "parthenon/meshblock/nx2=16", "parthenon/meshblock/nx3=16"]
    seen = set()
    for r in range(8):
        p = _plan("synthetic_mhd", ov, rank=r, nranks=8)
        assert p.info.nblocks_local == 8
        locs = np.array([p.block_gid(lb)[1] for lb in range(8)])
        assert np.all(locs.max(axis=0) - locs.min(axis=0) == 1)  # a 2x2x2 brick
        seen |= {p.block_gid(lb)[0] for lb in range(8)}
        # a periodic 2x2x2 rank grid: every other rank is a neighbour (faces, edges, corners)
        assert sorted(q[0] for q in p.peers()) == [q for q in range(8) if q != r]
    assert seen == set(range(64))


# ---- ghost-exchange plans, executed on the host -------------------------------------------------------
def _run_plans(deck, overrides, nranks, fill, nvar, pack="pack", unpack="unpack", msgs="uniform"):
    """Emulates all ranks in one process: returns {gid: block array} after the exchange."""
    plans = [_plan(deck, overrides, rank=r, nranks=nranks) for r in range(nranks)]
    i0 = plans[0].info
    shape = (nvar, i0.mb[2] + 2 * i0.ng if i0.mb[2] > 1 else 1, i0.mb[1] + 2 * i0.ng if i0.mb[1] > 1 else 1,
             i0.mb[0] + 2 * i0.ng)
    blocks, sendb, recvb = [], [], []
    for r, p in enumerate(plans):
        blk = []
        for lb in range(p.info.nblocks_local):
            gid, loc = p.block_gid(lb)
            blk.append(fill(gid, loc, shape))
        blocks.append(blk)
        sendb.append({q: np.zeros(sc) for q, sc, rc in p.messages(msgs)})
        recvb.append({q: np.zeros(rc) for q, sc, rc in p.messages(msgs)})

    def base(r, kind, idx):
        p = plans[r]
        if kind == 0:
            return blocks[r][idx].reshape(-1)
        peer = p.peers()[idx][0]
        return (sendb if kind == 1 else recvb)[r][peer]

    def run(r, phase):
        for reg in plans[r].regions(phase):
            src, dst = base(r, reg.src_kind, reg.src_block), base(r, reg.dst_kind, reg.dst_block)
            ii, jj, kk, vv = np.meshgrid(np.arange(reg.ext[0]), np.arange(reg.ext[1]), np.arange(reg.ext[2]),
                                         np.arange(reg.nvar), indexing="ij")
            so = reg.src_off + ii * reg.src_stride[0] + jj * reg.src_stride[1] + kk * reg.src_stride[2] + vv * reg.src_stride[3]
            do = reg.dst_off + ii * reg.dst_stride[0] + jj * reg.dst_stride[1] + kk * reg.dst_stride[2] + vv * reg.dst_stride[3]
            val = src[so]
            if reg.flip_var >= 0:
                val = np.where(vv == reg.flip_var, -val, val)
            dst[do] = val

    for r in range(nranks):
        run(r, pack)
        run(r, "local")
    for r, p in enumerate(plans):  # the "wire": my send buffer to q is q's recv buffer from me
        for q, sc, rc in p.messages(msgs):
            assert sendb[r][q].size == recvb[q][r].size
            recvb[q][r][:] = sendb[r][q]
    for r in range(nranks):
        run(r, unpack)
        for ph in ("bc1", "bc2", "bc3"):
            run(r, ph)
    out = {}
    for r, p in enumerate(plans):
        for lb in range(p.info.nblocks_local):
            out[p.block_gid(lb)[0]] = blocks[r][lb]
    return out, plans[0].info


@pytest.mark.parametrize("nranks", [2, 3, 8])
def test_one_layer_plans_fill_the_first_ghost_layer(nranks):
    """The one-layer exchange (mesh.hpp PH_PACK_THIN / PH_UNPACK_THIN; the exchange in front of VL2's donor-cell
    predictor): the first layer of ghost cells all round every block equals the full exchange's, deeper layers filled by
    messages keep what they held, and a face message is a third of the full one (nghost = 3)."""
    ov = ["parthenon/mesh/nx1=24", "parthenon/mesh/nx2=16", "parthenon/mesh/nx3=16", "parthenon/meshblock/nx1=12",
          "parthenon/meshblock/nx2=8", "parthenon/meshblock/nx3=8"]

    def fill(gid, loc, shape):
        return np.random.default_rng(100 + gid).standard_normal(shape)

    full, info = _run_plans("synthetic_mhd", ov, nranks, fill, 9)
    thin, _ = _run_plans("synthetic_mhd", ov, nranks, fill, 9, pack="pack_thin", unpack="unpack_thin", msgs="uniform_thin")
    ng = info.ng
    for gid in full:
        a, b, c = full[gid], thin[gid], fill(gid, None, full[gid].shape)
        first = (slice(None),) + tuple(slice(ng - 1, n - ng + 1) for n in a.shape[1:])
        assert np.array_equal(a[first], b[first])
        assert np.all((b == a) | (b == c))  # deeper: the full exchange's value (same-rank copy) or untouched
    p = _plan("synthetic_mhd", ov, rank=0, nranks=nranks)
    sizes, sizes1 = p.messages("uniform"), p.messages("uniform_thin")
    assert [q[0] for q in sizes] == [q[0] for q in sizes1] and all(0 < t[1] < f[1] and 0 < t[2] < f[2] for f, t in zip(sizes, sizes1))
    assert p.peers() == sizes  # (introspection leaves the full set selected)


@pytest.mark.parametrize("nranks", [1, 2, 3, 8])
@pytest.mark.parametrize("case", ["periodic3d", "sod_outflow", "ot2d"])
def test_ghost_plans_reproduce_the_oracle_ghost_fill(oracle, case, nranks):
    if case == "periodic3d":
        deck, ov = "synthetic_mhd", ["parthenon/mesh/nx1=24", "parthenon/mesh/nx2=16", "parthenon/mesh/nx3=16",
                                     "parthenon/meshblock/nx1=12", "parthenon/meshblock/nx2=8",
                                     "parthenon/meshblock/nx3=8"]
        okw = dict(fluid="glmmhd", recon="ppm", riemann="hlld", integrator="vl2", nx=(24, 16, 16), mb=(12, 8, 8), ng=3)
        pg = "synthetic"
    elif case == "sod_outflow":
        deck, ov = "sod", ["parthenon/mesh/nx1=32", "parthenon/mesh/nx2=8", "parthenon/mesh/nx3=8",
                           "parthenon/meshblock/nx1=8", "parthenon/meshblock/nx2=4", "parthenon/meshblock/nx3=8"]
        okw = dict(fluid="euler", recon="plm", riemann="hllc", integrator="rk2", nx=(32, 8, 8), mb=(8, 4, 8), ng=2,
                   bc=("outflow", "periodic", "periodic"), xmin=(0.0, -0.5, -0.5), xmax=(1.0, 0.5, 0.5), gamma=1.4)
        pg = "sod"
    else:
        deck, ov = "orszag_tang", ["parthenon/mesh/nx1=32", "parthenon/mesh/nx2=32", "parthenon/meshblock/nx1=8",
                                   "parthenon/meshblock/nx2=16"]
        okw = dict(fluid="glmmhd", recon="ppm", riemann="hlld", integrator="vl2", nx=(32, 32, 1), mb=(8, 16, 1), ng=3,
                   xmin=(-0.5, -0.5, -0.5), xmax=(0.5, 0.5, 0.5))
        pg = "orszag_tang"
    o = oracle.Sim(**okw)
    o.pgen(pg)
    nvar = o.geom.nvar
    rng = np.random.default_rng(0)
    # distinct interior data per block, garbage in the ghosts
    for b in range(o.nblocks):
        o.cons(b)[...] = rng.uniform(1.0, 2.0, o.cons(b).shape)
    start = {b: o.cons(b).copy() for b in range(o.nblocks)}
    o.lib.orc_sim_exchange_ghosts(o.h)

    def fill(gid, loc, shape):
        assert shape == start[gid].shape
        return start[gid].copy()

    got, info = _run_plans(deck, ov, nranks, fill, nvar)
    assert len(got) == o.nblocks
    for gid, arr in got.items():
        assert np.array_equal(arr, o.cons(gid)), "block %d differs" % gid


def test_reflecting_boundary_plan(oracle):
    ov = ["parthenon/mesh/nx1=16", "parthenon/mesh/nx2=8", "parthenon/mesh/nx3=8", "parthenon/meshblock/nx1=8",
          "parthenon/meshblock/nx2=8", "parthenon/meshblock/nx3=8", "parthenon/mesh/ix1_bc=reflecting",
          "parthenon/mesh/ox1_bc=reflecting"]
    o = oracle.Sim(fluid="euler", recon="plm", riemann="hllc", integrator="rk2", nx=(16, 8, 8), mb=(8, 8, 8), ng=2,
                   bc=("reflecting", "periodic", "periodic"), xmin=(0.0, -0.5, -0.5), xmax=(1.0, 0.5, 0.5), gamma=1.4)
    o.pgen("sod")
    rng = np.random.default_rng(1)
    for b in range(o.nblocks):
        o.cons(b)[...] = rng.uniform(1.0, 2.0, o.cons(b).shape)
    start = {b: o.cons(b).copy() for b in range(o.nblocks)}
    o.lib.orc_sim_exchange_ghosts(o.h)
    got, _ = _run_plans("sod", ov, 2, lambda gid, loc, shape: start[gid].copy(), 5)
    for gid, arr in got.items():
        assert np.array_equal(arr, o.cons(gid))


# ---- few-modes turbulence driver: host spectral state (FewModesFT) ---------------------------------------
def _deck_k_vec(p):
    from athenapk_amd import decks
    kv, block = {}, None
    for line in decks.load("turbulence").splitlines():
        line = line.split("#")[0].strip()
        if line.startswith("<"):
            block = line.strip("<>")
        elif "=" in line and block == "modes":
            k, v = [x.strip() for x in line.split("=")]
            kv[k] = int(v)
    n = len(kv) // 3
    return np.array([[kv["k_%d_%d" % (m + 1, d)] for m in range(n)] for d in range(3)], dtype=np.float64)


def test_turbulence_host_state_matches_oracle_bitwise(oracle):
    """std::mt19937 + std::uniform_real_distribution in the product vs the oracle's restatement
    of both: identical spectral coefficients over 50 OU steps, identical phase tables."""
    p = _plan("turbulence")
    assert p.fmft_num_modes() == 30
    kv = _deck_k_vec(p)
    f = oracle.Fmft(oracle.load(), kv, k_peak=2.0, sol_weight=1.0, t_corr=1.0, rseed=20190729)
    assert np.all(p.fmft_var_hat() == 0.0)
    for n in range(50):
        dt = 0.002 + 1e-4 * n
        p.fmft_evolve(dt)
        f.evolve(dt)
        assert np.array_equal(p.fmft_var_hat(), f.var_hat())
    for ax, (n, g0) in enumerate(((64, 0), (32, 32), (32, 0))):
        # the product keeps the reference's variable shape (re|im, mode, cell)
        assert np.array_equal(p.fmft_phases(ax, n, g0), f.phases(ax, n, g0, 64).transpose(2, 1, 0))


@pytest.mark.parametrize("overrides,msg", [
    (["modes/k_3_0=40"], "k_vec x1 mode too large"),                              # few_modes_ft.cpp:49-53
    (["problem/turbulence/sol_weight=1.5"], "sol_weight for projection"),       # few_modes_ft.cpp:83-86
    (["parthenon/mesh/pack_size=4"], "pack_size=-1"),                            # few_modes_ft.cpp:113-116
    (["parthenon/mesh/x1max=2.0"], "cubic meshes"),                              # few_modes_ft.cpp:132-135
    (["problem/turbulence/b_config=3"], "Random B fields not implemented yet"),  # turbulence.cpp:264
])
def test_turbulence_deck_errors(overrides, msg):
    from athenapk_amd import lib as L
    with pytest.raises(L.ApkError) as e:
        _plan("turbulence", overrides)
    assert e.value.code == L.APK_ERR_INVALID and msg in str(e.value)


# ---- decks of the problems added for the reference's other regression suites ---------------------------------
@pytest.mark.parametrize("deck,fluid,recon,riemann,nx", [("blast", "euler", "plm", "hlle", [64, 64, 64]),
                                                         ("lw_implode", "euler", "plm", "hllc", [256, 256, 1]),
                                                         ("cpaw", "glmmhd", "plm", "hlld", [64, 32, 32]),
                                                         ("turbulence", "glmmhd", "plm", "hlle", [64, 64, 64]),
                                                         ("field_loop", "glmmhd", "plm", "hlle", [128, 64, 1]),
                                                         ("kh-shear-lecoanet_2d", "euler", "plm", "hlle", [128, 256, 1]),
                                                         ("blast_3d_amr", "euler", "plm", "hlle", [32, 32, 32]),
                                                         ("advection_3d", "euler", "plm", "hlle", [32, 32, 32])])
def test_problem_decks_parse(deck, fluid, recon, riemann, nx):
    from athenapk_amd import lib as L
    p = _plan(deck)
    assert (p.info.fluid, p.info.recon, p.info.riemann) == (L.FLUID[fluid], L.RECON[recon], L.RIEMANN[riemann])
    assert list(p.info.nx) == nx


@pytest.mark.parametrize("deck,overrides,msg", [
    ("cpaw", ["hydro/fluid=euler", "hydro/riemann=hllc"], "cpaw requires hydro/fluid = glmmhd"),
    ("cpaw", ["parthenon/mesh/nx3=1", "parthenon/meshblock/nx3=1"], "3-D"),
    ("lw_implode", ["hydro/fluid=glmmhd", "hydro/riemann=hlld"], "Only hydro runs are supported"),
    ("kh-shear-lecoanet_2d", ["problem/kh/iprob=1"], "Unknow iprob for KHI pgen."),
    ("kh-shear-lecoanet_2d", ["problem/kh/iprob=5"], "a"),
    ("field_loop", ["hydro/fluid=euler"], "field_loop requires hydro/fluid = glmmhd"),
    ("blast_3d_amr", ["parthenon/meshblock/nx1=2", "parthenon/meshblock/nx2=2", "parthenon/meshblock/nx3=2"],
     "at least 2 * nghost"),
    ("blast_3d_amr", ["parthenon/static_refinement0/level=1", "parthenon/static_refinement0/x1min=0.6",
                      "parthenon/static_refinement0/x1max=0.7", "parthenon/static_refinement0/x2min=0",
                      "parthenon/static_refinement0/x2max=0.1", "parthenon/static_refinement0/x3min=0",
                      "parthenon/static_refinement0/x3max=0.1"], "outside of the mesh"),
])
def test_problem_deck_errors(deck, overrides, msg):
    from athenapk_amd import lib as L
    with pytest.raises(L.ApkError) as e:
        _plan(deck, overrides)
    assert msg in str(e.value)
