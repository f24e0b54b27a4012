"""A refined-mesh restatement of the stage loop, written against the FOREST (level, position of every leaf) and
nothing else of the product: test infrastructure, the checker of the multilevel ghost exchange, the coarse-fine flux
correction and the time loop on refined meshes (tests/test_gpu_amr.py).

Unlike tests/amr_emulator.py -- which executes the index-box plans the driver built -- nothing here reads a plan:
every ghost region is classified by looking up who covers the slot next to the block (same level / finer / coarser),
in the order Parthenon's boundary communication + refinement tasks produce (un-vendored: SURVEY App. A; call sites
hydro_driver.cpp:506-577):
  1. every block of level >= 1 restricts its interior into its coarse buffer            (RestrictAverage)
  2. ghost zones from same-level neighbours' interiors and finer neighbours' restricted interiors; coarse-buffer
     ghost zones from the coarser neighbour's interior and same-level neighbours' coarse buffers
  3. prolongation of the coarse buffer into the ghost regions that face a coarser block   (ProlongateCellMinMod)
and the flux correction replaces a coarse block's face flux by the area average of the fine fluxes on that face
(hydro_driver.cpp:527-531) before the flux divergence.  The pointwise operators are the oracle's (oracle/amr.c,
oracle/block.c); periodic boxes only."""
import itertools

import numpy as np

import helpers as H


class RefinedMeshOracle:
    def __init__(self, oracle, fluid, recon, riemann, integrator, nrb, mb, ng, xmin, xmax, leaves, gamma, cfl, alpha=0.1,
                 tlim=1e300):
        self.o, self.fluid, self.recon, self.riemann, self.integrator = oracle, fluid, recon, riemann, integrator
        self.nrb, self.mb, self.ng = tuple(nrb), tuple(mb), ng
        self.act = [True, mb[1] > 1, mb[2] > 1]
        assert all(self.act), "3-D forests only"
        assert ng % 2 == 0
        self.cng = (ng + 1) // 2 + 1
        self.xmin, self.xmax = tuple(xmin), tuple(xmax)
        self.leaves = [(int(l), tuple(int(x) for x in lx)) for l, lx in leaves]
        self.index = {key: n for n, key in enumerate(self.leaves)}
        self.gamma, self.cfl, self.alpha, self.tlim = gamma, cfl, alpha, tlim
        self.nv = H.NHYDRO[fluid]
        self.fs = [ng] * 3
        self.fe = [ng + mb[d] - 1 for d in range(3)]
        self.cs = [self.cng] * 3
        self.ce = [self.cng + mb[d] // 2 - 1 for d in range(3)]
        self.shape = (self.nv,) + tuple(mb[d] + 2 * ng for d in (2, 1, 0))
        self.cshape = (self.nv,) + tuple(mb[d] // 2 + 2 * self.cng for d in (2, 1, 0))
        nb = len(self.leaves)
        self.cons = [np.zeros(self.shape) for _ in range(nb)]
        self.prim = [np.zeros(self.shape) for _ in range(nb)]
        self.coarse = [np.zeros(self.cshape) for _ in range(nb)]
        self.levels = sorted({l for l, _ in self.leaves})
        self.time, self.dt, self.ncycle = 0.0, 1.7976931348623157e308, 0
        self.dt_hyp = 1.7976931348623157e308
        self.c_h = 0.0
        self.nstages, self.beta, self.gam0, self.gam1 = oracle.integrator_coeffs(integrator)

    # ---- geometry ------------------------------------------------------------------------------------------------
    def dx(self, level):
        return tuple((self.xmax[d] - self.xmin[d]) / (self.nrb[d] * self.mb[d] * 2 ** level) for d in range(3))

    def corner(self, n):
        level, lx = self.leaves[n]
        dx = self.dx(level)
        return tuple(self.xmin[d] + lx[d] * self.mb[d] * dx[d] for d in range(3))

    def wrap(self, level, pos):
        return tuple(pos[d] % (self.nrb[d] << level) for d in range(3))

    def classify(self, level, pos):
        """who covers the block-sized slot `pos` of `level`: ('same', n) / ('finer', None) / ('coarser', n)"""
        w = self.wrap(level, pos)
        if (level, w) in self.index:
            return "same", self.index[(level, w)]
        if level > 0:
            p = tuple(x >> 1 for x in w)
            if (level - 1, p) in self.index:
                return "coarser", self.index[(level - 1, p)]
        return "finer", None

    @staticmethod
    def box(lo, ext):
        """numpy index of the (i, j, k) box lo .. lo + ext - 1 in a [nvar][k][j][i] array"""
        return (slice(None),) + tuple(slice(lo[d], lo[d] + ext[d]) for d in (2, 1, 0))

    def geom_of(self, level):
        return H.geom(self.fluid, self.mb, self.ng, 0, self.dx(level))

    def rgeom(self, n):
        level, _ = self.leaves[n]
        return self.o.make_refine_geom(self.mb, self.ng, self.cng, self.corner(n), self.dx(level))

    # ---- the multilevel ghost exchange ---------------------------------------------------------------------------
    def exchange(self, field=None):
        u = self.cons if field is None else field
        ng, cng, mb = self.ng, self.cng, self.mb
        fs, fe, cs, ce = self.fs, self.fe, self.cs, self.ce
        # 1. own interior -> coarse buffer
        for n, (level, lx) in enumerate(self.leaves):
            if level >= 1:
                self.coarse[n][...] = 0.0
                self.o.restrict(self.rgeom(n), 0, u[n], self.coarse[n], tuple(cs), tuple(ce))
        # 2. copies, region by region
        needs_prolongation = []
        for n, (level, lx) in enumerate(self.leaves):
            for o in itertools.product((-1, 0, 1), repeat=3):
                if o == (0, 0, 0):
                    continue
                pos = tuple(lx[d] + o[d] for d in range(3))
                kind, nb = self.classify(level, pos)
                if kind == "same":
                    dlo = [fs[d] if o[d] == 0 else (fs[d] - ng if o[d] < 0 else fe[d] + 1) for d in range(3)]
                    slo = [fs[d] if o[d] == 0 else (fe[d] - ng + 1 if o[d] < 0 else fs[d]) for d in range(3)]
                    ext = [mb[d] if o[d] == 0 else ng for d in range(3)]
                    u[n][self.box(dlo, ext)] = u[nb][self.box(slo, ext)]
                    if level >= 1:
                        dlo = [cs[d] if o[d] == 0 else (cs[d] - cng if o[d] < 0 else ce[d] + 1) for d in range(3)]
                        slo = [cs[d] if o[d] == 0 else (ce[d] - cng + 1 if o[d] < 0 else cs[d]) for d in range(3)]
                        ext = [mb[d] // 2 if o[d] == 0 else cng for d in range(3)]
                        self.coarse[n][self.box(dlo, ext)] = self.coarse[nb][self.box(slo, ext)]
                elif kind == "finer":
                    w = self.wrap(level, pos)
                    for c in itertools.product((0, 1), repeat=3):
                        if any(o[d] != 0 and c[d] != (1 if o[d] < 0 else 0) for d in range(3)):
                            continue
                        child = self.index[(level + 1, tuple(2 * w[d] + c[d] for d in range(3)))]
                        dlo = [fs[d] + c[d] * (mb[d] // 2) if o[d] == 0 else (fs[d] - ng if o[d] < 0 else fe[d] + 1) for d in range(3)]
                        slo = [cs[d] if o[d] == 0 else (ce[d] - ng + 1 if o[d] < 0 else cs[d]) for d in range(3)]
                        ext = [mb[d] // 2 if o[d] == 0 else ng for d in range(3)]
                        u[n][self.box(dlo, ext)] = self.coarse[child][self.box(slo, ext)]
                else:  # coarser: my coarse-buffer ghost region <- its interior
                    clo = [cs[d] if o[d] == 0 else (cs[d] - cng if o[d] < 0 else ce[d] + 1) for d in range(3)]
                    ext = [mb[d] // 2 if o[d] == 0 else cng for d in range(3)]
                    # global (level - 1) cell coordinate of my coarse cell c along d: lx * mb / 2 + (c - cs); the coarser
                    # block covering the slot starts (unwrapped) at floor((lx + o) / 2) * mb
                    slo = [fs[d] + lx[d] * (mb[d] // 2) + (clo[d] - cs[d]) - ((lx[d] + o[d]) >> 1) * mb[d] for d in range(3)]
                    self.coarse[n][self.box(clo, ext)] = u[nb][self.box(slo, ext)]
                    plo = [cs[d] if o[d] == 0 else (cs[d] - ng // 2 if o[d] < 0 else ce[d] + 1) for d in range(3)]
                    pext = [mb[d] // 2 if o[d] == 0 else ng // 2 for d in range(3)]
                    needs_prolongation.append((n, plo, [plo[d] + pext[d] - 1 for d in range(3)]))
        # 3. prolongation into the ghost regions facing coarser blocks
        for n, lo, hi in needs_prolongation:
            self.o.prolongate(self.rgeom(n), self.coarse[n], u[n], tuple(lo), tuple(hi))

    # ---- regridding: the state on a new forest -------------------------------------------------------------------
    def regrid(self, new_leaves):
        """Mesh::LoadBalancingAndAdaptiveMeshRefinement's data movement, given the new forest: surviving blocks keep
        their data; a new fine block is the minmod prolongation of its parent's octant (through a coarse buffer cut
        out of the parent: the octant plus cng cells of the parent's -- valid -- ghost zones and interior all round);
        a merged block collects the restricted interiors of its children.  Then exchange + FillDerived on the new
        mesh.  Returns the oracle of the new forest (time, dt and c_h carried over)."""
        new = RefinedMeshOracle(self.o, self.fluid, self.recon, self.riemann, self.integrator, self.nrb, self.mb, self.ng,
                                self.xmin, self.xmax, new_leaves, self.gamma, self.cfl, self.alpha, self.tlim)
        mb, cng, fs, cs, ce = self.mb, self.cng, self.fs, self.cs, self.ce
        for n, (level, lx) in enumerate(new.leaves):
            if (level, lx) in self.index:
                new.cons[n] = np.array(self.cons[self.index[(level, lx)]], copy=True)
                continue
            parent = (level - 1, tuple(x >> 1 for x in lx)) if level > 0 else None
            if parent in self.index:
                p = self.index[parent]
                c = [lx[d] & 1 for d in range(3)]
                clo = [cs[d] - cng for d in range(3)]
                ext = [mb[d] // 2 + 2 * cng for d in range(3)]
                slo = [fs[d] + c[d] * (mb[d] // 2) - cng for d in range(3)]
                new.coarse[n][self.box(clo, ext)] = self.cons[p][self.box(slo, ext)]
                self.o.prolongate(new.rgeom(n), new.coarse[n], new.cons[n], tuple(cs), tuple(ce))
            else:
                for c in itertools.product((0, 1), repeat=3):
                    child = self.index[(level + 1, tuple(2 * lx[d] + c[d] for d in range(3)))]
                    tmp = np.zeros(self.cshape)
                    self.o.restrict(self.rgeom(child), 0, self.cons[child], tmp, tuple(cs), tuple(ce))
                    dlo = [fs[d] + c[d] * (mb[d] // 2) for d in range(3)]
                    ext = [mb[d] // 2 for d in range(3)]
                    new.cons[n][self.box(dlo, ext)] = tmp[self.box(cs, ext)]
        new.exchange()
        new.fill_derived()
        new.time, new.dt, new.ncycle, new.dt_hyp, new.c_h = self.time, self.dt, self.ncycle, self.dt_hyp, self.c_h
        return new

    # ---- coarse-fine flux correction ----------------------------------------------------------------------------
    def restricted_face_flux(self, d, fine):
        """area average of the 2 x 2 fine faces of direction d behind every coarse face, in the pairwise order of the
        operator (j pairs, then i, then k; the face direction has one member): [nvar][mb/2 (+1 along d)] per axis"""
        fs, mb = self.fs, self.mb
        n = [mb[q] // 2 for q in range(3)]
        out = {}
        for side, plane in ((0, fs[d]), (1, self.fe[d] + 1)):
            def f(ok, oj, oi):
                idx = [slice(None)] * 4
                for q, off in ((2, ok), (1, oj), (0, oi)):
                    if q == d:
                        idx[3 - q] = slice(plane, plane + 1)
                    else:
                        idx[3 - q] = slice(fs[q] + off, fs[q] + mb[q], 2)
                return fine[tuple(idx)]
            ar = 1.0
            dxl = self._dx_for_flux
            for q in range(3):
                if q != d:
                    ar *= dxl[q]
            z = 0.0
            t = {}
            for ok in range(2):
                for oj in range(2):
                    for oi in range(2):
                        inside = not ((d == 2 and ok) or (d == 1 and oj) or (d == 0 and oi))
                        t[ok, oj, oi] = ar * f(ok, oj, oi) if inside else z
                        t["v", ok, oj, oi] = ar if inside else z
            tot = ((t[0, 0, 0] + t[0, 1, 0]) + (t[0, 0, 1] + t[0, 1, 1])) + ((t[1, 0, 0] + t[1, 1, 0]) + (t[1, 0, 1] + t[1, 1, 1]))
            vol = ((t["v", 0, 0, 0] + t["v", 0, 1, 0]) + (t["v", 0, 0, 1] + t["v", 0, 1, 1])) + \
                  ((t["v", 1, 0, 0] + t["v", 1, 1, 0]) + (t["v", 1, 0, 1] + t["v", 1, 1, 1]))
            out[side] = tot / vol
        return out

    def flux_correction(self, flux):
        """flux[d][n]: [nvar][k][j][i] face fluxes of block n (lower d-face of each cell)"""
        fs, fe, mb = self.fs, self.fe, self.mb
        for n, (level, lx) in enumerate(self.leaves):
            for d in range(3):
                for side in (0, 1):
                    o = [0, 0, 0]
                    o[d] = 1 if side else -1
                    pos = tuple(lx[q] + o[q] for q in range(3))
                    kind, _ = self.classify(level, pos)
                    if kind != "finer":
                        continue
                    w = self.wrap(level, pos)
                    for c in itertools.product((0, 1), repeat=3):
                        if c[d] != (0 if side else 1):
                            continue
                        child = self.index[(level + 1, tuple(2 * w[q] + c[q] for q in range(3)))]
                        self._dx_for_flux = self.dx(level + 1)
                        avg = self.restricted_face_flux(d, flux[d][child])[0 if side else 1]  # the child's face towards me
                        dlo = [fs[q] + c[q] * (mb[q] // 2) for q in range(3)]
                        dlo[d] = fe[d] + 1 if side else fs[d]
                        ext = [mb[q] // 2 for q in range(3)]
                        ext[d] = 1
                        flux[d][n][self.box(dlo, ext)] = avg

    # ---- the time loop (hydro_driver.cpp:474-603; flux-array task order) ----------------------------------------------
    def by_level(self, arrs, level):
        idx = [n for n, (l, _) in enumerate(self.leaves) if l == level]
        return idx, np.stack([arrs[n] for n in idx])

    def fill_derived(self):
        eos = self.o.make_eos(self.gamma)
        for level in self.levels:
            idx, c = self.by_level(self.cons, level)
            c2, w, bad = H.orc_c2p(self.fluid, self.geom_of(level), c, eos)
            assert bad == 0
            for m, n in enumerate(idx):
                self.cons[n], self.prim[n] = c2[m], w[m]

    def estimate_dt(self):
        m = 1.7976931348623157e308
        for level in self.levels:
            idx, w = self.by_level(self.prim, level)
            m = min(m, H.orc_min_dt(self.fluid, self.geom_of(level), w, self.gamma))
        return self.cfl * m

    def set_global_dt(self, est):
        dt = self.dt
        if dt < 0.1 * 1.7976931348623157e308:
            dt *= 2.0
        dt = min(dt, est)
        if self.time < self.tlim and (self.tlim - self.time) < dt:
            dt = self.tlim - self.time
        self.dt = dt

    def initialize(self, cons):
        for n, c in enumerate(cons):
            self.cons[n] = np.array(c, copy=True)
        self.exchange()
        self.fill_derived()
        est = self.estimate_dt()
        if self.fluid == "glmmhd":
            self.dt_hyp = min(self.dt_hyp, est)
        self.set_global_dt(est)
        return self

    def step(self):
        if self.time < self.tlim and (self.tlim - self.time) < self.dt:
            self.dt = self.tlim - self.time
        mhd = self.fluid == "glmmhd"
        mindx = min(self.dx(self.levels[-1]))
        if mhd:
            self.c_h = self.cfl * mindx / self.dt_hyp
        u1 = None
        for stage in range(1, self.nstages + 1):
            g0, g1, bdt = self.gam0[stage - 1], self.gam1[stage - 1], self.beta[stage - 1] * self.dt
            if stage == 1:
                u1 = [np.array(c, copy=True) for c in self.cons]
            recon = "dc" if (self.integrator == "vl2" and stage == 1) else self.recon
            flux = [[None] * len(self.leaves) for _ in range(3)]
            for level in self.levels:
                idx, w = self.by_level(self.prim, level)
                fl = H.orc_fluxes(self.fluid, recon, self.riemann, self.geom_of(level), w, self.gamma, self.c_h)
                for m, n in enumerate(idx):
                    for d in range(3):
                        flux[d][n] = fl[d][m]
            self.flux_correction(flux)
            for level in self.levels:
                g = self.geom_of(level)
                idx, c0 = self.by_level(self.cons, level)
                _, c1 = self.by_level(u1, level)
                fl = [np.stack([flux[d][n] for n in idx]) for d in range(3)]
                new = H.orc_update(g, c0, c1, fl, g0, g1, bdt)
                if mhd:
                    _, w = self.by_level(self.prim, level)
                    new = H.orc_dedner(g, new, w, 0, self.alpha, self.c_h, mindx, bdt)
                for m, n in enumerate(idx):
                    self.cons[n] = new[m]
            self.exchange()
            self.fill_derived()
        if mhd:
            self.dt_hyp = 1.7976931348623157e308
        self.time += self.dt
        self.ncycle += 1
        est = self.estimate_dt()
        if mhd:
            self.dt_hyp = min(self.dt_hyp, est)
        self.set_global_dt(est)
