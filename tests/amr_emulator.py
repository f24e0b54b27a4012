"""Host emulation of the multilevel ghost exchange and flux correction: executes the index-box plans
the driver builds (athenapk_amd/csrc/host/amr.hpp) with numpy copies and the ORACLE's refinement
operators (oracle/amr.c).  Test infrastructure: the checker for the plans on the CPU and for the
device exchange on the GPU."""
import numpy as np


def block_shape(info):
    ng = info.ng
    return (info.nhydro + info.nscalars, info.mb[2] + 2 * ng if info.mb[2] > 1 else 1,
            info.mb[1] + 2 * ng if info.mb[1] > 1 else 1, info.mb[0] + 2 * ng)


def placement(view):
    """per block: (level, lx, lower corner, cell widths)"""
    i = view.refresh_info()
    out = []
    for lb in range(i.nblocks_local):
        lev = view.block_level(lb)
        _, loc = view.block_gid(lb)
        dx = [i.dx[d] / 2 ** lev if i.mb[d] > 1 else i.dx[d] for d in range(3)]
        x0 = [i.xmin[d] + loc[d] * i.mb[d] * dx[d] for d in range(3)]
        out.append((lev, loc, x0, dx))
    return out


class Emulator:
    def __init__(self, view, oracle):
        self.v, self.o = view, oracle
        i = view.refresh_info()
        self.info = i
        self.nb = i.nblocks_local
        self.shape = block_shape(i)
        self.nvar = self.shape[0]
        ops = view.amr_ops("restrict_own") + view.amr_ops("prolongate")
        self.cng = ops[0].cng if ops else (i.ng + 1) // 2 + 1
        self.coarse_doubles = ops[0].coarse_doubles if ops else 1
        act = [True, i.mb[1] > 1, i.mb[2] > 1]
        self.cshape = (self.nvar,) + tuple((i.mb[d] // 2 + 2 * self.cng) if act[d] else 1 for d in (2, 1, 0))
        self.cons = [np.zeros(self.shape) for _ in range(self.nb)]
        self.coarse = [np.zeros(self.coarse_doubles) for _ in range(self.nb)]
        self.flux = [[np.zeros(self.shape) for _ in range(self.nb)] for _ in range(3)]

    def _base(self, kind, idx):
        if kind == 0:
            return self.cons[idx].reshape(-1)
        if kind == 1:
            return self.send[idx]
        if kind == 2:
            return self.recv[idx]
        if kind == 3:
            return self.coarse[idx]
        return self.flux[kind - 4][idx].reshape(-1)

    def use_messages(self, which):
        """allocate the per-peer buffers of the "halo" / "flux" message set; returns the peer list"""
        self.peers = self.v.messages(which)
        self.send = [np.full(sc, np.nan) for _, sc, _ in self.peers]
        self.recv = [np.full(rc, np.nan) for _, _, rc in self.peers]
        return self.peers

    def _copy(self, phase):
        for reg in self.v.regions(phase):
            src, dst = self._base(reg.src_kind, reg.src_block), self._base(reg.dst_kind, reg.dst_block)
            ii, jj, kk, vv = np.meshgrid(np.arange(reg.ext[0]), np.arange(reg.ext[1]), np.arange(reg.ext[2]),
                                         np.arange(reg.nvar), indexing="ij")
            so = reg.src_off + ii * reg.src_stride[0] + jj * reg.src_stride[1] + kk * reg.src_stride[2] + vv * reg.src_stride[3]
            do = reg.dst_off + ii * reg.dst_stride[0] + jj * reg.dst_stride[1] + kk * reg.dst_stride[2] + vv * reg.dst_stride[3]
            val = src[so]
            if reg.flip_var >= 0:
                val = np.where(vv == reg.flip_var, -val, val)
            dst[do] = val

    def _coarse_view(self, lb):
        n = int(np.prod(self.cshape))
        return self.coarse[lb][:n].reshape(self.cshape)

    def _ops(self, which):
        i = self.info
        for op in self.v.amr_ops(which):
            g = self.o.make_refine_geom(tuple(i.mb), i.ng, self.cng, tuple(op.xmin), tuple(op.dx))
            lo, hi = tuple(op.lo), tuple(op.hi)
            if op.kind == 0:      # prolongate: coarse buffer -> block
                self.o.prolongate(g, self._coarse_view(op.src_block), self.cons[op.dst_block], lo, hi)
            elif op.kind == 1:    # restrict cells: block -> coarse buffer
                self.o.restrict(g, 0, self.cons[op.src_block], self._coarse_view(op.dst_block), lo, hi)
            else:                 # restrict the cell-shaped flux array of direction kind - 5
                d = op.kind - 5
                self._restrict_flux(d, self.flux[d][op.src_block], self._coarse_view(op.dst_block), lo, hi, tuple(op.dx))

    def _restrict_flux(self, d, fine, coarse, lo, hi, dx):
        """area-weighted average of the fine face fluxes (numpy restatement of RestrictAverage on
        a face: weights = face areas, pairwise sums)"""
        w = 1.0
        for q in range(3):
            if q != d:
                w *= dx[q]
        i = self.info
        act = [True, i.mb[1] > 1, i.mb[2] > 1]
        fs = [i.ng if act[q] else 0 for q in range(3)]
        cs = [self.cng if act[q] else 0 for q in range(3)]
        for ck in range(lo[2], hi[2] + 1):
            for cj in range(lo[1], hi[1] + 1):
                for ci in range(lo[0], hi[0] + 1):
                    c = (ci, cj, ck)
                    f = [(c[q] - cs[q]) * 2 + fs[q] if act[q] else 0 for q in range(3)]
                    rng = [range(1) if (q == d or not act[q]) else range(2) for q in range(3)]
                    # pairwise sums in the order of the device operator: (j pairs) then i, then k
                    t = {}
                    for ok in range(2):
                        for oj in range(2):
                            for oi in range(2):
                                inside = (ok in rng[2]) and (oj in rng[1]) and (oi in rng[0])
                                t[ok, oj, oi] = w * fine[:, f[2] + ok, f[1] + oj, f[0] + oi] if inside else 0.0
                                t["v", ok, oj, oi] = w if inside else 0.0
                    tot = ((t[0, 0, 0] + t[0, 1, 0]) + (t[0, 0, 1] + t[0, 1, 1])) + ((t[1, 0, 0] + t[1, 1, 0]) + (t[1, 0, 1] + t[1, 1, 1]))
                    vol = ((t["v", 0, 0, 0] + t["v", 0, 1, 0]) + (t["v", 0, 0, 1] + t["v", 0, 1, 1])) + \
                          ((t["v", 1, 0, 0] + t["v", 1, 1, 0]) + (t["v", 1, 0, 1] + t["v", 1, 1, 1]))
                    coarse[:, ck, cj, ci] = tot / vol

    def exchange(self):
        self._ops("restrict_own")
        self._copy("amr_fill")
        for d in (1, 2, 3):
            self._copy("amr_coarse_bc%d" % d)
        self._ops("prolongate")
        for d in (1, 2, 3):
            self._copy("amr_bc%d" % d)

    def flux_correction(self):
        for d in range(self.info.ndim):
            self._ops("flux_restrict%d" % (d + 1))
            self._copy("amr_flux%d" % (d + 1))


def _wire(ems):
    """deliver every send buffer into the matching receive buffer (the comm callback's job)"""
    for r, e in enumerate(ems):
        for p, (peer, sc, rc) in enumerate(e.peers):
            q = [n for n, (pr, _, _) in enumerate(ems[peer].peers) if pr == r]
            assert len(q) == 1 and ems[peer].peers[q[0]][2] == sc, "message sizes disagree between ranks %d and %d" % (r, peer)
            ems[peer].recv[q[0]][:] = e.send[p]


def exchange_on_ranks(ems):
    """the multilevel exchange with the blocks distributed over len(ems) ranks (one Emulator per rank
    over that rank's HostPlan): every rank executes its share of the plans, messages are wired"""
    for e in ems:
        e.use_messages("halo")
        e._ops("my_restrict_own")
        e._copy("my_fill_pack")
        e._copy("my_fill")
    _wire(ems)
    for e in ems:
        e._copy("my_fill_unpack")
        for d in (1, 2, 3):
            e._copy("my_coarse_bc%d" % d)
        e._ops("my_prolongate")
        for d in (1, 2, 3):
            e._copy("my_bc%d" % d)


def stage_loop_exchange_on_ranks(ems, mode):
    """the exchanges of the stage loop (sim_amr.cpp amr_exchange): mode "faces" = without the boxes behind edges and
    corners, "direct" = nor the copies between same-rank blocks of one level, "shell" = every ghost zone two layers deep"""
    tag = "shell" if mode == "shell" else "faces"
    for e in ems:
        e.use_messages("halo_" + tag)
        e._ops("my_restrict_own")
        e._copy("my_fill_pack_" + tag)
        e._copy({"faces": "my_fill_faces", "direct": "my_fill_direct", "shell": "my_fill_shell"}[mode])
    _wire(ems)
    for e in ems:
        e._copy("my_fill_unpack_" + tag)
        for d in (1, 2, 3):
            e._copy("my_coarse_bc%d" % d)
        e._ops("my_prolongate_" + tag)
        for d in (1, 2, 3):
            e._copy(("my_bc_shell%d" if mode == "shell" else "my_bc%d") % d)


def flux_correction_on_ranks(ems):
    for e in ems:
        e.use_messages("flux")
        for d in range(e.info.ndim):
            e._ops("my_flux_restrict%d" % (d + 1))
            e._copy("my_flux%d" % (d + 1))
            e._copy("my_flux_pack%d" % (d + 1))
    _wire(ems)
    for e in ems:
        for d in range(e.info.ndim):
            e._copy("my_flux_unpack%d" % (d + 1))
