"""Generates tests/golden/edge_cases.npz + edge_cases.json: the crafted special cases of SURVEY 8(c)(2)
-- identical L/R states, Bx = 0, B = 0, supersonic either way, every HLLD region and degenerate
branch (src/hydro/rsolvers/glmmhd_hlld.hpp:196-213, 228-243, 252-294, 330-387), HLLC with a clipped
contact pressure / shocks either side (hydro_hllc.hpp:63-70, 112, 126), PPM's extremum limiters, its
round-off guard and its overshoot limiters (src/recon/ppm_simple.hpp:66-98, 129-157), every floor
and ceiling of ConsToPrim (src/eos/adiabatic_glmmhd.hpp:78-160) -- with the oracle's outputs and, per
case, the set of reference branches it takes (recorded by the oracle's branch tracing,
oracle/apk_oracle.h ORC_TR_*).  The generator refuses to write the fixture unless every traced
branch is hit by at least one case.

The cases are stored as DATA (inputs + expected outputs + branch masks); tests/test_edge_cases.py
checks the oracle against them on the CPU and both HIP builds against them on the GPU, through the
real kernels: Riemann cases are laid out as neighbouring cells of donor-cell blocks (a donor-cell
face flux is the Riemann solver applied to the two adjacent cells), PPM stencils as pencils of a
block, ConsToPrim cases as cells of a block.

    python tests/golden/make_edge_cases.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import oracle as O  # noqa: E402

GAMMA, C_H = 5.0 / 3.0, 1.3


def mhd(d, v1, v2, v3, p, b1, b2, b3, psi=0.0):
    return [d, v1, v2, v3, p, b1, b2, b3, psi]


def riemann_cases():
    """[(label, fluid, riemann, wl, wr)]: states are in NATURAL variable order for a sweep along x1"""
    c = []
    q = mhd(1.0, 0.3, -0.2, 0.1, 0.8, 0.6, 0.4, -0.3, 0.05)
    c.append(("mhd identical states, moving", "glmmhd", "hlld", q, q))
    q0 = mhd(1.0, 0.0, 0.0, 0.0, 1.0, 0.5, 0.3, 0.2)
    c.append(("mhd identical states, at rest", "glmmhd", "hlld", q0, q0))
    c.append(("mhd B = 0 everywhere, subsonic", "glmmhd", "hlld", mhd(1.0, 0.2, 0.1, 0.0, 1.0, 0, 0, 0), mhd(0.8, -0.1, 0.0, 0.1, 0.7, 0, 0, 0)))
    c.append(("mhd B = 0, flow to the left", "glmmhd", "hlld", mhd(1.0, -0.4, 0.1, 0.0, 1.0, 0, 0, 0), mhd(0.8, -0.5, 0.0, 0.1, 0.7, 0, 0, 0)))
    c.append(("mhd Bx = 0, transverse field", "glmmhd", "hlld", mhd(1.0, 0.1, 0.2, 0.0, 1.0, 0.0, 0.7, 0.2), mhd(0.9, 0.05, -0.1, 0.1, 0.8, 0.0, 0.5, -0.3)))
    c.append(("mhd supersonic to the right", "glmmhd", "hlld", mhd(1.0, 5.0, 0.1, 0.0, 1.0, 0.5, 0.3, 0.1), mhd(0.9, 4.5, 0.0, 0.1, 0.8, 0.5, 0.2, 0.0)))
    c.append(("mhd supersonic to the left", "glmmhd", "hlld", mhd(1.0, -5.0, 0.1, 0.0, 1.0, 0.5, 0.3, 0.1), mhd(0.9, -4.5, 0.0, 0.1, 0.8, 0.5, 0.2, 0.0)))
    # weak field: the contact lies outside the Alfven fan -> single-star regions
    c.append(("mhd left star region (weak Bx, flow to the right)", "glmmhd", "hlld", mhd(1.0, 0.4, 0.1, 0.0, 1.0, 0.05, 0.3, 0.1), mhd(0.9, 0.35, 0.0, 0.1, 0.9, 0.05, 0.2, 0.0)))
    c.append(("mhd right star region (weak Bx, flow to the left)", "glmmhd", "hlld", mhd(1.0, -0.35, 0.1, 0.0, 1.0, 0.05, 0.3, 0.1), mhd(0.9, -0.4, 0.0, 0.1, 0.9, 0.05, 0.2, 0.0)))
    # strong field: contact between the Alfven waves -> double-star regions
    c.append(("mhd left double-star region", "glmmhd", "hlld", mhd(1.0, 0.1, 0.3, 0.0, 1.0, 1.5, 0.5, 0.2), mhd(0.9, 0.08, -0.2, 0.1, 0.9, 1.5, -0.4, 0.1)))
    c.append(("mhd right double-star region", "glmmhd", "hlld", mhd(1.0, -0.08, 0.3, 0.0, 1.0, 1.5, 0.5, 0.2), mhd(0.9, -0.1, -0.2, 0.1, 0.9, 1.5, -0.4, 0.1)))
    # By = Bz = 0 with Bx^2 > gamma p: fast speed = Alfven speed, rho (s-v)(s-s*) = Bx^2 -> degenerate star states
    c.append(("mhd degenerate star states (purely normal strong field)", "glmmhd", "hlld", mhd(1.0, 0.0, 0.1, 0.0, 0.1, 2.0, 0.0, 0.0), mhd(1.0, 0.0, -0.1, 0.05, 0.1, 2.0, 0.0, 0.0)))
    c.append(("mhd degenerate on the left only", "glmmhd", "hlld", mhd(1.0, 0.0, 0.1, 0.0, 0.1, 2.0, 0.0, 0.0), mhd(1.0, 0.0, -0.1, 0.05, 0.1, 2.0, 0.3, 0.0)))
    c.append(("mhd strong shock", "glmmhd", "hlld", mhd(1.0, 3.0, 0.0, 0.0, 10.0, 0.7, 1.0, 0.0), mhd(0.125, -3.0, 0.0, 0.0, 0.1, 0.7, -1.0, 0.0)))
    c.append(("mhd GLM jump: Bx and psi discontinuous", "glmmhd", "hlld", mhd(1.0, 0.1, 0.0, 0.0, 1.0, 0.4, 0.3, 0.1, 0.2), mhd(1.0, 0.1, 0.0, 0.0, 1.0, 0.8, 0.3, 0.1, -0.3)))
    c.append(("mhd hlle identical states", "glmmhd", "hlle", q, q))
    c.append(("mhd hlle B = 0", "glmmhd", "hlle", mhd(1.0, 0.2, 0.1, 0.0, 1.0, 0, 0, 0), mhd(0.8, -0.1, 0.0, 0.1, 0.7, 0, 0, 0)))
    c.append(("mhd hlle supersonic to the right", "glmmhd", "hlle", mhd(1.0, 5.0, 0.1, 0.0, 1.0, 0.5, 0.3, 0.1), mhd(0.9, 4.5, 0.0, 0.1, 0.8, 0.5, 0.2, 0.0)))
    c.append(("mhd llf strong shock", "glmmhd", "llf", mhd(1.0, 3.0, 0.0, 0.0, 10.0, 0.7, 1.0, 0.0), mhd(0.125, -3.0, 0.0, 0.0, 0.1, 0.7, -1.0, 0.0)))
    h = [1.0, 0.3, -0.2, 0.1, 0.8]
    c.append(("hydro hllc identical states", "euler", "hllc", h, h))
    c.append(("hydro hllc contact moving right, no shocks", "euler", "hllc", [1.0, 0.5, 0.1, 0.0, 1.0], [0.5, 0.5, 0.0, 0.2, 1.0]))
    c.append(("hydro hllc contact moving left", "euler", "hllc", [1.0, -0.5, 0.1, 0.0, 1.0], [0.5, -0.5, 0.0, 0.2, 1.0]))
    c.append(("hydro hllc colliding flows: shocks both sides", "euler", "hllc", [1.0, 2.0, 0.0, 0.0, 1.0], [1.0, -2.0, 0.0, 0.0, 1.0]))
    c.append(("hydro hllc strong rarefaction: pmid < 0, contact pressure clipped", "euler", "hllc", [1.0, -10.0, 0.0, 0.0, 0.01], [1.0, 10.0, 0.0, 0.0, 0.01]))
    c.append(("hydro hllc supersonic to the right", "euler", "hllc", [1.0, 5.0, 0.0, 0.0, 1.0], [0.9, 5.0, 0.0, 0.0, 0.9]))
    c.append(("hydro hllc supersonic to the left", "euler", "hllc", [1.0, -5.0, 0.0, 0.0, 1.0], [0.9, -5.0, 0.0, 0.0, 0.9]))
    c.append(("hydro hllc Sod", "euler", "hllc", [1.0, 0.0, 0.0, 0.0, 1.0], [0.125, 0.0, 0.0, 0.0, 0.1]))
    c.append(("hydro hlle identical states", "euler", "hlle", h, h))
    c.append(("hydro hlle Sod", "euler", "hlle", [1.0, 0.0, 0.0, 0.0, 1.0], [0.125, 0.0, 0.0, 0.0, 0.1]))
    c.append(("hydro hlle negative Roe enthalpy argument guarded (q < 0 -> a = 0)", "euler", "hlle", [1.0, 30.0, 0.0, 0.0, 1e-6], [1.0, -30.0, 0.0, 0.0, 1e-6]))
    c.append(("hydro llf Sod", "euler", "llf", [1.0, 0.0, 0.0, 0.0, 1.0], [0.125, 0.0, 0.0, 0.0, 0.1]))
    return c


def ppm_stencils():
    """[(label, five cell values)]"""
    s = [("linear data", [1.0, 2.0, 3.0, 4.0, 5.0]),
         ("constant data: every second difference is exactly zero (round-off guard)", [1.0, 1.0, 1.0, 1.0, 1.0]),
         ("parabola minimum", [4.0, 1.0, 0.0, 1.0, 4.0]),
         ("smooth maximum", [0.9, 1.0, 1.02, 1.0, 0.9]),
         ("isolated spike", [0.0, 0.0, 1.0, 0.0, 0.0]),
         ("step up", [0.0, 0.0, 0.0, 1.0, 1.0]),
         ("step down", [1.0, 1.0, 0.0, 0.0, 0.0]),
         ("second differences at round-off of the data (|d2| ~ 1e-13 max|q|)", [1.0, 1.0 + 1e-13, 1.0, 1.0 + 1e-13, 1.0]),
         ("overshoot on the lower side", [0.0, 0.1, 0.2, 1.5, 3.5]),
         ("overshoot on the upper side", [3.5, 1.5, 0.2, 0.1, 0.0]),
         ("interface extremum below the cell", [1.0, 0.0, 0.9, 1.0, 1.1]),
         ("interface extremum above the cell", [1.1, 1.0, 0.9, 0.0, 1.0]),
         ("sawtooth", [1.0, -1.0, 1.0, -1.0, 1.0]),
         ("negative values, monotone", [-5.0, -3.0, -2.5, -1.0, -0.2]),
         ("large dynamic range", [1e-8, 1e-4, 1.0, 1e4, 1e8]),
         ("near-extremum kept by the ratio test", [1.0, 1.5, 1.75, 1.5, 1.0])]
    rng = np.random.default_rng(20240929)
    for n in range(24):
        s.append(("random %d" % n, list(np.round(rng.normal(size=5), 3))))
    return s


def c2p_cases():
    """[(label, eos kwargs, conserved state)] GLM-MHD"""
    def cons(d, v, p, b=(0.0, 0.0, 0.0), psi=0.0):
        e = p / (GAMMA - 1.0) + 0.5 * d * sum(x * x for x in v) + 0.5 * sum(x * x for x in b) + 0.5 * psi * psi
        # (the reference's GLM-MHD ConsToPrim does not subtract psi^2/2: keep psi = 0 so that p is what we set)
        return [d, d * v[0], d * v[1], d * v[2], e, b[0], b[1], b[2], psi]
    fl = dict(pfloor=1e-3, dfloor=1e-2)
    return [("regular cell, no floor acts", fl, cons(1.0, (0.1, 0.2, -0.1), 1.0, (0.3, 0.2, 0.1))),
            ("density below the floor", fl, cons(1e-3, (0.1, 0.0, 0.0), 1.0)),
            ("negative density, floored", fl, [-0.5, 0.1, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0]),
            ("pressure below the floor", fl, cons(1.0, (0.1, 0.0, 0.0), 1e-5, (0.1, 0.0, 0.0))),
            ("negative pressure (kinetic + magnetic energy exceed E), floored", fl, [1.0, 2.0, 0.0, 0.0, 1.0, 1.0, 0.0, 0.0, 0.0]),
            ("velocity above the ceiling", dict(vceil=2.0), cons(1.0, (3.0, 4.0, 0.0), 1.0)),
            ("internal energy below the floor", dict(efloor=0.5), cons(1.0, (0.1, 0.0, 0.0), 0.01)),
            ("internal energy above the ceiling", dict(eceil=2.0), cons(1.0, (0.1, 0.0, 0.0), 10.0)),
            ("velocity ceiling and pressure floor together", dict(vceil=1.0, pfloor=1e-2), cons(1.0, (2.0, 0.0, 0.0), 1e-4))]


PPM_BLOCK = dict(nx=(32, 6, 6), ng=3, dx=(0.1, 0.07, 0.13), gam0=0.5, gam1=0.5, beta_dt=0.002, alpha=0.1, mindx=0.07)


def ppm_block():
    """One GLM-MHD block whose pencils along EVERY direction are the crafted stencils above, strung
    together: w(k,j,i) = base + a S[i] + b S[j'] + c S[k'] per variable with different offsets into
    the sequence S, so that the x1 sweep, the x2 / x3 marches and the two-kernel stage all meet them
    (adding a constant along a pencil changes no difference, hence no limiter decision except the
    scale-dependent round-off guard).  Returns prim [1][9][Nk][Nj][Ni]."""
    nx, ng = PPM_BLOCK["nx"], PPM_BLOCK["ng"]
    seq = np.concatenate([np.clip(np.array(v, dtype=np.float64), -3.0, 3.0) for _, v in ppm_stencils()])
    N = [n + 2 * ng for n in nx]
    w = np.zeros((1, 9, N[2], N[1], N[0]))
    base = [1.5, 0.0, 0.0, 0.0, 1.2, 0.4, 0.2, -0.3, 0.0]
    amp = [0.12, 0.05, 0.05, 0.05, 0.1, 0.08, 0.08, 0.08, 0.01]
    k, j, i = np.meshgrid(np.arange(N[2]), np.arange(N[1]), np.arange(N[0]), indexing="ij")
    for v in range(9):
        w[0, v] = base[v] + amp[v] * (seq[(i + 7 * v) % seq.size] + 0.5 * seq[(j + 11 * v + 3) % seq.size]
                                      + 0.5 * seq[(k + 13 * v + 5) % seq.size])
    assert w[0, 0].min() > 0.5 and w[0, 4].min() > 0.3
    return w


REQUIRED = {"hlld": ["hlld_fl", "hlld_fr", "hlld_lstar", "hlld_ldstar", "hlld_rdstar", "hlld_rstar", "hlld_deg_l", "hlld_deg_r",
                     "hlld_deg_dst"],
            "hllc": ["hllc_am_pos", "hllc_cp_clip", "hllc_ql", "hllc_qr"],
            "ppm": ["ppm_lim_m", "ppm_lim_p", "ppm_extremum", "ppm_roundoff", "ppm_ratio_big", "ppm_over_m", "ppm_over_p"],
            "c2p": ["c2p_dfloor", "c2p_vceil", "c2p_pfloor", "c2p_efloor", "c2p_eceil"]}


def main():
    O.build()
    out, meta = {}, {"gamma": GAMMA, "c_h": C_H, "riemann": [], "ppm": [], "c2p": []}
    hit = set()
    # ---- Riemann: flux for sweeps along x1, x2, x3 (the case's states are permuted accordingly)
    rc = riemann_cases()
    for n, (label, fluid, riemann, wl, wr) in enumerate(rc):
        wl, wr = np.array(wl, dtype=np.float64), np.array(wr, dtype=np.float64)
        nv = wl.size
        fluxes, masks = [], []
        for ivx in (1, 2, 3):
            # rotate the vector components so that the case's "x1" becomes direction ivx
            def rot(w):
                r = w.copy()
                src = [1, 2, 3]
                dst = [1 + (ivx - 1 + k) % 3 for k in range(3)]
                for a, b in zip(src, dst):
                    r[b] = w[a]
                    if nv == 9:
                        r[4 + b] = w[4 + a]
                return r
            f, m = O.riemann_many_traced(fluid, riemann, ivx, rot(wl), rot(wr), GAMMA, C_H)
            fluxes.append(f[0])
            masks.append(int(m[0]))
            out["riemann_%02d_wl_dir%d" % (n, ivx)] = rot(wl)
            out["riemann_%02d_wr_dir%d" % (n, ivx)] = rot(wr)
            out["riemann_%02d_flux_dir%d" % (n, ivx)] = f[0]
        assert len(set(masks)) == 1, (label, masks)          # the direction must not change the branch
        names = O.trace_names(masks[0])
        hit.update(names)
        meta["riemann"].append({"label": label, "fluid": fluid, "riemann": riemann, "mask": masks[0], "branches": names})
    # ---- PPM
    ps = ppm_stencils()
    q = np.array([s[1] for s in ps], dtype=np.float64)
    ql, qr, masks = O.recon_many_traced("ppm", q)
    out["ppm_q"], out["ppm_ql"], out["ppm_qr"] = q, ql, qr
    for (label, _), m in zip(ps, masks):
        names = O.trace_names(int(m))
        hit.update(names)
        meta["ppm"].append({"label": label, "mask": int(m), "branches": names})
    # the other reconstructions on the same stencils (no special-case branches to trace, but frozen too)
    for rec in ("plm", "wenoz", "weno3", "limo3"):
        a, b = O.recon_many(rec, q, dx=0.1, n=0)
        out["%s_ql" % rec], out["%s_qr" % rec] = a, b
    # ---- ConsToPrim
    for n, (label, ekw, u) in enumerate(c2p_cases()):
        eos = O.make_eos(GAMMA, **ekw)
        u2, w, st, m = O.c2p_many_traced("glmmhd", eos, [u])
        out["c2p_%02d_u" % n], out["c2p_%02d_u_after" % n], out["c2p_%02d_w" % n] = np.array(u, dtype=np.float64), u2[0], w[0]
        names = O.trace_names(int(m[0]))
        hit.update(names)
        meta["c2p"].append({"label": label, "eos": ekw, "status": int(st[0]), "mask": int(m[0]), "branches": names})
    # ---- a block made of the crafted stencils, through fluxes and one full general stage
    import helpers as H
    w = ppm_block()
    nx, ng = PPM_BLOCK["nx"], PPM_BLOCK["ng"]
    g = H.geom("glmmhd", nx, ng, 0, PPM_BLOCK["dx"])
    cons = H.prim_to_cons("glmmhd", w, GAMMA)
    fl = H.orc_fluxes("glmmhd", "ppm", "hlld", g, w, GAMMA, C_H)
    stage = H.orc_stage("glmmhd", "ppm", "hlld", g, cons, cons * 1.01, w, GAMMA, C_H, PPM_BLOCK["gam0"], PPM_BLOCK["gam1"],
                        PPM_BLOCK["beta_dt"], dedner=1, alpha=PPM_BLOCK["alpha"], mindx=PPM_BLOCK["mindx"])
    out["block_prim"], out["block_stage"] = w, stage
    for d in range(3):
        out["block_flux%d" % (d + 1)] = fl[d]
    # which PPM branches the block's pencils take, per sweep direction (every cell whose stencil is in the block)
    bc = {}
    for d, ax in ((1, 4), (2, 3), (3, 2)):
        q = np.moveaxis(w[0], ax - 1, -1)                      # pencil direction last
        win = np.lib.stride_tricks.sliding_window_view(q, 5, axis=-1).reshape(-1, 5)
        _, _, m = O.recon_many_traced("ppm", win)
        for name, bit in O.TRACE_BITS.items():
            if name.startswith("ppm_"):
                bc["dir%d_%s" % (d, name)] = int(((m & bit) != 0).sum())
    meta["block"] = dict(PPM_BLOCK, branch_hit_counts=bc)
    assert all(bc["dir%d_%s" % (d, n)] > 0 for d in (1, 2, 3) for n in REQUIRED["ppm"] if n != "ppm_roundoff"), bc
    # ---- coverage
    missing = [b for grp in REQUIRED.values() for b in grp if b not in hit]
    if missing:
        raise SystemExit("crafted cases do not reach: %s" % missing)
    counts = {}
    for grp in ("riemann", "ppm", "c2p"):
        for c in meta[grp]:
            for b in c["branches"]:
                counts[b] = counts.get(b, 0) + 1
    meta["branch_hit_counts"] = dict(sorted(counts.items()))
    np.savez_compressed(os.path.join(HERE, "edge_cases.npz"), **out)
    with open(os.path.join(HERE, "edge_cases.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print("cases: %d riemann x 3 directions, %d ppm stencils, %d c2p" % (len(rc), len(ps), len(meta["c2p"])))
    print("branch hit counts:", meta["branch_hit_counts"])


if __name__ == "__main__":
    main()
